"""Generalized Morse wavelet object (reference: ghost/wave/morse.py:12-211).

Holds ``fs, gamma, beta`` and the peak frequency; gives the frequency bounds and
per-scale kernel lengths ``transform`` needs.  ``__call__`` returns the
reference's L-point ``(psi, psif)`` pair for inspection/plotting; the engine
itself never calls it (its filter bank is built on the GPU).
"""
import copy

import numpy as np

from .wavelet import Positive, Wavelet
from . import morseutils

__all__ = ["Morse"]


class Morse(Wavelet):

    def __init__(self, *, fs=None, freq=None, gamma=None, beta=None):
        super().__init__()
        self.fs = 1 if fs is None else fs
        # the reference stores a plain attribute here (morse.py:41) and so never
        # initialises the peak frequency; we set it as evidently intended
        self.frequency = 0.25 * self.fs if freq is None else freq
        self.gamma = 3 if gamma is None else gamma
        self.beta = 20 if beta is None else beta

    def __call__(self, length, *, normalization=None):
        """(psi, psif) of ``length`` samples (morse.py:53-91 -> morseutils.py:22-198):
        'bandpass' (default; the spectrum peaks at 2) or 'energy' normalisation."""
        n = 16384 if length is None else length
        if n < 1:
            raise ValueError("length must at least 1 but got {}".format(n))
        norm = normalization or "bandpass"
        if norm not in ("bandpass", "energy"):
            raise ValueError("normalization must be 'bandpass' or 'energy' but got {}"
                             .format(norm))
        psi, psif = morseutils.morsewave(n, self._gamma, self._beta, self._norm_radian_freq,
                                         normalization=norm)
        return psi[:, 0, 0], psif[:, 0, 0]

    def compute_freq_bounds(self, N, *, p=None, **kwargs):
        """[lowest, highest] peak frequency (rad/sample) for N samples (morse.py:93-106)."""
        p = 5 if p is None else p
        wh = morseutils.morsehigh(self._gamma, self._beta, **kwargs)
        w0 = morseutils.morsefreq(self._gamma, self._beta)
        max_scale = int(np.floor(N / p)) / morseutils.base_length(self._gamma, self._beta)
        return [w0 / max_scale, wh]

    def compute_lengths(self, norm_radian_freqs):
        """Kernel length per frequency: ceil(w0/omega * base) (morse.py:108-122)."""
        w0 = morseutils.morsefreq(self._gamma, self._beta)
        scale = w0 / norm_radian_freqs
        return np.ceil(scale * morseutils.base_length(self._gamma, self._beta)).astype(int)

    def copy(self):
        return copy.deepcopy(self)

    def _norm_radians_to_hz(self, val):
        return val / np.pi * self._fs / 2

    def _hz_to_norm_radians(self, val):
        return val / (self._fs / 2) * np.pi

    # Validated parameters.  The range tests of the reference's frequency setters only ever
    # reject values <= 0 (morse.py:158, :174), so "positive" is the whole contract.
    fs = Positive("fs must be positive but got {}")
    gamma = Positive("gamma must be positive")
    beta = Positive("beta must be positive")
    frequency = Positive("The frequency must be between 0 and the Nyquist frequency"
                         " but got {} (fs = {fs})", after="_peak_from_hz", store="_freq")
    norm_radian_freq = Positive("The normalized radian frequency must be between 0 and the"
                                " Nyquist frequency pi but got {}", after="_peak_from_radians")

    def _peak_from_hz(self, hz):
        self._norm_radian_freq = self._hz_to_norm_radians(hz)

    def _peak_from_radians(self, omega):
        self._freq = self._norm_radians_to_hz(omega)

    @property
    def time_bandwidth(self):
        return self._gamma * self._beta

"""Morlet wavelet object (reference: ghost/wave/morlet.py:10-139).

A time-domain wavelet sampled at ``fs`` with its spectral peak at ``freq`` Hz.  The
reference's ``transform()`` cannot take it (it has neither ``compute_freq_bounds`` nor
``compute_lengths``); like there it is a kernel factory -- convolve a signal with
``get_wavelet()`` through ``ghost_amd.sigtools.fastconv_hip`` for one Morlet scale.
"""
import copy

import numpy as np

from .wavelet import Positive, Wavelet

__all__ = ["Morlet"]


class Morlet(Wavelet):

    def __init__(self, *, w0=None, freq=None, fs=None):
        super().__init__()
        self._w0 = 6 if w0 is None else w0          # non-dimensional frequency (> 5: admissible)
        self._freq = 1 if freq is None else freq    # Hz at the spectral peak
        self._fs = 1 if fs is None else fs
        self._scale = None
        self._time_repr = None
        self._recompute()

    def get_wavelet(self):
        return self._time_repr

    def _recompute(self):
        """Scale from the peak frequency (morlet.py:52-54), then the sampled, energy-
        normalised wavelet over 15 scales' worth of samples (morlet.py:56-76)."""
        w0 = self._w0
        self._scale = (w0 + np.sqrt(2 + w0 ** 2)) / (4 * np.pi * self._freq)
        dt = 1 / self._fs
        span = 15 * self._fs * self._scale
        eta = np.arange(-(span + 1) / 2, (span + 1) / 2) * dt / self._scale
        carrier = np.exp(1j * w0 * eta) - np.exp(-0.5 * w0 ** 2)   # zero-mean correction
        self._time_repr = (np.pi ** -0.25 * np.exp(-0.5 * eta ** 2) * carrier
                           * np.sqrt(dt / self._scale))

    def copy(self):
        return copy.deepcopy(self)

    # every parameter is positive and re-samples the wavelet when it changes
    fs = Positive("Sampling rate must be positive", after="_changed")
    w0 = Positive("Frequency ratio must be positive", after="_changed")
    freq = Positive("The wavelet frequency must be positive", after="_changed")

    def _changed(self, _value):
        self._recompute()

    @property
    def scale(self):
        return self._scale

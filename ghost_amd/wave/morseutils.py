"""Scalar properties of generalized Morse wavelets used when setting up a
transform (reference: ghost/wave/morseutils.py).  Host-side, evaluated once per
call; the per-sample filter arithmetic lives in the HIP library."""
import numpy as np

__all__ = ["morsefreq", "morsehigh", "morseprops", "base_length"]


def _check(gamma, beta):
    if gamma < 0:
        raise ValueError("Gamma must be positive")
    if beta < 0:
        raise ValueError("Beta must be positive")


def morsefreq(gamma, beta):
    """Peak radian frequency (beta/gamma)**(1/gamma)  (morseutils.py:315)."""
    _check(gamma, beta)
    return np.exp((np.log(beta) - np.log(gamma)) / gamma)


def morsehigh(gamma, beta, eta=None):
    """Largest peak frequency whose wavelet is below ``eta`` of its peak at
    Nyquist, searched on the reference's 10 000-point grid (morseutils.py:607-624)."""
    _check(gamma, beta)
    eta = 0.1 if eta is None else eta
    if eta < 0 or eta > 1:
        raise ValueError("eta must be between 0 and 1")
    grid = np.linspace(1e-12, np.pi, 10000)
    w = morsefreq(gamma, beta) * np.pi / grid
    ln_psi = (beta / gamma) * np.log(np.e * gamma / beta) + beta * np.log(w) - w ** gamma
    return grid[np.flatnonzero(np.log(eta) - ln_psi < 0)[0]]


def morseprops(gamma, beta):
    """(window width, skewness, kurtosis) of the demodulate (morseutils.py:697-703)."""
    _check(gamma, beta)
    p = np.sqrt(gamma * beta)
    skew = (gamma - 3) / beta
    return p, skew, 3 - np.square(skew) - 2 / np.square(p)


def base_length(gamma, beta):
    """Four footprints of the mother wavelet in samples (morse.py:101, :115-116)."""
    return (2 * np.sqrt(2) * np.sqrt(gamma * beta)) / morsefreq(gamma, beta) * 4

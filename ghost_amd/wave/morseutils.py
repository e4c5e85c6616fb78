"""Generalized Morse wavelet utilities (reference: ghost/wave/morseutils.py).

The scalar properties (``morsefreq``, ``morsehigh``, ``morseprops``, ``base_length``) set
up a transform; the rest -- ``morsewave`` with both normalisations and higher orders,
``morseafunc``, ``morsemom``, ``morsef``, ``morsespace``, ``morselow`` -- is the
inspection layer around it.  All host-side and evaluated once per call: the per-sample
filter arithmetic of ``transform()`` lives in the HIP library.  Pinned against the
reference by tests/golden/g10_morse_utils.npz.
"""
import numpy as np
from scipy.special import binom, comb, gamma as _gamma_fn, gammaln

__all__ = ["morsefreq", "morsehigh", "morseprops", "base_length", "morsewave", "morseafunc",
           "morsemom", "morsef", "morsespace", "morselow", "laguerre"]

_NORMALIZATIONS = ("bandpass", "energy")


def _check(gamma, beta):
    if gamma < 0:
        raise ValueError("Gamma must be positive")
    if beta < 0:
        raise ValueError("Beta must be positive")


def morsefreq(gamma, beta, *, nout=None):
    """Characteristic radian frequencies (morseutils.py:275-337).  ``nout`` = 1 (default):
    the peak frequency (beta/gamma)**(1/gamma); 2: + the energy frequency; 3: + the
    instantaneous frequency at the wavelet centre; 4: + its curvature."""
    _check(gamma, beta)
    nout = 1 if nout is None else nout
    if nout not in (1, 2, 3, 4):
        raise ValueError("nout must be 1, 2, 3, or 4")
    res = [np.exp((np.log(beta) - np.log(gamma)) / gamma)]
    if nout >= 2:
        res.append(_gamma_fn((2 * beta + 2) / gamma) / _gamma_fn((2 * beta + 1) / gamma)
                   / 2 ** (1 / gamma))
    if nout >= 3:
        res.append(_gamma_fn((beta + 2) / gamma) / _gamma_fn((beta + 1) / gamma))
    if nout == 4:
        k2 = morsemom(2, gamma, beta, nout=3)[2]
        k3 = morsemom(3, gamma, beta, nout=3)[2]
        res.append(-k3 / np.sqrt(k2 ** 3))
    return res[0] if nout == 1 else tuple(res)


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


def morselow(gamma, beta, pack_num, N):
    """Lowest peak frequency for an N-sample series: the wavelet spans ``pack_num`` window
    widths at the series' ends (morseutils.py:626-667)."""
    _check(gamma, beta)
    pack_num = 5 if pack_num is None else pack_num
    if not pack_num > 0:
        raise ValueError("pack_num must be positive")
    if N < 2:
        raise ValueError("N must be at least 2")
    return 2 * np.sqrt(2) * morseprops(gamma, beta)[0] * pack_num / N


def morsespace(gamma, beta, N, *, high=None, eta=None, pack_num=None, low=None, density=None):
    """Log-spaced peak frequencies (rad/sample, ascending) between ``morselow`` and
    ``morsehigh``, neighbours a factor 1 + 1/(density*P) apart (morseutils.py:473-571; the
    density default is 2 as in the reference's code)."""
    _check(gamma, beta)
    if N < 2:
        raise ValueError("N must be at least 2")
    eta = 0.1 if eta is None else eta
    if eta < 0 or eta > 1:
        raise ValueError("eta must be between 0 and 1")
    high = np.pi if high is None else high
    if high < 0 or high > np.pi:
        raise ValueError("high must be between 0 and pi")
    pack_num = 5 if pack_num is None else pack_num
    if not pack_num > 0:
        raise ValueError("pack_num must be positive")
    low = 0 if low is None else low
    if low < 0 or low > np.pi:
        raise ValueError("low must be between 0 and pi")
    density = 2 if density is None else density
    if not density > 0:
        raise ValueError("density must be positive")
    top = min(high, morsehigh(gamma, beta, eta))
    bottom = max(low, morselow(gamma, beta, pack_num, N))
    ratio = 1 + 1 / (density * morseprops(gamma, beta)[0])
    count = int(np.floor(np.log(top / bottom) / np.log(ratio)))
    return (top / ratio ** np.arange(count + 1))[::-1]


def morsef(gamma, beta):
    """Normalised first frequency-domain moment Gamma((beta+1)/gamma) / (2 pi gamma)
    (morseutils.py:449-471)."""
    return _gamma_fn((beta + 1) / gamma) / (2 * np.pi * gamma)


def morseafunc(gamma, beta, *, normalization=None, order=None):
    """Amplitude coefficient: 'bandpass' makes the spectrum peak at 2, 'energy' gives the
    order-``order`` wavelet unit energy (morseutils.py:200-254)."""
    _check(gamma, beta)
    normalization = "bandpass" if normalization is None else normalization
    if normalization not in _NORMALIZATIONS:
        raise ValueError("Normalization must be 'bandpass' or 'energy'")
    order = 1 if order is None else order
    if order < 0:
        raise ValueError("Order must non-negative")
    if normalization == "bandpass":
        if beta == 0:
            return 2
        wp = morsefreq(gamma, beta)
        return 2 / np.exp(beta * np.log(wp) - wp ** gamma)
    r = (2 * beta + 1) / gamma
    return np.sqrt(2 * np.pi * gamma * 2 ** r * np.exp(gammaln(order) - gammaln(order + r - 1)))


def _moment(p, gamma, beta):
    return morseafunc(gamma, beta) * morsef(gamma, beta + p)


def _cumulants(moments):
    """kappa_0 = ln m_0; kappa_n = m_n/m_0 - sum_{k=1}^{n-1} C(n-1,k-1) kappa_k m_{n-k}/m_0
    (morseutils.py:419-447)."""
    m = np.atleast_1d(np.asarray(moments, dtype=float))
    if m.ndim != 1:
        raise ValueError("Moments must be either a scalar or array with only one"
                         " non-singleton dimension")
    kappa = np.zeros(m.size)
    kappa[0] = np.log(m[0])
    for n in range(1, m.size):
        k = np.arange(1, n)
        kappa[n] = (m[n] - np.sum(comb(n - 1, k - 1) * kappa[k] * m[n - k])) / m[0]
    return kappa


def morsemom(p, gamma, beta, *, nout=None):
    """Frequency-domain moments under the bandpass normalisation (morseutils.py:339-413):
    the p-th moment; with ``nout`` >= 2 the energy moment, >= 3 the p-th cumulant, 4 the
    p-th energy cumulant."""
    _check(gamma, beta)
    if p < 0:
        raise ValueError("p must be non-negative")
    nout = 1 if nout is None else nout
    if nout not in (1, 2, 3, 4):
        raise ValueError("nout must be 1, 2, 3, or 4")

    def energy_moment(q):
        return 2 / 2 ** ((1 + q) / gamma) * _moment(q, gamma, 2 * beta)

    res = [_moment(p, gamma, beta)]
    if nout >= 2:
        res.append(energy_moment(p))
    orders = np.arange(p + 1)
    if nout >= 3:
        res.append(_cumulants(_moment(orders, gamma, beta))[p])
    if nout == 4:
        res.append(_cumulants(energy_moment(orders))[p])
    return res[0] if nout == 1 else tuple(res)


def laguerre(x, k, c):
    """Generalised Laguerre polynomial L_k^(c)(x) = sum_m (-1)^m C(k+c, k-m) x^m / m!
    (morseutils.py:256-273)."""
    x = np.atleast_1d(np.asarray(x, dtype=np.float64).squeeze())
    if x.ndim != 1:
        raise ValueError("The input x must have only one non-singleton dimension")
    y = np.zeros_like(x)
    for m in range(k + 1):
        y += (-1) ** m * binom(k + c, k - m) * x ** m / _gamma_fn(m + 1)
    return y


def _one_frequency(N, n_wavelets, gamma, beta, freq, normalization):
    """(psi, psif), each (N, n_wavelets), for one positive peak frequency
    (morseutils.py:93-198)."""
    w0 = morsefreq(gamma, beta)
    stretch = freq / w0
    w = 2 * np.pi * np.linspace(0, 1 - 1 / N, N) / stretch
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        if normalization == "energy":
            # the reference writes beta ** log(w) here (morseutils.py:124), not
            # beta * log(w); its numbers are the contract, so the same expression is used
            base = np.exp(-w ** gamma) if beta == 0 else np.exp(beta ** np.log(w) - w ** gamma)
        elif beta == 0:
            base = 2 * np.exp(-w ** gamma)
        else:
            base = 2 * np.exp(-beta * np.log(w0) + w0 ** gamma + beta * np.log(w) - w ** gamma)
    base[0] /= 2                                  # unit step at zero frequency
    r = (2 * beta + 1) / gamma
    half = round(N / 2)                           # one-sided: bins 0 .. round(N/2)-1
    psif = np.zeros((N, n_wavelets))
    for order in range(n_wavelets):
        if normalization == "energy":
            coeff = np.sqrt(1 / stretch) * morseafunc(gamma, beta, order=order + 1,
                                                      normalization="energy")
        elif beta != 0:
            coeff = np.sqrt(np.exp(gammaln(r) + gammaln(order + 1) - gammaln(order + r)))
        else:
            coeff = 1
        poly = np.zeros(N)
        poly[:half] = laguerre(2 * w[:half] ** gamma, order, r - 1)
        with np.errstate(invalid="ignore"):
            psif[:, order] = coeff * base * poly
    psif[psif == np.inf] = 0
    centred = psif * np.exp(1j * w * (N + 1) / 2 * stretch)[:, None]
    return np.fft.ifft(centred, axis=0), psif


def morsewave(N, gamma, beta, freqs, *, n_wavelets=None, normalization=None):
    """Time- and frequency-domain generalized Morse wavelets of N samples
    (morseutils.py:22-91): ``psi`` complex and ``psif`` real, both (N, len(freqs),
    n_wavelets); ``freqs`` are peak radian frequencies (negative ones give the conjugate
    wavelet).  'bandpass' (default): the spectrum peaks at 2; 'energy': unit energy."""
    _check(gamma, beta)
    freqs = np.atleast_1d(np.asarray(freqs, dtype=float).squeeze())
    if freqs.ndim != 1:
        raise ValueError("Freqs must be either a scalar or an array"
                         " with one non-singleton dimension")
    n_wavelets = 1 if n_wavelets is None else n_wavelets
    if not n_wavelets > 0:
        raise ValueError("n_wavelets must be positive")
    normalization = "bandpass" if normalization is None else normalization
    if normalization not in _NORMALIZATIONS:
        raise ValueError("Normalization must be 'energy' or 'bandpass'")
    psi = np.zeros((N, freqs.size, n_wavelets), dtype=complex)
    psif = np.zeros((N, freqs.size, n_wavelets))
    for i, f in enumerate(freqs):
        t, s = _one_frequency(N, n_wavelets, gamma, beta, abs(f), normalization)
        if f < 0:
            t = t.conj()
            s[1:] = s[:0:-1].copy()
        psi[:, i, :], psif[:, i, :] = t, s
    return psi, psif

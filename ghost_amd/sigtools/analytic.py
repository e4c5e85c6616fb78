"""Analytic signal on the GPU (reference: ghost/sigtools/analytic.py).

``analytic_signal_hip`` plays the role of ``analytic_signal_fftw`` (:22-112): same
arguments, same checks, the numbers of ``scipy.signal.hilbert(x, N=fft_length)[:len(x)]``.
It runs ``gcwt_analytic_signal``; float32 arithmetic, complex64 result.
"""
import ctypes as C

import numpy as np

from .._lib import lib, check

__all__ = ["analytic_signal_hip"]

lib.gcwt_analytic_signal.restype = C.c_int
lib.gcwt_analytic_signal.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int]


def analytic_signal_hip(signal, *, fft_length=None, device=-1, precision=None):
    """x_a = x + i*H(x) for a real 1-D ``signal``.  ``fft_length`` (default: no padding) is
    the DFT length used, any integer >= len(signal) and <= 2**21 (2**23 with precision='high': float64 transforms
    and a complex128 result, what analytic.py:22-112 returns)."""
    signal = np.asarray(signal)
    if np.iscomplexobj(signal):
        raise ValueError("The input data must be real")
    if signal.size == 0:
        raise ValueError("Cannot compute analytic signal on an empty array")
    if signal.ndim != 1:
        raise ValueError("Input data must be 1-dimensional")
    n = signal.shape[-1]
    if fft_length is None:
        fft_length = n
    fft_length = int(fft_length)
    if fft_length < n:
        raise ValueError("'fft_length' must be at least the length of the"
                         " input data")
    from .fourier import _check_precision, _dft
    if _check_precision(precision) == "high":
        # the DFT, the one-sided mask of scipy.signal.hilbert (analytic.py:80-98), the inverse DFT: float64 on the device
        x = np.zeros(fft_length, dtype=np.float64)
        x[:n] = signal
        spec = _dft(x, False, device, "high")
        h = np.zeros(fft_length)
        if fft_length % 2 == 0:
            h[0] = h[fft_length // 2] = 1.0
            h[1:fft_length // 2] = 2.0
        else:
            h[0] = 1.0
            h[1:(fft_length + 1) // 2] = 2.0
        return _dft(spec * h, True, device, "high")[:n]
    x = np.ascontiguousarray(signal, dtype=np.float32)
    out = np.empty(n, dtype=np.complex64)
    check(lib.gcwt_analytic_signal(x.ctypes.data_as(C.c_void_p), n, fft_length,
                                   out.ctypes.data_as(C.c_void_p), int(device)))
    return out

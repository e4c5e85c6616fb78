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


def analytic_signal_hip(signal, *, fft_length=None, device=-1):
    """x_a = x + i*H(x) for a real 1-D ``signal``.  ``fft_length`` (default: no padding) is
    the DFT length used, any integer >= len(signal) and <= 2**21."""
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
    x = np.ascontiguousarray(signal, dtype=np.float32)
    out = np.empty(n, dtype=np.complex64)
    check(lib.gcwt_analytic_signal(x.ctypes.data_as(C.c_void_p), n, fft_length,
                                   out.ctypes.data_as(C.c_void_p), int(device)))
    return out

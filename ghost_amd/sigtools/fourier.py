"""Arbitrary-length DFT on the GPU (reference: ghost/sigtools/fourier.py).

``chirpz_dft_hip`` plays the role of ``chirpz_dft`` (:9-52): the DFT of a 1-D signal of any
length (prime lengths included) through the chirp-z identity, here on the device's
power-of-two FFTs (``gcwt_dft``).  float32 arithmetic, complex64 result.
"""
import ctypes as C

import numpy as np

from .._lib import lib, check

__all__ = ["chirpz_dft_hip", "chirpz_idft_hip"]

lib.gcwt_dft.restype = C.c_int
lib.gcwt_dft.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int]


lib.gcwt_dft_f64.restype = C.c_int
lib.gcwt_dft_f64.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int]


def _check_precision(precision):
    if precision not in (None, "fast", "high"):
        raise ValueError("'precision' must be 'fast' (float32 arithmetic, complex64 result) or 'high' (float64, complex128: "
                         "what the reference returns)")
    return "fast" if precision is None else precision


def _dft(x, inverse, device, precision=None):
    x = np.asarray(x)
    if x.ndim != 1:
        raise ValueError("Data must be 1-dimensional")
    if x.size == 0:
        raise ValueError("Data must not be empty")
    cplx = np.iscomplexobj(x)
    if _check_precision(precision) == "high":
        v = np.ascontiguousarray(x, dtype=np.complex128 if cplx else np.float64)
        out = np.empty(x.shape[0], dtype=np.complex128)
        check(lib.gcwt_dft_f64(v.ctypes.data_as(C.c_void_p), x.shape[0], 1 if cplx else 0,
                               1 if inverse else 0, out.ctypes.data_as(C.c_void_p), int(device)))
        return out
    v = np.ascontiguousarray(x, dtype=np.complex64 if cplx else np.float32)
    out = np.empty(x.shape[0], dtype=np.complex64)
    check(lib.gcwt_dft(v.ctypes.data_as(C.c_void_p), x.shape[0], 1 if cplx else 0,
                       1 if inverse else 0, out.ctypes.data_as(C.c_void_p), int(device)))
    return out


def chirpz_dft_hip(x, *, device=-1, precision=None):
    """The DFT of ``x`` (real or complex, 1-D, up to 2**21 points; 2**23 with precision='high': float64
    arithmetic and a complex128 result, what fourier.py:9-52 returns)."""
    return _dft(x, False, device, precision)


def chirpz_idft_hip(x, *, device=-1, precision=None):
    """The normalised inverse DFT of ``x``."""
    return _dft(x, True, device, precision)

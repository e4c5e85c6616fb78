"""FFT convolution on the GPU (reference: ghost/sigtools/convolution.py).

``fastconv_hip`` plays the role of ``fastconv_scipy`` / ``fastconv_fftw`` (:16-216),
``fastconv_freq_hip`` that of ``fastconv_freq_scipy`` / ``fastconv_freq_fftw`` (:218-402).
Both run ``gcwt_fastconv``: one FFT of length 2^k >= N + M - 1 instead of chunked
overlap-add (the results are the same linear convolution).  float32 arithmetic; the
result is complex64 for a complex kernel and float32 for a real one.
"""
import ctypes as C

import numpy as np

from .._lib import lib, check

__all__ = ["fastconv_hip", "fastconv_freq_hip"]

_MODES = {"full": 0, "same": 1, "valid": 2}

lib.gcwt_fastconv.restype = C.c_int
lib.gcwt_fastconv.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                              C.c_void_p, C.c_int]


def fastconv_hip(signal, kernel, *, mode=None, device=-1):
    """Linear convolution of a real 1-D ``signal`` with a real or complex 1-D ``kernel``.
    ``mode``: 'full', 'same' (default, centred as convolution.py:85) or 'valid'."""
    signal = np.asarray(signal)
    kernel = np.asarray(kernel)
    if signal.ndim != 1:
        raise ValueError("Signal must be 1D")
    if kernel.ndim != 1:
        raise ValueError("Kernel must be 1D")
    if np.iscomplexobj(signal):
        raise TypeError("signal must be real")
    if mode is None:
        mode = "same"
    if mode not in _MODES:
        raise ValueError("Mode must be 'full', 'same', or 'valid'")
    n, m = signal.shape[0], kernel.shape[0]
    if mode == "valid" and n < m:
        raise ValueError("Cannot do a 'valid' convolution because "
                         "the input is shorter than the kernel")
    cplx = np.iscomplexobj(kernel)
    x = np.ascontiguousarray(signal, dtype=np.float32)
    k = np.ascontiguousarray(kernel, dtype=np.complex64 if cplx else np.float32)
    count = {"full": n + m - 1, "same": n, "valid": n - m + 1}[mode]
    out = np.empty(count, dtype=np.complex64)
    check(lib.gcwt_fastconv(x.ctypes.data_as(C.c_void_p), n, k.ctypes.data_as(C.c_void_p), m,
                            1 if cplx else 0, _MODES[mode], out.ctypes.data_as(C.c_void_p),
                            int(device)))
    return out if cplx else np.ascontiguousarray(out.real)


def fastconv_freq_hip(signal_td, kernel_fd, kernel_len, *, mode=None, device=-1):
    """Convolution with a kernel given by its DFT (any length >= ``kernel_len``), as
    ``fastconv_freq_scipy(signal_td, kernel_fd, kernel_len, mode=...)``.  The kernel is
    taken back to the time domain on the host (it is short) and ``fastconv_hip`` does the
    rest."""
    kernel_fd = np.asarray(kernel_fd)
    if kernel_fd.ndim != 1:
        raise ValueError("Kernel must be 1D")
    kernel_td = np.fft.ifft(kernel_fd)[:int(kernel_len)]
    if np.abs(kernel_td.imag).max() <= 1e-12 * max(np.abs(kernel_td).max(), 1e-300):
        kernel_td = kernel_td.real
    return fastconv_hip(signal_td, kernel_td, mode=mode, device=device)

"""FFT convolution on the GPU (reference: ghost/sigtools/convolution.py).

``fastconv_hip`` plays the role of ``fastconv_scipy`` / ``fastconv_fftw`` (:16-216),
``fastconv_freq_hip`` that of ``fastconv_freq_scipy`` / ``fastconv_freq_fftw`` (:218-402).
Both run on a ``ConvPlan`` -- the library's reusable convolution operator
(``gcwt_conv_plan_*``): stream, FFT tables, kernel spectrum and workspace made once, batches
of signals ``(C, N)``, device-resident input / output if wanted, signals of any length
(overlap-save chunks of one power-of-two FFT, the reference's chunked overlap-add).
float32 arithmetic; the result is complex64 for a complex kernel and float32 for a real one.
"""
import ctypes as C

import numpy as np

from .._lib import lib, check

__all__ = ["fastconv_hip", "fastconv_freq_hip", "ConvPlan"]

_MODES = {"full": 0, "same": 1, "valid": 2}
_vp, _i64p = C.c_void_p, C.POINTER(C.c_int64)
for _name, _args in (("gcwt_conv_plan_create", [C.POINTER(_vp), C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
                     ("gcwt_conv_plan_info", [_vp, _i64p, _i64p, _i64p]),
                     ("gcwt_conv_plan_set_kernel", [_vp, _vp, C.c_int, C.c_int]),
                     ("gcwt_conv_plan_set_kernel_fd", [_vp, _vp, C.c_int]),
                     ("gcwt_conv_plan_execute", [_vp, _vp, C.c_int, _vp, C.c_int]),
                     ("gcwt_fastconv", [_vp, C.c_int64, _vp, C.c_int64, C.c_int, C.c_int, _vp, C.c_int])):
    getattr(lib, _name).restype = C.c_int
    getattr(lib, _name).argtypes = _args
lib.gcwt_conv_plan_destroy.restype = None
lib.gcwt_conv_plan_destroy.argtypes = [_vp]


def _check_mode(mode, n, m):
    if mode is None:
        mode = "same"
    if mode not in _MODES:
        raise ValueError("Mode must be 'full', 'same', or 'valid'")
    if mode == "valid" and n < m:
        raise ValueError("Cannot do a 'valid' convolution because "
                         "the input is shorter than the kernel")
    return mode


class ConvPlan:
    """Convolution of ``n_channels`` real signals of ``n_samples`` with one kernel of
    ``kernel_len`` taps.  ``fft_length``: None (the smallest power of two that holds the
    whole convolution as one overlap-save chunk, ``n + 2 (m - 1)`` points, at most 2^22; longer
    signals are chunked) or the reference's ``fft_length`` (chunks of ``fft_length - kernel_len + 1``):
    any value up to 2^22 is taken and rounded UP to a power of two of at least 4096 (the
    reference accepts any length; the result does not depend on it)."""

    def __init__(self, n_samples, kernel_len, n_channels=1, *, fft_length=None, device=-1):
        self._handle = _vp()
        log2 = 0
        if fft_length is not None:
            if int(fft_length) < 1 or int(fft_length) > (1 << 22):
                raise ValueError("fft_length must be between 1 and 2**22")
            log2 = max(12, (int(fft_length) - 1).bit_length())      # rounded up to a power of two >= 4096
        check(lib.gcwt_conv_plan_create(C.byref(self._handle), int(n_samples), int(kernel_len),
                                        int(n_channels), log2, int(device)))
        self.n_samples, self.kernel_len, self.n_channels = int(n_samples), int(kernel_len), int(n_channels)
        f, c, k = C.c_int64(), C.c_int64(), C.c_int64()
        check(lib.gcwt_conv_plan_info(self._handle, C.byref(f), C.byref(c), C.byref(k)))
        self.fft_length, self.chunk, self.n_chunks = f.value, c.value, k.value
        self.kernel_is_complex = True

    def set_kernel(self, kernel):
        """Kernel taps (time domain), real or complex, ``kernel_len`` of them."""
        kernel = np.asarray(kernel)
        if kernel.ndim != 1 or kernel.shape[0] != self.kernel_len:
            raise ValueError("Kernel must be 1D with %d taps" % self.kernel_len)
        self.kernel_is_complex = bool(np.iscomplexobj(kernel))
        k = np.ascontiguousarray(kernel, dtype=np.complex64 if self.kernel_is_complex else np.float32)
        check(lib.gcwt_conv_plan_set_kernel(self._handle, k.ctypes.data_as(_vp),
                                            1 if self.kernel_is_complex else 0, 0))
        return self

    def set_kernel_fd(self, kernel_fd, *, real_kernel=False):
        """The kernel by its DFT on the plan's own ``fft_length``-point grid
        (``fastconv_freq_*``'s ``kernel_fd``; ``real_kernel``: its taps are real)."""
        kernel_fd = np.asarray(kernel_fd)
        if kernel_fd.ndim != 1 or kernel_fd.shape[0] != self.fft_length:
            raise ValueError("kernel_fd must have the plan's fft_length (%d) bins" % self.fft_length)
        self.kernel_is_complex = not real_kernel
        k = np.ascontiguousarray(kernel_fd, dtype=np.complex64)
        check(lib.gcwt_conv_plan_set_kernel_fd(self._handle, k.ctypes.data_as(_vp), 0))
        return self

    def count(self, mode):
        n, m = self.n_samples, self.kernel_len
        return {"full": n + m - 1, "same": n, "valid": n - m + 1}[mode]

    def execute(self, signals, *, mode=None):
        """signals (C, N) or (N,) real -> (C, count) or (count,); complex64, or float32 when
        the kernel is real."""
        mode = _check_mode(mode, self.n_samples, self.kernel_len)
        x = np.asarray(signals)
        if np.iscomplexobj(x):
            raise TypeError("signal must be real")
        one = x.ndim == 1
        want = (self.n_samples,) if one else (self.n_channels, self.n_samples)
        if x.shape != want or (one and self.n_channels != 1):
            raise ValueError("signals must have shape (%d, %d)%s, got %s"
                             % (self.n_channels, self.n_samples,
                                " or (%d,)" % self.n_samples if self.n_channels == 1 else "", x.shape))
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(self.n_channels, self.n_samples)
        out = np.empty((self.n_channels, self.count(mode)), dtype=np.complex64)
        check(lib.gcwt_conv_plan_execute(self._handle, x.ctypes.data_as(_vp), _MODES[mode],
                                         out.ctypes.data_as(_vp), 0))
        res = out if self.kernel_is_complex else np.ascontiguousarray(out.real)
        return res[0] if one else res

    def execute_device(self, x_buf, out_buf, *, mode=None):
        """Device-resident form: float32 [C][N] in, (re, im) float32 pairs [C][count] out
        (``ghost_amd.engine.DeviceBuffer`` or raw pointers)."""
        from .._lib import X_ON_DEVICE, OUT_ON_DEVICE
        mode = _check_mode(mode, self.n_samples, self.kernel_len)
        xp = getattr(x_buf, "ptr", x_buf)
        op = getattr(out_buf, "ptr", out_buf)
        check(lib.gcwt_conv_plan_execute(self._handle, xp, _MODES[mode], op, X_ON_DEVICE | OUT_ON_DEVICE))

    def close(self):
        if self._handle:
            lib.gcwt_conv_plan_destroy(self._handle)
            self._handle = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_cache = {}          # the last few plans by layout: repeated calls reuse tables and workspace


def _plan_for(n, m, fft_length, device):
    key = (n, m, fft_length, device)
    plan = _cache.pop(key, None)
    if plan is None:
        plan = ConvPlan(n, m, 1, fft_length=fft_length, device=device)
    _cache[key] = plan                  # most recently used last
    while len(_cache) > 4:
        _cache.pop(next(iter(_cache))).close()
    return plan


lib.gcwt_fastconv_f64.restype = C.c_int
lib.gcwt_fastconv_f64.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int]


def _fastconv_f64(signal, kernel, mode, device):
    """float64 on the device: the reference's arithmetic and result (convolution.py:68: complex128 chunks)."""
    sc, kc = np.iscomplexobj(signal), np.iscomplexobj(kernel)
    x = np.ascontiguousarray(signal, dtype=np.complex128 if sc else np.float64)
    k = np.ascontiguousarray(kernel, dtype=np.complex128 if kc else np.float64)
    n, m = x.shape[0], k.shape[0]
    out = np.empty({"full": n + m - 1, "same": n, "valid": n - m + 1}[mode], dtype=np.complex128)
    check(lib.gcwt_fastconv_f64(x.ctypes.data_as(C.c_void_p), n, 1 if sc else 0, k.ctypes.data_as(C.c_void_p), m,
                                1 if kc else 0, {"full": 0, "same": 1, "valid": 2}[mode],
                                out.ctypes.data_as(C.c_void_p), int(device)))
    return out if (sc or kc) else out.real.copy()


def fastconv_hip(signal, kernel, *, mode=None, fft_length=None, device=-1, precision=None):
    """Linear convolution of a real 1-D ``signal`` with a real or complex 1-D ``kernel``.
    ``mode``: 'full', 'same' (default, centred as convolution.py:85) or 'valid';
    ``fft_length`` as in the reference (a power of two here).  precision='high': float64 arithmetic and a float64 /
    complex128 result, what the reference returns (one FFT of the whole result up to 2**24 samples, overlap-add over chunks of the signal beyond, as convolution.py:70-77)."""
    signal = np.asarray(signal)
    kernel = np.asarray(kernel)
    if signal.ndim != 1:
        raise ValueError("Signal must be 1D")
    if kernel.ndim != 1:
        raise ValueError("Kernel must be 1D")
    if np.iscomplexobj(signal):
        raise TypeError("signal must be real")
    n, m = signal.shape[0], kernel.shape[0]
    mode = _check_mode(mode, n, m)
    if fft_length is not None and fft_length < m:
        raise ValueError("FFT length must be at least the kernel size")
    from .fourier import _check_precision
    if _check_precision(precision) == "high":
        return _fastconv_f64(signal, kernel, mode, device)
    plan = _plan_for(n, m, fft_length, int(device))
    return plan.set_kernel(kernel).execute(signal, mode=mode)


def fastconv_freq_hip(signal_td, kernel_fd, kernel_len, *, mode=None, device=-1, precision=None):
    """Convolution with a kernel given by its DFT (any length >= ``kernel_len``), as
    ``fastconv_freq_scipy(signal_td, kernel_fd, kernel_len, mode=...)``.  When the DFT is
    on a power-of-two grid the plan can take (4096 .. 2^22 bins) it is used as it is and the
    signal is chunked exactly as the reference does (``len(kernel_fd) - kernel_len + 1``
    samples per chunk, convolution.py:262); any other grid goes back to the time domain on
    the host first (the kernel is short)."""
    signal_td = np.asarray(signal_td)
    kernel_fd = np.asarray(kernel_fd)
    if signal_td.ndim != 1:
        raise ValueError("Signal must be 1D")
    if kernel_fd.ndim != 1:
        raise ValueError("Kernel must be 1D")
    n, m, f = signal_td.shape[0], int(kernel_len), kernel_fd.shape[0]
    mode = _check_mode(mode, n, m)
    from .fourier import _check_precision, _dft
    if _check_precision(precision) == "high":
        # the kernel's taps from its DFT (any length) in float64 on the device, then the float64 convolution
        kernel_td = _dft(np.asarray(kernel_fd, dtype=np.complex128), True, device, "high")[:m]
        herm = np.abs(kernel_fd[1:] - np.conj(kernel_fd[:0:-1])).max() <= 1e-12 * np.abs(kernel_fd).max() if f > 1 else True
        return _fastconv_f64(signal_td, kernel_td.real.copy() if herm else kernel_td, mode, device)
    if f >= 4096 and f <= (1 << 22) and (f & (f - 1)) == 0 and f >= m:
        # real taps <=> Hermitian spectrum: hand back float32 like fastconv_hip does
        herm = np.abs(kernel_fd[1:] - np.conj(kernel_fd[:0:-1])).max() <= 1e-6 * np.abs(kernel_fd).max()
        plan = _plan_for(n, m, f, int(device))
        return plan.set_kernel_fd(kernel_fd, real_kernel=bool(herm)).execute(signal_td, mode=mode)
    kernel_td = np.fft.ifft(kernel_fd)[:m]
    if np.abs(kernel_td.imag).max() <= 1e-12 * max(np.abs(kernel_td).max(), 1e-300):
        kernel_td = kernel_td.real
    return fastconv_hip(signal_td, kernel_td, mode=mode, device=device)

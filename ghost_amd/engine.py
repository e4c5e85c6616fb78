"""Thin Python handle on a libghostcwt plan (host side of the C ABI).

``CwtPlan`` is what ``ContinuousWaveletTransform.transform`` drives; bench.py
and the tests use it directly for device-resident runs.  All arithmetic happens
in the HIP library.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import lib, check

__all__ = ["CwtPlan", "DeviceBuffer", "DeviceResult", "set_option", "device_count", "device_name", "device_memory"]


def set_option(name, value=None):
    """Test / measurement switch of the library (include/ghostcwt_debug.h: gcwt_debug_set_option), read by
    plans created afterwards; ``value=None`` restores the default."""
    check(lib.gcwt_debug_set_option(name.encode(), 0 if value is None else int(value), 1 if value is None else 0))


def device_count():
    n = C.c_int(0)
    rc = lib.gcwt_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def device_name(device=0):
    buf = C.create_string_buffer(256)
    check(lib.gcwt_device_name(device, buf, 256))
    return buf.value.decode()


def device_memory():
    """(free, total) bytes of the current device."""
    free, total = C.c_size_t(0), C.c_size_t(0)
    check(lib.gcwt_device_memory(C.byref(free), C.byref(total)))
    return free.value, total.value


class DeviceBuffer:
    """hipMalloc'd bytes owned by Python."""

    def __init__(self, nbytes):
        self.ptr = C.c_void_p()
        self.nbytes = int(nbytes)
        check(lib.gcwt_device_malloc(C.byref(self.ptr), self.nbytes))

    def upload(self, array, offset_bytes=0):
        a = np.ascontiguousarray(array)
        assert offset_bytes + a.nbytes <= self.nbytes
        dst = C.c_void_p(self.ptr.value + offset_bytes)
        check(lib.gcwt_memcpy_h2d(dst, a.ctypes.data_as(C.c_void_p), a.nbytes))

    def download(self, shape, dtype, offset_bytes=0):
        out = np.empty(shape, dtype=dtype)
        assert offset_bytes + out.nbytes <= self.nbytes
        src = C.c_void_p(self.ptr.value + offset_bytes)
        check(lib.gcwt_memcpy_d2h(out.ctypes.data_as(C.c_void_p), src, out.nbytes))
        return out

    def zero(self):
        check(lib.gcwt_device_memset(self.ptr, 0, self.nbytes))

    def free(self):
        if self.ptr:
            lib.gcwt_device_free(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceResult:
    """A transform's result left on the device: ``shape`` (C, S, N) rows of float32 (complex64 for complex
    output), ``pitch`` samples apart.  ``to_host`` brings over the whole result or any (scale, sample) range
    of it -- straight into page-locked memory at the link's rate (ghost_amd.hostmem), float64 widened on the
    device -- and ``buffer.ptr`` is the handle for whoever keeps working on the device."""

    def __init__(self, buffer, shape, pitch, complex_):
        self.buffer, self.shape, self.pitch, self.is_complex = buffer, tuple(int(v) for v in shape), int(pitch), bool(complex_)

    @property
    def nbytes(self):
        return self.shape[0] * self.shape[1] * self.pitch * (8 if self.is_complex else 4)

    def to_host(self, dtype=None, scales=None, start=0, stop=None):
        """ndarray (C, S', n): scales ``scales`` (slice or None = all), samples [start, stop) of every channel.
        dtype: float32 / float64 (complex64 / complex128 for complex results); default the device's."""
        from . import hostmem
        c, s, n = self.shape
        k = 2 if self.is_complex else 1
        narrow = np.complex64 if self.is_complex else np.float32
        wide = np.complex128 if self.is_complex else np.float64
        dtype = np.dtype(narrow if dtype is None else dtype)
        if dtype not in (np.dtype(narrow), np.dtype(wide)):
            raise ValueError("dtype must be %s or %s" % (np.dtype(narrow), np.dtype(wide)))
        if isinstance(scales, (int, np.integer)):
            scales = slice(int(scales), int(scales) + 1) if scales != -1 else slice(-1, None)
        sl = range(s)[slice(None) if scales is None else scales]
        if sl.step != 1 and len(sl) > 1:
            raise ValueError("scales must be a contiguous range")
        stop = n if stop is None else min(int(stop), n)
        start = max(0, int(start))
        cols = max(0, stop - start)
        out_shape = (c, len(sl), cols)
        out = hostmem.empty(out_shape, dtype)
        pinned = out is not None
        if not pinned:
            out = np.empty(out_shape, dtype=dtype)
        if out.size == 0:
            return out
        flags = (_lib.OUT_F64 if dtype == np.dtype(wide) else 0) | (_lib.HOST_PINNED if pinned else 0)
        esz = 4 * k
        if len(sl) == s:                       # all scales: the channels' rows follow each other
            groups = [(0, c * s, out.reshape(c * s, cols))]
        else:
            groups = [(ch * s + sl.start, len(sl), out[ch]) for ch in range(c)]
        for row0, n_rows, dst in groups:
            src = C.c_void_p(self.buffer.ptr.value + (row0 * self.pitch + start) * esz)
            check(lib.gcwt_rows_to_host(src, self.pitch * k, n_rows, cols * k, dst.ctypes.data_as(C.c_void_p),
                                        cols * k, flags))
        return out

    def free(self):
        if self.buffer is not None:
            self.buffer.free()
            self.buffer = None


_OUT_DTYPE = {_lib.OUT_AMPLITUDE: np.float32, _lib.OUT_POWER: np.float32,
              _lib.OUT_COMPLEX: np.complex64}
_OUT_MODES = {"amplitude": _lib.OUT_AMPLITUDE, "power": _lib.OUT_POWER,
              "complex": _lib.OUT_COMPLEX}


class CwtPlan:
    """One (n_channels, n_samples, frequencies, epochs) transform layout.

    Parameters mirror ``gcwt_params``; ``freqs_hz`` are the Morse peak
    frequencies in the order the output rows are wanted."""

    def __init__(self, n_samples, n_channels, fs, freqs_hz, *, gamma=3.0, beta=20.0,
                 epoch_bounds=None, output="amplitude", device=-1, band_eps=0.0, block=0,
                 max_fft_log2=0, normalization=None, order=0, precision=None, support_tol=0.0):
        self._handle = C.c_void_p()
        self.freqs = np.ascontiguousarray(freqs_hz, dtype=np.float64)
        if epoch_bounds is None:
            epoch_bounds = [[0, n_samples]]
        self.bounds = np.ascontiguousarray(epoch_bounds, dtype=np.int64).reshape(-1, 2)
        self.out_mode = _OUT_MODES[output] if isinstance(output, str) else int(output)
        p = _lib.Params()
        p.n_samples = int(n_samples)
        p.n_channels = int(n_channels)
        p.n_freqs = int(self.freqs.size)
        p.fs = float(fs)
        p.gamma = float(gamma)
        p.beta = float(beta)
        p.freqs_hz = self.freqs.ctypes.data_as(C.POINTER(C.c_double))
        p.n_epochs = int(self.bounds.shape[0])
        p.out_mode = self.out_mode
        p.epoch_bounds = self.bounds.ctypes.data_as(C.POINTER(C.c_int64))
        p.device = int(device)
        p.block = int(block)
        p.band_eps = float(band_eps)
        p.max_fft_log2 = int(max_fft_log2)
        # other members of the Morse family (morseutils.py:119-124, :181-196); transform()
        # itself always uses the first 'bandpass' wavelet (morse.py:84-91)
        if normalization not in (None, "bandpass", "energy"):
            raise ValueError("Normalization must be 'bandpass', or 'energy'")
        if not 0 <= int(order) <= 32:
            raise ValueError("order must be between 0 and 32")
        p.wavelet_flags = int(order) | (_lib.WAVELET_ENERGY if normalization == "energy" else 0)
        # 'auto' (default): float64 forward transform and per-level low cut, the reference's dynamic range (it
        # computes in float64: transforms.py:142-143), with the scales at risk recomputed exactly; 'high': the same
        # without the recomputation; 'fast': float32 throughout; 'exact': no decimated path
        if precision not in (None, "default", "auto", "fast", "high", "exact"):
            raise ValueError("precision must be 'auto', 'fast', 'high' or 'exact'")
        p.precision = {None: 0, "default": 0, "auto": 4, "fast": 1, "high": 2, "exact": 3}[precision]
        p.support_tol = float(support_tol)
        check(lib.gcwt_plan_create(C.byref(self._handle), C.byref(p)))
        self.n_samples, self.n_channels = int(n_samples), int(n_channels)
        self.n_freqs = int(self.freqs.size)
        self.out_shape = (self.n_channels, self.n_freqs, self.n_samples)
        self.out_dtype = _OUT_DTYPE[self.out_mode]

    # -- description ------------------------------------------------------
    @property
    def info(self):
        i = _lib.PlanInfo()
        check(lib.gcwt_plan_get_info(self._handle, C.byref(i)))
        return {k: getattr(i, k) for k, _ in _lib.PlanInfo._fields_}

    def scale_info(self):
        s = self.n_freqs
        method = np.zeros(s, np.int32)
        dec = np.zeros(s, np.int32)
        halo = np.zeros(s, np.int32)
        hop = np.zeros(s, np.int32)
        length = np.zeros(s, np.int64)
        i32p, i64p = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        check(lib.gcwt_plan_scale_info(self._handle, method.ctypes.data_as(i32p),
                                       dec.ctypes.data_as(i32p), halo.ctypes.data_as(i32p),
                                       hop.ctypes.data_as(i32p), length.ctypes.data_as(i64p)))
        theta_hi = np.zeros(s, np.float64)
        support = np.zeros(s, np.float64)
        n_bins = np.zeros(s, np.int32)
        f64p = C.POINTER(C.c_double)
        check(lib.gcwt_plan_scale_support(self._handle, theta_hi.ctypes.data_as(f64p),
                                          support.ctypes.data_as(f64p), n_bins.ctypes.data_as(i32p)))
        theta_neg = np.zeros(s, np.float64)
        check(lib.gcwt_debug_scale_theta_neg(self._handle, theta_neg.ctypes.data_as(f64p)))
        theta_lo = np.zeros(s, np.float64)
        check(lib.gcwt_debug_scale_theta_lo(self._handle, theta_lo.ctypes.data_as(f64p)))
        return {"method": method, "decimation": dec, "halo": halo, "hop": hop, "length": length,
                "theta_hi": theta_hi, "theta_neg": theta_neg, "theta_lo": theta_lo, "support": support,
                "n_bins": n_bins}

    # -- device -----------------------------------------------------------
    def upload(self):
        check(lib.gcwt_plan_upload(self._handle))

    def set_profiling(self, on=True):
        """True / 1: every stage between HIP events; 2: the synthesis kernels only (include/ghostcwt.h); False: none."""
        check(lib.gcwt_plan_set_profiling(self._handle, 2 if on == 2 and on is not True else (1 if on else 0)))

    def set_row_pitch(self, pitch_samples):
        """Row pitch (samples) of device output buffers; 0 = dense.  Use a multiple of 32
        when the row length is not one (see gcwt_plan_set_row_pitch)."""
        check(lib.gcwt_plan_set_row_pitch(self._handle, int(pitch_samples)))

    def precision_report(self):
        """After an execute with precision 'auto' (the default) or 'high': {"predicted": per scale, the loss to the
        float32 stages of its decimation level predicted from the recording's spectrum (relative to the scale's own
        output; 0 for scales on the exact paths), "worst", "rerouted": how many scales the last execute made again by
        the exact paths, "watched": False where the detector does not look -- plans whose segments take FFTs of 2^23 /
        2^24 points (kernels of millions of taps: DESIGN.md 8), every scale of which is then 'high''s}."""
        pred = np.zeros(self.n_freqs, np.float32)
        worst, n = C.c_float(0), C.c_int32(0)
        check(lib.gcwt_plan_precision_report(self._handle, pred.ctypes.data_as(C.POINTER(C.c_float)), C.byref(worst), C.byref(n)))
        if getattr(self, "_watched", None) is None:
            self._watched = all(p <= (1 << 22) for _, _, p in self.segments())
        return {"predicted": pred, "worst": float(worst.value), "rerouted": int(n.value), "watched": self._watched}

    def debug_precision_terms(self):
        """The two terms of the last execute's prediction (slot 0 of its last batch): float32 rounding of the level's
        stages, and what the level's slice of the spectrum leaves out; plus the energy each level's x_R held and the
        spectrum's band energies (sixteen bands per octave of the bin index) all of it is predicted from."""
        a, b = np.zeros(self.n_freqs, np.float32), np.zeros(self.n_freqs, np.float32)
        lv = np.zeros(max(1, self.info["n_levels"]), np.float32)
        bands = np.zeros(384, np.float32)
        f32p = C.POINTER(C.c_float)
        check(lib.gcwt_debug_precision_terms(self._handle, a.ctypes.data_as(f32p), b.ctypes.data_as(f32p), lv.ctypes.data_as(f32p),
                                             bands.ctypes.data_as(f32p)))
        return {"rounding": a, "left_out": b, "level_energy": lv, "band_energy": bands}

    def timings(self):
        t = _lib.Timings()
        check(lib.gcwt_get_timings(self._handle, C.byref(t)))
        return {k: getattr(t, k) for k, _ in _lib.Timings._fields_}

    def _host_result(self, shape, wide):
        """(array, flags) for a host result: float32/complex64, or the reference's
        float64/complex128 when ``wide`` (widened by the library while it copies)."""
        if not wide:
            return np.empty(shape, dtype=self.out_dtype), 0
        dt = np.complex128 if self.out_dtype == np.complex64 else np.float64
        return np.empty(shape, dtype=dt), _lib.OUT_F64

    def execute(self, x, wide=False):
        """x: array-like (C, N) -> ndarray out_shape (host in, host out)."""
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(self.n_channels, self.n_samples)
        out, flags = self._host_result(self.out_shape, wide)
        check(lib.gcwt_execute(self._handle, x.ctypes.data_as(C.c_void_p),
                               out.ctypes.data_as(C.c_void_p), flags))
        return out

    def execute_resident(self, x, result=None):
        """x: array-like (C, N) on the host; the result stays on the device: a DeviceResult (``result`` is reused
        when it is one of this shape).  Rows are padded to 32 samples (128-byte row starts: DESIGN.md 5)."""
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(self.n_channels, self.n_samples)
        self.upload()                           # (without a GPU this is where GCWT_ERR_NO_DEVICE is raised)
        pitch = (self.n_samples + 31) & ~31
        cplx = self.out_dtype == np.complex64
        if (result is None or result.buffer is None or result.shape != self.out_shape or result.pitch != pitch
                or result.is_complex != cplx):
            if result is not None:
                result.free()
            nbytes = self.n_channels * self.n_freqs * pitch * (8 if cplx else 4)
            result = DeviceResult(DeviceBuffer(nbytes), self.out_shape, pitch, cplx)
        self.set_row_pitch(pitch)
        try:
            check(lib.gcwt_execute(self._handle, x.ctypes.data_as(C.c_void_p), result.buffer.ptr, _lib.OUT_ON_DEVICE))
        finally:
            self.set_row_pitch(0)
        return result

    def execute_device(self, x_buf, out_buf):
        """Both buffers are DeviceBuffer (or raw c_void_p); returns when done."""
        xp = x_buf.ptr if isinstance(x_buf, DeviceBuffer) else x_buf
        op = out_buf.ptr if isinstance(out_buf, DeviceBuffer) else out_buf
        check(lib.gcwt_execute(self._handle, xp, op, _lib.X_ON_DEVICE | _lib.OUT_ON_DEVICE))

    def segments(self):
        """[(core_start, core_stop, fft_length)] of the time blocks the plan works in."""
        res = []
        for i in range(lib.gcwt_plan_segment_count(self._handle)):
            a, b, p = C.c_int64(), C.c_int64(), C.c_int64()
            check(lib.gcwt_plan_segment_info(self._handle, i, C.byref(a), C.byref(b), C.byref(p)))
            res.append((a.value, b.value, p.value))
        return res

    def execute_block(self, x, start, length, reuse_means=False, wide=False):
        """Samples [start, start+length) of every channel and scale from the whole
        recording x (C, N): ndarray (C, S, length).  Host in, host out."""
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(self.n_channels, self.n_samples)
        out, flags = self._host_result((self.n_channels, self.n_freqs, int(length)), wide)
        check(lib.gcwt_execute_block(self._handle, x.ctypes.data_as(C.c_void_p),
                                     out.ctypes.data_as(C.c_void_p), int(start), int(length),
                                     flags | (_lib.REUSE_MEANS if reuse_means else 0)))
        return out

    def execute_block_device(self, x_buf, out_buf, start, length, reuse_means=False):
        xp = x_buf.ptr if isinstance(x_buf, DeviceBuffer) else x_buf
        op = out_buf.ptr if isinstance(out_buf, DeviceBuffer) else out_buf
        flags = _lib.X_ON_DEVICE | _lib.OUT_ON_DEVICE | (_lib.REUSE_MEANS if reuse_means else 0)
        check(lib.gcwt_execute_block(self._handle, xp, op, int(start), int(length), flags))

    def filter_bank(self):
        b = self.info["block"]
        out = np.empty((self.n_freqs, b), dtype=np.complex64)
        check(lib.gcwt_filter_bank(self._handle, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def direct_kernel(self, scale):
        n = int(self.scale_info()["length"][scale])
        out = np.empty(n, dtype=np.complex64)
        check(lib.gcwt_direct_kernel(self._handle, int(scale),
                                     out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    # -- test hooks ---------------------------------------------------------
    def debug_levels(self, epoch=0):
        n = lib.gcwt_debug_level_count(self._handle)
        res = []
        for l in range(n):
            d, h, hp, nb = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
            m = C.c_int64()
            check(lib.gcwt_debug_level_info(self._handle, epoch, l, C.byref(d), C.byref(h),
                                            C.byref(hp), C.byref(nb), C.byref(m)))
            sh = C.c_int32()
            check(lib.gcwt_debug_level_band_shift(self._handle, l, C.byref(sh)))
            cut = C.c_double()
            check(lib.gcwt_debug_level_low_cut(self._handle, l, C.byref(cut)))
            res.append({"decimation": d.value, "halo": h.value, "hop": hp.value,
                        "nblk": nb.value, "m": m.value, "band_shift": sh.value, "low_cut": cut.value})
        return res

    def debug_mean_folded(self):
        """True when the last run summed the channels inside the forward column pass (include/ghostcwt_debug.h)."""
        return bool(lib.gcwt_debug_mean_folded(self._handle))

    def debug_graph_state(self):
        """1: executes replay a HIP graph, 0: not (yet), -1: capture failed, eager from then on."""
        return int(lib.gcwt_debug_graph_state(self._handle))

    def debug_blockconv(self):
        """Groups of the block-convolution scales: [{"scales": rows in order of kernel length, "hop", "back"}]."""
        i32p = C.POINTER(C.c_int32)
        n_bc = max(1, self.info["n_blockconv"])       # at most one group per scale
        arr = [np.zeros(n_bc, np.int32) for _ in range(4)]
        order = np.zeros(n_bc, np.int32)
        n = lib.gcwt_debug_blockconv_groups(self._handle, *[a.ctypes.data_as(i32p) for a in arr],
                                            order.ctypes.data_as(i32p), n_bc, n_bc)
        if n < 0:
            check(n)
        return [{"scales": order[arr[0][g]:arr[0][g] + arr[1][g]].tolist(), "hop": int(arr[2][g]), "back": int(arr[3][g])}
                for g in range(min(n, n_bc))]

    def debug_interp(self):
        """Per level: None when it is made by the FFT-per-sample kernels, else the interpolating
        synthesis' design -- q, factor I = R / q, alpha, err_bound, coef[2][I][8] -- and, under
        "demod", the demodulation bin of every scale of the plan (csrc/synthi.hip)."""
        n = lib.gcwt_debug_level_count(self._handle)
        f32p, i32p = C.POINTER(C.c_float), C.POINTER(C.c_int32)
        levels = []
        for l in range(n):
            q, fac, al, eb = C.c_int32(), C.c_int32(), C.c_double(), C.c_double()
            check(lib.gcwt_debug_interp_level(self._handle, l, C.byref(q), C.byref(fac), C.byref(al),
                                              C.byref(eb), None, 0))
            if q.value == 0:
                levels.append(None)
                continue
            coef = np.empty((2, fac.value, 8), dtype=np.float32)
            check(lib.gcwt_debug_interp_level(self._handle, l, None, None, None, None,
                                              coef.ctypes.data_as(f32p), coef.size))
            levels.append({"q": q.value, "factor": fac.value, "alpha": al.value,
                           "err_bound": eb.value, "coef": coef})
        demod = np.zeros(self.n_freqs, dtype=np.int32)
        check(lib.gcwt_debug_scale_demod(self._handle, demod.ctypes.data_as(i32p)))
        return {"levels": levels, "demod": demod}

    def debug_batches(self):
        """[(first_segment, count)] of the launch batches the plan's segments form."""
        res, seg, n = [], 0, lib.gcwt_plan_segment_count(self._handle)
        while seg < n:
            first, count = C.c_int32(), C.c_int32()
            check(lib.gcwt_debug_batch_of(self._handle, seg, C.byref(first), C.byref(count)))
            res.append((first.value, count.value))
            seg = first.value + max(1, count.value)
        return res

    def debug_exact_gain(self, scale, a, b):
        """G of one scale's reference kernel at theta = 2 pi a / b (host evaluation)."""
        a = np.ascontiguousarray(a, dtype=np.int64)
        out = np.empty(a.size, dtype=np.float64)
        check(lib.gcwt_debug_exact_gain(self._handle, int(scale), a.ctypes.data_as(C.POINTER(C.c_int64)),
                                        int(b), a.size, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def debug_fetch(self, what, channel=0, epoch=0, level=0):
        lv = self.debug_levels(epoch)
        if what == 0:
            n = self.info["fft_length"]
        elif what == 1:
            n = lv[level]["m"]
        else:
            n = lv[level]["nblk"] * self.info["block"]
        out = np.empty(n, dtype=np.complex64)
        check(lib.gcwt_debug_fetch(self._handle, what, channel, epoch, level,
                                   out.ctypes.data_as(C.POINTER(C.c_float)), n))
        return out

    def close(self):
        if self._handle:
            lib.gcwt_plan_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

"""ctypes binding of libghostcwt.so (include/ghostcwt.h).  No torch, no numpy C-API.

The library is the product: there is no CPU fallback.  Importing this module
without the built library raises; computing without a GPU raises GhostCwtError
(GCWT_ERR_NO_DEVICE).
"""
import ctypes as C
import os

__all__ = ["lib", "GhostCwtError", "check", "Params", "PlanInfo", "Timings", "LIB_PATH",
           "OUT_AMPLITUDE", "OUT_POWER", "OUT_COMPLEX", "X_ON_DEVICE", "OUT_ON_DEVICE", "OUT_F64", "HOST_PINNED",
           "SCALE_SPECTRAL", "SCALE_DIRECT", "SCALE_FULLBAND", "SCALE_BLOCKCONV", "ERR_INVALID", "ERR_UNSUPPORTED", "ERR_NO_DEVICE"]

LIB_PATH = os.environ.get("GHOSTCWT_LIB") or os.path.join(
    os.path.dirname(os.path.abspath(__file__)), "libghostcwt.so")

OUT_AMPLITUDE, OUT_POWER, OUT_COMPLEX = 0, 1, 2
X_ON_DEVICE, OUT_ON_DEVICE, REUSE_MEANS, OUT_F64, HOST_PINNED = 1, 2, 4, 8, 16
SCALE_SPECTRAL, SCALE_DIRECT, SCALE_FULLBAND, SCALE_BLOCKCONV = 0, 1, 2, 3
WAVELET_ENERGY = 0x100
ERR_INVALID, ERR_UNSUPPORTED, ERR_NO_DEVICE, ERR_HIP, ERR_NOMEM, ERR_COMM, ERR_COMM_INCOMPLETE = -1, -2, -3, -4, -5, -6, -7
COMM_ID_BYTES = 128


class GhostCwtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libghostcwt error %d: %s" % (code, message))
        self.code = code


class Params(C.Structure):
    _fields_ = [("n_samples", C.c_int64), ("n_channels", C.c_int32), ("n_freqs", C.c_int32),
                ("fs", C.c_double), ("gamma", C.c_double), ("beta", C.c_double),
                ("freqs_hz", C.POINTER(C.c_double)), ("n_epochs", C.c_int32),
                ("out_mode", C.c_int32), ("epoch_bounds", C.POINTER(C.c_int64)),
                ("device", C.c_int32), ("block", C.c_int32), ("band_eps", C.c_double),
                ("max_fft_log2", C.c_int32), ("wavelet_flags", C.c_int32),
                ("precision", C.c_int32), ("reserved0", C.c_int32), ("support_tol", C.c_double)]


class PlanInfo(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("n_levels", C.c_int32), ("n_spectral", C.c_int32),
                ("n_direct", C.c_int32), ("block", C.c_int32), ("max_decimation", C.c_int32),
                ("fft_length", C.c_int64), ("workspace_bytes", C.c_int64),
                ("out_bytes", C.c_int64), ("n_fullband", C.c_int32), ("n_interp", C.c_int32),
                ("n_blockconv", C.c_int32), ("reserved", C.c_int32)]


class Timings(C.Structure):
    _fields_ = [("mean_ms", C.c_float), ("fwd_fft_ms", C.c_float), ("decimate_ms", C.c_float),
                ("block_fft_ms", C.c_float), ("synth_ms", C.c_float), ("direct_ms", C.c_float),
                ("total_ms", C.c_float), ("synth_launches", C.c_int32), ("fullband_ms", C.c_float),
                ("interp_ms", C.c_float), ("blockconv_ms", C.c_float)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "ghost_amd: %s is missing. Build it with `make -C ghost_amd/csrc` (needs hipcc); "
            "there is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, i32p, i64p, f32p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_float)
    sig = {
        "gcwt_abi_version": (C.c_int, []),
        "gcwt_last_error": (C.c_char_p, []),
        "gcwt_device_count": (C.c_int, [C.POINTER(C.c_int)]),
        "gcwt_device_name": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
        "gcwt_device_pci_bus_id": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
        "gcwt_set_device": (C.c_int, [C.c_int]),
        "gcwt_current_device": (C.c_int, [C.POINTER(C.c_int)]),
        "gcwt_device_malloc": (C.c_int, [C.POINTER(vp), C.c_size_t]),
        "gcwt_device_free": (C.c_int, [vp]),
        "gcwt_memcpy_h2d": (C.c_int, [vp, vp, C.c_size_t]),
        "gcwt_memcpy_d2h": (C.c_int, [vp, vp, C.c_size_t]),
        "gcwt_device_memset": (C.c_int, [vp, C.c_int, C.c_size_t]),
        "gcwt_device_synchronize": (C.c_int, []),
        "gcwt_device_memory": (C.c_int, [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
        "gcwt_host_alloc": (C.c_int, [C.POINTER(vp), C.c_size_t]),
        "gcwt_host_free": (C.c_int, [vp]),
        "gcwt_rows_to_host": (C.c_int, [vp, C.c_int64, C.c_int64, C.c_int64, vp, C.c_int64, C.c_int]),
        "gcwt_plan_create": (C.c_int, [C.POINTER(vp), C.POINTER(Params)]),
        "gcwt_plan_destroy": (None, [vp]),
        "gcwt_plan_get_info": (C.c_int, [vp, C.POINTER(PlanInfo)]),
        "gcwt_plan_scale_info": (C.c_int, [vp, i32p, i32p, i32p, i32p, i64p]),
        "gcwt_plan_scale_support": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), i32p]),
        "gcwt_plan_set_profiling": (C.c_int, [vp, C.c_int]),
        "gcwt_plan_precision_report": (C.c_int, [vp, f32p, f32p, i32p]),
        "gcwt_plan_set_row_pitch": (C.c_int, [vp, C.c_int64]),
        "gcwt_plan_upload": (C.c_int, [vp]),
        "gcwt_execute": (C.c_int, [vp, vp, vp, C.c_int]),
        "gcwt_execute_block": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int64, C.c_int]),
        "gcwt_plan_segment_count": (C.c_int, [vp]),
        "gcwt_plan_segment_info": (C.c_int, [vp, C.c_int, i64p, i64p, i64p]),
        "gcwt_filter_bank": (C.c_int, [vp, f32p]),
        "gcwt_direct_kernel": (C.c_int, [vp, C.c_int, f32p]),
        "gcwt_get_timings": (C.c_int, [vp, C.POINTER(Timings)]),
        "gcwt_comm_unique_id": (C.c_int, [vp]),
        "gcwt_comm_create": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, vp]),
        "gcwt_comm_destroy": (None, [vp]),
        "gcwt_comm_abort": (None, [vp]),
        "gcwt_comm_barrier": (C.c_int, [vp]),
        "gcwt_comm_allreduce_max": (C.c_int, [vp, C.POINTER(C.c_double)]),
        "gcwt_comm_broadcast_bank": (C.c_int, [vp, vp, C.c_int]),
        # test-only hooks (include/ghostcwt_debug.h)
        "gcwt_debug_level_count": (C.c_int, [vp]),
        "gcwt_debug_level_band_shift": (C.c_int, [vp, C.c_int, i32p]),
        "gcwt_debug_set_option": (C.c_int, [C.c_char_p, C.c_int64, C.c_int]),
        "gcwt_debug_level_low_cut": (C.c_int, [vp, C.c_int, C.POINTER(C.c_double)]),
        "gcwt_debug_scale_theta_lo": (C.c_int, [vp, C.POINTER(C.c_double)]),
        "gcwt_debug_graph_state": (C.c_int, [vp]),
        "gcwt_debug_mean_folded": (C.c_int, [vp]),
        "gcwt_debug_precision_terms": (C.c_int, [vp, f32p, f32p, f32p, f32p]),
        "gcwt_debug_blockconv_groups": (C.c_int, [vp] + [C.POINTER(C.c_int32)] * 5 + [C.c_int, C.c_int]),
        "gcwt_debug_batch_of": (C.c_int, [vp, C.c_int, i32p, i32p]),
        "gcwt_debug_level_info": (C.c_int, [vp, C.c_int, C.c_int, i32p, i32p, i32p, i32p, i64p]),
        "gcwt_debug_exact_gain": (C.c_int, [vp, C.c_int, i64p, C.c_int64, C.c_int64, C.POINTER(C.c_double)]),
        "gcwt_debug_measure_build": (C.c_int, []),
        "gcwt_debug_interp_level": (C.c_int, [vp, C.c_int, i32p, i32p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                            f32p, C.c_int64]),
        "gcwt_debug_scale_demod": (C.c_int, [vp, i32p]),
        "gcwt_debug_scale_theta_neg": (C.c_int, [vp, C.POINTER(C.c_double)]),
        "gcwt_debug_clock": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "gcwt_debug_check_output": (C.c_int, [vp, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, i64p, i64p]),
        "gcwt_debug_bandwidth": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(C.c_double)]),
        "gcwt_debug_fetch": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, f32p, C.c_int64]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(code):
    if code != 0:
        raise GhostCwtError(code, lib.gcwt_last_error().decode("utf-8", "replace"))

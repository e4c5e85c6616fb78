"""Where a k_synth7 workgroup's life goes, per level R = 2, 4, 8 of the headline grid (measure build:
`make -C ghost_amd/csrc measure`): mean microseconds from a workgroup's start to its marks -- tables and
samples parked in LDS, block spectra exchanged, spectra done, scale loop starts, end.
  GHOSTCWT_LIB=$PWD/ghost_amd/libghostcwt_measure.so GHOSTCWT_CLOCK_PROBE=1 GHOSTCWT_CLOCK_PHASES=1 \
      python tools/wg_phases.py          (+ GHOSTCWT_SYNTH_DROP_STORES=1: the same without HBM writes)
The probe's own waits and atomics slow the kernel (R = 2: 1.4 -> 1.9 ms); read the marks relative to
each other."""
import sys, os, ctypes; sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
import numpy as np
from ghost_amd.engine import CwtPlan, DeviceBuffer
from _opts import apply_env_options; apply_env_options()
from ghost_amd.synthetic import lfp
from ghost_amd._lib import lib, check
fs, N, C = 1000.0, 1000000, 128
f_all = np.geomspace(200.0, 2.0, 100)
dec = CwtPlan(N, C, fs, f_all).scale_info()["decimation"]
x = lfp(4, N); x = np.tile(x, (C // 4 + 1, 1))[:C]
xb = DeviceBuffer(x.nbytes); xb.upload(x)
for R in (2, 4, 8):
    plan = CwtPlan(N, C, fs, f_all[dec == R]); plan.set_profiling(True)
    ob = DeviceBuffer(plan.info["out_bytes"])
    g = ctypes.c_double(); w = ctypes.c_double()
    for i in range(3): plan.execute_device(xb, ob)
    os.environ.pop("GHOSTCWT_CLOCK_PHASES", None)          # (the warm-up runs are read without printing)
    check(lib.gcwt_debug_clock(plan._handle, ctypes.byref(g), ctypes.byref(w)))
    os.environ["GHOSTCWT_CLOCK_PHASES"] = "1"
    ts = []
    for i in range(5): plan.execute_device(xb, ob); ts.append(plan.timings()["synth_ms"])
    sys.stderr.write("R=%d synth %.3f ms: " % (R, min(ts))); sys.stderr.flush()
    check(lib.gcwt_debug_clock(plan._handle, ctypes.byref(g), ctypes.byref(w)))

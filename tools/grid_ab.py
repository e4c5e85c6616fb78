"""k_synthi's grid order on the headline workload: items fastest (0) against channels fastest (1), alternating
in one process (same buffers), timings from the plan's own HIP events.  GRID_AB_ROUNDS, GRID_AB_ALLOC=n extra
allocations before the result buffer (moves the placement of the 51 GB)."""
import os, sys; sys.path.insert(0, '.')
import numpy as np
from ghost_amd.engine import CwtPlan, DeviceBuffer, set_option
from ghost_amd.synthetic import lfp
C, N, fs = 128, 1000000, 1000.0
f = np.geomspace(200.0, 2.0, 100)
base = lfp(4, N, fs)
xb = DeviceBuffer(4 * C * N)
for c in range(C):
    xb.upload(base[c % 4], offset_bytes=4 * c * N)
pad = [DeviceBuffer(int(os.environ.get("GRID_AB_PAD_MB", "0")) << 20)] if os.environ.get("GRID_AB_PAD_MB") else []
out = DeviceBuffer(4 * C * 100 * N)
print("out at 0x%x" % out.ptr.value, flush=True)
plans = {}
for g in (0, 1):
    set_option("interp_grid", g)
    p = CwtPlan(N, C, fs, f)
    p.set_profiling(True)
    for _ in range(3):
        p.execute_device(xb, out)
    plans[g] = p
set_option("interp_grid", None)
for rnd in range(int(os.environ.get("GRID_AB_ROUNDS", "4"))):
    for g in (0, 1):
        ts = []
        for _ in range(6):
            plans[g].execute_device(xb, out)
            ts.append(plans[g].timings())
        print("grid %d: k_synthi %.3f ms  synth %.3f  total %.3f" % (
            g, np.median([t["interp_ms"] for t in ts]), np.median([t["synth_ms"] for t in ts]),
            np.median([t["total_ms"] for t in ts])), flush=True)

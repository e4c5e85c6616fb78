import sys, os; sys.path.insert(0, '.')
import numpy as np
from ghost_amd.engine import CwtPlan, DeviceBuffer
from ghost_amd.synthetic import lfp
fs = 30000.0; N = 3700000; C = 16
x = lfp(2, N, fs); x = np.tile(x, (C // 2 + 1, 1))[:C]
xb = DeviceBuffer(x.nbytes); xb.upload(x)
for name, f in [("200 scales 500-1", np.geomspace(500, 1, 200)), ("100 scales 500-16", np.geomspace(500, 16, 100)),
                ("100 scales 15-1", np.geomspace(15, 1, 100)), ("22 scales 2-1", np.geomspace(2, 1.02, 22))]:
    plan = CwtPlan(N, C, fs, f); plan.set_profiling(True)
    ob = DeviceBuffer(plan.info["out_bytes"])
    ts = []
    for i in range(3):
        plan.execute_device(xb, ob); ts.append(plan.timings()["synth_ms"])
    t = min(ts); S = len(f)
    si = plan.scale_info()
    print("%-20s segs %d R %s: synth %.2f ms -> %.2f ps/output, %.0f GB/s" %
          (name, len(plan.segments()), sorted(set(si["decimation"])), t, t * 1e9 / (C * N * S), C * N * S * 4 / t / 1e6))
    plan.close(); ob.free()

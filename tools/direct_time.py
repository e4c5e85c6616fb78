"""Time of the time-domain (direct) scales: the top of a grid that reaches 0.39 fs, 128 ch x 1e6."""
import sys, os; sys.path.insert(0, '.')
import numpy as np
from ghost_amd.engine import CwtPlan, DeviceBuffer
from ghost_amd.synthetic import lfp
fs = 1000.; N = 1000000; C = int(os.environ.get("QB_C", "128"))
f = np.geomspace(391.9, 2.0, 100)
plan = CwtPlan(N, C, fs, f); plan.set_profiling(True)
si = plan.scale_info()
nd = int((si["method"] == 1).sum())
x = lfp(4, N); x = np.tile(x, (C // 4 + 1, 1))[:C]
xb = DeviceBuffer(x.nbytes); xb.upload(x)
ob = DeviceBuffer(plan.info['out_bytes'])
best = None
for it in range(5):
    plan.execute_device(xb, ob); tm = plan.timings()
    best = tm if best is None or tm["direct_ms"] < best["direct_ms"] else best
print("direct scales %d (L %s): direct %.3f ms, synth %.3f ms, total %.3f ms -> %.2f ps per direct output, %.0f GB/s of direct rows" %
      (nd, si["length"][si["method"] == 1].tolist(), best["direct_ms"], best["synth_ms"], best["total_ms"],
       best["direct_ms"] * 1e9 / (C * N * max(nd, 1)), C * N * nd * 4 / max(best["direct_ms"], 1e-9) / 1e6))

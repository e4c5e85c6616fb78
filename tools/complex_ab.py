"""Complex-output mode of the headline workload (128 ch x 1e6 x 100 scales, 102.9 GB per step):
every execute timed one by one from the very first on a FRESH result buffer -- first touch of its
pages against steady state -- for the library in GHOSTCWT_LIB (store-policy builds)."""
import sys, os, time; sys.path.insert(0, '.')
import numpy as np
from ghost_amd.engine import CwtPlan, DeviceBuffer
from ghost_amd.synthetic import lfp
fs, N, S, C = 1000.0, 1000000, 100, 128
plan = CwtPlan(N, C, fs, np.geomspace(200, 2, S), output="complex"); plan.set_profiling(True)
x = lfp(4, N); x = np.tile(x, (C // 4 + 1, 1))[:C]
xb = DeviceBuffer(x.nbytes); xb.upload(x)
ob = DeviceBuffer(plan.info["out_bytes"])
syn, tot = [], []
for it in range(int(os.environ.get("CX_REPS", "12"))):
    t0 = time.perf_counter(); plan.execute_device(xb, ob); tot.append((time.perf_counter() - t0) * 1e3)
    syn.append(plan.timings()["synth_ms"])
print(os.environ.get("QB_TAG", ""), "synth ms: first %.2f second %.2f | steady min %.2f med %.2f max %.2f | wall med %.2f" %
      (syn[0], syn[1], min(syn[2:]), float(np.median(syn[2:])), max(syn[2:]), float(np.median(tot[2:]))))

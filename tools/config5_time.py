"""Stage times of BASELINE config 5 per GPU (48 ch x 18e6 @ 30 kHz x 200 scales, streamed) as
bench.py --config 5 runs it, without the checks: C5_REPS passes, min / median."""
import sys, os, ctypes; sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
import numpy as np
from ghost_amd.engine import CwtPlan, DeviceBuffer
from _opts import apply_env_options; apply_env_options()
from ghost_amd.synthetic import lfp
fs, N, S = 30000.0, int(os.environ.get("C5_N", "18000000")), 200
C, group = int(os.environ.get("C5_C", "48")), int(os.environ.get("C5_GROUP", "24"))
reps = int(os.environ.get("C5_REPS", "4"))
f = np.geomspace(500.0, 1.0, S)
if os.environ.get("C5_LEVEL"):          # only the scales of one decimation level (per-level rates)
    dec = CwtPlan(N, group, fs, f).scale_info()["decimation"]
    f = f[dec == int(os.environ["C5_LEVEL"])]; S = f.size
plan = CwtPlan(N, group, fs, f, output=os.environ.get("C5_OUT", "amplitude")); plan.set_profiling(True)
segs = plan.segments()
x = lfp(2, N, fs, seed=1234)
xb = DeviceBuffer(4 * C * N)
for c in range(C):
    xb.upload(x[c % 2], offset_bytes=4 * c * N)
core = max(b - a for a, b, _ in segs)
ring = [DeviceBuffer(4 * group * S * core) for _ in range(2)]
tot = []
for it in range(reps + 1):
    acc = {}
    k = 0
    for g in range(C // group):
        xg = ctypes.c_void_p(xb.ptr.value + 4 * g * group * N)
        for i, (a, b, _) in enumerate(segs):
            plan.execute_block_device(xg, ring[k & 1], a, b - a, reuse_means=i > 0); k += 1
            for key, v in plan.timings().items():
                acc[key] = acc.get(key, 0) + v
    if it:
        tot.append(acc)
med = {k: round(float(np.median([t[k] for t in tot])), 3) for k in tot[0] if k.endswith("_ms")}
print(os.environ.get("QB_TAG", ""), "%d scales" % S, "%.0f GB/s of rows |" % (C * N * S * 4 / max(1e-9, med["synth_ms"]) / 1e6), "config 5 per step: total min %.2f med %.2f ms |" % (min(t["total_ms"] for t in tot), med["total_ms"]), med)

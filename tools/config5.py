"""BASELINE config 5 regime at a size one GPU holds: 30 kHz, 200 scales 1-500 Hz,
streamed block by block with gcwt_execute_block (device-resident ring of one block)."""
import sys, time, os; sys.path.insert(0, '.')
import numpy as np
from ghost_amd.engine import CwtPlan, DeviceBuffer
from ghost_amd.synthetic import lfp
fs = 30000.0
C = int(os.environ.get("C5_C", "32")); N = int(os.environ.get("C5_N", str(9000000)))   # 5 min
S = 200
f = np.geomspace(500.0, 1.0, S)
plan = CwtPlan(N, C, fs, f)
plan.set_profiling(True)
segs = plan.segments()
si = plan.scale_info()
print("segments", len(segs), "fft", segs[0][2], "levels", plan.info["n_levels"], "max R", si["decimation"].max(),
      "halos", sorted(set(si["halo"])), "workspace GB", plan.info["workspace_bytes"] / 1e9)
x = lfp(2, N, fs); x = np.tile(x, (C // 2 + 1, 1))[:C]
xb = DeviceBuffer(x.nbytes); xb.upload(x)
core = max(b - a for a, b, _ in segs)
ob = DeviceBuffer(C * S * core * 4)
t0 = time.time()
tot_synth = 0.0
for i, (a, b, _) in enumerate(segs):
    plan.execute_block_device(xb, ob, a, b - a, reuse_means=i > 0)
    tm = plan.timings(); tot_synth += tm["synth_ms"]
    if i < 2: print({k: round(v, 2) for k, v in tm.items()})
dt = time.time() - t0
print("streamed %d blocks in %.3f s -> %.1f Msamples/s (%.1f GB/s out), synth %.1f ms" %
      (len(segs), dt, C * N / dt / 1e6, C * N * S * 4 / dt / 1e9, tot_synth))

"""Synthesis throughput per decimation level: 16 scales inside one level each."""
import sys, os; sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
import numpy as np
from ghost_amd.engine import CwtPlan, DeviceBuffer
from _opts import apply_env_options; apply_env_options()
from ghost_amd.synthetic import lfp
fs = float(os.environ.get("LB_FS", "1000")); N = int(os.environ.get("LB_N", "1000000")); C = int(os.environ.get("LB_C", "64"))
x = lfp(2, N, fs); x = np.tile(x, (C // 2 + 1, 1))[:C]
xb = DeviceBuffer(x.nbytes); xb.upload(x)
for R in [int(v) for v in os.environ.get("LB_R", "2,4,16,32,64,256").split(",")]:
    fhi = fs / (1.805 * R) * 0.999; flo = fhi / 2 * 1.03
    f = np.geomspace(fhi, flo, 16)
    plan = CwtPlan(N, C, fs, f); plan.set_profiling(True)
    assert set(plan.scale_info()["decimation"]) == {R}, (R, set(plan.scale_info()["decimation"]))
    ob = DeviceBuffer(plan.info["out_bytes"])
    ts = []
    for i in range(5):
        plan.execute_device(xb, ob); ts.append(plan.timings()["synth_ms"])
    t = min(ts)
    print("R %6d halo %d hop %d: synth %.3f ms -> %.2f ps/output, %.0f GB/s" %
          (R, plan.scale_info()["halo"][0], plan.scale_info()["hop"][0], t, t * 1e9 / (C * N * 16), C * N * 16 * 4 / t / 1e6))
    plan.close(); ob.free()

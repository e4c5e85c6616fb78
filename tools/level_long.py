"""k_synth7 vs k_synth8 in steady state: many scales inside ONE decimation level, so that a
workgroup's prologue (block spectra, twiddles) is amortised over a long scale walk.
LL_S scales (default 96), LL_R levels."""
import sys, os; sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
import numpy as np
from ghost_amd.engine import CwtPlan, DeviceBuffer
from _opts import apply_env_options; apply_env_options()
from ghost_amd.synthetic import lfp
fs = 1000.0; N = 1000000; C = int(os.environ.get("LL_C", "32")); S = int(os.environ.get("LL_S", "96"))
x = lfp(2, N, fs); x = np.tile(x, (C // 2 + 1, 1))[:C]
xb = DeviceBuffer(x.nbytes); xb.upload(x)
for R in [int(v) for v in os.environ.get("LL_R", "4,32,128").split(",")]:
    fhi = fs / (1.72 * R) * 0.999; flo = fhi / 2 * 1.06
    f = np.geomspace(fhi, flo, S)
    plan = CwtPlan(N, C, fs, f); plan.set_profiling(True)
    si = plan.scale_info()
    assert set(si["decimation"]) == {R}, (R, set(si["decimation"]))
    ob = DeviceBuffer(plan.info["out_bytes"])
    ts = []
    for i in range(4):
        plan.execute_device(xb, ob); ts.append(plan.timings()["synth_ms"])
    t = min(ts)
    print("kernel %s R %4d halo %d hop %d S %d: synth %.3f ms -> %.2f ps/output, %.0f GB/s" %
          (os.environ.get("GHOSTCWT_SYNTH_KERNEL", "8"), R, si["halo"][0], si["hop"][0], S, t,
           t * 1e9 / (C * N * S), C * N * S * 4 / t / 1e6))
    plan.close(); ob.free()

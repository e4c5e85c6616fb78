import sys, time; sys.path.insert(0,'.')
import numpy as np
from ghost_amd.engine import CwtPlan, DeviceBuffer
from ghost_amd.synthetic import lfp
fs, N, C, S = 30000.0, 18000000, 8, 200
f = np.geomspace(500.0, 1.0, S)
for prec in ("high", "exact"):
    plan = CwtPlan(N, C, fs, f, precision=prec); plan.set_profiling(True)
    info = plan.info
    segs = plan.segments()
    a, b, _ = segs[len(segs)//2]
    x = lfp(2, N, fs); x = np.tile(x, (C // 2, 1))
    xb = DeviceBuffer(x.nbytes); xb.upload(x)
    ob = DeviceBuffer(4 * C * S * (b - a))
    for it in range(2):
        t0 = time.perf_counter(); plan.execute_block_device(xb, ob, a, b - a); el = time.perf_counter() - t0
    tm = plan.timings()
    print(prec, "scales", info['n_spectral'], info['n_direct'], info['n_blockconv'], info['n_fullband'], "block of %d samples x %d ch: %.1f ms -> %.0f Msamples/s" % (b - a, C, el * 1e3, C * (b - a) / el / 1e6), {k: round(v, 1) for k, v in tm.items() if k.endswith('_ms') and v > 0.05})
    xb.free(); ob.free(); plan.close()

"""Wall time per device-resident execute against the device time of its stages: what the host adds between two steps
(the bench's timed loop is back-to-back executes).  SG_PROFILING=0/1, QB_PRECISION."""
import sys, time, os; sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import numpy as np
from ghost_amd.engine import CwtPlan, DeviceBuffer
from ghost_amd._lib import lib, check
def device_synchronize(): check(lib.gcwt_device_synchronize())
from _opts import apply_env_options; apply_env_options()
from ghost_amd.synthetic import lfp
fs = 1000.; N = 1000000; S = 100; C = 128
f = np.geomspace(200, 2, S)
x = lfp(4, N); x = np.tile(x, (C // 4 + 1, 1))[:C]
xb = DeviceBuffer(x.nbytes); xb.upload(x)
for prec in ("auto", "high", "fast"):
    for prof in (1, 0):
        plan = CwtPlan(N, C, fs, f, precision=prec)
        plan.set_profiling(bool(prof))
        ob = DeviceBuffer(plan.info['out_bytes'])
        for _ in range(3):
            plan.execute_device(xb, ob)
        device_synchronize()
        K = 20
        t0 = time.perf_counter(); tot = 0.0
        for _ in range(K):
            plan.execute_device(xb, ob)
            if prof:
                tot += plan.timings()['total_ms']
        device_synchronize()
        wall = (time.perf_counter() - t0) / K * 1e3
        print("precision %-5s profiling %d: wall %.3f ms per step%s" % (prec, prof, wall, ", device stages %.3f, host gap %.3f" % (tot / K, wall - tot / K) if prof else ""))
        plan.close(); ob.free()

"""Latency of BASELINE config 1 (1 ch x 16 384 samples x 32 scales) device-resident, eager against the HIP graph that
execute_range captures for small plans (option graphs = 0 / 1), alternating on one box."""
import sys, time; sys.path.insert(0,'.')
import numpy as np
from ghost_amd.engine import CwtPlan, DeviceBuffer, set_option
from ghost_amd.synthetic import lfp_channel
from ghost_amd._lib import lib, check
fs, N = 1000.0, 16384
f1 = 200.0 / 2.0 ** (np.arange(32) / 6.0)
x = lfp_channel(N, fs, channel=0, seed=99)
for g in (0, 1, 0, 1):
    set_option("graphs", g)
    plan = CwtPlan(N, 1, fs, f1, output="amplitude"); plan.upload()
    xb, ob = DeviceBuffer(4 * N), DeviceBuffer(plan.info["out_bytes"]); xb.upload(x)
    for _ in range(10): plan.execute_device(xb, ob)
    w = []
    for _ in range(300):
        t0 = time.perf_counter(); plan.execute_device(xb, ob); w.append(time.perf_counter() - t0)
    print("graphs", g, "state", plan.debug_graph_state(), "median %.1f us  min %.1f us" % (np.median(w) * 1e6, np.min(w) * 1e6))

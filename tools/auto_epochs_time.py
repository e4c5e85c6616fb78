"""precision='auto' on a recording cut into epochs, a mains line on all of it: how many different scale sets the
segments' verdicts ask for, and what the first / later executes cost (sub-plans are made on first use, four kept)."""
import sys, time; sys.path.insert(0, '.')
import numpy as np
from ghost_amd.engine import CwtPlan, DeviceBuffer
from ghost_amd.synthetic import lfp_channel
fs, n, C = 1000.0, 1200000, int(sys.argv[1]) if len(sys.argv) > 1 else 4
n_ep = int(sys.argv[2]) if len(sys.argv) > 2 else 12
f = np.geomspace(200.0, 2.0, 100)
t = np.arange(n) / fs
rng = np.random.default_rng(3)
x = np.stack([lfp_channel(n, fs, 20 + c) for c in range(C)]).astype(np.float64)
import os
if os.environ.get("AE_VARY"):       # the line's strength differs from epoch to epoch: 3 .. 300 x the spread (other verdicts per epoch)
    amp = np.zeros(n)
    for k in range(n_ep):
        amp[n * k // n_ep: n * (k + 1) // n_ep] = 3.0 * 100.0 ** (rng.random())
else:
    amp = 100.0 * (1.0 + 0.5 * np.sin(2 * np.pi * t / 97.0))
x += (x.std() * amp * np.sin(2 * np.pi * 60.0 * t))[None]
x = x.astype(np.float32)
edges = np.linspace(0, n, n_ep + 1).astype(int)
eb = [[int(a) + 5, int(b)] for a, b in zip(edges[:-1], edges[1:])]
for prec in ("high", "auto"):
    p = CwtPlan(n, C, fs, f, epoch_bounds=eb, precision=prec)
    ts = []
    xb = DeviceBuffer(x.nbytes); xb.upload(x)
    ob = DeviceBuffer(p.info['out_bytes'])
    for it in range(5):
        t0 = time.perf_counter(); p.execute_device(xb, ob); ts.append(time.perf_counter() - t0)
    rep = p.precision_report()
    print(prec, "executes ms:", [round(1e3 * v, 1) for v in ts], "rerouted", rep["rerouted"], "worst %.2e" % rep["worst"])
    p.close()

"""Quick A/B of the headline step's stages for precision = fast / high (device-resident, profiling on)."""
import sys, time; sys.path.insert(0, '.')
import numpy as np
from ghost_amd.engine import CwtPlan, DeviceBuffer
from ghost_amd.synthetic import lfp
C, N, fs = 128, 1000000, 1000.0
f = np.geomspace(200.0, 2.0, 100)
base = lfp(4, N, fs)
xb = DeviceBuffer(4 * C * N)
for c in range(C):
    xb.upload(base[c % 4], offset_bytes=4 * c * N)
out = DeviceBuffer(4 * C * 100 * N)
for prec in ("fast", "high", "fast", "high"):
    p = CwtPlan(N, C, fs, f, precision=prec)
    p.set_profiling(True)
    for _ in range(3):
        p.execute_device(xb, out)
    ts = []
    for _ in range(8):
        p.execute_device(xb, out)
        ts.append(p.timings())
    med = {k: float(np.median([t[k] for t in ts])) for k in ts[0]}
    print(prec, " ".join("%s %.3f" % (k, v) for k, v in med.items() if isinstance(v, float) and v > 0), flush=True)
    p.close()

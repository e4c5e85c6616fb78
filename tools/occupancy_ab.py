"""k_synthi's launch time against its occupancy (measure build: GHOSTCWT_LIB=ghost_amd/libghostcwt_measure.so): unused LDS
on top of the kernel's 42 KB leaves 3 / 2 / 1 workgroups per CU."""
import sys; sys.path.insert(0, '.')
import numpy as np
from ghost_amd.engine import CwtPlan, DeviceBuffer, set_option
from ghost_amd.synthetic import lfp
C, N, fs = 128, 1000000, 1000.0
f = np.geomspace(200.0, 2.0, 100)
base = lfp(4, N, fs)
xb = DeviceBuffer(4 * C * N)
for c in range(C):
    xb.upload(base[c % 4], offset_bytes=4 * c * N)
out = DeviceBuffer(4 * C * 100 * N)
p = CwtPlan(N, C, fs, f); p.set_profiling(True)
for pad, wgs in ((0, 3), (30, 2), (100, 1), (0, 3), (30, 2)):
    set_option("synthi_pad_kb", pad)
    for _ in range(3): p.execute_device(xb, out)
    ts = []
    for _ in range(8):
        p.execute_device(xb, out); ts.append(p.timings())
    print("k_synthi with %3d KB of padding (%d workgroups = %2d waves per CU): %.3f ms" % (pad, wgs, 4 * wgs, np.median([t["interp_ms"] for t in ts])), flush=True)

"""Where the public call's time goes (config 2: 1 ch x 1e6 x 100 scales): plan, upload, execute (host in / host out)."""
import sys, time; sys.path.insert(0, '.')
import numpy as np
from ghost_amd.engine import CwtPlan
from ghost_amd.synthetic import lfp_channel
from ghost_amd.wave import ContinuousWaveletTransform
fs, N = 1000.0, 1000000
f = np.geomspace(200.0, 2.0, 100)
x = lfp_channel(N, fs)
for rep in range(3):
    t0 = time.perf_counter(); p = CwtPlan(N, 1, fs, f); t1 = time.perf_counter(); p.upload(); t2 = time.perf_counter()
    out = p.execute(x[None]); t3 = time.perf_counter(); out2 = p.execute(x[None]); t4 = time.perf_counter()
    o64 = p.execute(x[None], wide=True); t5 = time.perf_counter(); p.close(); t6 = time.perf_counter()
    print("plan %.1f upload %.1f execute(first) %.1f execute(second) %.1f execute(f64) %.1f close %.1f ms" % tuple(
        1e3 * (b - a) for a, b in ((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5), (t5, t6))), flush=True)
    t0 = time.perf_counter(); e = np.empty((1, 100, N), np.float32); e[:] = 0; print("  np.empty + touch 400 MB: %.1f ms" % (1e3 * (time.perf_counter() - t0)))
cwt = ContinuousWaveletTransform()
for rep in range(3):
    t0 = time.perf_counter(); cwt.transform(x, fs=fs, freqs=f[::-1].copy(), dtype=np.float32); print("transform f32 %.1f ms" % (1e3 * (time.perf_counter() - t0)))

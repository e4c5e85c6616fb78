"""float64 results on the host: widened on the device and sent as float64 (production) against float32 over the link
and widened by host threads into the page-locked destination (options host_widen, host_threads)."""
import sys, time; sys.path.insert(0, '.')
import numpy as np
from ghost_amd.wave import ContinuousWaveletTransform
from ghost_amd.engine import set_option
from ghost_amd.synthetic import lfp_channel
fs, N = 1000., 1000000
x = lfp_channel(N, fs, channel=0, seed=4321)
f = np.geomspace(2, 200, 100)
cwt = ContinuousWaveletTransform()
ts = []
for i in range(5):
    t0 = time.perf_counter(); cwt.transform(x, fs=fs, freqs=f.copy(), dtype=np.float32); a = cwt.amplitude; ts.append(time.perf_counter() - t0); del a
print("float32 result: %s ms -> link about %.1f GB/s" % (" ".join("%.1f" % (1e3 * t) for t in ts[1:]), 0.4 / (min(ts) - 0.0016)))
for rnd in range(2):
    for tag, opts in (("device widen", {"host_widen": 0}), ("host widen 8 thr", {"host_widen": 1, "host_threads": 8}),
                      ("host widen 16 thr", {"host_widen": 1, "host_threads": 16}), ("host widen 4 thr", {"host_widen": 1, "host_threads": 4})):
        for k, v in opts.items(): set_option(k, v)
        cwt = ContinuousWaveletTransform()
        ts = []
        for i in range(6):
            t0 = time.perf_counter(); cwt.transform(x, fs=fs, freqs=f.copy()); a = cwt.amplitude; ts.append(time.perf_counter() - t0)
            chk = float(a[50, 123456]); del a
        print("%-20s transform + float64 amplitude ms: %s  (a[50,123456] = %.9g)" % (tag, " ".join("%.1f" % (1e3 * t) for t in ts[1:]), chk))

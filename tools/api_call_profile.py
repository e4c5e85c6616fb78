"""Where the 1.6 ms of a repeated transform() go (config 2, result left on the device): cProfile of 200 calls."""
import sys, time, cProfile, pstats; sys.path.insert(0, '.')
import numpy as np
from ghost_amd.synthetic import lfp_channel
from ghost_amd.wave import ContinuousWaveletTransform
fs, N = 1000.0, 1000000
f = np.geomspace(2.0, 200.0, 100)
x = lfp_channel(N, fs).astype(np.float32)
cwt = ContinuousWaveletTransform()
for _ in range(3): cwt.transform(x, fs=fs, freqs=f.copy())
t0 = time.perf_counter()
for _ in range(200): cwt.transform(x, fs=fs, freqs=f.copy())
print("per call %.3f ms" % (1e3 * (time.perf_counter() - t0) / 200))
pr = cProfile.Profile(); pr.enable()
for _ in range(200): cwt.transform(x, fs=fs, freqs=f.copy())
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)

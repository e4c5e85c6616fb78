"""Many short epochs (the nelpy use: one transform over an epoch array): per-epoch launch
overhead vs work.  300 epochs x 4000 samples, 1 channel, default grid."""
import sys, time; sys.path.insert(0, '.')
import numpy as np
from ghost_amd.wave import ContinuousWaveletTransform
from ghost_amd.synthetic import lfp_channel
fs = 1000.0; ne, le = 300, 4000
x = lfp_channel(ne * le, fs, 1).astype(np.float64)
t = np.arange(ne * le) / fs + np.repeat(np.arange(ne) * 3.0, le)      # 3 s gap after every epoch
cwt = ContinuousWaveletTransform()
for it in range(7):
    t0 = time.perf_counter()
    cwt.transform(x, fs=fs, timestamps=t, verbose=(it == 6))
    print("run %d: %.3f s, %d scales, %d epochs" % (it, time.perf_counter() - t0, cwt.frequencies.size, ne), flush=True)

import sys, time; sys.path.insert(0,'.')
import numpy as np
from ghost_amd.wave import ContinuousWaveletTransform
from ghost_amd.synthetic import lfp_channel
from ghost_amd import hostmem
fs=1000.; N=1000000
x=lfp_channel(N, fs, channel=0, seed=4321)
f=np.geomspace(2,200,100)
for name,kw in (("float64",{}),("float32",{"dtype":np.float32})):
    cwt=ContinuousWaveletTransform()
    for i in range(5):
        t0=time.perf_counter(); cwt.transform(x, fs=fs, freqs=f.copy(), **kw); t1=time.perf_counter()
        a=cwt.amplitude; t2=time.perf_counter()
        print(name, i, "transform %.2f ms  materialize %.2f ms  total %.2f" % ((t1-t0)*1e3,(t2-t1)*1e3,(t2-t0)*1e3), a.dtype, a.shape, hostmem.is_pinned(a))
        del a
    t0=time.perf_counter(); cwt.transform(x, fs=fs, freqs=f.copy(), **kw); p=cwt.fetch(start=500000, stop=501000); print(name,"slice %.2f ms"%((time.perf_counter()-t0)*1e3), p.shape)
    # eager unpinned for comparison
    lim=hostmem.limit_bytes; hostmem.limit_bytes=0
    for i in range(3):
        t0=time.perf_counter(); cwt.transform(x, fs=fs, freqs=f.copy(), lazy=False, **kw); print(name,"pageable eager %.2f ms"%((time.perf_counter()-t0)*1e3))
    hostmem.limit_bytes=lim

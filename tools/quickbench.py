import sys, time; sys.path.insert(0,'.')
import numpy as np
from ghost_amd.engine import CwtPlan, DeviceBuffer, device_name
from ghost_amd.synthetic import lfp
print(device_name(0))
fs=1000.; N=1000000; S=100
f=np.geomspace(200,2,S)
for C in (1, 16, 128):
    plan=CwtPlan(N,C,fs,f)
    plan.set_profiling(True)
    x=lfp(min(C,4),N)
    x=np.tile(x,(C//x.shape[0]+1,1))[:C]
    xb=DeviceBuffer(x.nbytes); xb.upload(x)
    ob=DeviceBuffer(plan.info['out_bytes'])
    plan.upload()
    for it in range(3):
        t0=time.time(); plan.execute_device(xb,ob); dt=time.time()-t0
        tm=plan.timings()
        print(C, "wall %.2f ms"%(dt*1e3), {k:round(v,3) for k,v in tm.items()})
    print("Msamples/s", C*N/dt/1e6, "GB/s out", C*N*S*4/dt/1e9)
    plan.close(); xb.free(); ob.free()

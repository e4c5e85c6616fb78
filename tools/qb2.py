import sys, time, os; sys.path.insert(0,'.')
import numpy as np
from ghost_amd.engine import CwtPlan, DeviceBuffer
from ghost_amd.synthetic import lfp
fs=1000.; N=1000000; S=100; C=int(os.environ.get("QB_C","128"))
f=np.geomspace(200,2,S)
plan=CwtPlan(N,C,fs,f)
plan.set_profiling(True)
x=lfp(4,N); x=np.tile(x,(C//4+1,1))[:C]
xb=DeviceBuffer(x.nbytes); xb.upload(x)
ob=DeviceBuffer(plan.info['out_bytes'])
for it in range(3):
    plan.execute_device(xb,ob)
tm=plan.timings()
print(os.environ.get("GHOSTCWT_DEBUG_FLAGS","0"), {k:round(v,3) for k,v in tm.items() if k in ('synth_ms','total_ms')})

import sys, time, os; sys.path.insert(0,'.')
sys.path.insert(0, 'tools')
import numpy as np
from ghost_amd.engine import CwtPlan, DeviceBuffer
from _opts import apply_env_options; apply_env_options()
from ghost_amd.synthetic import lfp
fs=1000.; N=1000000; S=100; C=int(os.environ.get("QB_C","128")); reps=int(os.environ.get("QB_REPS","10"))
f=np.geomspace(200,2,S)
plan=CwtPlan(N,C,fs,f,output=os.environ.get("QB_OUT","amplitude"),precision=os.environ.get("QB_PRECISION","auto"))
plan.set_profiling(True)
x=lfp(4,N); x=np.tile(x,(C//4+1,1))[:C]
xb=DeviceBuffer(x.nbytes); xb.upload(x)
ob=DeviceBuffer(plan.info['out_bytes'])
syn=[]; tot=[]
for it in range(reps+2):
    plan.execute_device(xb,ob)
    tm=plan.timings()
    if it>=2: syn.append(tm['synth_ms']); tot.append(tm['total_ms'])
clk=""
if os.environ.get("GHOSTCWT_CLOCK_PROBE"):
    import ctypes
    from ghost_amd._lib import lib, check
    g=ctypes.c_double(); w=ctypes.c_double()
    check(lib.gcwt_debug_clock(plan._handle, ctypes.byref(g), ctypes.byref(w)))
    clk=" | synth clock %.3f GHz, workgroup-seconds/launch %.4f"%(g.value, w.value/(reps+2))
print(os.environ.get("QB_TAG","")+clk, "synth min %.3f med %.3f | total min %.3f med %.3f"%(min(syn), np.median(syn), min(tot), np.median(tot)),
      {k:round(v,3) for k,v in tm.items() if k.endswith('_ms') and k not in ('synth_ms','total_ms')})

import sys, time; sys.path.insert(0,'.')
import numpy as np
from ghost_amd.engine import CwtPlan
for name,args in (("headline",(1000000,128,1000.,np.geomspace(200.,2.,100))),("config5",(18000000,48,30000.,np.geomspace(500.,1.,200))),("headline again",(1000000,128,1000.,np.geomspace(200.,2.,100)))):
    t0=time.perf_counter(); p=CwtPlan(*args); t1=time.perf_counter(); p.upload(); t2=time.perf_counter()
    print("%-15s host plan %.1f ms, upload (allocations, tables, bank) %.1f ms"%(name,(t1-t0)*1e3,(t2-t1)*1e3)); p.close()

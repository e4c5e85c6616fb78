"""Cost of the exact paths that heavy-tailed wavelets take, headline shape: 128 ch x 1e6 x 100
scales 200..2 Hz with Morse(3, 4) (time-domain scales for L <= 256, full-band above)."""
import sys, os; sys.path.insert(0, '.')
import numpy as np
from ghost_amd.engine import CwtPlan, DeviceBuffer
from ghost_amd.synthetic import lfp
from ghost_amd.engine import set_option
if os.environ.get('HT_GROUP'): set_option('fullband_group', int(os.environ['HT_GROUP']))
for k in ('blockconv', 'direct_max_len'):
    if os.environ.get('HT_' + k.upper()): set_option(k, int(os.environ['HT_' + k.upper()]))
fs = 1000.; N = 1000000; C = int(os.environ.get("QB_C", "128")); S = 100
g, b = float(os.environ.get("HT_GAMMA", "3")), float(os.environ.get("HT_BETA", "4"))
f = np.geomspace(200.0, 2.0, S)
plan = CwtPlan(N, C, fs, f, gamma=g, beta=b, precision=os.environ.get('HT_PRECISION')); plan.set_profiling(True)
info = plan.info
x = lfp(4, N); x = np.tile(x, (C // 4 + 1, 1))[:C]
xb = DeviceBuffer(x.nbytes); xb.upload(x)
ob = DeviceBuffer(info['out_bytes'])
for it in range(3):
    plan.execute_device(xb, ob); tm = plan.timings()
print("Morse(%g,%g): spectral %d direct %d blockconv %d fullband %d | direct %.1f ms blockconv %.1f ms fullband %.1f ms synth %.1f ms total %.1f ms -> %.0f Msamples/s" %
      (g, b, info['n_spectral'], info['n_direct'], info['n_blockconv'], info['n_fullband'], tm['direct_ms'], tm['blockconv_ms'], tm['fullband_ms'], tm['synth_ms'],
       tm['total_ms'], C * N / tm['total_ms'] / 1e3))

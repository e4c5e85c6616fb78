"""Debug helper: the pipelined interpolating kernel (option synthp) against k_synthi on block requests; prints where
they differ (channel, scale, sample), several repeats (a race shows as run-to-run differences)."""
import sys; sys.path.insert(0, '.')
import numpy as np
from ghost_amd.engine import CwtPlan, set_option
from ghost_amd.synthetic import lfp

fs, n = 1000.0, 30000
x = lfp(2, n, fs) + 0.75
f = [300.0, 150.0, 40.0, 12.0]

def run(flag):
    set_option("synthp", flag)
    p = CwtPlan(n, 2, fs, np.asarray(f), output="amplitude", max_fft_log2=13)
    full = p.execute(x)
    segs = p.segments()
    outs = [full]
    for (a, b, _) in (segs[0], segs[2], segs[-1]):
        outs.append((a, b, p.execute_block(x, a, b - a)))
    outs.append((7777, 7777 + 9001, p.execute_block(x, 7777, 9001, reuse_means=True)))
    print("levels", p.debug_levels() if hasattr(p, "debug_levels") else "")
    return outs

ref = run(0)
import os
for rep in range(2):
    set_option("synthp_turns", rep)
    got = run(1)
    if rep == 0:
        p = CwtPlan(n, 2, fs, np.asarray(f), output="amplitude", max_fft_log2=13); print("segments", p.segments())
    d = np.argwhere(got[0] != ref[0])
    print("rep", rep, "full: mismatches", len(d), d[:12].tolist())
    for k in range(1, len(got)):
        a, b, blk = got[k]
        d = np.argwhere(blk != ref[0][:, :, a:b])
        print("  block [%d, %d): mismatches %d" % (a, b, len(d)), [(int(c), int(s), int(a + i)) for c, s, i in d[:12]])
set_option("synthp_turns", 0)
got = run(1)
a, b, blk = got[4]
np.set_printoptions(precision=6, linewidth=200)
print("ref ", ref[0][0, 3, 7770:7790])
print("blk ", blk[0, 3, 0:13])
print("ref1", ref[4][2][0, 3, 0:13])

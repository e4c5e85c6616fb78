"""The engine's dynamic-range envelope, measured (round 4): gate metric against the amplitude of an interferer -- a 60 Hz
line (inside the bands of the levels R = 2, 4, 16 of the headline grid) and a 0.05 Hz drift (below every level's low
cut) -- added to the pink LFP workload data, N = 2^19 @ 1 kHz, the headline's 100 scales, complex output, the three
precisions ('exact': no decimated path).  D = interferer amplitude over the smallest row maximum of the clean recording's coefficients: how far the
quietest analysed band lies below the interferer.  Writes gpurun_out/dynamic_range.json."""
import json, sys; sys.path.insert(0, '.')
import numpy as np
from ghost_amd.engine import CwtPlan
from ghost_amd.synthetic import lfp_channel
from oracle import ghost_oracle as orc
fs, n = 1000.0, 1 << 19
f = np.geomspace(200.0, 2.0, 100)
t = np.arange(n) / fs
base = lfp_channel(n, fs, 2).astype(np.float64)
clean = orc.cwt_complex(base, fs, f, n_threads=8)
# the interferer fades in and out over the first and last tenth of the recording (half a cosine): switched on abruptly
# its edge transient would dominate every row's maximum and the gate metric would not see the interior
win = np.ones(n); m = n // 10
win[:m] = 0.5 - 0.5 * np.cos(np.pi * np.arange(m) / m); win[-m:] = win[:m][::-1]
quiet = np.abs(clean).max(axis=1).min()
rows = []
for kind, freq in (("line 60 Hz", 60.0), ("drift 0.05 Hz", 0.05)):
    for amp in (10.0, 30.0, 100.0, 300.0, 1000.0, 3000.0, 10000.0):
        x = (base + amp * base.std() * win * np.sin(2 * np.pi * freq * t + 0.7)).astype(np.float32)
        ref = orc.cwt_complex(x.astype(np.float64), fs, f, n_threads=8)
        res = {}
        for prec in ("high", "fast", "exact"):
            p = CwtPlan(n, 1, fs, f, output="complex", precision=prec)
            got = p.execute(x[None])[0]
            si = p.scale_info()
            p.close()
            err = np.abs(got - ref).max(axis=1) / np.abs(ref).max(axis=1)
            res[prec] = (float(err.max()), int(si["decimation"][err.argmax()]), float(f[err.argmax()]))
        rows.append(dict(kind=kind, amp_over_std=amp, D=float(amp * base.std() / quiet), high=res["high"], fast=res["fast"],
                         exact=res["exact"]))
        print("%-14s A = %6.0f std  D = %8.0f   high %.2e (R %d, %.1f Hz)   fast %.2e (R %d, %.1f Hz)   exact %.2e (%.1f Hz)" % (
            kind, amp, rows[-1]["D"], *res["high"], *res["fast"], res["exact"][0], res["exact"][2]), flush=True)
json.dump(dict(fs=fs, n=n, quiet_row_max_over_std=float(quiet / base.std()), rows=rows), open("gpurun_out/dynamic_range.json", "w"), indent=1)

"""Steep backgrounds without an edge transient (round 4): 1/f^2 and 1/f^3 noise faded in and out over a tenth of the
recording at each end, so that a row's maximum is the band's own level and not the response to the recording's first
sample.  N = 1e6 @ 1 kHz, the headline's 100 scales, complex output; both precisions; worst scale per decimation."""
import json, sys; sys.path.insert(0, '.')
import numpy as np
from ghost_amd.engine import CwtPlan
from ghost_amd.synthetic import power_law_noise
from oracle import ghost_oracle as orc
fs, n = 1000.0, 1000000
f = np.geomspace(200.0, 2.0, 100)
win = np.ones(n); m = n // 10
win[:m] = 0.5 - 0.5 * np.cos(np.pi * np.arange(m) / m); win[-m:] = win[:m][::-1]
out = {}
for expo in (1.0, 2.0, 2.5, 3.0):
    x = (power_law_noise(n, expo, 11) * win).astype(np.float32)
    ref = orc.cwt_complex(x.astype(np.float64), fs, f, n_threads=8)
    D = x.astype(np.float64).std() / np.abs(ref).max(axis=1)
    row = {}
    for prec in ("high", "fast"):
        p = CwtPlan(n, 1, fs, f, output="complex", precision=prec)
        got = p.execute(x[None])[0]; si = p.scale_info(); p.close()
        err = np.abs(got - ref).max(axis=1) / np.abs(ref).max(axis=1)
        row[prec] = {int(R): float(err[si["decimation"] == R].max()) for R in sorted(set(si["decimation"].tolist()))}
    out["1/f^%g" % expo] = dict(D_top=float(D[0]), D_bottom=float(D[-1]), **row)
    print("1/f^%-3g D(200 Hz) %7.0f D(2 Hz) %5.1f  high: %s" % (expo, D[0], D[-1], " ".join("R%d %.1e" % kv for kv in row["high"].items())))
    print("%36s fast: %s" % ("", " ".join("R%d %.1e" % kv for kv in row["fast"].items())), flush=True)
json.dump(out, open("gpurun_out/steep_interior.json", "w"), indent=1)

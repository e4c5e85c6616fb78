"""Time blocks on steep spectra (round 4): config 5's geometry (30 kHz, 200 scales 500 .. 1 Hz, blocks of 2^22 / 2^21 samples),
1/f^2 and 1/f^3 recordings with an offset, rows across a seam between blocks against the oracle."""
import sys; sys.path.insert(0, '.')
import numpy as np
from ghost_amd.engine import CwtPlan
from ghost_amd.synthetic import power_law_noise
from oracle import ghost_oracle as orc
fs, n, S = 30000.0, 6000000, 200
f = np.geomspace(500.0, 1.0, S)
for expo in (2.0, 3.0):
    x = (power_law_noise(n, expo, 5) + 3.0).astype(np.float32)
    for mfl in (0, 21):
        p = CwtPlan(n, 1, fs, f, output="amplitude", max_fft_log2=mfl)
        segs = p.segments()
        seam = segs[0][1] if len(segs) > 1 else n // 2
        a, ln = seam - 150000, 300000
        got = p.execute_block(x[None], a, ln)[0]
        om = orc.hz_to_rad(f, fs); lengths = orc.morse_lengths(om)
        xc = x.astype(np.float64); xc -= xc.mean()
        res = []
        for sc in (0, 80, 140, 170, 199):
            L = int(lengths[sc]); psi, _ = orc.morse_kernel(L, om[sc])
            w0, w1 = max(0, a - L), min(n, a + ln + L)
            ref = np.abs(orc.overlap_add_convolve(xc[w0:w1], psi)[a - w0:a - w0 + ln])
            res.append(float(np.abs(got[sc] - ref).max() / ref.max()))
        print("1/f^%g, max_fft_log2 %d (%d blocks of %d): %s" % (expo, mfl, len(segs), segs[0][2], " ".join("%.1e" % e for e in res)), flush=True)
        p.close()

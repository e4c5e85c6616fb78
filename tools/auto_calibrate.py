"""precision = auto: the detector's prediction against the measured error (gate metric, oracle in float64) on the
inputs of the precision work -- LFP with a 60 Hz line of 10 .. 1000 x its spread, the spectrum classes -- with the
fast path alone (precision='high') and rerouted (the default)."""
import sys, os, time; sys.path.insert(0, '.')
import numpy as np
from ghost_amd.engine import CwtPlan
from ghost_amd.synthetic import lfp_channel
from oracle import ghost_oracle as orc

fs, n = 1000.0, int(os.environ.get("AC_N", "200000"))
f = np.geomspace(200.0, 2.0, 100)
t = np.arange(n) / fs
base = lfp_channel(n, fs, channel=3).astype(np.float64)
sd = base.std()
win = np.sin(np.pi * np.arange(n) / n) ** 2          # the line fades in and out (profiles/r04_dynamic_range.md)
cases = [("lfp", base)]
for amp in (10, 30, 100, 300, 1000):
    cases.append(("line60 x%d" % amp, base + amp * sd * win * np.sin(2 * np.pi * 60.0 * t)))
cases.append(("line17 x300", base + 300 * sd * win * np.sin(2 * np.pi * 17.0 * t)))
cases.append(("drift x1000", base + 1000 * sd * win * np.sin(2 * np.pi * 0.05 * t)))
for name, x in cases:
    x = x.astype(np.float32).astype(np.float64)      # what the device is given: the oracle sees the same samples
    ref = orc.cwt_amplitude(x, fs, f)
    row = []
    for prec in ("high", "auto"):
        p = CwtPlan(n, 1, fs, f, precision=prec)
        t0 = time.time(); got = p.execute(x.astype(np.float32)[None])[0]; dt = time.time() - t0
        t0 = time.time(); got = p.execute(x.astype(np.float32)[None])[0]; dt2 = time.time() - t0
        err = np.abs(got - ref).max(axis=1) / ref.max(axis=1)
        rep = p.precision_report()
        if prec == "high":
            rep["terms"] = p.debug_precision_terms()
        row.append((prec, err, rep, dt2))
        p.close()
    (_, e_h, r_h, _), (_, e_a, r_a, dt_a) = row
    ratio = e_h / np.maximum(r_h["predicted"], 1e-12)
    sel = (r_h["terms"]["left_out"] > r_h["terms"]["rounding"]) & (e_h > 1.5e-6)
    if sel.any():
        print("   left-out term dominant on %d scales with err > 1.5e-6: err / term percentiles %s" % (sel.sum(), np.array2string(np.percentile(e_h[sel] / r_h["terms"]["left_out"][sel], [10, 50, 90]), precision=2)))
    worst = int(np.argmax(e_h))
    print("%-12s high: worst err %.2e (scale %d, %.1f Hz) predicted there %.2e; max predicted %.2e; err/pred over scales with err>1e-6: %s | auto: worst err %.2e rerouted %d (%.1f ms)"
          % (name, e_h.max(), worst, f[worst], r_h["predicted"][worst], r_h["worst"],
             np.array2string(np.percentile(ratio[e_h > 1e-6], [10, 50, 90]), precision=2) if (e_h > 1e-6).any() else "-",
             e_a.max(), r_a["rerouted"], dt_a * 1e3))
if os.environ.get("AC_DETAIL"):
    amp = float(os.environ["AC_DETAIL"])
    x = (base + amp * sd * win * np.sin(2 * np.pi * 60.0 * t)).astype(np.float32).astype(np.float64)
    ref = orc.cwt_amplitude(x, fs, f)
    out = {}
    for prec in ("high", "auto", "exact"):
        p = CwtPlan(n, 1, fs, f, precision=prec)
        got = p.execute(x.astype(np.float32)[None])[0]
        out[prec] = (np.abs(got - ref).max(axis=1) / ref.max(axis=1), p.precision_report()["predicted"] if prec != "exact" else None, p.scale_info()["decimation"])
        p.close()
    for s in range(len(f)):
        print("%3d %7.2f Hz R=%3d  err high %.2e pred %.2e | auto %.2e | exact %.2e" % (s, f[s], out["high"][2][s], out["high"][0][s], out["high"][1][s], out["auto"][0][s], out["exact"][0][s]))

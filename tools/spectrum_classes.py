"""Parity of the engine on recordings with steep spectra, mains interference and drift (round 4,
VERDICT r03 task 1): input class x scale -> gate metric max|y - ref| / max|ref| against the oracle.
Headline scales (100 log-spaced 200..2 Hz) at fs = 1 kHz, N = 1e6; SC_N / SC_OUT override.
SC_GAMMA / SC_BETA: another Morse wavelet; SC_BLOCKCONV=0: the older exact paths (time domain / one FFT per segment).
Writes gpurun_out/spectrum_classes.json (copy under profiles/ to track)."""
import json, os, sys, time; sys.path.insert(0, '.')
import numpy as np
from ghost_amd.engine import CwtPlan
from ghost_amd.synthetic import spectrum_class, SPECTRUM_CLASSES
from oracle import ghost_oracle as orc

fs = 1000.0
n = int(float(os.environ.get("SC_N", "1e6")))
f = np.geomspace(200.0, 2.0, 100)
kw = {}
if os.environ.get("SC_PRECISION"):
    kw["precision"] = os.environ["SC_PRECISION"]
okw = {}
if os.environ.get("SC_GAMMA"):
    okw = dict(gamma=float(os.environ["SC_GAMMA"]), beta=float(os.environ["SC_BETA"]))
    kw.update(okw)
if os.environ.get("SC_BLOCKCONV"):
    from ghost_amd.engine import set_option
    set_option("blockconv", int(os.environ["SC_BLOCKCONV"]))
table = {}
for name in SPECTRUM_CLASSES:
    x = spectrum_class(name, n, fs)
    t0 = time.time()
    ref = orc.cwt_complex(x.astype(np.float64), fs, f, n_threads=8, **okw)
    t1 = time.time()
    row = {}
    for output in ("complex", "amplitude"):
        p = CwtPlan(n, 1, fs, f, output=output, **kw)
        got = p.execute(x[None])[0]
        r = ref if output == "complex" else np.abs(ref)
        err = np.abs(got - r).max(axis=1) / np.abs(r).max(axis=1)
        si = p.scale_info()
        p.close()
        row[output] = err
    worst = max(row["complex"].max(), row["amplitude"].max())
    print("%-10s oracle %.1f s  complex max %.2e (scale %d)  amplitude max %.2e (scale %d)" % (
        name, t1 - t0, row["complex"].max(), row["complex"].argmax(), row["amplitude"].max(), row["amplitude"].argmax()), flush=True)
    by_level = {}
    for R in sorted(set(si["decimation"].tolist())):
        m = si["decimation"] == R
        by_level[int(R)] = [float(row["complex"][m].max()), float(row["amplitude"][m].max())]
    print("           per decimation (complex, amplitude): " + "  ".join("R%d %.1e %.1e" % (R, a, b) for R, (a, b) in by_level.items()), flush=True)
    table[name] = dict(worst=float(worst), complex=row["complex"].tolist(), amplitude=row["amplitude"].tolist(), by_level=by_level)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(dict(fs=fs, n=n, frequencies=f.tolist(), kw=kw, classes=table), open("gpurun_out/spectrum_classes%s.json" % os.environ.get("SC_TAG", ""), "w"), indent=1)

"""precision = auto on the input classes of the precision work (ghost_amd/synthetic.py: spectrum_class) at the headline
scales: worst gate-metric error of 'high' and 'auto', how many scales 'auto' made again, and the time of each."""
import sys, os, time; sys.path.insert(0, '.')
import numpy as np
from ghost_amd.engine import CwtPlan
from ghost_amd.synthetic import spectrum_class, SPECTRUM_CLASSES
from oracle import ghost_oracle as orc
fs, n = 1000.0, int(os.environ.get("AC_N", "200000"))
f = np.geomspace(200.0, 2.0, 100)
for name in SPECTRUM_CLASSES:
    x = spectrum_class(name, n, fs).astype(np.float32)
    ref = orc.cwt_amplitude(x.astype(np.float64), fs, f)
    line = "%-12s" % name
    for prec in ("high", "auto"):
        p = CwtPlan(n, 1, fs, f, precision=prec)
        p.execute(x[None]); t0 = time.time(); got = p.execute(x[None])[0]; dt = time.time() - t0
        err = np.abs(got - ref).max(axis=1) / ref.max(axis=1)
        rep = p.precision_report()
        line += " | %s: err %.2e predicted %.2e rerouted %3d %.1f ms" % (prec, err.max(), rep["worst"], rep["rerouted"], dt * 1e3)
        if prec == "high":
            tm = p.debug_precision_terms()
            line += " (rounding %.1e left-out %.1e)" % (tm["rounding"].max(), tm["left_out"].max())
        p.close()
    print(line)

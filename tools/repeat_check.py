"""Race check: the same plan executed many times must give bit-identical results (the level
passes run on three streams; a missing dependency would show up as run-to-run differences).
REPEAT_N runs (default 20) of 4 ch x 300 000 samples x 100 scales, amplitude and complex."""
import os, sys; sys.path.insert(0, '.')
import numpy as np
from ghost_amd.engine import CwtPlan
from ghost_amd.synthetic import lfp

n_runs = int(os.environ.get("REPEAT_N", "20"))
fs, C, n = 1000.0, 4, 300000
f = np.geomspace(200.0, 2.0, 100)
x = lfp(C, n, fs, seed=77)
for output in ("amplitude", "complex"):
    for eb in (None, [[0, 120000], [120500, 300000]]):
        kw = dict(output=output)
        if eb is not None:
            kw["epoch_bounds"] = eb
        p = CwtPlan(n, C, fs, f, **kw)
        first = p.execute(x)
        bad = 0
        for i in range(n_runs):
            bad += not np.array_equal(p.execute(x), first)
        print("%-9s epochs %d: %d of %d repeats differ" % (output, 1 if eb is None else 2, bad, n_runs), flush=True)
        if bad:
            sys.exit(1)
        p.close()
print("ok")

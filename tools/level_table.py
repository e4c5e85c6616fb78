"""Per decimation level of the headline grid (128 ch x 1e6 x the scales of that level only):
which kernel makes it, ms per launch, GB/s of result rows, and package power / sclk read from
sysfs while the level is run for LT_SECONDS (bench.py's PowerSampler: no HIP call in the thread).
GHOSTCWT_INTERP=0 gives the FFT-per-sample kernel for every level (A/B on the same box)."""
import sys, os, time; sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
import numpy as np
from ghost_amd.engine import CwtPlan, DeviceBuffer
from _opts import apply_env_options; apply_env_options()
from ghost_amd.synthetic import lfp
import importlib.util
spec = importlib.util.spec_from_file_location("bench_mod", "bench.py")
fs, N, C = 1000.0, 1000000, int(os.environ.get("LT_C", "128"))
seconds = float(os.environ.get("LT_SECONDS", "1.0"))
f_all = np.geomspace(200.0, 2.0, 100)
full = CwtPlan(N, C, fs, f_all)
dec = full.scale_info()["decimation"]
x = lfp(4, N); x = np.tile(x, (C // 4 + 1, 1))[:C]
xb = DeviceBuffer(x.nbytes); xb.upload(x)

class Sampler:                      # (bench.py redirects stdout on import: a copy of its PowerSampler's reads)
    def __init__(self):
        import glob
        self.cards = [hw for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
                      if os.path.exists(os.path.join(hw, "power1_average")) or os.path.exists(os.path.join(hw, "power1_input"))]
    def read(self):
        best = (0.0, 0.0)
        for hw in self.cards:
            try:
                pf = os.path.join(hw, "power1_average")
                if not os.path.exists(pf): pf = os.path.join(hw, "power1_input")
                p = float(open(pf).read()) / 1e6
                c = float(open(os.path.join(hw, "freq1_input")).read()) / 1e9
                if p > best[0]: best = (p, c)
            except (OSError, ValueError):
                pass
        return best
smp = Sampler()
print("| R | scales | kernel | ms | GB/s of rows | W | sclk GHz |")
print("|---|---|---|---|---|---|---|")
tot = 0.0
for R in sorted(set(dec.tolist())):
    f = f_all[dec == R]
    plan = CwtPlan(N, C, fs, f); plan.set_profiling(True)
    assert set(plan.scale_info()["decimation"].tolist()) <= {R, R // 2, 2 * R}, (R, set(plan.scale_info()["decimation"].tolist()))
    ob = DeviceBuffer(plan.info["out_bytes"])
    ts = []
    for i in range(4):
        plan.execute_device(xb, ob); ts.append(plan.timings()["synth_ms"])
    t0 = time.time(); pw = []
    while time.time() - t0 < seconds:
        plan.execute_device(xb, ob); ts.append(plan.timings()["synth_ms"])
        if time.time() - t0 > 0.4 * seconds: pw.append(smp.read())
    t = float(np.median(ts[4:])) if len(ts) > 4 else min(ts)
    tot += t
    w = np.mean([p[0] for p in pw]) if pw else float("nan"); ck = np.mean([p[1] for p in pw]) if pw else float("nan")
    kern = "k_synthi" if plan.info["n_interp"] else "k_synth7"
    print("| %d | %d | %s | %.3f | %.0f | %.0f | %.2f |" % (R, len(f), kern, t, C * N * len(f) * 4 / t / 1e6, w, ck))
    plan.close(); ob.free()
print("sum of the levels %.2f ms" % tot)

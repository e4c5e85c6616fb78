"""Round 6, VERDICT r05 task 1(a): what bounds the synthesis -- package power or issue / latency?

The synthesis kernels of the headline workload (128 ch x 1e6 x 100 scales) run ALONE on streams made with a CU mask
(option cu_count -> hipExtStreamCreateWithCUMask in gcwt_plan_upload) at 256 / 224 / 192 / 160 / 128 CUs; per point: ms
per launch, package W, sclk, mclk, fclk (sysfs, sampled while the kernel runs back to back for BS_SECONDS).
  energy-bound        -> the time stays about flat while CUs drop and the clock rises towards 2.4 GHz;
  issue/latency-bound -> the time goes as 1 / CUs at once, at an unchanged or higher clock and lower power.
Groups: k_synthi (R >= 16, 63 scales), k_synth7 (R = 2, 4, 8; 37 scales) and each level alone (BS_LEVELS=1).
Prints a markdown table; tools/cu_mask_probe.hip gives the same sweep for a pure store kernel and a pure FMA kernel.
With the measure build (GHOSTCWT_LIB=ghost_amd/libghostcwt_measure.so BS_PROBE=1) the k_synth7 rows also carry the clock
the workgroups themselves saw (s_memtime over s_memrealtime, gcwt_debug_clock) next to the sysfs reading."""
import glob
import os
import sys
import time

sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
import numpy as np
from ghost_amd.engine import CwtPlan, DeviceBuffer, set_option
from _opts import apply_env_options; apply_env_options()
from ghost_amd.synthetic import lfp

fs, N, C = 1000.0, 1000000, int(os.environ.get("BS_C", "128"))
seconds = float(os.environ.get("BS_SECONDS", "2.0"))
counts = [int(v) for v in os.environ.get("BS_CUS", "256,224,192,160,128").split(",")]
f_all = np.geomspace(200.0, 2.0, 100)


class Sysfs:
    """Package power / clocks of the card under load: a box may list several cards in sysfs; the one this process drives
    is the one that draws the most while the kernel runs (bench.py's PowerSampler reads the same way)."""
    def __init__(self):
        self.cards = []
        for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            for name in ("power1_average", "power1_input"):
                if os.path.exists(os.path.join(hw, name)):
                    self.cards.append((hw, os.path.join(hw, name)))
                    break
        # the card this process drives, by PCI address (HIP runtime, tools only); else the busiest card at each read
        try:
            import ctypes
            hip = ctypes.CDLL("libamdhip64.so")
            buf = ctypes.create_string_buffer(64)
            if hip.hipDeviceGetPCIBusId(buf, 64, 0) == 0:
                mine = glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % buf.value.decode().lower())
                own = [(hw, pf) for hw, pf in self.cards if mine and os.path.realpath(hw) == os.path.realpath(mine[0])]
                if not own and mine:
                    own = [(mine[0], os.path.join(mine[0], n)) for n in ("power1_average", "power1_input")
                           if os.path.exists(os.path.join(mine[0], n))][:1]
                if own:
                    self.cards = own
        except OSError:
            pass
        self.hw = "%d card(s): %s" % (len(self.cards), self.cards[0][0] if len(self.cards) == 1 else "busiest at each read")

    @staticmethod
    def dpm(dev, name):
        try:
            for line in open(os.path.join(dev, name)):
                if "*" in line:
                    return float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
        except (OSError, ValueError, IndexError, TypeError):
            pass
        return float("nan")

    def read(self):
        best = (float("nan"),) * 4
        top = -1.0
        for hw, pf in self.cards:
            try:
                w = float(open(pf).read()) / 1e6
                ck = float(open(os.path.join(hw, "freq1_input")).read()) / 1e9
            except (OSError, ValueError):
                continue
            if w > top:
                dev = os.path.realpath(hw).split("/hwmon")[0]
                top, best = w, (w, ck, self.dpm(dev, "pp_dpm_mclk"), self.dpm(dev, "pp_dpm_fclk"))
        return best


smp = Sysfs()
print("sysfs:", smp.hw, flush=True)
full = CwtPlan(N, C, fs, f_all)
dec = full.scale_info()["decimation"]
full.close()
x = lfp(4, N); x = np.tile(x, (C // 4 + 1, 1))[:C]
xb = DeviceBuffer(x.nbytes); xb.upload(x)

groups = [("k_synthi (R >= 16)", f_all[dec >= 16]), ("k_synth7 (R = 2, 4, 8)", f_all[dec <= 8])]
if os.environ.get("BS_LEVELS", "0") == "1":
    groups += [("R = %d" % R, f_all[dec == R]) for R in sorted(set(dec.tolist()))]

probe = os.environ.get("BS_PROBE", "0") == "1"
if probe:
    import ctypes
    from ghost_amd._lib import lib, check
    set_option("clock_probe", 1)
only = os.environ.get("BS_ONLY")            # substring of the group names to keep ("synth7", "R = 2", ...)
if only:
    groups = [g for g in groups if only in g[0]]
print("| kernels | scales | CUs | ms | TB/s of rows | ms x CUs / 256 | W | sclk GHz | mclk MHz | fclk MHz | in-kernel GHz |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for name, f in groups:
    for n_cu in counts:
        set_option("cu_count", 0 if n_cu >= 256 else n_cu)
        plan = CwtPlan(N, C, fs, f); plan.set_profiling(True)
        ob = DeviceBuffer(plan.info["out_bytes"])
        ts, pw = [], []
        for i in range(3):
            plan.execute_device(xb, ob)
        t0 = time.time()
        while time.time() - t0 < seconds:
            plan.execute_device(xb, ob); ts.append(plan.timings()["synth_ms"])
            if time.time() - t0 > 0.4 * seconds:
                pw.append(smp.read())
        t = float(np.median(ts[len(ts) // 3:]))
        m = np.nanmean(np.array(pw), axis=0) if pw else [float("nan")] * 4
        ghz = float("nan")
        if probe and plan.info["n_interp"] == 0:
            g, w = ctypes.c_double(), ctypes.c_double()
            check(lib.gcwt_debug_clock(plan._handle, ctypes.byref(g), ctypes.byref(w)))
            ghz = g.value
        print("| %s | %d | %d | %.3f | %.2f | %.3f | %.0f | %.2f | %.0f | %.0f | %.3f |" % (
            name, len(f), n_cu, t, C * N * len(f) * 4 / t / 1e9, t * n_cu / 256.0, m[0], m[1], m[2], m[3], ghz), flush=True)
        plan.close(); ob.free()
set_option("cu_count", 0)

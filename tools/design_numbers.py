"""Rewrites the generated fragments of DESIGN.md -- the paragraphs between `<!-- gen:NAME -->` and `<!-- /gen -->` and the
tracked-numbers block -- from the round's tracked evidence under profiles/, so that the text and the files cannot drift
(tests/test_docs.py checks the block).  Usage: python tools/design_numbers.py [r04]"""
import json, os, re, sys
rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
j = json.loads(open(os.path.join(root, "profiles", rnd + "_bench.json")).read().strip().splitlines()[-1])
j5 = json.loads(open(os.path.join(root, "profiles", rnd + "_bench_config5.json")).read().strip().splitlines()[-1])
tr = json.load(open(os.path.join(root, "profiles", "traffic.json")))
rl, st, cx = j["roofline"], j["stages_ms"], j["other_modes"]["complex"]
c2, c5 = j["other_configs"]["config2"], j["other_configs"]["config5"]
stats = {}
import csv
for row in csv.DictReader(open(os.path.join(root, "profiles", rnd + "_kernel_stats.csv"))):
    stats[row["Name"].split("(")[0].replace("void ", "")] = (int(row["Calls"]), float(row["AverageNs"]) / 1e6)
# k_synth7: its 32-column launch (R = 4, 8) and, since round 6, the 16-column one (R = 2): one launch each per step
ki = stats["gcwt::k_synthi<0>"]
k7_parts = [v for k, v in stats.items() if k.startswith("gcwt::k_synth7<0,")]
k7 = (k7_parts[0][0], sum(v[1] for v in k7_parts))
gen = {}
te = c2["transform_end_to_end"]
cb = j["cpu_baseline"]
gen["headline"] = (
    "**Measured, round %s** (`profiles/%s_bench.json`, `tools/prof_round.sh %s`: `bench.py --steps 10 --warmup 2` on one MI355X;\n"
    "boxes of the pool differ by 2 - 4 %%): synthesis %.2f ms from HIP events (k_synthi %.2f + k_synth7 %.2f) = %.2f TB/s =\n"
    "**%.4f** of the 8 TB/s peak (`roofline.frac`); rocprofv3 of the same command (`profiles/%s_kernel_stats.csv`, %d launches\n"
    "each): %.3f + %.3f = %.2f ms.  Whole step %.2f ms = **%.0f Msamples/s** = %.3f of the peak on algorithmic bytes (mean\n"
    "%.2f, forward FFT + band pass %.2f, level passes %.2f ms).  PMC traffic of both kernels %.2f GB per step = %.2f x the\n"
    "algorithmic bytes (`profiles/traffic.json`, %s); everything before the synthesis moves %.1f GB.  Checked in the same run:\n"
    "%.1e against the oracle.  `cpu_baseline`: the oracle (a port) on this box's host cores, %.2f Msamples/s with\n"
    "ThreadPool(%d), %.2f with ThreadPool(%d) (what the reference's `parallel=True` starts), %.2f serial."
    % (rnd[1:], rnd, rnd, rl["kernel_ms"], rl["kernels"]["k_synthi"]["ms"], rl["kernels"]["k_synth7"]["ms"], rl["achieved"] / 1e3, rl["frac"],
       rnd, ki[0], ki[1], k7[1], ki[1] + k7[1], j["ms_per_step"], j["value"], j["whole_job_frac_of_hbm_peak"], st["mean_ms"],
       st["fwd_fft_ms"], st["decimate_ms"], rl["traffic"] / 1e9, rl["traffic"] / rl["algorithmic_bytes"], tr["round"],
       tr["non_synth_hbm_bytes_per_step"] / 1e9, j["check"]["worst_rel_err"], cb["value"], cb["cores"],
       cb["legs"]["config2_threadpool_all_cores"]["value"], cb["legs"]["config2_threadpool_all_cores"]["threads"],
       cb["legs"]["config2_serial"]["value"]))
gen["others"] = (
    "complex output %.2f ms per step = %.0f Msamples/s (804 B per channel-sample: %.3f of the peak for the launch), checked\n"
    "%.1e; config 5 per GPU (48 ch x 18e6 @ 30 kHz x 200 scales, streamed; `profiles/%s_bench_config5.json`) %.1f ms per step =\n"
    "%.0f Msamples/s, synthesis **%.3f** of the peak, checked %.1e; config 2 (1 channel) device-resident %.3f ms of device time,\n"
    "`transform()` + the whole amplitude on the host %.1f ms float64 / %.1f ms float32 (the call alone %.1f ms, the call + one\n"
    "second of every scale %.1f ms); config 1 (16 384 samples x 32 scales) %.0f us per execute; `Morse(3, 2)` at the headline\n"
    "shape %.1f ms per step."
    % (cx["ms_per_step"], cx["value"], cx["kernel_frac_of_hbm_peak"], cx["worst_rel_err"], rnd, j5["ms_per_step"], j5["value"],
       j5["roofline"]["frac"], j5["check"]["worst_rel_err"], c2["device_resident"]["device_ms"], te["float64"]["ms_per_call"],
       te["float32"]["ms_per_call"], te["float64"]["transform_returns_ms"], te["float64"]["transform_plus_1s_slice_ms"],
       j["other_configs"]["config1"]["device_resident"]["us_per_call"], j["other_configs"]["heavy_tailed_wavelet"]["ms_per_step"]))
b = "profiles/%s_bench.json" % rnd
b5 = "profiles/%s_bench_config5.json" % rnd
rows = [(b, "value", "%.2f" % j["value"]), (b, "ms_per_step", "%.4f" % j["ms_per_step"]), (b, "roofline.frac", "%.4f" % rl["frac"]),
        (b, "roofline.kernel_ms", "%.4f" % rl["kernel_ms"]), (b, "roofline.kernels.k_synthi.ms", "%.4f" % rl["kernels"]["k_synthi"]["ms"]),
        (b, "roofline.kernels.k_synth7.ms", "%.4f" % rl["kernels"]["k_synth7"]["ms"]), (b, "roofline.traffic", "%d" % rl["traffic"]),
        (b, "stages_ms.fwd_fft_ms", "%.4f" % st["fwd_fft_ms"]), (b, "other_modes.complex.ms_per_step", "%.4f" % cx["ms_per_step"]),
        (b, "other_modes.complex.value", "%.2f" % cx["value"]),
        (b, "other_configs.config2.device_resident.device_ms", "%.4f" % c2["device_resident"]["device_ms"]),
        (b, "other_configs.config5.ms_per_step", "%.4f" % c5["ms_per_step"]),
        (b, "other_configs.config5.roofline.frac", "%.4f" % c5["roofline"]["frac"]), (b, "cpu_baseline.value", "%.4f" % j["cpu_baseline"]["value"]),
        (b, "other_configs.config2.transform_end_to_end.float64.ms_per_call", "%.2f" % te["float64"]["ms_per_call"]),
        (b, "other_configs.config2.transform_end_to_end.float32.ms_per_call", "%.2f" % te["float32"]["ms_per_call"]),
        (b, "stages_ms.decimate_ms", "%.4f" % st["decimate_ms"]),
        ("profiles/traffic.json", "non_synth_hbm_bytes_per_step", "%d" % tr["non_synth_hbm_bytes_per_step"]),
        (b5, "ms_per_step", "%.4f" % j5["ms_per_step"]), (b5, "value", "%.2f" % j5["value"]), (b5, "roofline.frac", "%.4f" % j5["roofline"]["frac"]),
        ("profiles/traffic.json", "k_synth_hbm_bytes_per_launch", "%d" % tr["k_synth_hbm_bytes_per_launch"])]
# README.md: the measured paragraph, from the same files
oc = j["other_configs"]
readme_text = (
    "Measured (round %s, one MI355X," % rnd[1:].lstrip("0") + " `profiles/%s_bench.json`; the boxes of the pool differ by a few per cent): headline\n"
    "128 ch x 1e6 samples x 100 scales, amplitude, device-resident %.0f Msamples/s (%.2f ms per step; the synthesis\n"
    "kernels at %.3f of the 8 TB/s HBM peak on algorithmic bytes, PMC traffic %.2f x those bytes), checked against the\n"
    "oracle in the same run; complex output %.0f Msamples/s; config 5 (48 ch x 18e6 samples @ 30 kHz x 200 scales,\n"
    "streamed in time blocks) %.0f Msamples/s at %.3f; config 2 (1 channel) %.2f ms per execute; config 1 (16 384 samples x\n"
    "32 scales) %.0f us; Morse(3, 2) at the headline shape %.1f ms per step; the oracle on %d of the host's cores %.2f Msamples/s."
    % (rnd, j["value"], j["ms_per_step"], rl["frac"], rl["traffic"] / rl["algorithmic_bytes"], cx["value"], j5["value"], j5["roofline"]["frac"],
       oc["config2"]["device_resident"]["ms_per_step"], oc["config1"]["device_resident"]["us_per_call"],
       oc["heavy_tailed_wavelet"]["ms_per_step"], j["cpu_baseline"]["cores"], j["cpu_baseline"]["value"]))
rpath = os.path.join(root, "README.md")
r = open(rpath).read()
rpat = re.compile(r"(<!-- gen:measured -->).*?(<!-- /gen -->)", re.S)
assert rpat.search(r), "README.md has no generated fragment"
open(rpath, "w").write(rpat.sub(lambda m: m.group(1) + readme_text + m.group(2), r))

path = os.path.join(root, "DESIGN.md")
s = open(path).read()
import textwrap
for name, text in gen.items():
    text = "\n" + textwrap.fill(" ".join(text.split()), 118) + "\n"
    pat = re.compile(r"(<!-- gen:%s -->).*?(<!-- /gen -->)" % re.escape(name), re.S)
    assert pat.search(s), "DESIGN.md has no generated fragment " + name
    s = pat.sub(lambda m: m.group(1) + text + m.group(2), s)
blk = re.compile(r"(<!-- tracked-numbers %s:[^>]*-->\s*```\n).*?(```)" % rnd, re.S)
assert blk.search(s)
s = blk.sub(lambda m: m.group(1) + "\n".join(" | ".join(r) for r in rows) + "\n" + m.group(2), s)
open(path, "w").write(s)
print("DESIGN.md: %d fragments and %d tracked numbers rewritten from profiles/%s_*" % (len(gen), len(rows), rnd))

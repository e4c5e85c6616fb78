"""profiles/traffic.json and the round's rocprof summaries from the newest files that
tools/prof_round.sh left under gpurun_out/<round>/.  Usage: python tools/traffic.py r01"""
import collections, csv, glob, json, os, re, shutil, sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = os.path.join(root, "gpurun_out", rnd)


def newest(pattern):
    files = glob.glob(os.path.join(src, pattern))
    return max(files, key=os.path.getmtime)


def counter(kind):
    f = newest("%s/*/*counter_collection.csv" % kind)
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
        acc[name].append(float(row["Counter_Value"]))
    return {k: {"calls": len(v), "avg_KB": sum(v) / len(v)} for k, v in acc.items()}


fetch, write = counter("fetch"), counter("write")
# the synthesis: k_synth7 (FFT per sample) and k_synthi (interpolating), one launch each per step
keys = [k for k in fetch if "k_synth7" in k or "k_synthi" in k]
f_kb = sum(fetch[k]["avg_KB"] for k in keys)
w_kb = sum(write[k]["avg_KB"] for k in keys)
steps = max(1, fetch[keys[0]]["calls"])
out = {
    "round": rnd,
    "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, bench.py 128ch x "
            "1e6 x 100 scales amplitude; values are KB per launch. On gfx950 FETCH_SIZE counts 64 B "
            "per 128 B request for coalesced streams, so fetched bytes = 2 * FETCH_SIZE * 1024 "
            "(guides/MI355X_MICROARCH.md, HBM section); WRITE_SIZE reads the bytes exactly for the "
            "16-byte-per-lane stores of k_synthi and the 4-byte, 256-B-per-wave stores of k_synth7 "
            "(with the nt policy it reads about 2 % above the bytes stored). The synthesis is the two "
            "kernels together: " + ", ".join(keys),
    "k_synth_hbm_bytes_per_launch": int(2 * f_kb * 1024 + w_kb * 1024),
    "synthesis_kernels": {k: {"fetch_KB": fetch[k]["avg_KB"], "write_KB": write[k]["avg_KB"]} for k in keys},
    "non_synth_hbm_bytes_per_step": int(sum(2 * fetch[k]["avg_KB"] * 1024 * fetch[k]["calls"] for k in fetch if k not in keys) / steps
                                        + sum(write[k]["avg_KB"] * 1024 * write[k]["calls"] for k in write if k not in keys) / steps),
    "counters": {"FETCH_SIZE": fetch, "WRITE_SIZE": write},
}
json.dump(out, open(os.path.join(root, "profiles", "traffic.json"), "w"), indent=1)
shutil.copy(newest("stats/*/*kernel_stats.csv"), os.path.join(root, "profiles", rnd + "_kernel_stats.csv"))
# the tracked line carries the traffic of THIS round's counter passes (bench.py itself can only read the
# traffic.json that was in the tree when it ran, i.e. the previous collection)
line = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
if isinstance(line.get("roofline"), dict):
    line["roofline"]["traffic"] = out["k_synth_hbm_bytes_per_launch"]
    line["roofline"]["traffic_source"] = ("profiles/traffic.json (%s): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                          "command in the same gpurun call as this line (tools/prof_round.sh), both synthesis "
                                          "kernels; filled in by tools/traffic.py" % rnd)
open(os.path.join(root, "profiles", rnd + "_bench.json"), "w").write(json.dumps(line) + "\n")
print("synthesis (%s): fetch %.1f MB (x2 corrected) + write %.1f MB = %d bytes per step" %
      (" + ".join(keys), 2 * f_kb / 1024, w_kb / 1024, out["k_synth_hbm_bytes_per_launch"]))

# config 5: the line and the kernel stats of the same command
import glob as _g
c5 = os.path.join(src, "bench_config5.json")
if os.path.exists(c5) and os.path.getsize(c5) > 0:
    shutil.copy(c5, os.path.join(root, "profiles", rnd + "_bench_config5.json"))
    st5 = _g.glob(os.path.join(src, "stats5/*/*kernel_stats.csv"))
    if st5:
        shutil.copy(max(st5, key=os.path.getmtime), os.path.join(root, "profiles", rnd + "_config5_kernel_stats.csv"))
    print("config 5 line and kernel stats copied")

"""profiles/<round>_pmc_<kernel>.json from the three counter passes tools/pmc.sh <tag> left
under gpurun_out/pmc_<tag>_{1,2,3}.   Usage: python tools/pmc_summary.py <tag> <round> [kernel]"""
import collections, csv, glob, json, os, sys
tag, rnd = sys.argv[1], sys.argv[2]
kern = sys.argv[3] if len(sys.argv) > 3 else "k_synth7<0, 32, false>"
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
vals = {}
for d in ("1", "2", "3"):
    files = sorted(glob.glob(os.path.join(root, "gpurun_out", "pmc_%s_%s" % (tag, d), "*", "*counter_collection.csv")), key=os.path.getmtime)
    for f in files[-1:]:                  # the newest run of this pass
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if kern.replace(" ", "") in row["Kernel_Name"].replace(" ", ""):
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            vals[k] = sum(v) / len(v)
cyc = vals["GRBM_GUI_ACTIVE"] / 8
wc = vals["SQ_WAVE_CYCLES"]
out = {
    "kernel": "gcwt::" + kern, "round": rnd,
    "note": "rocprofv3 --pmc passes of tools/pmc.sh (one counter group per pass, kernel-trace only), averaged over the "
            "launches of one tools/stage_times.py run (128 ch x 1e6 x 100 scales, amplitude); SQ counters are summed over "
            "the 8 XCDs; VALU busy = SQ_ACTIVE_INST_VALU (quad-cycles) * 4 / (1024 SIMDs * GRBM_GUI_ACTIVE/8)",
    "counters": vals,
    "derived": {
        "gpu_cycles_per_launch": cyc,
        "valu_busy_pct": 100 * vals["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc,
        "wait_any_pct_of_wave_cycles": 100 * vals["SQ_WAIT_ANY"] / wc,
        "wait_inst_any_pct_of_wave_cycles": 100 * vals["SQ_WAIT_INST_ANY"] / wc,
        "active_inst_any_pct_of_wave_cycles": 100 * vals["SQ_ACTIVE_INST_ANY"] / wc,
        "lds_idx_active_pct_per_cu": 100 * vals["SQ_LDS_IDX_ACTIVE"] / 256 / cyc,
        "lds_bank_conflict_share_pct": 100 * vals["SQ_LDS_BANK_CONFLICT"] / max(1.0, vals["SQ_LDS_IDX_ACTIVE"]),
        "valu_insts_per_wave": vals["SQ_INSTS_VALU"] / vals["SQ_WAVES"],
        "lds_insts_per_wave": vals["SQ_INSTS_LDS"] / vals["SQ_WAVES"],
    },
}
path = os.path.join(root, "profiles", "%s_pmc_%s.json" % (rnd, kern.split("<")[0]))
json.dump(out, open(path, "w"), indent=1)
print(path, json.dumps(out["derived"], indent=1))

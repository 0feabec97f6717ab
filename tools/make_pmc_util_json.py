"""Build profiles/rNN_pmc_wave_states.json (MFMA utilisation + wave states of the dominant kernel) from ONE rocprofv3
counter pass:  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --kernel-trace -- python3 tools/bench_conv.py --only=12,13
usage: make_pmc_util_json.py <dir> <out.json> <kernel-substring> [<kernel-substring-2> <out2.json>]"""
import collections, csv, glob, json, sys

d, out_path, needle = sys.argv[1:4]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if needle in r["Kernel_Name"]:
        agg[(r["Kernel_Name"].split("(")[0][:90], int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"kernel": needle,
       "method": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY "
                 "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --kernel-trace over `python3 tools/bench_conv.py --only=12,13` (counters "
                 "only, no other trace domain); GRBM_GUI_ACTIVE is summed over the 8 XCDs; mfma_util = MFMA-busy SIMD-cycles / "
                 "(kernel cycles x 256 CUs x 4 SIMDs); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed "
                 "over waves (MI355X_MICROARCH.md)",
       "launches": {}}
for (k, g), c in sorted(agg.items(), key=lambda kv: kv[0][1]):
    v = {n: sum(x) / len(x) for n, x in c.items()}
    v["launches_averaged"] = len(next(iter(c.values())))
    cyc = v["GRBM_GUI_ACTIVE"] / 8.0
    v["kernel_cycles_per_xcd"] = cyc
    v["mfma_util"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 256 * 4)
    if "SQ_WAVE_CYCLES" in v and "SQ_WAIT_ANY" in v:
        w = v["SQ_WAVE_CYCLES"]
        v["frac_wave_parked (s_waitcnt / barrier)"] = v["SQ_WAIT_ANY"] / w
        v["frac_issue_stalled (MFMA pipe / dependency)"] = v["SQ_WAIT_INST_ANY"] / w
        v["frac_issue_stalled_on_LDS"] = v["SQ_WAIT_INST_LDS"] / w
        v["frac_issuing"] = v["SQ_ACTIVE_INST_ANY"] / w
    res["launches"][f"{k} grid_threads={g}"] = v
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps(res, indent=1))

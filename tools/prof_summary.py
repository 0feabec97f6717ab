"""Summarise a rocprofv3 --kernel-trace --stats CSV directory: per-kernel totals, kernel families and
per-(kernel, grid) averages.  usage: python tools/prof_summary.py <dir> [n_steps_in_run] [--grid]"""
import collections, csv, glob, re, sys
d = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms over the run = {tot/1e6/steps:.2f} ms/step at {steps} steps; "
      f"{calls/steps:.0f} launches/step, mean {tot/1e3/calls:.1f} us")
FAM = [("conv_halo", r"k_conv_halo"), ("wgrad_halo", r"k_wgrad_halo"), ("conv_gemm2", r"k_conv_gemm2|k_conv_img"),
       ("conv wgrad (small) + slab reduce", r"k_conv_wgrad2|k_slab_reduce|k_conv_smallk|k_conv_1ch|k_conv_thin|k_conv_direct"),
       ("BN family", r"k_bn_|k_stripreduce"), ("SE", r"k_se_"), ("depthwise", r"k_dw5"),
       ("SN + prep + Adamax", r"k_sn_|k_weight_prep|k_adamax"), ("RCCL", r"[nr]ccl(?!r)|Ccl|CCL"),
       ("loss / sampler / elementwise", r".")]
fam = collections.OrderedDict((n, [0, 0.0]) for n, _ in FAM)
for r in rows:
    for n, pat in FAM:
        if re.search(pat, r["Name"]):
            fam[n][0] += int(r["Calls"]); fam[n][1] += float(r["TotalDurationNs"])
            break
for n, (c, t) in fam.items():
    print(f"  family {n:36s} {c/steps:7.1f} launches/step {t/1e6/steps:7.3f} ms/step avg {t/1e3/max(c,1):7.1f} us")
js = [a for a in sys.argv if a.startswith("--json=")]
if js:
    import json
    with open(js[0].split("=", 1)[1], "w") as fh:
        json.dump({"source": "rocprofv3 --kernel-trace --stats of bench.py", "steps_in_run": steps,
                   "kernel_time_ms_per_step": round(tot / 1e6 / steps, 3), "launches_per_step": round(calls / steps, 1),
                   "families": {n: {"launches_per_step": round(c / steps, 1), "ms_per_step": round(t / 1e6 / steps, 3),
                                    "avg_us": round(t / 1e3 / max(c, 1), 1)} for n, (c, t) in fam.items()}}, fh, indent=1)
for r in rows[:40]:
    print(f'{r["Name"][:70]:70s} calls={int(r["Calls"]):6d} total={float(r["TotalDurationNs"])/1e6/steps:8.3f} ms/step avg={float(r["AverageNs"])/1e3:8.1f} us {float(r["Percentage"]):5.1f}%')
if "--grid" in sys.argv:
    t = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(t)):
        k = (r["Kernel_Name"][:52], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]))
        agg[k][0] += 1
        agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
        print(f"{k[0]:54s} blocks=({k[1]},{k[2]}) calls={v[0]:5d} total={v[1]/1e3/steps:8.3f} ms/step avg={v[1]/v[0]:8.1f} us")

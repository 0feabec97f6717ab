"""Summarise a rocprofv3 --kernel-trace --stats CSV directory: per-kernel totals and per-(kernel, grid)
averages.  usage: python tools/prof_summary.py <dir> [n_steps_in_run]"""
import collections, csv, glob, sys
d = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms over the run = {tot/1e6/steps:.2f} ms/step at {steps} steps")
for r in rows[:28]:
    print(f'{r["Name"][:70]:70s} calls={int(r["Calls"]):6d} total={float(r["TotalDurationNs"])/1e6/steps:8.3f} ms/step avg={float(r["AverageNs"])/1e3:8.1f} us {float(r["Percentage"]):5.1f}%')
if "--grid" in sys.argv:
    t = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(t)):
        k = (r["Kernel_Name"][:44], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]))
        agg[k][0] += 1
        agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"{k[0]:46s} blocks=({k[1]},{k[2]}) calls={v[0]:5d} total={v[1]/1e3/steps:8.3f} ms/step avg={v[1]/v[0]:8.1f} us")

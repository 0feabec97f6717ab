"""Where does a data-parallel step lose time on ONE GPU?  Reads a rocprofv3 --kernel-trace CSV of
`bench.py --force-dp` and prints, for the RCCL kernels of a few steps in the middle of the DP-configured run, the idle
time before and after each (gap to the previous / next kernel on any stream), plus the largest idle gaps of a step.
usage: python tools/dp_gaps.py <dir with *kernel_trace.csv>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
import re
is_ccl = lambda n: re.search(r"nccl|rccl|oneRankReduce", n, re.I) is not None and "rocclr" not in n
adamax = [i for i, r in enumerate(rows) if "k_adamax" in r[2]]
print(f"{len(rows)} kernels, {len(adamax)} optimizer steps, {sum(is_ccl(r[2]) for r in rows)} RCCL kernels")
# per step (adamax end to adamax end): wall time, time with NO kernel running on any stream, kernels
steps = []
for a, b in zip(adamax[:-1], adamax[1:]):
    busy_end, idle, gaps = rows[a][1], 0.0, []
    for i in range(a + 1, b + 1):
        s_, e_, n_ = rows[i]
        if s_ > busy_end:
            idle += (s_ - busy_end) / 1e3
            gaps.append(((s_ - busy_end) / 1e3, rows[i - 1][2][:48], n_[:48]))
        busy_end = max(busy_end, e_)
    ccl = [(rows[i][1] - rows[i][0]) / 1e3 for i in range(a + 1, b + 1) if is_ccl(rows[i][2])]
    steps.append(((rows[b][1] - rows[a][1]) / 1e3, idle, b - a, len(ccl), sorted(gaps, reverse=True)[:6], sum(ccl)))
for i, (wall, idle, n, nccl, gaps, ccl_us) in enumerate(steps):
    print(f"step {i:2d}: wall {wall:9.1f} us  chip idle {idle:8.1f} us  kernels {n:5d}  rccl kernels {nccl} ({ccl_us:7.1f} us)")
for i in (len(steps) // 4, 3 * len(steps) // 4):
    print(f"--- largest idle gaps of step {i}:")
    for g in steps[i][4]:
        print(f"    {g[0]:7.1f} us   after {g[1]:48s} before {g[2]}")

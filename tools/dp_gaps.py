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
is_ccl = lambda n: "ccl" in n.lower()
adamax = [i for i, r in enumerate(rows) if "k_adamax" in r[2]]
print(f"{len(rows)} kernels, {len(adamax)} optimizer steps, {sum(is_ccl(r[2]) for r in rows)} RCCL kernels")
# steps of the DP-configured model are the ones that contain RCCL kernels
steps = []
for a, b in zip(adamax[:-1], adamax[1:]):
    n_ccl = sum(is_ccl(rows[i][2]) for i in range(a + 1, b + 1))
    steps.append((a + 1, b, n_ccl, (rows[b][1] - rows[a][1]) / 1e3))
plain = [s for s in steps if s[2] == 0]
dp = [s for s in steps if s[2] > 0]
med = lambda v: sorted(v)[len(v) // 2] if v else float("nan")
print(f"step wall (adamax end to adamax end): plain median {med([s[3] for s in plain]):.1f} us over {len(plain)}, "
      f"DP-configured median {med([s[3] for s in dp]):.1f} us over {len(dp)}")
for kind, sel in (("plain", plain), ("DP", dp)):
    if len(sel) < 3:
        continue
    lo, hi = sel[len(sel) // 2][:2]
    busy_end = rows[lo][0]
    idle, gaps = 0.0, []
    for i in range(lo, hi + 1):
        s, e, n = rows[i]
        if s > busy_end:
            idle += (s - busy_end) / 1e3
            gaps.append(((s - busy_end) / 1e3, rows[i - 1][2][:50], n[:50]))
        busy_end = max(busy_end, e)
    gaps.sort(reverse=True)
    print(f"--- {kind} step: {hi - lo + 1} kernels, chip completely idle for {idle:.1f} us in {len(gaps)} gaps; the largest:")
    for g in gaps[:8]:
        print(f"    {g[0]:7.1f} us   after {g[1]:50s} before {g[2]}")
    if kind == "DP":
        for i in range(lo, hi + 1):
            if is_ccl(rows[i][2]):
                s, e, n = rows[i]
                prev_end = max(r[1] for r in rows[max(lo, i - 40):i]) if i > lo else s
                nxt = rows[i + 1][0] if i + 1 <= hi else e
                print(f"    RCCL {n[:40]:40s} dur {(e - s) / 1e3:7.1f} us, starts {(s - prev_end) / 1e3:7.1f} us after the last earlier "
                      f"kernel ended, next kernel starts {(nxt - e) / 1e3:7.1f} us after it ends")

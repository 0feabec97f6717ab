"""Build profiles/rNN_pmc_dominant.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over
`python tools/bench_conv.py --only=12,13`.  usage: make_pmc_json.py <fetch_dir> <write_dir> <out.json> <kernel-substring>"""
import collections, csv, glob, json, sys
fdir, wdir, out_path, needle = sys.argv[1:5]
out = {}
for d, name in ((fdir, "FETCH_SIZE"), (wdir, "WRITE_SIZE")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
    for (k, g), v in agg.items():
        if needle in k:
            out.setdefault(g, {})[name] = sum(v) / len(v)
            out[g]["kernel"] = k.split("(")[0][:80]
shapes = {16: ("B128 16x16 384->384 5x5", (128 * 16 * 16 * 384 * 2) * 2 + 25 * 384 * 384 * 2),
          32: ("B128 32x32 192->192 5x5", (128 * 32 * 32 * 192 * 2) * 2 + 25 * 192 * 192 * 2)}
res = {"kernel": needle, "unit": "bytes per launch",
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python tools/bench_conv.py "
                 "--only=12,13`; counters are in KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half "
                 "the bytes of 16-B/lane streams); Infinity-Cache hits are included in FETCH_SIZE",
       "launches": {}}
for g, v in sorted(out.items(), key=lambda kv: int(kv[0])):
    name, alg = shapes[16] if len(res["launches"]) == 0 else shapes[32]
    t = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
    res["launches"][name] = {"grid_threads": int(g), "FETCH_SIZE_KiB": v["FETCH_SIZE"], "WRITE_SIZE_KiB": v["WRITE_SIZE"],
                             "traffic_bytes": t, "algorithmic_bytes": alg, "ratio": t / alg}
res["traffic_bytes_avg"] = sum(x["traffic_bytes"] for x in res["launches"].values()) / max(len(res["launches"]), 1)
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps(res, indent=1))

"""Condense rocprofv3 outputs (kernel stats + FETCH_SIZE / WRITE_SIZE PMC passes) into small files
that are committed under profiles/<tag>/ :  kernel_stats_top.csv, pmc_traffic.json"""
import collections, csv, glob, json, os, sys

out, tag = sys.argv[1], sys.argv[2]
res = {"tag": tag, "note": "FETCH_SIZE/WRITE_SIZE are in KiB per rocprofv3; gfx950 FETCH_SIZE reads 1/2 of a wide coalesced "
       "stream (MI355X_MICROARCH.md, HBM) -> fetch_bytes_corrected = 2 * FETCH_SIZE * 1024", "kernels": {}}
stats = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(out, "kernel_stats_top.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
        for r in rows[:60]:
            w.writerow([r["Name"][:160], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]])
for key, sub, fn in (("fetch", "pmc_fetch", "FETCH_SIZE"), ("write", "pmc_write", "WRITE_SIZE")):
    files = glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] != fn:
            continue
        k = r["Kernel_Name"]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    for k, (c, v) in agg.items():
        e = res["kernels"].setdefault(k[:200], {})
        e[key + "_launches"] = c
        e[key + "_kib_total"] = round(v, 1)
top = sorted(res["kernels"].items(), key=lambda kv: -(kv[1].get("fetch_kib_total", 0) + kv[1].get("write_kib_total", 0)))[:60]
res["kernels"] = dict(top)
json.dump(res, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
print("summaries written to", out)

"""Condense rocprofv3 outputs (kernel stats + FETCH_SIZE / WRITE_SIZE PMC passes) into small files that are committed
under profiles/<tag>/ :  kernel_stats_top.csv, pmc_traffic.json.

    python3 tools/summarize_profiles.py <dir> <tag> <builds in each PMC run>

HBM bytes per kernel = FETCH_SIZE * f + WRITE_SIZE (rocprofv3 reports KiB).  f = 2 for STREAMING kernels -- gfx950 counts a
wide coalesced read at half its bytes (MI355X_MICROARCH.md, section HBM) -- and 1 for gather kernels, whose access widths the
guide calls uncalibrated (the correction is only established for 16-byte-per-lane streams)."""
import collections
import csv
import glob
import json
import os
import sys

out, tag = sys.argv[1], sys.argv[2]
builds = int(sys.argv[3]) if len(sys.argv) > 3 else 1
STREAMING = ("k_rs_scatter", "k_rs_hist", "k_scan_tiles", "k_scan_tile_sums", "k_start_bits", "k_byte_hist", "k_rs_chunk_sums",
             "k_rs_tile_offsets", "k_reduce", "PackRunsFn", "DiffFn", "k_sm_sums")      # (k_sm_merge / k_sm_wide read 4-8 bytes per lane
                                                                                       # forward and gather a few words per segment: counted as gather kernels, factor 1)
res = {"tag": tag, "builds_profiled": builds,
       "note": "per kernel over the whole PMC run (builds_profiled builds of the bench workload): FETCH_SIZE/WRITE_SIZE in KiB as rocprofv3 "
               "reports them; hbm_bytes_total = fetch_factor * FETCH * 1024 + WRITE * 1024 with fetch_factor 2 for streaming kernels "
               "(gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md) and 1 for gather kernels (uncalibrated widths)", "kernels": {}}
stats = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(out, "kernel_stats_top.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
        for r in rows[:80]:
            w.writerow([r["Name"][:200], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]])
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
        e = res["kernels"].setdefault(k[:240], {})
        e[key + "_launches"] = c
        e[key + "_kib_total"] = round(v, 1)
for k, e in res["kernels"].items():
    f = 2 if any(s in k for s in STREAMING) else 1
    e["fetch_factor"] = f
    e["hbm_bytes_total"] = round((f * e.get("fetch_kib_total", 0.0) + e.get("write_kib_total", 0.0)) * 1024)
res["kernels"] = {k: v for k, v in res["kernels"].items() if not k.startswith("void at::")}      # (the workload generator's torch kernels)
# (ALL kernels, largest first: bench.py's traffic_stale reads "not in this file" as "not launched by the workload")
res["kernels"] = dict(sorted(res["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_total"]))
json.dump(res, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
print("summaries written to", out)

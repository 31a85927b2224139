#!/usr/bin/env python3
"""Per-kernel sums of the counters in a rocprofv3 --pmc counter_collection.csv (and the dispatches' duration from the start / end
timestamps).  Usage: tools/pmc_summary.py <csv> [substring of the kernel name ...]"""
import csv
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void prim::", "").replace("prim::", "")
    return name[:110]


def main():
    path = sys.argv[1]
    filt = sys.argv[2:]
    sums = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(dict)
    with open(path) as f:
        for row in csv.DictReader(f):
            k = short(row["Kernel_Name"])
            if filt and not any(x in k for x in filt):
                continue
            sums[k][row["Counter_Name"]] += float(row["Counter_Value"])
            disp[k][row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"]), int(row["Grid_Size"]))
    counters = sorted({c for v in sums.values() for c in v})
    print("%-112s %6s %10s %14s " % ("kernel", "calls", "ms", "threads") + " ".join("%16s" % c[-16:] for c in counters))
    for k, v in sorted(sums.items(), key=lambda kv: -sum(d[0] for d in disp[kv[0]].values())):
        ms = sum(d[0] for d in disp[k].values()) / 1e6
        thr = sum(d[1] for d in disp[k].values())
        print("%-112s %6d %10.3f %14d " % (k, len(disp[k]), ms, thr) + " ".join("%16.4g" % v.get(c, 0) for c in counters))


if __name__ == "__main__":
    main()

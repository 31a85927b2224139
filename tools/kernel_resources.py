#!/usr/bin/env python3
"""Registers / scratch / occupancy / LDS of every kernel from `hipcc ... -Rpass-analysis=kernel-resource-usage 2> file`.
  tools/kernel_resources.py NEW [OLD]     lists the kernels that spill to scratch, and (with OLD) the ones whose VGPR count moved
by >= 4 or whose occupancy changed.  (An out-of-line path nobody executes can still cost the hot path its registers: the level-0
hashing kernel went from 105 to 148 VGPRs + 328 bytes of scratch -- 46 -> 122 ms -- when the long-phrase code grew.)"""
import re
import subprocess
import sys


def parse(path):
    out, cur = {}, None
    for line in open(path):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            out[cur] = {}
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur:
                out[cur][key] = int(m.group(1))
    return out


def demangle(names):
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
        return dict(zip(names, r.stdout.split("\n")))
    except OSError:
        return {n: n for n in names}


def main():
    new = parse(sys.argv[1])
    old = parse(sys.argv[2]) if len(sys.argv) > 2 else {}
    dn = demangle(list(new))
    print("kernels with scratch:")
    for k, v in new.items():
        if v.get("scratch", 0) > 0:
            print("  vgpr %3d scratch %4d occ %d  %s" % (v.get("vgpr", 0), v["scratch"], v.get("occ", 0), dn[k][:150]))
    if old:
        print("moved (old -> new):")
        for k, v in new.items():
            o = old.get(k)
            if o and (abs(o.get("vgpr", 0) - v.get("vgpr", 0)) >= 4 or o.get("occ") != v.get("occ") or o.get("scratch") != v.get("scratch")):
                print("  vgpr %3d -> %3d  occ %d -> %d  scratch %d -> %d  %s" % (o.get("vgpr", 0), v.get("vgpr", 0), o.get("occ", 0), v.get("occ", 0),
                                                                                o.get("scratch", 0), v.get("scratch", 0), dn[k][:140]))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Lease script: behind which launch sites does the host wait during a build?  (GRLBWT_SYNC_SITES=1: the engine's profile
then carries one "@sync_after:<site>" entry per host synchronisation, named by the last launch before it.)
  python tools/gpu_sync_sites.py [reads] [read_len]        default: 1,000,000 x 100 bp (BASELINE configs[1], 101 MB)"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["GRLBWT_SYNC_SITES"] = "1"


def main():
    import torch
    import __graft_entry__ as g
    from grlbwt_amd import engine, workloads
    reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    rl = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    dev = torch.device("cuda", 0)
    text = workloads.uniform_reads_torch(reads, rl, seed=20260001, device=dev)
    lib = g.build_hip()
    with engine.Context(0, 0, lib) as ctx:
        for rep in range(2):
            ctx.profile_enable(rep == 1)
            ctx.attach_device(text.data_ptr(), text.numel(), 1, keepalive=text)
            ctx.build()
        prof = ctx.profile()
    syncs = collections.Counter()
    launches = collections.Counter()
    for k, (c, ms, nb) in prof.items():
        if k.startswith("@sync_after:"):
            syncs[k[len("@sync_after:"):].split("#")[0]] += c
        elif not k.startswith("@"):
            launches[k.split("#")[0]] += c
    print("host synchronisations: %d, launches: %d" % (sum(syncs.values()), sum(launches.values())))
    for k, c in syncs.most_common():
        print("  sync after %-34s %4d" % (k, c))
    print("launches by site:")
    for k, c in launches.most_common(40):
        print("  %-34s %4d" % (k, c))


if __name__ == "__main__":
    main()

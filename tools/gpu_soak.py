#!/usr/bin/env python3
"""Lease script: the same inputs built over and over, every image's md5 compared with the first (races show as rare mismatches).
  python tools/gpu_soak.py [builds]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import __graft_entry__ as g
    from grlbwt_amd import dist as gdist
    from grlbwt_amd import engine, workloads
    lib = g.build_hip()
    builds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda", 0)
    texts = {
        "illumina_1GB": (workloads.sampled_reads_torch(6622517, 150, 33000000, seed=20260003, device=dev), 1),
        "uniform_101MB": (workloads.uniform_reads_torch(1000000, 100, seed=20260001, device=dev), 1),
        "repetitive_240MB": (workloads.repetitive_copies_torch(20, 12000000, device=dev), 1),
    }
    bad = 0
    for name, (t, w) in texts.items():
        first = None
        with engine.Context(0, 0, lib) as ctx:
            for k in range(builds):
                ctx.attach_device(t.data_ptr(), t.numel(), w, keepalive=t)
                ctx.build()
                nb, nr = ctx.result_size()
                md5 = workloads.md5_device(gdist._view(ctx.result_device_ptr(), nb, dev))
                if first is None:
                    first = md5
                elif md5 != first:
                    bad += 1
                    print("MISMATCH", name, k, md5, first, flush=True)
        print("%-18s %d builds, md5 %s" % (name, builds, first), flush=True)
    print("soak done, mismatches:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

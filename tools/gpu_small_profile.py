#!/usr/bin/env python3
"""Lease script: BASELINE configs[1] (1 M x 100 bp uniform reads, 101 MB) -- per-site kernel time, launches and host
synchronisations of one build (HIP events on the engine's stream), after warm-up builds; and the wall time of a build."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import __graft_entry__ as g  # noqa: E402
from grlbwt_amd import engine, workloads  # noqa: E402

t = workloads.uniform_reads_torch(1000000, 100, seed=20260001, device="cuda:0")
torch.cuda.synchronize()
with engine.Context(0, 0, g.build_hip()) as ctx:
    def step():
        ctx.attach_device(t.data_ptr(), t.numel(), 1, keepalive=t)
        ctx.build()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    print("ms per build: %.3f" % ((time.perf_counter() - t0) * 100))
    ctx.profile_enable(True)
    step()
    prof = ctx.profile()
    syncs = prof.pop("@host_sync", (0, 0, 0))[0]
    sites = {}
    for k, (c, ms, nb) in prof.items():
        s = k.partition("#")[0]
        e = sites.setdefault(s, [0, 0.0])
        e[0] += c; e[1] += ms
    print("launches %d, host syncs %d, kernel ms %.3f" % (sum(v[0] for v in sites.values()), syncs, sum(v[1] for v in sites.values())))
    for s, (c, ms) in sorted(sites.items(), key=lambda kv: -kv[1][1])[:70]:
        print("  %-30s %4d launches %8.3f ms" % (s, c, ms))

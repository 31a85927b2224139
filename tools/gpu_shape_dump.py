#!/usr/bin/env python3
"""Lease script: one profiled build of a workload; dumps round/level shapes, stage clocks and the
per-launch-site HIP-event profile as JSON (gpurun_out/shape_<tag>.json).  Used to size kernels."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import __graft_entry__ as g
    from grlbwt_amd import engine, workloads
    kind = sys.argv[1] if len(sys.argv) > 1 else "illumina"
    reads = int(sys.argv[2]) if len(sys.argv) > 2 else 66225166
    tag = sys.argv[3] if len(sys.argv) > 3 else "%s_%d" % (kind, reads)
    dev = torch.device("cuda", 0)
    lib = os.environ.get("GRLBWT_HIP_LIB") or g.build_hip()
    t0 = time.time()
    if kind == "illumina":
        text = workloads.sampled_reads_torch(reads, 150, 330000000, seed=20260003, device=dev)
    else:
        text = workloads.uniform_reads_torch(reads, 100, seed=20260001, device=dev)
    torch.cuda.synchronize()
    print("generated %d bytes in %.1f s" % (text.numel(), time.time() - t0), file=sys.stderr)
    ctx = engine.Context(0, 0, lib)
    for it in range(2):
        if it == 1:
            ctx.profile_enable(True)
        t0 = time.time()
        ctx.attach_device(text.data_ptr(), text.numel(), 1, keepalive=text)
        ctx.build()
        torch.cuda.synchronize()
        print("build %d: %.3f s" % (it, time.time() - t0), file=sys.stderr)
    prof = ctx.profile()
    rounds = []
    while True:
        try:
            rounds.append(ctx.round_info(len(rounds)))
        except engine.GrlbwtError:
            break
    levels = [ctx.level_info(l) for l in range(len(rounds) + 1)]
    out = {"bytes": int(text.numel()), "rounds": rounds, "levels": levels, "counters": ctx.counters(), "memory": ctx.memory_usage(),
           "stats": ctx.stats(), "profile": {k: v for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])}}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "shape_%s.json" % tag), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({"rounds": rounds, "levels": levels, "counters": out["counters"], "memory": out["memory"]}))


if __name__ == "__main__":
    main()

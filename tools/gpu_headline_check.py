#!/usr/bin/env python3
"""GPU lease script: the 10 GB headline image (BASELINE configs[3] on one GPU) built, checked (header, maximal runs), hashed
and decoded back to the input through grlbwt_invert_image's per-run form.  Writes gpurun_out/headline_10GB.json; its
md5 / size / runs are what tests/golden/headline_10GB.json holds (tests/test_gpu_parity.py::test_headline_10GB_round_trip)."""
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
    from grlbwt_amd import dist as gdist
    reads = int(sys.argv[1]) if len(sys.argv) > 1 else 66225166
    genome = int(sys.argv[2]) if len(sys.argv) > 2 else 330000000
    lib = g.build_hip()
    t0 = time.time()
    text = workloads.sampled_reads_torch(reads, 150, genome, seed=20260003, device="cuda:0")
    torch.cuda.synchronize()
    t_gen = time.time() - t0
    res = {"reads": reads, "genome": genome, "input_bytes": int(text.numel()), "generate_s": round(t_gen, 2)}
    with engine.Context(0, 0, lib) as ctx:
        t0 = time.time()
        ctx.attach_device(text.data_ptr(), text.numel(), 1, keepalive=text)
        ctx.build()
        res["build_s"] = round(time.time() - t0, 3)
        nb, nr = ctx.result_size()
        st = ctx.stats()
        img = torch.empty(nb, dtype=torch.uint8, device="cuda:0")
        img.copy_(gdist._view(ctx.result_device_ptr(), nb, torch.device("cuda:0")))
        torch.cuda.synchronize()
        res.update({"image_bytes": nb, "runs": nr, "sb": st["sb"], "fb": st["fb"], "build_peak_bytes": ctx.memory_usage()["peak_live_bytes"]})
    t0 = time.time()
    res["md5"] = workloads.md5_device(img)
    res["md5_s"] = round(time.time() - t0, 2)
    out = torch.zeros_like(text)
    with engine.Context(0, 0, lib) as ctx:
        ctx.profile_enable(True)
        t0 = time.time()
        n = ctx.invert_image(img.data_ptr(), nb, 1, out.data_ptr(), out.numel())
        torch.cuda.synchronize()
        res["invert_s"] = round(time.time() - t0, 3)
        res["invert_peak_bytes"] = ctx.memory_usage()["peak_live_bytes"]
        prof = ctx.profile()
        res["invert_sites_ms"] = {k: round(v[1], 2) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])[:8]}
    res["round_trip_equal"] = bool(n == text.numel() and torch.equal(out, text))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "headline_10GB.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))
    sys.exit(0 if res["round_trip_equal"] else 1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Lease script: BASELINE configs[2] (100 x 24 Mbp repetitive, 2.4 GB) and configs[4] (1 GB of uint16 Zipf tokens) at full
size: build, then invert the image on the device and compare with the input (timings of both)."""
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
    lib = g.build_hip()
    dev = torch.device("cuda", 0)
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    out = {}
    cases = []
    if which in ("both", "rep"):
        cases.append(("configs2_repetitive_100x24Mbp", lambda: workloads.repetitive_copies_torch(100, 24000000, device=dev), 1))
    if which == "chr1":
        # SURVEY 8(d) item 3 at chromosome scale: 100 copies of a 248,956,422 bp sequence (the length of human chr1), 24.9 GB --
        # the arena near device capacity (configs[2] exercised at the 2.4 GB pseudo-chromosome only)
        cases.append(("configs2_repetitive_100x248956422bp", lambda: workloads.repetitive_copies_torch(100, 248956422, device=dev), 1))
    if which in ("both", "tok"):
        cases.append(("configs4_u16_tokens_1GB", lambda: workloads.zipf_tokens_torch(500000000, device=dev), 2))
    for name, gen, w in cases:
        t0 = time.time()
        text = gen()
        torch.cuda.synchronize()
        tg = time.time() - t0
        back = torch.zeros_like(text)
        with engine.Context(0, 0, lib) as ctx:
            for rep in range(2):
                t0 = time.time()
                ctx.attach_device(text.data_ptr(), text.numel(), w, keepalive=text)
                ctx.build()
                torch.cuda.synchronize()
                tb = time.time() - t0
            nb, nr = ctx.result_size()
            rounds = 0
            while True:
                try:
                    ctx.round_info(rounds); rounds += 1
                except engine.GrlbwtError:
                    break
            t0 = time.time()
            n = ctx.invert_image(ctx.result_device_ptr(), nb, w, back.data_ptr(), back.numel())
            torch.cuda.synchronize()
            ti = time.time() - t0
            ok = n == text.numel() and bool(torch.equal(back, text))
            out[name] = {"cells": int(text.numel()), "bytes": int(text.numel()) * w, "generate_s": round(tg, 2), "build_s": round(tb, 3),
                         "MBps": round(text.numel() * w / tb / 1e6, 1), "runs": nr, "n_over_r": round(text.numel() / nr, 2), "rounds": rounds,
                         "image_bytes": nb, "invert_s": round(ti, 2), "round_trip_equal": ok, "memory": ctx.memory_usage()}
        print(name, json.dumps(out[name]), flush=True)
        del text, back
        torch.cuda.empty_cache()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "config_roundtrip%s.json" % ("" if which == "both" else "_" + which)), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()

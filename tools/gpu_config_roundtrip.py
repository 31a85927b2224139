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
        back = torch.zeros_like(text) if which != "chr1" else None
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
            if which == "chr1":
                # (the inverter walks a string serially: 100 strings of 249 M cells are 249 M dependent steps -- 20 minutes.  At this
                # size the image is checked through what torch can compute on its bytes: maximal runs, positive lengths that add up
                # to the input, and every symbol's total equal to its count in the text: the BWT is a permutation of the text)
                from grlbwt_amd import dist as gdist
                img = gdist._view(ctx.result_device_ptr(), nb, text.device)
                sb, fb = (int.from_bytes(bytes(img[o:o + 8].cpu().numpy()), "little") for o in (0, 8))
                rec = img[16:].view(nr, sb + fb)
                ok = nb == 16 + nr * (sb + fb)
                sym = torch.zeros(nr, dtype=torch.int64, device=text.device)
                for b in range(sb):
                    sym += rec[:, b].to(torch.int64) << (8 * b)
                ln = torch.zeros(nr, dtype=torch.int64, device=text.device)
                for b in range(fb):
                    ln += rec[:, sb + b].to(torch.int64) << (8 * b)
                ok = ok and bool((sym[1:] != sym[:-1]).all()) and int(ln.min().item()) > 0 and int(ln.sum().item()) == text.numel()
                for sv in torch.unique(sym).tolist():
                    cnt = 0
                    for a in range(0, text.numel(), 1 << 30):
                        cnt += int((text[a:a + (1 << 30)] == sv).sum().item())
                    ok = ok and int(ln[sym == sv].sum().item()) == cnt
                n = text.numel()
                del sym, ln
            else:
                n = ctx.invert_image(ctx.result_device_ptr(), nb, w, back.data_ptr(), back.numel())
                ok = n == text.numel() and bool(torch.equal(back, text))
            torch.cuda.synchronize()
            ti = time.time() - t0
            out[name] = {"cells": int(text.numel()), "bytes": int(text.numel()) * w, "generate_s": round(tg, 2), "build_s": round(tb, 3),
                         "MBps": round(text.numel() * w / tb / 1e6, 1), "runs": nr, "n_over_r": round(text.numel() / nr, 2), "rounds": rounds,
                         "image_bytes": nb, "invert_s": round(ti, 2), ("image_is_a_run_length_permutation_of_the_text" if which == "chr1" else "round_trip_equal"): ok, "memory": ctx.memory_usage()}
        print(name, json.dumps(out[name]), flush=True)
        del text, back
        torch.cuda.empty_cache()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "config_roundtrip%s.json" % ("" if which == "both" else "_" + which)), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()

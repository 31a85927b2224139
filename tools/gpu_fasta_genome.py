#!/usr/bin/env python3
"""Lease script: a genome-style FASTA (few records, sequences wrapped at 60 columns, an N gap) converted on the device and built:
conversion time, build time, refinement rounds.   python tools/gpu_fasta_genome.py [Mbp_per_record] [records] [gap_Mbp]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import __graft_entry__ as g
    from grlbwt_amd import engine
    mbp = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    nrec = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    gap = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    rng = np.random.default_rng(7)
    parts = []
    for r in range(nrec):
        seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=mbp * 1000000 + 37 * r)
        seq[len(seq) // 3: len(seq) // 3 + gap * 1000000] = ord("N")
        rows = (len(seq) + 59) // 60
        pad = np.full(rows * 60, 0, dtype=np.uint8)
        pad[:len(seq)] = seq
        lines = np.concatenate([pad.reshape(rows, 60), np.full((rows, 1), 10, dtype=np.uint8)], axis=1).reshape(-1)
        lines = lines[lines != 0]
        parts.append(np.frombuffer((">chr%d some description\n" % (r + 1)).encode(), dtype=np.uint8))
        parts.append(lines)
    fasta = np.concatenate(parts)
    lib = g.build_hip()
    src = torch.from_numpy(fasta).to("cuda:0")
    dst = torch.zeros(fasta.size + 16, dtype=torch.uint8, device="cuda:0")
    with engine.Context(0, 0, lib) as ctx:
        torch.cuda.synchronize()
        t0 = time.time()
        n_out, n_str = ctx.fastx_convert(src.data_ptr(), src.numel(), False, dst.data_ptr(), dst.numel())
        torch.cuda.synchronize()
        tc = time.time() - t0
        text = dst[:n_out]
        for rep in range(2):                       # (the second build: the process's arena is backed by then)
            ctx.profile_enable(rep == 1 and bool(os.environ.get("GRLBWT_FASTA_PROFILE")))
            t0 = time.time()
            ctx.attach_device(text.data_ptr(), n_out, 1, keepalive=text)
            ctx.build()
            torch.cuda.synchronize()
            tb = time.time() - t0
        cnt = ctx.counters()
        if os.environ.get("GRLBWT_FASTA_PROFILE"):
            for k, (c, ms, nb) in sorted(ctx.profile().items(), key=lambda kv: -kv[1][1])[:10]:
                print("  %-32s %4d %10.2f ms" % (k, c, ms))
        iters = []
        r = 0
        while True:
            try:
                iters.append(ctx.round_info(r)["sort_iters"])
                r += 1
            except engine.GrlbwtError:
                break
        nb, nr = ctx.result_size()
    print("FASTA %.1f MB, %d records, %d Mbp N gap each -> %d cells in %d strings: convert %.3f s, build %.3f s (%.1f MB/s; hashing %.2f s), runs %d, refinement rounds %s"
          % (fasta.size / 1e6, nrec, gap, n_out, n_str, tc, tb, n_out / 1e6 / tb, cnt["t_hash"], nr, iters))


if __name__ == "__main__":
    main()

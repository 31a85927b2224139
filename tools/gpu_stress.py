#!/usr/bin/env python3
"""Lease script: inputs chosen to hit slow paths (long runs, tandem repeats, monotone ramps over a large alphabet, one giant string,
millions of identical strings, one symbol only).  Each is built on the device and inverted back (device round trip); prints the
build time, the refinement rounds per level and the verdict.   python tools/gpu_stress.py [case ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cases():
    rng = np.random.default_rng(2026)
    dna = lambda n: rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n)
    nl = np.array([10], dtype=np.uint8)
    out = {}
    out["tandem_ACG_x700k_5_strings"] = (np.concatenate([np.concatenate([np.tile(np.frombuffer(b"ACG", dtype=np.uint8), 700000 + k), dna(50), nl]) for k in range(5)]), 1)
    out["tandem_period7_x300k"] = (np.concatenate([np.concatenate([np.tile(np.frombuffer(b"ACGGTCA", dtype=np.uint8), 300000 + 3 * k), dna(20), nl]) for k in range(4)]), 1)
    ramp = np.arange(1, 400001, dtype=np.uint32)
    out["u32_shared_ramps_400k_x6"] = (np.concatenate([np.concatenate([ramp, np.array([500000 + k, 0], dtype=np.uint32)]) for k in range(6)]).view(np.uint8), 4)
    out["one_string_64MB"] = (np.concatenate([dna(64 << 20), nl]), 1)
    out["identical_strings_8M_x_ACGTACGT"] = (np.tile(np.frombuffer(b"ACGTACGT\n", dtype=np.uint8), 8 << 20), 1)
    out["one_symbol_100M"] = (np.concatenate([np.full(100 << 20, ord("A"), dtype=np.uint8), nl]), 1)
    tok = rng.integers(1, 300, size=20 << 20).astype(np.uint16)
    tok[5 << 20: 9 << 20] = 77
    tok[(np.arange(1, 40) * (1 << 19))] = 0
    tok[-1] = 0
    out["u16_tokens_with_a_4M_run"] = (tok.view(np.uint8), 2)
    out["tiny_strings_40M_x_1_symbol"] = (np.tile(np.frombuffer(b"A\nC\nG\nT\nA\n", dtype=np.uint8), 8 << 20), 1)
    out["random_bytes_100MB_255_symbols"] = (np.concatenate([rng.integers(1, 256, size=100 << 20).astype(np.uint8), np.zeros(1, dtype=np.uint8)]), 1)
    prot = rng.choice(np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8), size=400 << 20)
    prot[(np.arange(1, 800000) * 524)] = 10
    prot[-1] = 10
    out["protein_like_400MB"] = (prot, 1)
    long_strings = dna(1 << 30)
    long_strings[(np.arange(1, 200) * ((1 << 30) // 200))] = 10
    long_strings[-1] = 10
    out["dna_1GB_in_200_strings"] = (long_strings, 1)
    big = rng.integers(1, 1 << 29, size=200 << 20, dtype=np.int64).astype(np.uint32)
    big[(np.arange(1, 2000) * 100003)] = 0
    big[-1] = 0
    out["u32_random_200M_cells_29_bits"] = (big.view(np.uint8), 4)
    zipf = (rng.zipf(1.3, size=100 << 20) % 50000 + 1).astype(np.uint32)
    zipf[(np.arange(1, 5000) * 20011)] = 0
    zipf[-1] = 0
    out["u32_zipf_100M_cells"] = (zipf.view(np.uint8), 4)
    return out


def main():
    import torch
    import __graft_entry__ as g
    from grlbwt_amd import engine
    lib = g.build_hip()
    want = sys.argv[1:]
    for name, (data, w) in cases().items():
        if want and name not in want:
            continue
        t = torch.from_numpy(np.ascontiguousarray(data)).to("cuda:0")
        view = {1: torch.uint8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[w]
        cells = t.view(view)
        back = torch.zeros_like(cells)
        with engine.Context(0, 0, lib) as ctx:
            t0 = time.time()
            ctx.attach_device(cells.data_ptr(), cells.numel(), w, keepalive=cells)
            ctx.build()
            torch.cuda.synchronize()
            tb = time.time() - t0
            nb, nr = ctx.result_size()
            iters = []
            r = 0
            while True:
                try:
                    iters.append(ctx.round_info(r)["sort_iters"])
                    r += 1
                except engine.GrlbwtError:
                    break
            if os.environ.get("GRLBWT_STRESS_PROFILE"):     # a second (warm) build with the per-site clocks: where does this input spend its time?
                ctx.profile_enable(True)
                t0 = time.time()
                ctx.attach_device(cells.data_ptr(), cells.numel(), w, keepalive=cells)
                ctx.build()
                torch.cuda.synchronize()
                print("  warm build with the profile on: %.3f s; rounds: %s" % (time.time() - t0, [ctx.round_info(k) for k in range(r)]))
                prof = sorted(((ms, c, k) for k, (c, ms, nb_) in ctx.profile().items() if not k.startswith("@")), reverse=True)
                for ms, c, k in prof[:25]:
                    print("    %-36s %5d launches %9.3f ms" % (k, c, ms))
                ctx.profile_enable(False)
            t0 = time.time()
            if os.environ.get("GRLBWT_STRESS_NO_INVERT"):      # (the test inverter walks a string serially: 100 MB in ONE string takes minutes)
                n, ok = cells.numel(), None
            else:
                n = ctx.invert_image(ctx.result_device_ptr(), nb, w, back.data_ptr(), back.numel())
            torch.cuda.synchronize()
            ti = time.time() - t0
        ok = None if os.environ.get("GRLBWT_STRESS_NO_INVERT") else (n == cells.numel() and bool(torch.equal(back, cells)))
        print("%-36s %8.1f MB  build %7.2f s (%8.1f MB/s)  runs %10d  rounds %2d  refinement rounds %s  invert %6.2f s  round trip %s"
              % (name, data.nbytes / 1e6, tb, data.nbytes / 1e6 / tb, nr, len(iters), iters, ti, ok), flush=True)


if __name__ == "__main__":
    main()

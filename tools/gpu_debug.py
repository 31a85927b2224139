#!/usr/bin/env python3
"""Lease script: build one input with the library named by GRLBWT_HIP_LIB and report per-round counters (debugging)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import __graft_entry__ as g  # noqa: E402
from grlbwt_amd import engine, workloads  # noqa: E402

lib = os.environ.get("GRLBWT_HIP_LIB") or g.build_hip()
data = open(os.path.join(ROOT, "tests", "golden", "test_byte_alphabet.txt"), "rb").read()
for flags in (0, engine.FLAG_FORCE_IDX64):
    with engine.Context(0, flags, lib) as ctx:
        ctx.upload(data, 1)
        try:
            for r in range(9):
                info, done = ctx.parse_round()
                print(flags, r, info["n_phrases"], info["dict_syms"], info["n_metasyms"], info["parse_size"], flush=True)
                if done:
                    break
        except engine.GrlbwtError as e:
            print("ERR", flags, e)

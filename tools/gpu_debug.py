#!/usr/bin/env python3
"""Lease script: compare the raw level-1 cells of the 32- and 64-bit index builds (debugging)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["GRLBWT_DBG_RAWCELLS"] = "1"
import torch  # noqa: E402
import __graft_entry__ as g  # noqa: E402
from grlbwt_amd import engine, workloads  # noqa: E402

lib = os.environ.get("GRLBWT_HIP_LIB") or g.build_hip()
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
data = workloads.uniform_reads(reads, 100)
texts = {}
for flags in (0, engine.FLAG_FORCE_IDX64):
    with engine.Context(0, flags | engine.FLAG_KEEP_LEVELS, lib) as ctx:
        ctx.upload(data, 1)
        info, done = ctx.parse_round()
        texts[flags] = ctx.level_text(1)
        info, done = ctx.parse_round()
        print(flags, info["n_phrases"], info["parse_size"])
a, b = texts[0], texts[engine.FLAG_FORCE_IDX64]
d = np.flatnonzero(a != b)
print("len", len(a), len(b), "mismatches", len(d), d[:10])
for i in d[:10]:
    print(i, hex(a[i]), hex(b[i]))
print("T cells", int((a & 1).sum()), int((b & 1).sum()))
ra, rb = a >> 2, b >> 2
for r in np.unique(ra[d])[:6]:
    ia, ib = np.flatnonzero(ra == r), np.flatnonzero(rb == r)
    print("rank", r, "occ32", len(ia), "T32", int((a[ia] & 1).sum()), "occ64", len(ib), "T64", int((b[ib] & 1).sum()), "rep", int(a[ia[0]] >> 1 & 1))
# the phrase itself: i-th phrase of the text
cells = np.frombuffer(data, dtype=np.uint8)
# phrase starts from the oracle-free rule are not at hand: print the neighbourhood of the string end instead
ends = np.flatnonzero(cells == 10)
# the mismatching cell is the LAST phrase of some string: find which string by counting T cells before it
tcount = np.cumsum(a & 1)
for i in d[:4]:
    sidx = int(tcount[i]) - 1
    e = ends[sidx]
    print("string", sidx, "tail", bytes(cells[e - 12:e + 1]))

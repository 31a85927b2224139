#!/usr/bin/env python3
"""Lease script: ONE build of an Illumina-style workload of <reads> reads (for rocprofv3 --pmc runs)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import __graft_entry__ as g  # noqa: E402
from grlbwt_amd import engine, workloads  # noqa: E402

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 6622517
genome = int(sys.argv[2]) if len(sys.argv) > 2 else 33000000
text = workloads.sampled_reads_torch(reads, 150, genome, seed=20260003, device="cuda:0")
torch.cuda.synchronize()
with engine.Context(0, 0, g.build_hip()) as ctx:
    ctx.attach_device(text.data_ptr(), text.numel(), 1, keepalive=text)
    ctx.build()
    print(ctx.result_size())

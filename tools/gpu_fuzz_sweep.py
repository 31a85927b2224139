"""GPU box: extended randomized parity sweep (one-off validation, same generators as tests/parity.py, other seeds,
plus mid-size mixed collections).  usage: gpu_fuzz_sweep.py [rounds]"""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (torch's HIP runtime first)
import __graft_entry__ as g
from grlbwt_amd import engine, workloads
from tests import parity
from oracle import oracle
oracle.build()
lib = g.build_hip()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
for kind in parity.KINDS:
    rng = np.random.default_rng(zlib.crc32(kind.encode()) + 20260101)
    for i in range(rounds):
        data, w = parity.rand_collection(rng, kind)
        flags = engine.FLAG_FORCE_IDX64 if i % 3 == 2 else 0
        try:
            parity.check_final(lib, data, w, flags)
        except AssertionError as e:
            bad += 1
            print("MISMATCH", kind, i, w, flags, len(data), e, flush=True)
# mid-size: reads with long homopolymers and duplicates mixed in (oracle finishes in seconds)
rng = np.random.default_rng(99)
for i in range(6):
    reads = workloads.sampled_reads(int(rng.integers(2000, 8000)), int(rng.integers(30, 200)), int(rng.integers(5000, 60000)), seed=int(rng.integers(1, 1 << 30)))
    extra = b"".join(bytes([b"ACGT"[int(rng.integers(0, 4))]]) * int(rng.integers(1, 5000)) + b"\n" for _ in range(20))
    data = reads.tobytes() + extra + reads.tobytes()[: len(reads) // 3 // 1]
    if not data.endswith(b"\n"):
        data = data[: data.rfind(b"\n") + 1]
    try:
        parity.check_final(lib, data, 1, engine.FLAG_FORCE_IDX64 if i % 2 else 0)
    except AssertionError as e:
        bad += 1
        print("MISMATCH mid", i, len(data), e, flush=True)
print("sweep done, mismatches:", bad)
sys.exit(1 if bad else 0)

"""GPU box: size-independent properties on large configs: idx32 and idx64 builds give the same
image; the BWT is a permutation of the text (per-symbol counts); runs are maximal; header widths.
usage: gpu_scale_check.py reads N | chr COPIES LEN | tokens NCELLS | illumina N_READS GENOME_LEN"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from grlbwt_amd import engine, workloads
from tests import bcr_check as bc

kind = sys.argv[1]
w = 1
if kind == "reads":
    text = workloads.uniform_reads_torch(int(sys.argv[2]), 100, device="cuda:0")
elif kind == "illumina":
    text = workloads.sampled_reads_torch(int(sys.argv[2]), 150, int(sys.argv[3]), device="cuda:0")
elif kind == "chr":
    text = torch.from_numpy(workloads.repetitive_copies(int(sys.argv[2]), int(sys.argv[3]))).to("cuda:0")
elif kind == "tokens":
    text = torch.from_numpy(workloads.zipf_tokens(int(sys.argv[2])).view(np.int16)).to("cuda:0")
    w = 2
torch.cuda.synchronize()
n_cells = text.numel()
host = text.cpu().numpy()
cells = host.view(np.uint16) if w == 2 else host
hist = np.bincount(cells.astype(np.int64), minlength=65536 if w == 2 else 256)
md5 = {}
builds = (("idx32", 0), ("idx64", engine.FLAG_FORCE_IDX64))
if n_cells >= 0xFFFFFF00:
    builds = (("idx64", 0),)                     # too large for 32-bit positions: one build
for name, flags in builds:
    with engine.Context(0, flags) as ctx:
        t0 = time.time()
        ctx.attach_device(text.data_ptr(), n_cells, w, keepalive=text)
        ctx.build()
        dt = time.time() - t0
        blob = ctx.result_bytes()
        rounds = 0
        while True:
            try:
                ri = ctx.round_info(rounds); rounds += 1
            except engine.GrlbwtError:
                break
    md5[name] = hashlib.md5(blob).hexdigest()
    sb, fb, sym, ln = bc.parse_rl_bwt(blob)
    assert (sb, fb) == bc.header_widths(cells, w), (sb, fb)
    assert bc.runs_are_maximal(sym) and int(ln.sum()) == n_cells
    got = np.bincount(sym.astype(np.int64), weights=ln.astype(np.float64), minlength=len(hist)).astype(np.int64)   # exact below 2^53
    assert np.array_equal(got, hist), "BWT is not a permutation of the text"
    print(name, "ok: %.2f s (%.1f MB/s), %d rounds, %d runs (n/r %.2f), sb=%d fb=%d md5=%s"
          % (dt, n_cells * w / 1e6 / dt, rounds, len(sym), n_cells / len(sym), sb, fb, md5[name]), flush=True)
    del blob, sym, ln
assert len(md5) == 1 or md5["idx32"] == md5["idx64"]
print("scale check passed:", kind, n_cells * w, "bytes")

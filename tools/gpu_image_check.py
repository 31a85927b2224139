"""GPU box: size-independent properties of a LARGE build checked entirely on the device through the image consumers
(grlbwt_image_stats_get): the BWT is a permutation of the text (per-symbol totals), its length, maximal runs.
usage: gpu_image_check.py illumina N_READS GENOME_LEN | uniform N_READS"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grlbwt_amd import engine, workloads
kind = sys.argv[1]
if kind == "uniform":
    text = workloads.uniform_reads_torch(int(sys.argv[2]), 100, device="cuda:0")
else:
    text = workloads.sampled_reads_torch(int(sys.argv[2]), 150, int(sys.argv[3]), device="cuda:0")
torch.cuda.synchronize()
n = text.numel()
# the text's symbol histogram, before the engine reserves its slab (chunked: no 8x temporaries)
hist = torch.zeros(256, dtype=torch.int64, device="cuda:0")
step = 1 << 28
for lo in range(0, n, step):
    hist += torch.bincount(text[lo:lo + step].to(torch.int64), minlength=256)
hist = hist.cpu().tolist()
torch.cuda.empty_cache()
with engine.Context(0, 0) as ctx:
    t0 = time.time()
    ctx.attach_device(text.data_ptr(), n, 1, keepalive=text)
    ctx.build()
    dt = time.time() - t0
    nb, nr = ctx.result_size()
    st = ctx.image_stats(ctx.result_device_ptr(), nb)
    assert st["n_runs"] == nr and st["text_size"] == n, (st["n_runs"], nr, st["text_size"], n)
    assert st["non_maximal"] == 0, st["non_maximal"]
    assert st["freq_of"] == hist, "BWT is not a permutation of the text"
    print("image check passed: %d bytes, %.2f s, %d runs (n/r %.2f), per-symbol totals equal the text's, runs maximal; "
          "run lengths min %d max %d, deciles %s" % (n, dt, nr, n / nr, st["min_run"], st["max_run"], st["deciles"]))

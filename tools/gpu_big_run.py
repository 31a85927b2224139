"""GPU box: one build of N x 100 bp uniform reads generated on the device; prints rate and memory."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grlbwt_amd import engine, workloads
reads = int(sys.argv[1])
text = workloads.uniform_reads_torch(reads, 100, device="cuda:0")
torch.cuda.synchronize()
print("text bytes", text.numel(), "torch allocated GB", torch.cuda.memory_allocated() / 1e9, flush=True)
torch.cuda.empty_cache()
with engine.Context(0, 0) as ctx:
    t0 = time.time()
    ctx.attach_device(text.data_ptr(), text.numel(), 1, keepalive=text)
    try:
        ctx.build()
        dt = time.time() - t0
        nb, nr = ctx.result_size()
        print("ok %.2f s  %.1f MB/s  runs %d image %d bytes" % (dt, text.numel() / 1e6 / dt, nr, nb))
        print({k: round(v, 3) for k, v in ctx.counters().items() if k.startswith("t_")})
    except engine.GrlbwtError as e:
        print("FAILED", e)
    print(ctx.memory_usage())

"""GPU box: one build of a large device-generated input; prints rate and memory.
usage: gpu_big_run.py uniform N_READS | illumina N_READS GENOME_LEN"""
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
print("text bytes", text.numel(), flush=True)
torch.cuda.empty_cache()
with engine.Context(0, 0) as ctx:
    t0 = time.time()
    ctx.attach_device(text.data_ptr(), text.numel(), 1, keepalive=text)
    try:
        ctx.build()
        dt = time.time() - t0
        nb, nr = ctx.result_size()
        print("ok %.2f s  %.1f MB/s  runs %d (n/r %.2f) image %d bytes" % (dt, text.numel() / 1e6 / dt, nr, text.numel() / nr, nb))
        print({k: round(v, 3) for k, v in ctx.counters().items() if k.startswith("t_")})
        r = 0
        while True:
            try:
                print("  round", r, ctx.round_info(r)); r += 1
            except engine.GrlbwtError:
                break
        if len(sys.argv) > 4 and sys.argv[4] == "verify":
            out = torch.zeros_like(text)
            n = ctx.invert_image(ctx.result_device_ptr(), nb, 1, out.data_ptr(), out.numel())
            torch.cuda.synchronize()
            print("round trip:", n == text.numel() and bool(torch.equal(out, text)))
    except engine.GrlbwtError as e:
        print("FAILED", e)
    print(ctx.memory_usage())

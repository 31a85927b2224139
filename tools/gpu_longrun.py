import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from grlbwt_amd import engine, workloads
rng = np.random.default_rng(5)
L1, LN = 1_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
copies = []
base_a = workloads.genome(L1, 11); base_b = workloads.genome(L1, 12)
for k in range(10):
    a = base_a.copy(); b = base_b.copy()
    idx = rng.integers(0, L1, size=1000); a[idx] = ord('A')
    copies.append(np.concatenate([a, np.full(LN + k, ord('N'), dtype=np.uint8), b, np.array([10], dtype=np.uint8)]))
data = np.concatenate(copies)
t = torch.from_numpy(data).to("cuda:0"); out = torch.zeros_like(t)
with engine.Context(0, 0) as ctx:
    t0 = time.time()
    ctx.attach_device(t.data_ptr(), t.numel(), 1, keepalive=t)
    ctx.build()
    torch.cuda.synchronize()
    dt = time.time() - t0
    nb, nr = ctx.result_size()
    print("build %.2f s (%.1f MB/s) runs %d" % (dt, data.size / 1e6 / dt, nr), {k: round(v, 2) for k, v in ctx.counters().items() if k.startswith("t_")})
    t0 = time.time()
    n = ctx.invert_image(ctx.result_device_ptr(), nb, 1, out.data_ptr(), out.numel())
    torch.cuda.synchronize()
    print("invert %.2f s, round trip %s" % (time.time() - t0, n == t.numel() and bool(torch.equal(out, t))))

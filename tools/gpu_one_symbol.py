import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import __graft_entry__ as g
from grlbwt_amd import engine
lib = g.build_hip()
n = int(sys.argv[1]) if len(sys.argv) > 1 else (100 << 20)
data = np.concatenate([np.full(n, ord("A"), dtype=np.uint8), np.array([10], dtype=np.uint8)])
t = torch.from_numpy(data).to("cuda:0")
with engine.Context(0, 0, lib) as ctx:
    ctx.profile_enable(True)
    t0 = time.time()
    ctx.attach_device(t.data_ptr(), t.numel(), 1, keepalive=t)
    ctx.build()
    torch.cuda.synchronize()
    print("build", round(time.time() - t0, 2), {k: round(v, 2) for k, v in ctx.counters().items() if k.startswith("t_")})
    prof = ctx.profile()
    for k, (c, ms, nb) in sorted(prof.items(), key=lambda kv: -kv[1][1])[:12]:
        print("  %-32s %4d %10.2f ms" % (k, c, ms))

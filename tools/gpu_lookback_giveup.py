#!/usr/bin/env python3
"""Lease script: the give-up path of pass C's one-walk form ON THE DEVICE.  The development library (tools/build_dev.sh) takes the
tiles' patience from GRLBWT_DEV_LB_PATIENCE: with 0 ticks every tile that has to wait for another one gives up, poisons its status
word (the tiles behind it give up at once), writes nothing, and the host takes count + emit -- the image must still be the oracle's.
  python3 tools/gpu_lookback_giveup.py        -> prints the levels that fell back and the comparison"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DEV = os.path.join(ROOT, "tools", "_build", "libgrlbwt_dev.so")

if len(sys.argv) > 1 and sys.argv[1] == "child":
    from grlbwt_amd import engine, workloads
    from oracle import oracle
    oracle.build()
    ok = True
    for name, data, w in (("reads 20 MB", workloads.sampled_reads(130000, 150, 660000, seed=5), 1),
                          ("repetitive", workloads.repetitive_copies(40, 50000, seed=3), 1),
                          ("tokens", workloads.zipf_tokens(200000, doc_len=500, vocab=20000), 2)):
        with engine.Context(0, 0, DEV) as ctx:
            ctx.upload(data.tobytes(), w)
            ctx.build()
            same = ctx.result_bytes() == oracle.rl_bwt(data.tobytes(), w)
        print("%-12s image == oracle: %s" % (name, same), flush=True)
        ok = ok and same
    sys.exit(0 if ok else 1)

env = dict(os.environ, GRLBWT_DEV_LB_PATIENCE="0", GRLBWT_ASM_ONE_WALK="1", GRLBWT_TABLE_TRACE="1", GRLBWT_QUIET_ENV="1")
p = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True, timeout=600)
gave_up = [l for l in p.stderr.splitlines() if "gave up" in l]
print(p.stdout, end="")
print("levels that gave up and took count + emit: %d" % len(gave_up))
print("\n".join(gave_up[:6]))
sys.exit(p.returncode if p.returncode else (0 if gave_up else 3))

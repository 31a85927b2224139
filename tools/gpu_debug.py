import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from grlbwt_amd import engine, workloads
from oracle import oracle
lib = engine.DEFAULT_LIB
data = workloads.uniform_reads(200, 100).tobytes()
o = oracle.OracleResult(data, 1, trace=True)
for trial in range(3):
    ctx = engine.Context(0, engine.FLAG_KEEP_LEVELS, lib)
    ctx.upload(data, 1)
    info, done = ctx.parse_round()
    cells = ctx.level_text(1)
    sym, rep = o.level_text(1)
    gs, gr = cells >> np.uint64(1), (cells & np.uint64(1)).astype(np.uint8)
    print("trial", trial, "sym mismatches", int((gs != sym).sum()), "rep mismatches", int((gr != rep).sum()), "of", len(sym),
          "rep ones gpu/oracle", int(gr.sum()), int(rep.sum()))
    bad = np.flatnonzero(gs != sym)[:5]
    print("  first bad", bad, gs[bad], sym[bad])
    ctx.close()

"""Shared parity checks: an engine library (HIP on the GPU box, or the serial
test stand-in for CPU logic tests) against the CPU oracle."""
import os

import numpy as np

from grlbwt_amd import engine
from oracle import oracle
from tests import bcr_check as bc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def run_engine(lib, data, cell_bytes=1, flags=0):
    return engine.grl_bwt_algo(data, cell_bytes, 0, flags, lib)


def check_final(lib, data, cell_bytes=1, flags=0):
    got = run_engine(lib, data, cell_bytes, flags)
    exp = oracle.rl_bwt(data, cell_bytes)
    assert got == exp, "rl_bwt differs (%d vs %d bytes)" % (len(got), len(exp))
    return got


def check_stagewise(lib, data, cell_bytes=1, flags=0):
    """Every stage boundary against the oracle: stats, per-round counters, every level's
    parse (rank<<1|rep) and every level's BWT runs, and the final bytes."""
    o = oracle.OracleResult(data, cell_bytes, trace=True)
    with engine.Context(0, flags | engine.FLAG_KEEP_LEVELS, lib) as ctx:
        ctx.upload(data, cell_bytes)
        st = ctx.stats()
        for k in ("n_strings", "n_syms", "min_sym", "max_sym", "max_sym_freq", "sb", "fb"):
            assert st[k] == o.stats[k], (k, st[k], o.stats[k])
        r = 0
        while True:
            info, done = ctx.parse_round()
            oc = o.counters(r)
            assert (info["n_in"], info["n_phrases"], info["dict_syms"], info["n_metasyms"], info["parse_size"], info["sigma"]) == \
                   (oc["n_in"], oc["D"], oc["S"], oc["M"], oc["parse_size"], oc["sigma"]), (r, info, oc)
            sym, rep = o.level_text(r + 1)
            cells = ctx.level_text(r + 1)
            assert np.array_equal(cells, (sym << np.uint64(1)) | rep.astype(np.uint64)), "parse of level %d differs" % (r + 1)
            r += 1
            if done:
                break
        assert r == o.n_rounds
        # a7/a8: every level's grammar cells, has_hocc flags and pre-BWT (the reference's dict_lev_r / pre_bwt_lev_r)
        for lvl in range(r):
            g0, g1, hh, ps, pl = ctx.level_grammar(lvl)
            og0, og1, ohh = o.level_grammar(lvl)
            assert np.array_equal(g0, og0) and np.array_equal(g1, og1), "grammar of level %d differs" % lvl
            assert np.array_equal(hh, ohh), "has_hocc of level %d differs" % lvl
            ops, opl = o.level_prebwt(lvl)
            assert _merged(ps, pl) == _merged(ops, opl), "pre-BWT of level %d differs" % lvl
        ctx.parse2bwt()
        s, l = ctx.level_bwt(r)
        os_, ol = o.level_bwt(r)
        assert np.array_equal(s, os_) and np.array_equal(l, ol), "deepest BWT differs"
        lvl = r
        while lvl > 0:
            lvl, li = ctx.infer_lvl_bwt()
            s, l = ctx.level_bwt(lvl)
            os_, ol = o.level_bwt(lvl)
            assert np.array_equal(s, os_) and np.array_equal(l, ol), "BWT of level %d differs" % lvl
            assert li["n_runs"] == len(os_) and li["n"] == int(ol.sum())
        assert ctx.result_bytes() == o.rl_bwt
    o.close()


def _merged(sym, ln):
    """Run list with adjacent equal symbols merged (the reference leaves a few pre-BWT runs unmerged -- the size()>1 quirk of
    exact_par_phase.cpp:212 -- which SURVEY A.9 lists as not part of the contract; the symbol sequence is what counts)."""
    out = []
    for s, l in zip(sym.tolist(), ln.tolist()):
        if out and out[-1][0] == s:
            out[-1][1] += l
        else:
            out.append([s, l])
    return out


def rand_collection(rng, kind):
    if kind == "dna":
        parts = [bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(0, 60))).astype(np.uint8)) + b"\n"
                 for _ in range(int(rng.integers(1, 40)))]
        return b"".join(parts), 1
    if kind == "binary":
        parts = [bytes(rng.choice(list(b"ab"), size=int(rng.integers(0, 80))).astype(np.uint8)) + b"\n"
                 for _ in range(int(rng.integers(1, 12)))]
        return b"".join(parts), 1
    if kind == "repeat":
        base = bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(5, 50))).astype(np.uint8))
        parts = []
        for _ in range(int(rng.integers(2, 15))):
            b = bytearray(base * int(rng.integers(1, 4)))
            if len(b) and rng.random() < 0.5:
                b[int(rng.integers(0, len(b)))] = ord("ACGT"[int(rng.integers(0, 4))])
            parts.append(bytes(b) + b"\n")
        return b"".join(parts), 1
    if kind == "dups":
        pool = [b"ACGT\n", b"A\n", b"\n", b"ACGTACGT\n", b"TTTT\n", b"GATTACA\n"]
        return b"".join(pool[int(rng.integers(0, len(pool)))] for _ in range(int(rng.integers(1, 30)))), 1
    if kind == "homopolymer":
        parts = []
        for _ in range(int(rng.integers(1, 8))):
            s = b""
            for _ in range(int(rng.integers(1, 6))):
                s += bytes([b"ACGT"[int(rng.integers(0, 4))]]) * int(rng.integers(1, 200))
            parts.append(s + b"\n")
        return b"".join(parts), 1
    if kind == "u16":
        cells = []
        for _ in range(int(rng.integers(1, 10))):
            cells += [int(x) for x in rng.integers(1, 300, size=int(rng.integers(0, 40)))] + [0]
        return np.array(cells, dtype=np.uint16).tobytes(), 2
    if kind == "u32":
        cells = []
        for _ in range(int(rng.integers(1, 8))):
            cells += [int(x) for x in rng.integers(8, 2000, size=int(rng.integers(0, 30)))] + [7]
        return np.array(cells, dtype=np.uint32).tobytes(), 4
    if kind == "u64":
        cells = []
        for _ in range(int(rng.integers(1, 8))):
            cells += [int(x) for x in rng.integers(3, 50, size=int(rng.integers(0, 30)))] + [2]
        return np.array(cells, dtype=np.uint64).tobytes(), 8
    raise ValueError(kind)


KINDS = ["dna", "binary", "repeat", "dups", "homopolymer", "u16", "u32", "u64"]


def lf_roundtrip(blob, data, cell_bytes=1):
    """Size-independent property: inverting the produced BWT gives back the collection."""
    dt = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[cell_bytes]
    cells = np.frombuffer(data, dtype=dt)
    sb, fb, sym, ln = bc.parse_rl_bwt(blob)
    esb, efb = bc.header_widths(cells, cell_bytes)
    assert (sb, fb) == (esb, efb)
    assert bc.runs_are_maximal(sym)
    assert int(ln.sum()) == len(cells)
    strings, sep = bc.split_strings(cells)
    rec = bc.lf_invert(sym, ln, sep)
    assert len(rec) == len(strings)
    for a, b in zip(rec, strings):
        assert len(a) == len(b) and np.array_equal(a.astype(dt), b)

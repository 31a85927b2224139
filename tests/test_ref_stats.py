"""Rows a1 / a16 / a17 of SURVEY 8a pinned by REFERENCE code: tests/golden/ref_stats.json holds what the reference's own
collection_stats<T> (external/cdt/lib/utils.cpp:100-189), sym_width / INT_CEIL as exact_ind_phase.cpp:274-276 combines them,
and bwt_buff_writer::push_back / inc_freq_last / close (include/bwt_io.h:448-550) produced for 21 inputs (the inputs of
ref_consumers.json + the header edges of SURVEY A.8), compiled from /root/reference by `make -C oracle ref` behind
oracle/ref_stats_driver.cpp and recorded by tests/golden/make_stats_fixtures.py.  The oracle -- and, under -m gpu, the
engine -- must agree with every field; where oracle/_ref/ref_stats is present the program is also run live."""
import hashlib
import json
import os
import sys
import zlib

import numpy as np
import pytest

from grlbwt_amd import engine, workloads  # noqa: F401
from tests import bcr_check as bc
from tests import parity

FIX = json.load(open(os.path.join(parity.GOLD, "ref_stats.json")))["cases"]
CONS = {c["name"]: c for c in json.load(open(os.path.join(parity.GOLD, "ref_consumers.json")))["cases"]}
sys.path.insert(0, parity.GOLD)


def case_input(c):
    """the input of a fixture case, rebuilt from its name (files, generators) or from the bytes the fixtures hold"""
    name = c["name"]
    if "input_hex" in c:
        data = bytes.fromhex(c["input_hex"])
    elif name.startswith("file:"):
        data = open(os.path.join(parity.GOLD, name[5:]), "rb").read()
    elif name.startswith("gen:"):
        data = eval("workloads." + name[4:]).tobytes()
    elif "input_hex" in CONS.get(name, {}):
        data = bytes.fromhex(CONS[name]["input_hex"])
    else:
        data = zlib.decompress(bytes.fromhex(CONS[name]["input_zlib_hex"]))
    assert hashlib.md5(data).hexdigest() == c["input_md5"], name
    return data


@pytest.mark.parametrize("c", FIX, ids=[c["name"] for c in FIX])
def test_oracle_agrees_with_reference_stats_header_and_writer(oracle_mod, c):
    data, w = case_input(c), c["cell_bytes"]
    cells = np.frombuffer(data, dtype={1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[w])
    assert bc.header_widths(cells, w) == (c["sb"], c["fb"])                       # the A.8 rule == the reference's arithmetic
    if "writer_md5" not in c:                                                     # (uint32 2^32-1: stats and header only)
        return
    o = oracle_mod.OracleResult(data, w)
    st = c["stats"]
    for k in ("n_strings", "n_syms", "min_sym", "max_sym", "max_sym_freq"):
        assert o.stats[k] == st[k], (k, o.stats[k], st[k])
    assert o.stats["longest"] == st["longest_string"]
    assert (o.stats["sb"], o.stats["fb"]) == (c["sb"], c["fb"])
    # what the reference's writer emitted for the oracle's run sequence cut into non-maximal pieces == the oracle's bytes
    assert len(o.rl_bwt) == c["writer_size"] and hashlib.md5(o.rl_bwt).hexdigest() == c["writer_md5"]
    sb, fb, sym, ln = bc.parse_rl_bwt(o.rl_bwt)
    assert len(sym) == c["runs"] < c["pieces_pushed"] or c["pieces_pushed"] == c["runs"]
    o.close()


def test_reference_program_live(oracle_mod):
    """where the reference tree was available at build time (oracle/_ref/ref_stats travels with the repository)"""
    if not oracle_mod.ref_prog("ref_stats"):
        pytest.skip("oracle/_ref/ref_stats not built (no /root/reference at build time)")
    import make_stats_fixtures as msf
    rng = np.random.default_rng(5)
    for kind in parity.KINDS:
        for _ in range(6):
            data, w = parity.rand_collection(rng, kind)
            blob = oracle_mod.rl_bwt(data, w)
            sb, fb, sym, ln = bc.parse_rl_bwt(blob)
            st, written = msf.ref_stats(data, w, msf.split_runs_randomly(sym, ln, int(rng.integers(1 << 30))))
            assert written == blob and (st["sb"], st["fb"]) == (sb, fb), (kind, data)
            o = oracle_mod.OracleResult(data, w)
            assert (o.stats["n_strings"], o.stats["longest"], o.stats["min_sym"], o.stats["max_sym"], o.stats["n_syms"], o.stats["max_sym_freq"]) == \
                   (st["n_strings"], st["longest_string"], st["min_sym"], st["max_sym"], st["n_syms"], st["max_sym_freq"])
            o.close()


@pytest.mark.gpu
def test_engine_agrees_with_reference_stats_and_writer():
    """the HIP path: grlbwt_get_stats (row a1, a17) and the image bytes (row a16) against the same reference-made values"""
    import __graft_entry__ as g
    lib = g.build_hip()
    for c in FIX:
        if "writer_md5" not in c:
            continue
        data, w = case_input(c), c["cell_bytes"]
        with engine.Context(0, 0, lib) as ctx:
            ctx.upload(data, w)
            st = ctx.stats()
            for k in ("n_strings", "n_syms", "min_sym", "max_sym", "max_sym_freq"):
                assert st[k] == c["stats"][k], (c["name"], k)
            assert (st["sb"], st["fb"]) == (c["sb"], c["fb"])
            ctx.build()
            blob = ctx.result_bytes()
        assert len(blob) == c["writer_size"] and hashlib.md5(blob).hexdigest() == c["writer_md5"], c["name"]

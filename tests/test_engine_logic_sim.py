"""CPU: host orchestration + per-element kernel bodies of the engine, run over the
serial test stand-in of the device primitives (tests/hostsim, TEST INFRASTRUCTURE),
against the oracle.  The real HIP kernels are covered by the -m gpu tests."""
import hashlib
import json
import os
import subprocess
import zlib

import numpy as np
import pytest

from grlbwt_amd import engine, workloads
from tests import parity

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def sim():
    from tests import simlib
    return simlib.sim_library()


def test_primitives_selftest(sim):
    with engine.Context(0, 0, sim) as ctx:
        assert ctx.selftest(30000, 5) == 0


def test_golden_table(sim, oracle_mod):
    tab = json.load(open(os.path.join(parity.GOLD, "golden_table.json")))
    for t in tab["tiny"]:
        assert parity.run_engine(sim, bytes.fromhex(t["input_hex"]), t["cell_bytes"]).hex() == t["rl_bwt_hex"]
    for name in ("test_2bytes_alphabet.txt", "test_byte_alphabet.txt"):
        g = tab[name]
        out = parity.run_engine(sim, open(os.path.join(parity.GOLD, name), "rb").read(), g["cell_bytes"])
        assert len(out) == g["size"] and hashlib.md5(out).hexdigest() == g["md5"]


def test_stagewise_2bytes(sim, oracle_mod):
    parity.check_stagewise(sim, open(os.path.join(parity.GOLD, "test_2bytes_alphabet.txt"), "rb").read(), 2)


def test_stagewise_reads(sim, oracle_mod):
    parity.check_stagewise(sim, workloads.sampled_reads(3000, 100, 20000, seed=11).tobytes(), 1)
    parity.check_stagewise(sim, workloads.uniform_reads(2000, 100, seed=5).tobytes(), 1)


def test_stagewise_repetitive(sim, oracle_mod):
    parity.check_stagewise(sim, workloads.repetitive_copies(30, 8000, seed=3).tobytes(), 1)


def test_stagewise_idx64(sim, oracle_mod):
    parity.check_stagewise(sim, workloads.sampled_reads(1500, 80, 10000, seed=2).tobytes(), 1, engine.FLAG_FORCE_IDX64)
    parity.check_stagewise(sim, workloads.zipf_tokens(30000, doc_len=100, vocab=3000).tobytes(), 2, engine.FLAG_FORCE_IDX64)


@pytest.mark.parametrize("kind", parity.KINDS)
def test_fuzz(sim, oracle_mod, kind):
    rng = np.random.default_rng(zlib.crc32(kind.encode()) + 1)
    for i in range(40):
        data, w = parity.rand_collection(rng, kind)
        flags = engine.FLAG_FORCE_IDX64 if i % 4 == 3 else 0
        parity.check_final(sim, data, w, flags)


def test_ill_formed(sim):
    with engine.Context(0, 0, sim) as ctx:
        with pytest.raises(engine.IllFormedInput):
            ctx.upload(b"AC\nGT", 1)
        with pytest.raises(engine.IllFormedInput):
            ctx.upload(b"B\nA\x01B\n", 1)
        with pytest.raises(engine.GrlbwtError):
            ctx.upload(b"", 1)
        with pytest.raises(engine.GrlbwtError):
            ctx.build()                      # nothing loaded: call out of order


def test_header_edges(sim, oracle_mod):
    for mx, w in [(251, 1), (252, 1), (255, 1), (65531, 2), (65532, 2), (65535, 2)]:
        dt = {1: np.uint8, 2: np.uint16}[w]
        parity.check_final(sim, np.array([3, mx, 5, 1, mx, 1], dtype=dt).tobytes(), w)


def test_lf_roundtrip_property(sim):
    data = workloads.sampled_reads(5000, 100, 30000, seed=21).tobytes()
    parity.lf_roundtrip(parity.run_engine(sim, data, 1), data, 1)


def test_long_phrases_saturated_length(sim, oracle_mod):
    # phrases longer than the 12-bit length field of the table key (verified through the start bits)
    data = b"A" * 6000 + b"\n" + b"A" * 6000 + b"\n" + b"A" * 7000 + b"\n" + b"C" * 5000 + b"A" * 4999 + b"\n"
    parity.check_final(sim, data, 1)
    parity.check_stagewise(sim, data, 1, engine.FLAG_FORCE_IDX64)


def test_giant_phrases_hashed_by_the_wave(sim, oracle_mod):
    """Phrases the walk has followed for HashInsertFn::kLongWalk = 4096 cells without reaching their end are listed and
    hashed in 64 pieces by the whole wave: the same phrase twice (one table entry, count 2), a phrase that differs in its last
    cell, one that is a cell longer, and a level that collapses into ONE phrase (a single long string of distinct-ish cells)."""
    g = b"A" * 90000
    data = g + b"\n" + g + b"\n" + b"A" * 89999 + b"C" + b"\n" + g + b"A\n" + b"CGT" * 10 + b"\n"
    parity.check_final(sim, data, 1)
    rng = np.random.default_rng(5)
    one = bytes(rng.integers(1, 256, size=300000).astype(np.uint8)) + b"\x00"       # one string: level 2 is a single phrase
    parity.check_final(sim, one, 1)


def test_table_growth_when_prefix_is_unrepresentative(sim, oracle_mod):
    # > 2^20 repetitive cells first (the capacity estimate sees almost no distinct phrases), then diverse reads
    rep = (b"ACGTTGCA" * 16 + b"\n") * 8500
    data = rep + workloads.uniform_reads(3000, 100, seed=77).tobytes()
    assert len(rep) > (1 << 20)
    with engine.Context(0, 0, sim) as ctx:
        ctx.upload(data, 1)
        ctx.build()
        got = ctx.result_bytes()
    assert got == oracle_mod.rl_bwt(data, 1)


@pytest.mark.parametrize("form", ["positions", "runs"])
@pytest.mark.parametrize("kind,w", [("reads", 1), ("tokens", 2), ("dups", 1), ("repetitive", 1)])
def test_invert_image_round_trip(sim, kind, w, form, monkeypatch):
    """reverse_bwt / grl2plain on the (stand-in) device: the image decodes back to the collection -- through the LF array
    over the positions and through the per-run records (the form the 10 GB headline image takes)."""
    monkeypatch.setenv("GRLBWT_INVERT", form)
    if kind == "reads":
        data = workloads.sampled_reads(3000, 100, 20000, seed=4)
    elif kind == "tokens":
        data = workloads.zipf_tokens(20000, doc_len=100, vocab=2000)
    elif kind == "repetitive":
        data = workloads.repetitive_copies(12, 6000, seed=5)
    else:
        data = np.frombuffer(b"A\n\nA\nGATTACA\nGATTACA\nT\n\n", dtype=np.uint8)
    for flags in (0, engine.FLAG_FORCE_IDX64):
        with engine.Context(0, flags, sim) as ctx:
            ctx.upload(data.tobytes(), w)
            ctx.build()
            nb, _ = ctx.result_size()
            out = np.zeros(data.size, dtype=data.dtype)
            n = ctx.invert_image(ctx.result_device_ptr(), nb, w, out.ctypes.data, data.size)   # stand-in: device == host
            assert n == data.size and np.array_equal(out, data)


@pytest.mark.parametrize("tail", [1, 3, 17, 400])
def test_invert_image_tails(sim, tail):
    """grlbwt_invert_image_tails: slot i of the output holds the last min(length, tail) cells of string i, right-aligned; what
    lies in front of them is left untouched (the form the chromosome-scale GPU test checks the order of a 24.9 GB image with)."""
    for data, w in ((np.frombuffer(b"A\n\nA\nGATTACA\nGATTACA\nT\n\nACGTACGTACGTACGTACGTAAAAAAC\n", dtype=np.uint8), 1),
                    (workloads.sampled_reads(300, 100, 2000, seed=14), 1), (workloads.zipf_tokens(5000, doc_len=60, vocab=500), 2)):
        sep = data[-1]
        ends = np.flatnonzero(data == sep)
        starts = np.concatenate([[0], ends[:-1] + 1])
        for flags in (0, engine.FLAG_FORCE_IDX64):
            with engine.Context(0, flags, sim) as ctx:
                ctx.upload(data.tobytes(), w)
                ctx.build()
                nb, _ = ctx.result_size()
                out = np.full(len(ends) * tail, 0x5A, dtype=data.dtype)
                k, n = ctx.invert_image_tails(ctx.result_device_ptr(), nb, w, tail, out.ctypes.data, out.size)
            assert k == len(ends)
            want = np.full(len(ends) * tail, 0x5A, dtype=data.dtype)
            tot = 0
            for i, (a, b) in enumerate(zip(starts, ends)):
                m = min(int(b - a + 1), tail)
                want[(i + 1) * tail - m:(i + 1) * tail] = data[b + 1 - m:b + 1]
                tot += m
            assert n == tot and np.array_equal(out, want)


def test_unfused_expansion_branch(sim, oracle_mod, monkeypatch):
    """The HIP engine falls back to count + scan + expand + split when a run would drop more cells than the fused
    kernel stages in LDS (> 32); the stand-in's limit is lowered to take that branch on ordinary inputs."""
    monkeypatch.setenv("GRLBWT_SIM_XS_MAXC", "1")
    parity.check_stagewise(sim, workloads.sampled_reads(3000, 100, 20000, seed=11).tobytes(), 1)
    parity.check_stagewise(sim, workloads.zipf_tokens(30000, doc_len=100, vocab=3000).tobytes(), 2, engine.FLAG_FORCE_IDX64)


def test_pass_c_walk_forms(sim, oracle_mod, monkeypatch):
    """Pass C in one walk (prim::stream_merge_onepass: run heads into arrays sized for the caller's bound -- the stand-in throws when
    the bound on the runs or on the queued segments does not hold) at every level, the two-pass form forced, and the fallback of
    a walk that gave up; the same outputs stage by stage."""
    reads = workloads.sampled_reads(3000, 100, 20000, seed=11).tobytes()
    reps = workloads.repetitive_copies(30, 8000, seed=4).tobytes()
    monkeypatch.setenv("GRLBWT_ASM_ONE_WALK", "1")
    parity.check_stagewise(sim, reads, 1)
    parity.check_stagewise(sim, reps, 1, engine.FLAG_FORCE_IDX64)
    parity.check_stagewise(sim, workloads.zipf_tokens(30000, doc_len=100, vocab=3000).tobytes(), 2)
    monkeypatch.setenv("GRLBWT_SIM_ONEPASS_GIVES_UP", "1")
    parity.check_stagewise(sim, reads, 1)
    parity.check_stagewise(sim, reps, 1)
    monkeypatch.delenv("GRLBWT_SIM_ONEPASS_GIVES_UP")
    monkeypatch.delenv("GRLBWT_ASM_ONE_WALK")
    monkeypatch.setenv("GRLBWT_ASM_TWO_PASS", "1")
    parity.check_stagewise(sim, reads, 1)


@pytest.mark.parametrize("layout", ["packed", "separate"])
def test_wider_cell_layouts(sim, oracle_mod, monkeypatch, layout):
    """Induced cells that do not fit one 64-bit word (bucket + run length + symbol > 64 bits) travel as bucket + packed
    payload, or as three arrays when a run length needs more than 32 bits; forced here on ordinary inputs."""
    monkeypatch.setenv("GRLBWT_CELL_LAYOUT", layout)
    parity.check_stagewise(sim, workloads.sampled_reads(3000, 100, 20000, seed=11).tobytes(), 1)
    parity.check_stagewise(sim, workloads.zipf_tokens(30000, doc_len=100, vocab=3000).tobytes(), 2, engine.FLAG_FORCE_IDX64)


def test_one_word_cells_in_64_bits(sim, oracle_mod, monkeypatch):
    """Induced cells whose bucket + run length + symbol fit 32 bits travel as 4-byte words (every small input does); with
    GRLBWT_NO_CELL32 they take the 8-byte one-word form the big levels use."""
    monkeypatch.setenv("GRLBWT_NO_CELL32", "1")
    parity.check_stagewise(sim, workloads.sampled_reads(3000, 100, 20000, seed=11).tobytes(), 1)
    parity.check_stagewise(sim, workloads.zipf_tokens(30000, doc_len=100, vocab=3000).tobytes(), 2, engine.FLAG_FORCE_IDX64)


def _long_run_collection(run, copies, seed):
    """strings that hold a run of `run` + k equal symbols between random flanks (every copy a little longer), plus runs that end
    a string, runs followed by a smaller and by a larger symbol, and a run that is a whole string"""
    rng = np.random.default_rng(seed)
    fl = lambda n: bytes(rng.choice(list(b"ACGT"), size=n).astype(np.uint8))
    a, b = fl(300), fl(300)
    parts = [a + b"N" * (run + k) + b + b"\n" for k in range(copies)]
    parts += [fl(50) + b"T" * (run // 2) + b"\n", fl(40) + b"C" * (run // 3) + b"A" + fl(30) + b"\n", b"G" * (run // 4) + b"T" + fl(20) + b"\n",
              b"A" * (run // 5) + b"\n", a + b"N" * run + b + b"\n"]
    return b"".join(parts)


def test_run_aware_suffix_keys(sim, oracle_mod, monkeypatch):
    """Phrases that hold long runs of one symbol (an N gap is ONE phrase as long as the gap): the dictionary suffix sort switches
    to run-aware keys (RunKeys: a class bit + the remaining run length below the key's window; a whole run is consumed per
    refinement round).  Without them the refinement needs run / K rounds for each of the run's suffixes -- quadratic.  Against
    the oracle: real long runs at the default threshold (and the number of refinement rounds stays small), then the run-aware
    keys forced on for ordinary inputs of every kind."""
    monkeypatch.delenv("GRLBWT_RUN_KEYS_MIN", raising=False)
    data = _long_run_collection(6000, 5, 3)
    with engine.Context(0, 0, sim) as ctx:
        ctx.upload(data, 1)
        ctx.build()
        nb, _ = ctx.result_size()
        assert ctx.result_bytes() == oracle_mod.rl_bwt(data, 1)
        assert max(ctx.round_info(r)["sort_iters"] for r in range(2)) <= 40          # (6000-cell runs: ~400 rounds of 16 symbols otherwise)
    monkeypatch.setenv("GRLBWT_RUN_KEYS_MIN", "0")
    parity.check_stagewise(sim, workloads.sampled_reads(3000, 100, 20000, seed=11).tobytes(), 1)
    parity.check_stagewise(sim, workloads.repetitive_copies(30, 8000, seed=3).tobytes(), 1)
    parity.check_stagewise(sim, workloads.zipf_tokens(30000, doc_len=100, vocab=3000).tobytes(), 2, engine.FLAG_FORCE_IDX64)
    parity.check_final(sim, _long_run_collection(900, 3, 5), 1)
    rng = np.random.default_rng(123)
    for kind in parity.KINDS:
        for _ in range(10):
            d, w = parity.rand_collection(rng, kind)
            parity.check_final(sim, d, w)


def _shared_ramps(n, copies, dtype=np.uint32):
    """strings that share one long strictly increasing ramp (ONE phrase each, no runs) and differ behind it, plus a decreasing one"""
    ramp = np.arange(1, n + 1, dtype=dtype)
    parts = [np.concatenate([ramp, np.array([n + 10 + 3 * k, 0], dtype=dtype)]) for k in range(copies)]
    parts.append(np.concatenate([ramp[::-1], np.array([0], dtype=dtype)]))
    parts.append(np.concatenate([ramp[: n // 2], np.array([n + 5, 0], dtype=dtype)]))
    return np.concatenate(parts)


def test_doubling_rounds_of_the_suffix_refinement(sim, oracle_mod, monkeypatch):
    """Long phrases that are not runs (monotone ramps over a large alphabet) shared by several strings: their suffix groups stay tied
    for the whole ramp, and the symbol extension would need ramp / K rounds.  After GRLBWT_DOUBLING_AFTER rounds (24) the refinement
    of a long-phrase level switches to doubling rounds (DoubleKeyFn: the key of a suffix is the current group of the suffix behind
    its sorted depth).  Real ramps at the defaults, with the rounds counted; then doubling from the first (third) round on for
    ordinary inputs of every kind."""
    monkeypatch.delenv("GRLBWT_RUN_KEYS_MIN", raising=False)
    monkeypatch.delenv("GRLBWT_DOUBLING_AFTER", raising=False)
    cells = _shared_ramps(3000, 5)
    with engine.Context(0, 0, sim) as ctx:
        ctx.upload(cells.tobytes(), 4)
        ctx.build()
        assert ctx.result_bytes() == oracle_mod.rl_bwt(cells.tobytes(), 4)
        assert ctx.round_info(0)["sort_iters"] <= 45          # (24 rounds of 2 symbols, then ~12 doubling rounds; 1500 otherwise)
    for after in ("0", "2"):               # doubling from the first refinement round on / after two rounds of symbol extension
        monkeypatch.setenv("GRLBWT_RUN_KEYS_MIN", "0")
        monkeypatch.setenv("GRLBWT_DOUBLING_AFTER", after)
        parity.check_stagewise(sim, workloads.sampled_reads(3000, 100, 20000, seed=11).tobytes(), 1)
        parity.check_stagewise(sim, workloads.repetitive_copies(30, 8000, seed=3).tobytes(), 1)
        parity.check_stagewise(sim, workloads.zipf_tokens(30000, doc_len=100, vocab=3000).tobytes(), 2, engine.FLAG_FORCE_IDX64)
        parity.check_final(sim, _long_run_collection(900, 3, 5), 1)
        parity.check_final(sim, _shared_ramps(700, 4, np.uint16).tobytes(), 2)
        rng = np.random.default_rng(321)
        for kind in parity.KINDS:
            for _ in range(8):
                d, w = parity.rand_collection(rng, kind)
                parity.check_final(sim, d, w)


def test_device_side_generators_match_host():
    """The torch generators of the large test inputs (run on the GPU there) produce the host generators' bytes."""
    a = workloads.repetitive_copies(5, 30011)
    assert np.array_equal(a, workloads.repetitive_copies_torch(5, 30011, device="cpu").numpy())
    a = workloads.zipf_tokens(50000, doc_len=100, vocab=3000)
    assert np.array_equal(a, workloads.zipf_tokens_torch(50000, doc_len=100, vocab=3000, device="cpu", chunk_docs=37).numpy().view(np.uint16))


def test_file_in_file_out(sim, oracle_mod, tmp_path):
    """grlbwt_text_load_file / grlbwt_result_write_file (the CLI's path): chunked staging, histogram taken per chunk."""
    for data, w in ((workloads.sampled_reads(3000, 100, 20000, seed=11).tobytes(), 1),
                    (workloads.zipf_tokens(30000, doc_len=100, vocab=3000).tobytes(), 2), (b"\n", 1)):
        fi, fo = tmp_path / "in.bin", tmp_path / "out.rl_bwt"
        fi.write_bytes(data)
        with engine.Context(0, 0, sim) as ctx:
            ctx.load_file(str(fi), w)
            st = ctx.stats()
            ctx.build()
            ctx.write_file(str(fo))
        assert fo.read_bytes() == oracle_mod.rl_bwt(data, w)
        assert st["n_syms"] == len(data) // w
    bad = tmp_path / "odd.bin"
    bad.write_bytes(b"\x01\x00\x02")
    with engine.Context(0, 0, sim) as ctx:
        with pytest.raises(engine.IllFormedInput):
            ctx.load_file(str(bad), 2)                # size not a multiple of the cell width
        with pytest.raises(engine.GrlbwtError):
            ctx.load_file(str(tmp_path / "missing.bin"), 1)


@pytest.mark.parametrize("cap", ["1", "3"])
def test_large_group_refinement_branch(sim, oracle_mod, monkeypatch, cap):
    """Suffix refinement: groups above the counting limit are re-sorted by two radix sorts (key, then group); the limit is
    lowered so that ordinary inputs take that path (cap 1: every group; cap 3: both paths in one round)."""
    monkeypatch.setenv("GRLBWT_SEG_CAP", cap)
    parity.check_stagewise(sim, workloads.sampled_reads(3000, 100, 20000, seed=11).tobytes(), 1)
    parity.check_stagewise(sim, workloads.repetitive_copies(30, 8000, seed=3).tobytes(), 1, engine.FLAG_FORCE_IDX64)
    parity.check_stagewise(sim, workloads.zipf_tokens(30000, doc_len=100, vocab=3000).tobytes(), 2)


def test_table_sizing_from_a_sample(sim, oracle_mod, capfd, monkeypatch):
    """Texts above 2^20 cells size their phrase tables from a strided sample (distinct phrases estimated from the sample's
    abundance classes): the sampled hashing functor, the estimate and the build behind it against the oracle -- reads from
    a small genome (few distinct phrases, many occurrences) and a text whose second half looks nothing like its first."""
    monkeypatch.setenv("GRLBWT_TABLE_TRACE", "1")
    data = workloads.sampled_reads(12000, 100, 30000, seed=13).tobytes()
    assert len(data) > (1 << 20)
    parity.check_final(sim, data, 1)
    err = capfd.readouterr().err
    assert "table" in err and "direct index on bits" in err      # at most 8 distinct cell values: phrases of <= 7 cells are their own slot number
    monkeypatch.setenv("GRLBWT_NO_DIRECT_INDEX", "1")
    parity.check_final(sim, data, 1)
    err = capfd.readouterr().err
    assert "hot table of" in err                                 # ... otherwise, few phrases dominate: they get the small dense table in front
    monkeypatch.setenv("GRLBWT_NO_HOT_TABLE", "1")
    parity.check_final(sim, data, 1)
    monkeypatch.delenv("GRLBWT_NO_HOT_TABLE")
    monkeypatch.delenv("GRLBWT_NO_DIRECT_INDEX")
    rep = (b"ACGTTGCA" * 16 + b"\n") * 6000
    parity.check_final(sim, rep + workloads.uniform_reads(6000, 100, seed=77).tobytes(), 1)


def test_hot_table_with_generic_keys(sim, oracle_mod, capfd, monkeypatch):
    """uint16 cells (no exact keys): a text above 2^20 cells made of few distinct documents -- the generic (hash tag + compare)
    keys through the hot table, and the mixed case where half of the text never shows up in the sample."""
    monkeypatch.setenv("GRLBWT_TABLE_TRACE", "1")
    rng = np.random.default_rng(3)
    docs = [np.concatenate([rng.integers(1, 50, size=int(rng.integers(20, 60))), [0]]).astype(np.uint16) for _ in range(40)]
    cells = np.concatenate([docs[int(i)] for i in rng.integers(0, 40, size=30000)])
    assert cells.size > (1 << 20)
    parity.check_final(sim, cells.tobytes(), 2)
    assert "hot table of" in capfd.readouterr().err


def test_partitioned_phrase_naming(sim, oracle_mod, monkeypatch, capfd):
    """Levels above 0 of single-GPU builds name their phrases through 128-bit records, a partition sort that can be undone and
    per-partition de-duplication (prim::RecSort / prim::rec_dedupe; by default from 2^20 occurrences per level on): forced
    on for small inputs here, stage by stage against the oracle -- reads, long repeats (phrases longer than a record: the
    mixed case with the hash table), uint16 tokens, 64-bit indices -- and with partitions that "overflow" (the level then
    falls back to the hash table)."""
    monkeypatch.setenv("GRLBWT_PART_MIN_OCC", "0")
    parity.check_stagewise(sim, workloads.sampled_reads(3000, 100, 20000, seed=11).tobytes(), 1)
    parity.check_stagewise(sim, workloads.uniform_reads(2000, 100, seed=5).tobytes(), 1)
    parity.check_stagewise(sim, workloads.repetitive_copies(30, 8000, seed=3).tobytes(), 1)
    parity.check_stagewise(sim, workloads.zipf_tokens(30000, doc_len=100, vocab=3000).tobytes(), 2, engine.FLAG_FORCE_IDX64)
    parity.check_stagewise(sim, open(os.path.join(parity.GOLD, "test_2bytes_alphabet.txt"), "rb").read(), 2)
    rng = np.random.default_rng(99)
    for kind in parity.KINDS:
        for _ in range(12):
            data, w = parity.rand_collection(rng, kind)
            parity.check_final(sim, data, w)
    # (the records by the general hashing pass instead of the lean record kernel + list of long phrases)
    monkeypatch.setenv("GRLBWT_PART_ONE_PASS", "1")
    parity.check_stagewise(sim, workloads.sampled_reads(3000, 100, 20000, seed=13).tobytes(), 1)
    parity.check_stagewise(sim, workloads.repetitive_copies(30, 8000, seed=4).tobytes(), 1)
    monkeypatch.delenv("GRLBWT_PART_ONE_PASS")
    monkeypatch.setenv("GRLBWT_SIM_PD_LIMIT", "3")
    monkeypatch.setenv("GRLBWT_TABLE_TRACE", "1")
    parity.check_stagewise(sim, workloads.sampled_reads(3000, 100, 20000, seed=12).tobytes(), 1)
    assert "falling back to the hash table" in capfd.readouterr().err


def test_direct_index_of_short_phrases(sim, oracle_mod, monkeypatch):
    """Byte texts with at most 8 distinct cell values: a phrase of <= 7 cells is named by its own number (three bits per cell,
    HashInsertFn::direct_index) -- no table.  Forced on texts too small for a sample, stage by stage against the oracle: DNA with
    N, an alphabet whose codes need non-adjacent bits, exactly 8 values, phrases of every length around 7 and at the very end of the
    text (the walk path must give the same slot as the batch path); 9 values fall back to the table."""
    monkeypatch.setenv("GRLBWT_FORCE_DIRECT_INDEX", "1")
    rng = np.random.default_rng(5)
    dna = workloads.sampled_reads(3000, 100, 20000, seed=11).copy()
    dna[rng.integers(0, dna.size, size=300)] = ord("N")
    dna[dna.size - 1] = 10
    dna[np.flatnonzero(workloads.sampled_reads(3000, 100, 20000, seed=11) == 10)] = 10
    parity.check_stagewise(sim, dna.tobytes(), 1)
    # (values told apart by bits 0, 3, 6 -- all 8 codes in use; by bits 1, 4, 7; three values; 9 values: no direct index;
    # 8 one-hot values: no three bits tell them apart, no direct index either)
    for alphabet in (bytes([0, 1, 8, 9, 64, 65, 72, 73]), bytes([0x20, 0x22, 0x30, 0x32, 0xA0, 0xA2]), b"\x00ab", bytes(range(10, 19)),
                     bytes([1, 2, 4, 8, 16, 32, 64, 128])):
        vals = np.frombuffer(alphabet, dtype=np.uint8)
        body = vals[1:][rng.integers(0, len(vals) - 1, size=40000)]
        body[rng.integers(0, body.size, size=700)] = vals[0]
        text = np.concatenate([body, vals[:1]])
        parity.check_stagewise(sim, text.tobytes(), 1)
    parity.check_final(sim, b"ACGTACG\nACGTACGT\nAC\nA\n\nACGTAC\nGATTACA\n", 1)


"""CPU, world_size 2 and 3 over gloo: the collection-level multi-GPU path (record shards,
dictionary all-gather + merge per parsing round) gives the BWT of the WHOLE collection,
bit-identical to the oracle and identical on every rank."""
import hashlib
import os
import subprocess
import sys

import pytest

from tests import parity

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def sim():
    from tests import simlib
    return simlib.sim_library()


def _run(world, lib, case, out_dir, port):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(HERE, "dist_worker.py"), lib, "gloo", case, str(out_dir)]
    # (the dictionaries of these inputs have a few thousand symbols: the size from which the dictionary stays sharded by owner --
    # 2^27 symbols by default -- is lowered to zero, so that from 4 ranks on, or wherever a test lowers that limit too, the
    # sharded form is what runs)
    env = dict(os.environ)
    env.setdefault("GRLBWT_DIST_SHARDED_DICT_MIN_SYMS", "0")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]


@pytest.mark.parametrize("world,case", [(2, "reads"), (2, "tokens"), (3, "uniform"), (2, "repetitive"), (3, "tiny"),
                                        (4, "reads"), (3, "longruns"), (4, "samechar")])
def test_sharded_collection_matches_oracle(sim, oracle_mod, tmp_path, world, case):
    port = 29500 + 7 * world + ['reads', 'tokens', 'uniform', 'repetitive', 'tiny', 'longruns', 'samechar'].index(case)
    _run(world, sim, case, tmp_path, port)
    data = open(tmp_path / (case + ".input"), "rb").read()
    out = open(tmp_path / (case + ".rl_bwt"), "rb").read()
    w = 2 if case == "tokens" else 1
    assert out == oracle_mod.rl_bwt(data, w)
    md5s = {open(tmp_path / ("%s.rank%d.md5" % (case, r))).read() for r in range(world)}
    assert md5s == {hashlib.md5(out).hexdigest()}
    # the exchanges really happened: dictionary/run all-gathers and the per-level atom routing (all-to-all)
    for r in range(world):
        n_ag, n_a2a, nbytes = map(int, open(tmp_path / ("%s.rank%d.comm" % (case, r))).read().split())
        assert n_ag > 0 and n_a2a > 0 and nbytes > 0


@pytest.mark.parametrize("layout", ["packed", "separate"])
def test_sharded_wider_cell_layouts(sim, oracle_mod, tmp_path, monkeypatch, layout):
    """The cell exchange of the collection-level induction in the two wider cell layouts (several arrays per cell)."""
    monkeypatch.setenv("GRLBWT_CELL_LAYOUT", layout)
    _run(3, sim, "reads", tmp_path, 29571 if layout == "packed" else 29572)
    data = open(tmp_path / "reads.input", "rb").read()
    assert open(tmp_path / "reads.rl_bwt", "rb").read() == oracle_mod.rl_bwt(data, 1)


@pytest.mark.parametrize("world,case,port", [(2, "reads", 29581), (3, "repetitive", 29582), (2, "tokens", 29583), (3, "uniform", 29584)])
def test_sharded_partitioned_phrase_naming(sim, oracle_mod, tmp_path, monkeypatch, world, case, port):
    """Levels above 0 of a collection-level round name their phrases through records as on one GPU (forced on for small
    inputs): a rank's short phrases reach their owners from the records (PhraseOwnerFn / SendCellsFn), long ones from the text."""
    monkeypatch.setenv("GRLBWT_PART_MIN_OCC", "0")
    _run(world, sim, case, tmp_path, port)
    data = open(tmp_path / (case + ".input"), "rb").read()
    assert open(tmp_path / (case + ".rl_bwt"), "rb").read() == oracle_mod.rl_bwt(data, 2 if case == "tokens" else 1)


@pytest.mark.parametrize("world,xmin,port", [(2, "2", 29586), (4, "9", 29587)])
def test_sharded_suffix_sort_both_forms(sim, oracle_mod, tmp_path, monkeypatch, world, xmin, port):
    """The sharded first suffix sort takes its records by a sample-sort exchange from 4 ranks on and by looking at all positions
    (no exchange) below; GRLBWT_SORT_EXCHANGE_MIN moves the switch: the exchange with 2 ranks, the local form with 4."""
    monkeypatch.setenv("GRLBWT_SORT_EXCHANGE_MIN", xmin)
    _run(world, sim, "reads", tmp_path, port)
    data = open(tmp_path / "reads.input", "rb").read()
    assert open(tmp_path / "reads.rl_bwt", "rb").read() == oracle_mod.rl_bwt(data, 1)


def test_sharded_cell_merge_by_radix_sort(sim, oracle_mod, tmp_path, monkeypatch):
    """The received cell blocks are merged by block offsets; GRLBWT_MERGE_CELLS=sort keeps the stable radix sort they replaced."""
    monkeypatch.setenv("GRLBWT_MERGE_CELLS", "sort")
    _run(3, sim, "reads", tmp_path, 29588)
    data = open(tmp_path / "reads.input", "rb").read()
    assert open(tmp_path / "reads.rl_bwt", "rb").read() == oracle_mod.rl_bwt(data, 1)


def test_sharded_8_byte_one_word_cells(sim, oracle_mod, tmp_path, monkeypatch):
    """The cell exchange in the 8-byte one-word form (small inputs otherwise send 4-byte cells)."""
    monkeypatch.setenv("GRLBWT_NO_CELL32", "1")
    _run(3, sim, "reads", tmp_path, 29585)
    data = open(tmp_path / "reads.input", "rb").read()
    assert open(tmp_path / "reads.rl_bwt", "rb").read() == oracle_mod.rl_bwt(data, 1)


def test_large_exchanges_go_in_rounds(sim, oracle_mod, tmp_path, monkeypatch):
    """All-to-all blocks above the engine's limit are sent in several rounds (the limit is lowered to 4 KiB here)."""
    monkeypatch.setenv("GRLBWT_A2A_BLOCK", "4096")
    _run(3, sim, "reads", tmp_path, 29573)
    data = open(tmp_path / "reads.input", "rb").read()
    assert open(tmp_path / "reads.rl_bwt", "rb").read() == oracle_mod.rl_bwt(data, 1)
    n_a2a = int(open(tmp_path / "reads.rank0.comm").read().split()[1])
    monkeypatch.delenv("GRLBWT_A2A_BLOCK")
    _run(3, sim, "reads", tmp_path, 29574)
    assert n_a2a > int(open(tmp_path / "reads.rank0.comm").read().split()[1])


def test_replicated_induction_fallback_agrees(sim, oracle_mod, tmp_path, monkeypatch):
    monkeypatch.setenv("GRLBWT_DIST_REPLICATED_INDUCTION", "1")
    _run(2, sim, "reads", tmp_path, 29590)
    data = open(tmp_path / "reads.input", "rb").read()
    assert open(tmp_path / "reads.rl_bwt", "rb").read() == oracle_mod.rl_bwt(data, 1)
    n_a2a = int(open(tmp_path / "reads.rank0.comm").read().split()[1])
    monkeypatch.delenv("GRLBWT_DIST_REPLICATED_INDUCTION")
    _run(2, sim, "reads", tmp_path, 29592)           # (the two modes exchange different things: the fallback all-gathers the
    assert open(tmp_path / "reads.rl_bwt", "rb").read() == oracle_mod.rl_bwt(data, 1)      # pre-BWT, the sharded induction windows and cells)
    assert n_a2a != int(open(tmp_path / "reads.rank0.comm").read().split()[1])


def test_sharded_large_group_refinement(sim, oracle_mod, tmp_path, monkeypatch):
    """The key-range-sharded suffix refinement with every group on the large-group path (limit lowered)."""
    monkeypatch.setenv("GRLBWT_SEG_CAP", "1")
    _run(2, sim, "tokens", tmp_path, 29594)
    data = open(tmp_path / "tokens.input", "rb").read()
    assert open(tmp_path / "tokens.rl_bwt", "rb").read() == oracle_mod.rl_bwt(data, 2)


def test_replicated_prebwt_fallback_agrees(sim, oracle_mod, tmp_path, monkeypatch):
    """Rounds 1-4 all-gathered every level's pre-BWT and cut the output pieces by symbol count; since round 5 a rank keeps the
    pre-BWT of its key range and that IS its piece.  The older form stays behind a switch and gives the same image."""
    monkeypatch.setenv("GRLBWT_DIST_REPLICATED_PREBWT", "1")
    _run(3, sim, "reads", tmp_path, 29596)
    data = open(tmp_path / "reads.input", "rb").read()
    assert open(tmp_path / "reads.rl_bwt", "rb").read() == oracle_mod.rl_bwt(data, 1)


def test_replicated_grammar_fallback_agrees(sim, oracle_mod, tmp_path, monkeypatch):
    """Since round 5 the grammar passes run where a dictionary position lives (marks and walk requests go to the owner of the
    position); the older form -- every rank applies every mark to a walk array of the whole dictionary -- stays behind a switch."""
    monkeypatch.setenv("GRLBWT_DIST_REPLICATED_GRAMMAR", "1")
    monkeypatch.setenv("GRLBWT_DIST_GATHERED_DICT", "1")        # (a dictionary sharded by owner has no replicated walk array)
    _run(3, sim, "reads", tmp_path, 29597)
    data = open(tmp_path / "reads.input", "rb").read()
    assert open(tmp_path / "reads.rl_bwt", "rb").read() == oracle_mod.rl_bwt(data, 1)


def test_owner_sharded_grammar_with_stop_bits(sim, oracle_mod, tmp_path, monkeypatch):
    """The walks of the owner-sharded grammar passes through the stop bit-vector (the form levels with very long phrases take)."""
    monkeypatch.setenv("GRLBWT_GRAMMAR_JUMP", "1")
    for kind, w, port in (("reads", 1, 29598), ("tokens", 2, 29599)):
        _run(4, sim, kind, tmp_path, port)
        data = open(tmp_path / ("%s.input" % kind), "rb").read()
        assert open(tmp_path / ("%s.rl_bwt" % kind), "rb").read() == oracle_mod.rl_bwt(data, w)


def test_group_fold_with_computed_suffix_records(sim, oracle_mod, tmp_path, monkeypatch):
    """From 8 ranks on the group fold computes what it needs of a suffix where it needs it (no record array over the whole
    dictionary); forced here at 2 and 3 ranks, with the large-group path taken by every group."""
    monkeypatch.setenv("GRLBWT_DIST_REC_FLY_MIN", "2")
    monkeypatch.setenv("GRLBWT_DIST_GATHERED_DICT", "1")        # (the gathered dictionary's fold; with the dictionary sharded by owner the records come from the owners)
    _run(2, sim, "reads", tmp_path, 29600)
    data = open(tmp_path / "reads.input", "rb").read()
    assert open(tmp_path / "reads.rl_bwt", "rb").read() == oracle_mod.rl_bwt(data, 1)
    monkeypatch.setenv("GRLBWT_SEG_CAP", "1")
    _run(3, sim, "tokens", tmp_path, 29601)
    data = open(tmp_path / "tokens.input", "rb").read()
    assert open(tmp_path / "tokens.rl_bwt", "rb").read() == oracle_mod.rl_bwt(data, 2)


@pytest.mark.parametrize("world", [2, 4])
def test_gathered_dictionary_fallback_agrees(sim, oracle_mod, tmp_path, monkeypatch, world):
    """Since round 5 the merged dictionary stays sharded by owner through the dictionary stage (symbols of positions a rank does not
    own are asked of their owners, round by round of the refinement); rounds 1-4 all-gathered it to every rank.  That form stays
    behind a switch -- it is also what levels with very long phrases take -- and gives the same image."""
    monkeypatch.setenv("GRLBWT_DIST_GATHERED_DICT", "1")
    for kind, w, port in (("reads", 1, 29602), ("tokens", 2, 29603)):
        _run(world, sim, kind, tmp_path, port + 2 * world)
        data = open(tmp_path / ("%s.input" % kind), "rb").read()
        assert open(tmp_path / ("%s.rl_bwt" % kind), "rb").read() == oracle_mod.rl_bwt(data, w)


def test_long_phrase_levels_take_the_gathered_dictionary(sim, oracle_mod, tmp_path, monkeypatch):
    """Levels whose longest phrase reaches GRLBWT_RUN_KEYS_MIN cells (run-aware suffix keys) gather the dictionary; the others of
    the same build keep it sharded: both forms in one build (limit lowered to 6 cells)."""
    monkeypatch.setenv("GRLBWT_RUN_KEYS_MIN", "6")
    _run(3, sim, "reads", tmp_path, 29612)
    data = open(tmp_path / "reads.input", "rb").read()
    assert open(tmp_path / "reads.rl_bwt", "rb").read() == oracle_mod.rl_bwt(data, 1)


def test_sharded_dictionary_at_two_and_three_ranks(sim, oracle_mod, tmp_path, monkeypatch):
    """By default the merged dictionary stays sharded by owner from 4 ranks on (below, the gathered form is faster: DESIGN.md
    section 6); forced here at 2 and 3 ranks, with every group of the refinement on the large-group path in one of the runs, and
    with a failure injected into one rank's part of the sharded sort (every rank raises the same error, nobody waits)."""
    monkeypatch.setenv("GRLBWT_DIST_SHARDED_DICT_MIN", "1")
    for world, kind, w, port in ((2, "reads", 1, 29620), (3, "tokens", 2, 29622), (3, "uniform", 1, 29624)):
        _run(world, sim, kind, tmp_path, port)
        data = open(tmp_path / ("%s.input" % kind), "rb").read()
        assert open(tmp_path / ("%s.rl_bwt" % kind), "rb").read() == oracle_mod.rl_bwt(data, w)
    monkeypatch.setenv("GRLBWT_SEG_CAP", "1")
    _run(2, sim, "tokens", tmp_path, 29626)
    data = open(tmp_path / "tokens.input", "rb").read()
    assert open(tmp_path / "tokens.rl_bwt", "rb").read() == oracle_mod.rl_bwt(data, 2)
    monkeypatch.delenv("GRLBWT_SEG_CAP")
    monkeypatch.setenv("GRLBWT_TEST_FAIL_RANK_SORT", "1")
    _run(3, sim, "injected", tmp_path, 29628)
    assert [open(tmp_path / ("injected.rank%d" % r)).read() for r in range(3)] == ["raised -71"] * 3


def test_sharded_dictionary_records_by_round_trip(sim, oracle_mod, tmp_path, monkeypatch):
    """With the dictionary sharded by owner, what the group fold reads about a suffix travels with its (key, position) record
    while every frequency fits 32 bits; collections beyond that (and this switch) ask the owners in a round trip of their own."""
    monkeypatch.setenv("GRLBWT_DIST_REC_ROUND_TRIP", "1")
    monkeypatch.setenv("GRLBWT_DIST_SHARDED_DICT_MIN", "1")
    for world, kind, w, port in ((2, "reads", 1, 29630), (4, "tokens", 2, 29632)):
        _run(world, sim, kind, tmp_path, port)
        data = open(tmp_path / ("%s.input" % kind), "rb").read()
        assert open(tmp_path / ("%s.rl_bwt" % kind), "rb").read() == oracle_mod.rl_bwt(data, w)


def test_replicated_dictionary_fallback_agrees(sim, oracle_mod, tmp_path, monkeypatch):
    monkeypatch.setenv("GRLBWT_DIST_REPLICATED_DICT", "1")
    _run(2, sim, "uniform", tmp_path, 29591)
    data = open(tmp_path / "uniform.input", "rb").read()
    assert open(tmp_path / "uniform.rl_bwt", "rb").read() == oracle_mod.rl_bwt(data, 1)


def test_ill_formed_shard_fails_on_every_rank(sim, tmp_path):
    """A shard-local validation error is agreed on before the first collective: all ranks raise, nobody hangs."""
    _run(3, sim, "illformed", tmp_path, 29593)
    got = [open(tmp_path / ("illformed.rank%d" % r)).read() for r in range(3)]
    assert got[1] == "raised -84" and got[0].startswith("raised") and got[2].startswith("raised"), got


def test_local_failure_inside_a_round_fails_on_every_rank(sim, tmp_path, monkeypatch):
    """A failure only one shard can have (e.g. its phrase table overflows) travels with the next exchange: all ranks raise
    the same error instead of waiting for the failed one."""
    monkeypatch.setenv("GRLBWT_TEST_FAIL_RANK", "1")
    _run(3, sim, "injected", tmp_path, 29595)
    assert [open(tmp_path / ("injected.rank%d" % r)).read() for r in range(3)] == ["raised -28"] * 3
    # ... and inside an induction level (rank 2 "runs out of memory" in its passes A+B)
    monkeypatch.delenv("GRLBWT_TEST_FAIL_RANK")
    monkeypatch.setenv("GRLBWT_TEST_FAIL_RANK_INDUCE", "2")
    _run(3, sim, "injected", tmp_path, 29596)
    assert [open(tmp_path / ("injected.rank%d" % r)).read() for r in range(3)] == ["raised -12"] * 3
    # ... in the merge of the phrases a rank owns, and in a rank's own key range of the sharded suffix sort
    monkeypatch.delenv("GRLBWT_TEST_FAIL_RANK_INDUCE")
    monkeypatch.setenv("GRLBWT_TEST_FAIL_RANK_MERGE", "0")
    _run(3, sim, "injected", tmp_path, 29597)
    assert [open(tmp_path / ("injected.rank%d" % r)).read() for r in range(3)] == ["raised -28"] * 3
    monkeypatch.delenv("GRLBWT_TEST_FAIL_RANK_MERGE")
    monkeypatch.setenv("GRLBWT_TEST_FAIL_RANK_SORT", "2")
    _run(3, sim, "injected", tmp_path, 29598)
    assert [open(tmp_path / ("injected.rank%d" % r)).read() for r in range(3)] == ["raised -71"] * 3


def test_fewer_strings_than_ranks_is_rejected():
    import numpy as np
    from grlbwt_amd import dist as gdist
    with pytest.raises(ValueError):
        gdist.shard_records(np.frombuffer(b"AC\nGT\n", dtype=np.uint8), 0, 3)


def test_sharded_dictionary_beyond_32_bit_positions(sim, oracle_mod, tmp_path, monkeypatch):
    """A dictionary position travels as (owner, offset in the owner's part) when the dictionary is sharded by owner and the records
    ride with the sort exchange: the parts must each be below 2^32 symbols, their sum need not.  GRLBWT_TEST_DICT_PART_PAD puts
    2^31 unused positions behind every rank's part, so the global numbering of these small dictionaries passes 2^32 from the third
    rank on -- a step that still used a global 32-bit position would wrap and lose parity.  The forms that do use global positions
    (records by round trip) refuse such a dictionary on every rank."""
    monkeypatch.setenv("GRLBWT_TEST_DICT_PART_PAD", str(1 << 31))
    monkeypatch.setenv("GRLBWT_RUN_KEYS_MIN", str(1 << 30))      # (levels with very long phrases take the gathered form, which has global positions)
    for world, kind, w, port in ((3, "reads", 1, 29640), (4, "tokens", 2, 29642), (3, "longruns", 1, 29644)):
        _run(world, sim, kind, tmp_path, port)
        data = open(tmp_path / ("%s.input" % kind), "rb").read()
        assert open(tmp_path / ("%s.rl_bwt" % kind), "rb").read() == oracle_mod.rl_bwt(data, w)
    monkeypatch.setenv("GRLBWT_DIST_REC_ROUND_TRIP", "1")
    _run(3, sim, "injected", tmp_path, 29646)
    assert [open(tmp_path / ("injected.rank%d" % r)).read() for r in range(3)] == ["raised -75"] * 3


def test_image_kept_in_parts(sim, oracle_mod, tmp_path, monkeypatch):
    """GRLBWT_COMM_KEEP_PARTS: no all-gather of the image; every rank holds the part whose runs it induced and writes it at its
    offset of one file (grlbwt_result_part / grlbwt_result_write_part).  The parts tile the image (checked in the worker) and
    the file equals the oracle's -- also where slices are single runs that merge into a neighbour's (parts of zero bytes)."""
    monkeypatch.setenv("GRLBWT_TEST_KEEP_PARTS", "1")
    for world, kind, w, port in ((3, "reads", 1, 29660), (4, "tokens", 2, 29662), (4, "samechar", 1, 29664), (3, "tiny", 1, 29666)):
        _run(world, sim, kind, tmp_path, port)
        data = open(tmp_path / ("%s.input" % kind), "rb").read()
        assert open(tmp_path / ("%s.rl_bwt" % kind), "rb").read() == oracle_mod.rl_bwt(data, w)


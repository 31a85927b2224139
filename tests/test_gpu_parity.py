"""GPU (-m gpu): the HIP path through the C-ABI against the oracle, the golden
fixtures and size-independent properties.  Bit-exact is the bar (integer path)."""
import hashlib
import json
import os
import sys
import zlib

import numpy as np
import pytest

from grlbwt_amd import engine, workloads
from tests import parity

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    import __graft_entry__ as g
    lib = g.build_hip()
    assert os.path.exists(lib)
    return lib


def test_native_library_is_loaded(hip):
    engine.load_library(hip)
    maps = open("/proc/self/maps").read()
    assert "libgrlbwt_hip.so" in maps


@pytest.mark.parametrize("n", [1, 63, 64, 65, 2048, 2049, 4096, 4097, 100000, 131072, 131073, 3000001, 20000001])
def test_primitives_selftest(hip, n):
    with engine.Context(0, 0, hip) as ctx:
        assert ctx.selftest(n, 11 + n) == 0


def test_golden_table(hip, oracle_mod):
    tab = json.load(open(os.path.join(parity.GOLD, "golden_table.json")))
    for t in tab["tiny"]:
        assert parity.run_engine(hip, bytes.fromhex(t["input_hex"]), t["cell_bytes"]).hex() == t["rl_bwt_hex"]
    for name in ("test_2bytes_alphabet.txt", "test_byte_alphabet.txt"):
        g = tab[name]
        out = parity.run_engine(hip, open(os.path.join(parity.GOLD, name), "rb").read(), g["cell_bytes"])
        assert len(out) == g["size"] and hashlib.md5(out).hexdigest() == g["md5"]


def test_stagewise_golden_files(hip, oracle_mod):
    parity.check_stagewise(hip, open(os.path.join(parity.GOLD, "test_2bytes_alphabet.txt"), "rb").read(), 2)
    parity.check_stagewise(hip, open(os.path.join(parity.GOLD, "test_byte_alphabet.txt"), "rb").read(), 1)


def test_stagewise_reads(hip, oracle_mod):
    parity.check_stagewise(hip, workloads.sampled_reads(20000, 100, 100000, seed=11).tobytes(), 1)
    parity.check_stagewise(hip, workloads.uniform_reads(20000, 100, seed=5).tobytes(), 1)


def test_stagewise_50MB_default_thresholds(hip, oracle_mod):
    """Stage by stage at a size where the DEFAULT thresholds switch the large-input paths on: 330 k reads of
    150 bp from a 1.7 Mbp genome, 49.8 MB -- levels 1-2 name their phrases through the partition sort (>= 2^20 occurrences), the
    dictionary sorts run on 16384-key tiles, large equal-suffix groups go through the chunked fold.  Every level's parse,
    grammar, has_hocc flags, pre-BWT and BWT against the oracle (about 15 s of one host core)."""
    data = workloads.sampled_reads(330000, 150, 1660000, seed=20260508)
    assert data.size == 330000 * 151
    parity.check_stagewise(hip, data.tobytes(), 1)


def test_stagewise_repetitive(hip, oracle_mod):
    parity.check_stagewise(hip, workloads.repetitive_copies(40, 50000, seed=3).tobytes(), 1)


def test_stagewise_tokens_u16(hip, oracle_mod):
    parity.check_stagewise(hip, workloads.zipf_tokens(400000, doc_len=500, vocab=20000).tobytes(), 2)


@pytest.mark.parametrize("layout", ["packed", "separate"])
def test_wider_cell_layouts(hip, oracle_mod, monkeypatch, layout):
    """The induced-cell layouts for levels whose bucket + run length + symbol exceed 64 bits (bucket array + packed payload,
    or three arrays), forced on ordinary inputs: the HIP kernels of those branches against the oracle, stage by stage."""
    monkeypatch.setenv("GRLBWT_CELL_LAYOUT", layout)
    parity.check_stagewise(hip, workloads.sampled_reads(20000, 100, 100000, seed=11).tobytes(), 1)
    parity.check_stagewise(hip, workloads.zipf_tokens(200000, doc_len=500, vocab=20000).tobytes(), 2, engine.FLAG_FORCE_IDX64)


@pytest.mark.parametrize("form", ["one_walk", "two_pass"])
def test_pass_c_walk_forms(hip, oracle_mod, monkeypatch, form):
    """Pass C on the device in both forms at EVERY level: one walk with the run index of a tile from the look-back across the tiles
    (prim::k_sm_merge<..., 2, ...>: by default the levels whose segments mix cells and pre-BWT runs), and count + emit (by default
    the levels of plain cells); stage by stage against the oracle -- reads, long runs whose TAKE segments span many runs of T (the
    queued segments carry tile-relative places in the one-walk form), tokens, 64-bit indices."""
    monkeypatch.setenv("GRLBWT_ASM_ONE_WALK" if form == "one_walk" else "GRLBWT_ASM_TWO_PASS", "1")
    parity.check_stagewise(hip, workloads.sampled_reads(20000, 100, 100000, seed=11).tobytes(), 1)
    parity.check_stagewise(hip, workloads.repetitive_copies(40, 50000, seed=3).tobytes(), 1, engine.FLAG_FORCE_IDX64)
    parity.check_stagewise(hip, workloads.zipf_tokens(200000, doc_len=500, vocab=20000).tobytes(), 2)
    parity.check_final(hip, workloads.sampled_reads(330000, 150, 1660000, seed=20260508).tobytes(), 1)


def test_unfused_expansion_branch(hip, oracle_mod, monkeypatch):
    """The count + scan + expand + split fallback of passes A+B (taken when a run drops more cells than the fused kernel
    stages): the limit is lowered so that ordinary inputs take it on the device."""
    monkeypatch.setenv("GRLBWT_XS_MAXC", "1")
    parity.check_stagewise(hip, workloads.sampled_reads(20000, 100, 100000, seed=11).tobytes(), 1)
    parity.check_stagewise(hip, workloads.repetitive_copies(40, 50000, seed=3).tobytes(), 1, engine.FLAG_FORCE_IDX64)


def test_one_word_cells_in_64_bits(hip, oracle_mod, monkeypatch):
    """The 8-byte one-word cells of the big levels (bucket + run length + symbol > 32 bits), forced on small inputs, whose
    cells otherwise fit the 4-byte form."""
    monkeypatch.setenv("GRLBWT_NO_CELL32", "1")
    parity.check_stagewise(hip, workloads.sampled_reads(20000, 100, 100000, seed=11).tobytes(), 1)
    parity.check_stagewise(hip, workloads.zipf_tokens(200000, doc_len=500, vocab=20000).tobytes(), 2, engine.FLAG_FORCE_IDX64)


@pytest.mark.parametrize("part_bits", ["", "9", "17", "20"])
def test_partitioned_phrase_naming(hip, oracle_mod, monkeypatch, capfd, part_bits):
    """prim::RecSort (forward passes over (key, hi) record words on 16384-record tiles, the digit taken from a hash of the key; the
    re-ranking passes back from the stored digits) and prim::k_rec_dedupe on the device: partitioned phrase naming forced on for
    small inputs (by default it starts at 2^20 phrase occurrences per level), stage by stage against the oracle -- with the
    engine's own number of partitions, with one 9-bit pass (per-wave counters in 16-bit halves), two passes (9 + 8) and three
    (7 + 7 + 6); then with partitions that cannot fit their LDS table (one bit of partition for 3 M distinct phrases)."""
    monkeypatch.setenv("GRLBWT_PART_MIN_OCC", "0")
    if part_bits:
        monkeypatch.setenv("GRLBWT_PART_BITS", part_bits)
    parity.check_stagewise(hip, workloads.sampled_reads(20000, 100, 100000, seed=11).tobytes(), 1)
    parity.check_stagewise(hip, workloads.repetitive_copies(40, 50000, seed=3).tobytes(), 1)
    parity.check_stagewise(hip, workloads.zipf_tokens(200000, doc_len=500, vocab=20000).tobytes(), 2, engine.FLAG_FORCE_IDX64)
    rng = np.random.default_rng(2026)
    for kind in parity.KINDS:
        for _ in range(6):
            data, w = parity.rand_collection(rng, kind)
            parity.check_final(hip, data, w)
    monkeypatch.setenv("GRLBWT_PART_ONE_PASS", "1")        # the records by the general hashing pass (HashInsertFn's own cut)
    parity.check_stagewise(hip, workloads.sampled_reads(20000, 100, 100000, seed=13).tobytes(), 1)
    monkeypatch.delenv("GRLBWT_PART_ONE_PASS")
    monkeypatch.setenv("GRLBWT_PART_BITS", "1")
    monkeypatch.setenv("GRLBWT_TABLE_TRACE", "1")
    capfd.readouterr()
    parity.check_stagewise(hip, workloads.uniform_reads(20000, 100, seed=5).tobytes(), 1)
    assert "falling back to the hash table" in capfd.readouterr().err


def test_run_aware_suffix_keys(hip, oracle_mod, monkeypatch):
    """Long runs of one symbol (an N gap of 150 k cells in six copies, runs ending a string, runs followed by smaller / larger
    symbols): ONE phrase per run, and the dictionary suffix sort must not take run / K refinement rounds for each of the run's
    suffixes (20 MB of such data took 311 s before the run-aware keys).  Against the oracle, with the rounds counted; then the
    run-aware keys forced on for ordinary inputs."""
    import time
    import torch
    from tests.test_engine_logic_sim import _long_run_collection
    monkeypatch.delenv("GRLBWT_RUN_KEYS_MIN", raising=False)
    data = _long_run_collection(10000, 6, 7)                # (the oracle is itself quadratic in the run length: 4 s here)
    with engine.Context(0, 0, hip) as ctx:
        ctx.upload(data, 1)
        ctx.build()
        assert ctx.result_bytes() == oracle_mod.rl_bwt(data, 1)
        assert max(ctx.round_info(r)["sort_iters"] for r in range(2)) <= 40
    # long runs proper: by the device round trip
    big = np.frombuffer(_long_run_collection(600000, 8, 9), dtype=np.uint8)
    t = torch.from_numpy(big.copy()).to("cuda:0")
    out = torch.zeros_like(t)
    with engine.Context(0, 0, hip) as ctx:
        t0 = time.time()
        ctx.attach_device(t.data_ptr(), t.numel(), 1, keepalive=t)
        ctx.build()
        torch.cuda.synchronize()
        assert time.time() - t0 < 30.0
        assert max(ctx.round_info(r)["sort_iters"] for r in range(2)) <= 40
        nb, _ = ctx.result_size()
        n = ctx.invert_image(ctx.result_device_ptr(), nb, 1, out.data_ptr(), out.numel())
    torch.cuda.synchronize()
    assert n == t.numel() and torch.equal(out, t)
    monkeypatch.setenv("GRLBWT_RUN_KEYS_MIN", "0")
    parity.check_stagewise(hip, workloads.sampled_reads(20000, 100, 100000, seed=11).tobytes(), 1)
    parity.check_stagewise(hip, workloads.zipf_tokens(200000, doc_len=500, vocab=20000).tobytes(), 2, engine.FLAG_FORCE_IDX64)
    rng = np.random.default_rng(77)
    for kind in parity.KINDS:
        for _ in range(6):
            d, w = parity.rand_collection(rng, kind)
            parity.check_final(hip, d, w)


def test_doubling_rounds_of_the_suffix_refinement(hip, oracle_mod, monkeypatch):
    """Shared monotone ramps over a u32 alphabet (long phrases without runs): the refinement switches to doubling rounds.  A small
    instance against the oracle, a 400 k-cell ramp in six strings by the device round trip (20 s before, with 200 001 rounds), then
    doubling forced from the first round on for ordinary inputs."""
    import time
    import torch
    from tests.test_engine_logic_sim import _long_run_collection, _shared_ramps
    monkeypatch.delenv("GRLBWT_RUN_KEYS_MIN", raising=False)
    monkeypatch.delenv("GRLBWT_DOUBLING_AFTER", raising=False)
    cells = _shared_ramps(3000, 5)
    with engine.Context(0, 0, hip) as ctx:
        ctx.upload(cells.tobytes(), 4)
        ctx.build()
        assert ctx.result_bytes() == oracle_mod.rl_bwt(cells.tobytes(), 4)
        assert ctx.round_info(0)["sort_iters"] <= 45
    big = _shared_ramps(400000, 6)
    t = torch.from_numpy(big.view(np.int32).copy()).to("cuda:0")
    out = torch.zeros_like(t)
    with engine.Context(0, 0, hip) as ctx:
        t0 = time.time()
        ctx.attach_device(t.data_ptr(), t.numel(), 4, keepalive=t)
        ctx.build()
        torch.cuda.synchronize()
        assert time.time() - t0 < 10.0
        assert ctx.round_info(0)["sort_iters"] <= 60
        nb, _ = ctx.result_size()
        n = ctx.invert_image(ctx.result_device_ptr(), nb, 4, out.data_ptr(), out.numel())
    torch.cuda.synchronize()
    assert n == t.numel() and torch.equal(out, t)
    monkeypatch.setenv("GRLBWT_RUN_KEYS_MIN", "0")
    monkeypatch.setenv("GRLBWT_DOUBLING_AFTER", "0")
    parity.check_stagewise(hip, workloads.sampled_reads(20000, 100, 100000, seed=11).tobytes(), 1)
    parity.check_stagewise(hip, workloads.zipf_tokens(200000, doc_len=500, vocab=20000).tobytes(), 2, engine.FLAG_FORCE_IDX64)
    parity.check_final(hip, _long_run_collection(2000, 4, 5), 1)
    rng = np.random.default_rng(78)
    for kind in parity.KINDS:
        for _ in range(6):
            d, w = parity.rand_collection(rng, kind)
            parity.check_final(hip, d, w)


def test_stagewise_idx64(hip, oracle_mod):
    parity.check_stagewise(hip, workloads.sampled_reads(5000, 80, 30000, seed=2).tobytes(), 1, engine.FLAG_FORCE_IDX64)


@pytest.mark.parametrize("kind", parity.KINDS)
def test_fuzz(hip, oracle_mod, kind):
    rng = np.random.default_rng(zlib.crc32(kind.encode()) + 7)
    for i in range(25):
        data, w = parity.rand_collection(rng, kind)
        flags = engine.FLAG_FORCE_IDX64 if i % 4 == 3 else 0
        parity.check_final(hip, data, w, flags)


def test_ill_formed_and_errors(hip):
    with engine.Context(0, 0, hip) as ctx:
        with pytest.raises(engine.IllFormedInput):
            ctx.upload(b"AC\nGT", 1)
        with pytest.raises(engine.GrlbwtError):
            ctx.upload(b"", 1)
        with pytest.raises(engine.GrlbwtError):
            ctx.build()


def test_long_equal_runs_and_long_phrases(hip, oracle_mod):
    # homopolymers (type scan walks), a single long string, and many empty strings
    data = b"A" * 50000 + b"C" * 30000 + b"\n" + b"ACGT" * 20000 + b"\n" + b"\n" * 1000 + b"T" * 70000 + b"\n"
    parity.check_final(hip, data, 1)


@pytest.mark.parametrize("w", [1, 2, 4, 8])
def test_equal_runs_across_word_and_tile_boundaries(hip, oracle_mod, w):
    # run-length structured strings: runs of 1..200 equal symbols end on every offset modulo 64 (the word of the
    # start-bit kernel) and modulo its LDS tile, with descents into and ascents out of the runs, plus runs that
    # reach the end of a string
    rng = np.random.default_rng(77 + w)
    dt = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[w]
    sep, hi = 1, {1: 6, 2: 300, 4: 70000, 8: 90000}[w]
    pieces = []
    for _ in range(40):
        syms = rng.integers(2, hi, size=400)
        lens = rng.integers(1, 200, size=400)
        lens[rng.integers(0, 400, size=5)] += 4096              # a few runs longer than a tile
        pieces.append(np.repeat(syms, lens).astype(dt))
        pieces.append(np.array([sep], dtype=dt))
    data = np.concatenate(pieces)
    parity.check_stagewise(hip, data.tobytes(), w)


def test_device_resident_input_torch(hip, oracle_mod):
    """The bench path: input already in HBM (torch tensor), output image stays in HBM."""
    import torch
    data = workloads.sampled_reads(30000, 100, 200000, seed=9)
    dev = torch.from_numpy(data.copy()).to("cuda:0")
    torch.cuda.synchronize()
    with engine.Context(0, 0, hip) as ctx:
        ctx.attach_device(dev.data_ptr(), dev.numel(), 1, keepalive=dev)
        ctx.build()
        got = ctx.result_bytes()
        assert ctx.result_device_ptr() != 0
    assert got == oracle_mod.rl_bwt(data.tobytes(), 1)


def test_lf_roundtrip_20mb(hip):
    data = workloads.sampled_reads(200000, 100, 2000000, seed=21).tobytes()
    parity.lf_roundtrip(parity.run_engine(hip, data, 1), data, 1)


def test_full_size_config_properties(hip):
    """BASELINE config[1] (1M x 100 bp, 101 MB): properties that do not need the oracle:
    header, maximal runs, symbol count, per-symbol histogram, and idx32 == idx64 builds."""
    from tests import bcr_check as bc
    data = workloads.uniform_reads(1000000, 100)
    a = parity.run_engine(hip, data, 1)
    sb, fb, sym, ln = bc.parse_rl_bwt(a)
    assert (sb, fb) == bc.header_widths(data, 1)
    assert bc.runs_are_maximal(sym)
    assert int(ln.sum()) == data.size
    hist = np.bincount(data, minlength=256)
    got = np.zeros(256, dtype=np.int64)
    np.add.at(got, sym.astype(np.int64), ln.astype(np.int64))
    assert np.array_equal(hist, got)          # the BWT is a permutation of the text
    b = parity.run_engine(hip, data, 1, engine.FLAG_FORCE_IDX64)
    assert hashlib.md5(a).hexdigest() == hashlib.md5(b).hexdigest()


def test_table_growth_when_prefix_is_unrepresentative(hip, oracle_mod):
    rep = (b"ACGTTGCA" * 16 + b"\n") * 8500
    data = rep + workloads.uniform_reads(30000, 100, seed=77).tobytes()
    parity.check_final(hip, data, 1)


def test_dist_path_single_rank_rccl(hip, oracle_mod, tmp_path):
    """The collection-level path over torch.distributed/RCCL (world_size 1 on this box): exercises the
    device-pointer callbacks (CUDA array interface views) and the merged-dictionary kernels on the GPU."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for case, w in (("reads", 1), ("tokens", 2)):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
               "--master-addr", "127.0.0.1", "--master-port", "29611",
               os.path.join(here, "dist_worker.py"), hip, "nccl", case, str(tmp_path)]
        # (with one rank every all-to-all block is the rank's own: route it through RCCL anyway)
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=os.path.dirname(here),
                           env=dict(os.environ, GRLBWT_A2A_SELF_VIA_COMM="1"))
        assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
        data = open(tmp_path / (case + ".input"), "rb").read()
        assert open(tmp_path / (case + ".rl_bwt"), "rb").read() == oracle_mod.rl_bwt(data, w)


def test_device_side_generator_matches_host(hip):
    import torch
    a = workloads.uniform_reads(5000, 100, seed=20260001)
    b = workloads.uniform_reads_torch(5000, 100, seed=20260001, device="cuda:0", chunk_reads=1300).cpu().numpy()
    assert np.array_equal(a, b)


def _expected_header_torch(text, w):
    """(sb, fb) of SURVEY A.8 from the input itself, on the device (no engine code): sb from max_sym + 4, fb from the largest
    byte frequency (-a 1) or the number of cells (-a 2/4/8)."""
    import torch
    n = text.numel()
    if w == 1:
        hist = torch.zeros(256, dtype=torch.int64, device=text.device)
        step = 1 << 28
        for o in range(0, n, step):
            hist += torch.bincount(text[o:o + step].to(torch.int64), minlength=256)
        mx = int(torch.nonzero(hist).max().item())
        F = int(hist.max().item())
    else:
        v = text.view(torch.int16).to(torch.int32) & 0xFFFF if w == 2 else text
        mx, F = int(v.max().item()), n
    return ((mx + 4).bit_length() + 7) // 8, (F.bit_length() + 7) // 8


def _assert_image_properties(ctx, text, w):
    """Header widths == the A.8 rule on the input, size == 16 + runs * (sb + fb), every run differs from its predecessor
    (maximal runs) and the run lengths add up to the input -- checked with torch on the image bytes, not by the engine."""
    import torch
    from grlbwt_amd import dist as gdist
    nb, nr = ctx.result_size()
    img = gdist._view(ctx.result_device_ptr(), nb, text.device)
    sb, fb = (int.from_bytes(bytes(img[o:o + 8].cpu().numpy()), "little") for o in (0, 8))
    assert (sb, fb) == _expected_header_torch(text, w), (sb, fb)
    assert nb == 16 + nr * (sb + fb)
    rec = img[16:].view(nr, sb + fb)
    step = 1 << 27
    total = 0
    for o in range(0, nr, step):
        blk = rec[o:o + step + 1]
        assert bool((blk[1:, :sb] != blk[:-1, :sb]).any(dim=1).all()), "a run repeats the symbol of the run before it"
        ln = torch.zeros(min(step, nr - o), dtype=torch.int64, device=text.device)
        for b in range(fb):
            ln += rec[o:o + step, sb + b].to(torch.int64) << (8 * b)
        assert int(ln.min().item()) > 0
        total += int(ln.sum().item())
    assert total == text.numel()
    return nb, nr


def test_full_size_encode_decode_round_trip(hip):
    """BASELINE config[1] at full size (101 MB): build the BWT, invert the image on the device
    (grl2plain + reverse_bwt kernels), compare with the input byte for byte."""
    import torch
    text = workloads.uniform_reads_torch(1000000, 100, device="cuda:0")
    out = torch.zeros_like(text)
    torch.cuda.synchronize()
    with engine.Context(0, 0, hip) as ctx:
        ctx.attach_device(text.data_ptr(), text.numel(), 1, keepalive=text)
        ctx.build()
        nb, _ = _assert_image_properties(ctx, text, 1)
        n = ctx.invert_image(ctx.result_device_ptr(), nb, 1, out.data_ptr(), out.numel())
    torch.cuda.synchronize()
    assert n == text.numel() and torch.equal(out, text)


@pytest.mark.parametrize("form", ["positions", "runs"])
def test_both_inverter_forms(hip, monkeypatch, form):
    """grlbwt_invert_image has two index forms (LF array over the positions; one record per run, which is what the 10 GB
    headline image takes): both forced on the same inputs, 32- and 64-bit positions, byte and uint16 cells."""
    import torch
    monkeypatch.setenv("GRLBWT_INVERT", form)
    for data, w, flags in ((workloads.sampled_reads(200000, 100, 2000000, seed=21), 1, 0),
                           (workloads.sampled_reads(50000, 100, 500000, seed=22), 1, engine.FLAG_FORCE_IDX64),
                           (workloads.zipf_tokens(500000, doc_len=300, vocab=20000), 2, 0),
                           (workloads.repetitive_copies(20, 100000, seed=9), 1, engine.FLAG_FORCE_IDX64)):
        t = torch.from_numpy(data.view(np.int16) if w == 2 else data).to("cuda:0")
        out = torch.zeros_like(t)
        with engine.Context(0, flags, hip) as ctx:
            ctx.attach_device(t.data_ptr(), data.size, w, keepalive=t)
            ctx.build()
            nb, _ = ctx.result_size()
            n = ctx.invert_image(ctx.result_device_ptr(), nb, w, out.data_ptr(), data.size)
        torch.cuda.synchronize()
        assert n == data.size and torch.equal(out, t)


def test_round_trip_u16_and_long_strings(hip):
    import torch
    for data, w in ((workloads.zipf_tokens(2000000, doc_len=1000, vocab=30000), 2),
                    (workloads.repetitive_copies(20, 300000, seed=9), 1)):
        t = torch.from_numpy(data.view(np.int16) if w == 2 else data).to("cuda:0")
        out = torch.zeros_like(t)
        with engine.Context(0, 0, hip) as ctx:
            ctx.attach_device(t.data_ptr(), data.size, w, keepalive=t)
            ctx.build()
            nb, _ = ctx.result_size()
            n = ctx.invert_image(ctx.result_device_ptr(), nb, w, out.data_ptr(), data.size)
        torch.cuda.synchronize()
        assert n == data.size and torch.equal(out, t)


@pytest.mark.parametrize("world,case", [(2, "reads_big"), (3, "tokens"), (2, "repetitive"), (4, "reads_big")])
def test_dist_kernels_multi_rank_on_one_gpu(hip, oracle_mod, tmp_path, world, case):
    """Collection-level path with world_size > 1 on this single-GPU box: the ranks share cuda:0 and exchange
    through gloo (host-staged), so the distributed kernels (merged dictionary, per-bucket rank counts, atom
    routing) run on the GPU with real multi-rank data.  RCCL itself is covered at world_size 1."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29650 + world),
           os.path.join(here, "dist_worker.py"), hip, "gloo-cuda", case, str(tmp_path)]
    # (the dictionary sharded by owner -- by default from 4 ranks and 2^27 dictionary symbols on -- is forced: its owner round
    # trips, the owner-side kernels and the sharded sort run on the GPU here; the gathered form runs in the 1 GB test below)
    env = dict(os.environ, GRLBWT_DIST_SHARDED_DICT_MIN="1", GRLBWT_DIST_SHARDED_DICT_MIN_SYMS="0")
    # (positions travel as (owner, offset) pairs in this form at any size; that nothing in it depends on the GLOBAL numbering staying
    # below 2^32 is checked on the CPU -- tests/test_dist_gloo.py pads every rank's part by 2^31 unused positions through a hook that
    # exists in the serial stand-in only)
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=os.path.dirname(here), env=env)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    data = open(tmp_path / (case + ".input"), "rb").read()
    w = 2 if case == "tokens" else 1
    assert open(tmp_path / (case + ".rl_bwt"), "rb").read() == oracle_mod.rl_bwt(data, w)


@pytest.mark.parametrize("config", ["configs2_repetitive_2.4GB", "configs4_u16_tokens_1GB"])
def test_baseline_configs_at_stated_size_round_trip(hip, config):
    """BASELINE.json configs[2] (100 copies x 24 Mbp pseudo-chromosome, 1e-3 substitutions, 2,400,000,100 bytes) and
    configs[4] (499,999,500 uint16 cells of Zipf tokens, 1 GB, -a 2) at FULL size: built, inverted on the device (grl2plain +
    reverse_bwt kernels) and compared with the input cell for cell.  (configs[2]: 100 strings of 24 M symbols, the
    inversion walks each of them sequentially: about 50 s.)"""
    import torch
    if config.startswith("configs2"):
        text, w = workloads.repetitive_copies_torch(100, 24000000, device="cuda:0"), 1
    else:
        text, w = workloads.zipf_tokens_torch(500000000, device="cuda:0"), 2
    back = torch.zeros_like(text)
    torch.cuda.synchronize()
    with engine.Context(0, 0, hip) as ctx:
        ctx.attach_device(text.data_ptr(), text.numel(), w, keepalive=text)
        ctx.build()
        st = ctx.stats()
        assert st["n_syms"] == text.numel() and st["n_strings"] == (100 if w == 1 else 499500)
        nb, nr = _assert_image_properties(ctx, text, w)
        assert nb == 16 + nr * (st["sb"] + st["fb"])
        n = ctx.invert_image(ctx.result_device_ptr(), nb, w, back.data_ptr(), back.numel())
    torch.cuda.synchronize()
    assert n == text.numel() and torch.equal(back, text)


def test_configs2_at_chromosome_scale_order_checked(hip):
    """BASELINE configs[2] at the size SURVEY 8(d) calls optional: 100 copies of a 248,956,422 bp sequence (the length of human
    chr1), 1e-3 substitutions per copy, 24.9 GB, 7.3 G phrase occurrences at level 0 (>= 2^32), 171 GB of device memory.  A full
    inversion walks 249 M dependent steps per string (minutes); the image is checked by (i) header widths, maximal
    runs and lengths through torch, (ii) every symbol's total against its count in the text, and (iii) **the ORDER: the last
    4 M cells of every string decoded by the per-run LF walk (grlbwt_invert_image_tails) and compared with the text** -- a BWT with
    the right symbol counts in a wrong order does not survive 400 M LF steps."""
    import torch
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    with engine.Context(0, 0, hip) as probe:        # (the arena earlier tests of this process grew is reused by this build: it counts as free)
        held = probe.memory_usage()["reserved_bytes"]
    if free + held < 215 << 30:
        pytest.skip("needs ~215 GB of device memory (free + what the engine's arena already holds): %d GB" % ((free + held) >> 30))
    L, k, tail = 248956422, 100, 4000000
    text = workloads.repetitive_copies_torch(k, L, device="cuda:0")
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    assert text.numel() == k * (L + 1)
    with engine.Context(0, 0, hip) as ctx:
        ctx.attach_device(text.data_ptr(), text.numel(), 1, keepalive=text)
        ctx.build()
        st = ctx.stats()
        assert st["n_syms"] == text.numel() and st["n_strings"] == k
        nb, nr = _assert_image_properties(ctx, text, 1)
        assert nb == 16 + nr * (st["sb"] + st["fb"])
        out = torch.zeros(k * tail, dtype=torch.uint8, device="cuda:0")
        ks, n = ctx.invert_image_tails(ctx.result_device_ptr(), nb, 1, tail, out.data_ptr(), out.numel())
    torch.cuda.synchronize()
    assert ks == k and n == k * tail
    assert torch.equal(out.view(k, tail), text.view(k, L + 1)[:, L + 1 - tail:])
    del text, out
    torch.cuda.empty_cache()


def test_result_pointer_is_complete_when_build_returns(hip, oracle_mod):
    """include/grlbwt_hip.h: "results are complete when a call returns" -- the image is read from ANOTHER stream (torch's
    default stream) right after build(), with no synchronisation of the engine's stream by the caller."""
    import torch
    from grlbwt_amd import dist as gdist
    data = workloads.sampled_reads(300000, 100, 3000000, seed=31)
    dev = torch.from_numpy(data.copy()).to("cuda:0")
    torch.cuda.synchronize()
    exp = oracle_mod.rl_bwt(data.tobytes(), 1)
    with engine.Context(0, 0, hip) as ctx:
        for _ in range(3):
            ctx.attach_device(dev.data_ptr(), dev.numel(), 1, keepalive=dev)
            ctx.build()
            nb, _ = ctx.result_size()
            img = gdist._view(ctx.result_device_ptr(), nb, torch.device("cuda:0")).clone()     # default stream, immediately
            assert bytes(img.cpu().numpy()) == exp


def test_giant_phrases_hashed_by_the_wave(hip, oracle_mod):
    """(see tests/test_engine_logic_sim.py) phrases of 90 k - 2 M cells: hashed in 64 pieces by the whole wave, duplicates found.
    Against the oracle where it finishes in seconds (it sorts the suffixes of a run of c equal cells in c^2 steps: 90 k yes, 2 M
    no); the 2 M-cell gaps through the device-side inverter instead (every string comes back at its input index)."""
    import torch
    g = b"A" * 90000
    data = g + b"\n" + g + b"\n" + b"A" * 89999 + b"C" + b"\n" + g + b"A\n" + b"CGT" * 10 + b"\n"
    parity.check_final(hip, data, 1)
    rng = np.random.default_rng(5)
    one = bytes(rng.integers(1, 256, size=2000000).astype(np.uint8)) + b"\x00"      # one string: the deepest level is a single phrase
    parity.check_final(hip, one, 1)
    two = b"N" * 2000000
    gaps = np.frombuffer(b"ACGT" + two + b"TTGA\n" + b"ACGT" + two + b"TTGA\n" + b"GG" + two + b"C\n" + b"ACGT" + two + b"TTGC\n", dtype=np.uint8)
    text = torch.from_numpy(gaps.copy()).to("cuda:0")
    back = torch.zeros_like(text)
    with engine.Context(0, 0, hip) as ctx:
        ctx.attach_device(text.data_ptr(), text.numel(), 1, keepalive=text)
        ctx.build()
        nb, nr = ctx.result_size()
        n = ctx.invert_image(ctx.result_device_ptr(), nb, 1, back.data_ptr(), back.numel())
    torch.cuda.synchronize()
    assert n == text.numel() and torch.equal(back, text)


def test_runs_and_parse_beyond_2_pow_32(hip):
    """4.4 G copies of the string "A": 8.8 G cells, a parse of 4.4 G phrase occurrences in ONE round (>= 2^32: the 64-bit
    build has no bound on them) and two BWT runs of 4.4 G symbols each (>= 2^32: the tiles of the stream merge that hold
    them take their 64-bit form).  The image is known in closed form: every "$"-suffix is preceded by A, every "A$"-suffix by
    the separator -- header (1, 5), then (A, k), (separator, k)."""
    import torch
    k = (1 << 32) + (1 << 27) + 12345
    text = torch.empty(2 * k, dtype=torch.uint8, device="cuda:0")
    text[0::2] = 65
    text[1::2] = 10
    torch.cuda.synchronize()
    with engine.Context(0, 0, hip) as ctx:
        ctx.attach_device(text.data_ptr(), text.numel(), 1, keepalive=text)
        ctx.build()
        st = ctx.stats()
        assert st["n_strings"] == k and st["n_syms"] == 2 * k and (st["sb"], st["fb"]) == (1, 5)
        got = ctx.result_bytes()
        assert ctx.round_info(0)["parse_size"] == k
    del text
    exp = (1).to_bytes(8, "little") + (5).to_bytes(8, "little") + bytes([65]) + k.to_bytes(5, "little") + bytes([10]) + k.to_bytes(5, "little")
    assert got == exp, (len(got), got[:40])


def test_final_bytes_vs_oracle_250MB(hip, oracle_mod):
    """The largest input compared byte for byte with the oracle ("What's weak" 3): 1.65 M Illumina-style reads of
    150 bp from an 8.3 Mbp genome (249 MB, 30x coverage, 0.5 % substitutions -- the headline distribution at 1/40 of its
    size), default switches, one build; the oracle takes about a minute of one host core."""
    import torch
    data = workloads.sampled_reads(1650000, 150, 8300000, seed=20260417)
    assert data.size == 1650000 * 151
    dev = torch.from_numpy(data).to("cuda:0")
    torch.cuda.synchronize()
    with engine.Context(0, 0, hip) as ctx:
        ctx.attach_device(dev.data_ptr(), dev.numel(), 1, keepalive=dev)
        ctx.build()
        got = ctx.result_bytes()
    del dev
    exp = oracle_mod.rl_bwt(data.tobytes(), 1)
    assert len(got) == len(exp)
    assert got == exp, "HIP .rl_bwt differs from the oracle on the 249 MB input"


def test_idx64_build_round_trip_4_3GB(hip):
    """A TRUE 64-bit index build (n >= 2^32 - 256 cells, no FORCE flag): 28,443,491 x 150 bp Illumina-style reads
    (4,294,967,141 bytes), built and inverted on the device (grl2plain + reverse_bwt kernels), compared byte for byte."""
    import torch
    reads = 28443491
    text = workloads.sampled_reads_torch(reads, 150, 140000000, seed=20260003, device="cuda:0")
    assert text.numel() >= 0xFFFFFF00
    torch.cuda.synchronize()
    with engine.Context(0, 0, hip) as ctx:
        ctx.attach_device(text.data_ptr(), text.numel(), 1, keepalive=text)
        ctx.build()
        assert ctx.counters()["idx_bytes"] == 8
        st = ctx.stats()
        assert (st["n_strings"], st["n_syms"]) == (reads, text.numel())
        nb, nr = _assert_image_properties(ctx, text, 1)
        img = torch.empty(nb, dtype=torch.uint8, device="cuda:0")
        from grlbwt_amd import dist as gdist
        img.copy_(gdist._view(ctx.result_device_ptr(), nb, torch.device("cuda:0")))
        torch.cuda.synchronize()
    # a fresh context: the build's buffers are gone, the inversion has the device to itself (36 B per symbol)
    out = torch.zeros_like(text)
    with engine.Context(0, 0, hip) as ctx:
        n = ctx.invert_image(img.data_ptr(), nb, 1, out.data_ptr(), out.numel())
    torch.cuda.synchronize()
    assert n == text.numel()
    assert torch.equal(out, text)


def test_arena_takes_a_multi_gib_first_step(hip):
    """The scratch arena in a FRESH process whose first request is several GiB (a host upload of 2.6 GB): it must keep
    growing afterwards.  hipMemSetAccess refuses a piece mapped behind a piece of another size (tools/arena_probe.hip); when
    the arena grew by "whatever the request needs" it stopped after such a first step and the rest of the build lived in
    hipMalloc slabs (reserved memory 2x the peak, seconds of allocation).  Checked through GRLBWT_POOL_TRACE and
    grlbwt_memory_usage in a child process."""
    import subprocess
    code = (
        "import os, sys, json\n"
        "sys.path.insert(0, %r)\n"
        "import torch\n"
        "from grlbwt_amd import engine, workloads\n"
        "t = workloads.sampled_reads_torch(17218543, 150, 86000000, seed=20260003, device='cuda:0').cpu().numpy()\n"
        "with engine.Context(0, 0, %r) as ctx:\n"
        "    ctx.upload(t, 1)\n"
        "    ctx.build()\n"
        "    m = ctx.memory_usage(); nb, nr = ctx.result_size()\n"
        "print('RESULT', json.dumps({'n': int(t.size), 'mem': m, 'runs': nr}))\n"
    ) % (ROOT, hip)
    env = dict(os.environ, GRLBWT_POOL_TRACE="1")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    assert res["n"] > (2 << 30)
    trace = [l for l in p.stderr.splitlines() if "pool:" in l and "arena steps" in l]
    assert trace, p.stderr[-2000:]
    assert "(0 refused), 0 hipMalloc slabs" in trace[-1], trace[-1]
    mem = res["mem"]
    assert mem["reserved_bytes"] - mem["peak_live_bytes"] < (4 << 30), mem


def test_headline_10GB_round_trip(hip):
    """BASELINE.json configs[3] / the metric's own configuration at its stated size on ONE GPU: 66,225,166 x 150 bp
    Illumina-style reads (10,000,000,066 bytes) -- what bench.py times.  The image must (a) carry the header of SURVEY A.8
    and maximal runs, (b) be the image bench.py hashes (tests/golden/headline_10GB.json), and (c) decode back to the input
    byte for byte: grlbwt_invert_image through the per-run LF records (scripts/reverse_bwt.cpp:36-52 + fm_index.h:79-83
    over the run-length BWT; the per-position LF array would need 360 GB here)."""
    import torch
    reads = 66225166
    text = workloads.sampled_reads_torch(reads, 150, 330000000, seed=20260003, device="cuda:0")
    assert text.numel() == 10000000066
    torch.cuda.synchronize()
    with engine.Context(0, 0, hip) as ctx:
        ctx.attach_device(text.data_ptr(), text.numel(), 1, keepalive=text)
        ctx.build()
        st = ctx.stats()
        assert (st["n_strings"], st["n_syms"]) == (reads, text.numel())
        nb, nr = _assert_image_properties(ctx, text, 1)
        from grlbwt_amd import dist as gdist
        img = torch.empty(nb, dtype=torch.uint8, device="cuda:0")
        img.copy_(gdist._view(ctx.result_device_ptr(), nb, torch.device("cuda:0")))
        torch.cuda.synchronize()
    gold = json.load(open(os.path.join(parity.GOLD, "headline_10GB.json")))
    assert (nb, nr) == (gold["image_bytes"], gold["runs"])
    assert workloads.md5_device(img) == gold["md5"]
    out = torch.zeros_like(text)
    with engine.Context(0, 0, hip) as ctx:       # a fresh context: the build's buffers are gone
        n = ctx.invert_image(img.data_ptr(), nb, 1, out.data_ptr(), out.numel())
    torch.cuda.synchronize()
    assert n == text.numel()
    assert torch.equal(out, text)


def test_sharded_equals_single_gpu_image(hip, tmp_path):
    """bench.py's N > 1 path on this single-GPU box: the SAME Illumina-style collection (a 1 GB instance of the headline
    distribution: 6,622,517 x 150 bp from a 33 Mbp genome) built by one context and by two record shards (two ranks
    sharing cuda:0, gloo transport) must give identical .rl_bwt bytes."""
    import subprocess
    import sys
    import torch
    reads, genome = 6622517, 33000000
    text = workloads.sampled_reads_torch(reads, 150, genome, seed=20260003, device="cuda:0")
    torch.cuda.synchronize()
    with engine.Context(0, 0, hip) as ctx:
        ctx.attach_device(text.data_ptr(), text.numel(), 1, keepalive=text)
        ctx.build()
        single = ctx.result_bytes()
    del text
    torch.cuda.empty_cache()
    here = os.path.dirname(os.path.abspath(__file__))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29671",
           os.path.join(here, "dist_worker.py"), hip, "gloo-cuda", "illumina_dev:%d:%d" % (reads, genome), str(tmp_path)]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=os.path.dirname(here))
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    want = "%s %d" % (hashlib.md5(single).hexdigest(), len(single))
    for r in range(2):
        assert open(tmp_path / ("illumina_dev.rank%d.md5" % r)).read() == want
    # the same collection through RCCL (one rank: all this box has).  Its level-0 cell exchange is a 2.1 GB block, the size
    # at which torch 2.10 / RCCL 2.26 deliver half of an all-to-all silently: the engine sends such blocks in rounds
    os.remove(tmp_path / "illumina_dev.rank0.md5")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", "29672",
           os.path.join(here, "dist_worker.py"), hip, "nccl", "illumina_dev:%d:%d" % (reads, genome), str(tmp_path)]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=os.path.dirname(here),
                       env=dict(os.environ, GRLBWT_A2A_SELF_VIA_COMM="1", GRLBWT_DIST_SHARDED_DICT_MIN="1", GRLBWT_DIST_SHARDED_DICT_MIN_SYMS="0"))
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]      # (this leg with the dictionary sharded by owner: its round trips through RCCL)
    assert open(tmp_path / "illumina_dev.rank0.md5").read() == want


def test_borrowed_text_that_changes_after_attach_is_refused(hip):
    """grlbwt_text_attach_device takes the statistics -- and with them the alphabet the level-0 direct index trusts -- at attach time; the
    buffer stays the caller's.  A cell value that was not there at attach time would take another symbol's 3-bit code: the build
    looks at a strided sample of a borrowed text again and refuses it.  The engine's own copies (upload, file) cannot change."""
    import torch
    data = workloads.sampled_reads(40000, 150, 300000, seed=77)            # 6 MB: large enough for the sampled direct index
    t = torch.from_numpy(data.copy()).to("cuda:0")
    with engine.Context(0, 0, hip) as ctx:
        ctx.attach_device(t.data_ptr(), t.numel(), 1, keepalive=t)
        ctx.build()
        good = ctx.result_bytes()
    with engine.Context(0, 0, hip) as ctx:
        ctx.attach_device(t.data_ptr(), t.numel(), 1, keepalive=t)
        t[1::2] = torch.where(t[1::2] == ord("A"), torch.full_like(t[1::2], ord("R")), t[1::2])      # a value the statistics did not see
        torch.cuda.synchronize()
        with pytest.raises(engine.GrlbwtError, match="has changed since its statistics"):
            ctx.build()
    with engine.Context(0, 0, hip) as ctx:               # the same bytes, attached as they are now: a text like any other
        ctx.attach_device(t.data_ptr(), t.numel(), 1, keepalive=t)
        ctx.build()
        assert ctx.result_bytes() != good

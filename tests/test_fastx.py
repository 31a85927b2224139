"""f3 ingestion: FASTA/FASTQ (optionally gzip) -> one string per line, with optional reverse complements.

The expectations in tests/golden/fastx_ref.json were produced by the REFERENCE's own converter (fastx2plain_format over kseq,
built from its sources by oracle/Makefile `ref`; tests/golden/make_fastx_fixtures.py).  They pin oracle/fastx_oracle.c; the
engine's device path is compared with the fixtures and with the oracle on random inputs."""
import hashlib
import json
import os
import subprocess
import tempfile
import zlib

import numpy as np
import pytest

from grlbwt_amd import engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
FIX = json.load(open(os.path.join(GOLD, "fastx_ref.json")))["cases"]


def md5(b):
    return hashlib.md5(bytes(b)).hexdigest()


def raw_bytes(name):
    raw = open(os.path.join(GOLD, "fastx", name), "rb").read()
    if raw[:2] == b"\x1f\x8b":
        out, rest = b"", raw
        while rest:
            o = zlib.decompressobj(31)
            out += o.decompress(rest)
            rest = o.unused_data
        return out
    return raw


@pytest.mark.parametrize("c", FIX, ids=[c["name"] for c in FIX])
def test_oracle_matches_reference_converter(oracle_mod, c):
    assert md5(open(os.path.join(GOLD, "fastx", c["name"]), "rb").read()) == c["input_md5"]
    raw = raw_bytes(c["name"])
    for mode, rc in (("plain", False), ("revcomp", True)):
        exp = c[mode]
        if exp["exit"] == 0:
            out, ns = oracle_mod.fastx2plain(raw, rc)
            assert (md5(out), len(out), ns) == (exp["md5"], exp["size"], exp["n_strings"])
        else:
            with pytest.raises(oracle_mod.NotDNA) as e:
                oracle_mod.fastx2plain(raw, rc)
            assert exp["exit"] == 1 and ("(invalid symbol:%s)" % e.value.args[0]) in exp["stderr"]


def rand_fastx(rng):
    """Random well-formed and damaged FASTA/FASTQ texts (line ends, blank lines, odd symbols, truncation)."""
    nl = b"\r\n" if rng.random() < 0.2 else b"\n"
    alpha = np.frombuffer(b"ACGT" if rng.random() < 0.7 else b"ACGTNacgt-*", dtype=np.uint8)
    parts = []
    fq = rng.random() < 0.5
    for i in range(int(rng.integers(1, 12))):
        L = int(rng.integers(0, 70))
        seq = bytes(rng.choice(alpha, size=L))
        if fq:
            qual = bytes(rng.integers(33, 74, size=L).astype(np.uint8))
            parts += [b"@r%d x" % i + nl, seq + nl, b"+" + nl, qual + nl]
        else:
            w = int(rng.integers(1, 40))
            parts.append((b">" if rng.random() < 0.9 else b"@") + b"r%d c" % i + nl)
            parts += [seq[k:k + w] + nl for k in range(0, L, w)]
            if rng.random() < 0.3:
                parts.append(nl)
    data = b"".join(parts)
    r = rng.random()
    if r < 0.15 and len(data) > 2:
        data = data[:int(rng.integers(1, len(data)))]           # truncated anywhere
    elif r < 0.25:
        data = data.rstrip(b"\r\n")
    elif r < 0.40 and fq:                                        # FASTQ over several lines / with garbage between records
        lines = data.split(nl)
        out = []
        for ln in lines:
            if len(ln) > 8 and rng.random() < 0.5:
                k = int(rng.integers(1, len(ln)))
                out += [ln[:k], ln[k:]]
            else:
                out.append(ln)
            if rng.random() < 0.05:
                out.append(b"junk" if rng.random() < 0.5 else b"")
        data = nl.join(out)
    return data


def test_oracle_matches_live_reference_on_random_inputs(oracle_mod):
    """The restatement against the reference's converter itself (oracle/_ref/fastx2plain), damaged inputs included."""
    prog = oracle_mod.ref_prog("fastx2plain")
    if not prog and oracle_mod.build_ref():
        prog = oracle_mod.ref_prog("fastx2plain")
    if not prog:
        pytest.skip("oracle/_ref/fastx2plain not built and /root/reference absent")
    rng = np.random.default_rng(99)
    with tempfile.TemporaryDirectory() as td:
        for it in range(150):
            data = rand_fastx(rng)
            fin, fout = os.path.join(td, "in.fx"), os.path.join(td, "out")
            open(fin, "wb").write(data)
            for rc in (False, True):
                p = subprocess.run([prog, fin, fout, "1" if rc else "0"], capture_output=True)      # (bytes: a '\r' may be the symbol named)
                try:
                    out, ns = oracle_mod.fastx2plain(data, rc)
                    assert p.returncode == 0, (it, data, p.stderr)
                    assert open(fout, "rb").read() == out and (b"n_strings %d" % ns) in p.stdout, (it, rc, data)
                except oracle_mod.NotDNA as e:
                    assert p.returncode == 1 and ("(invalid symbol:%s)" % e.args[0]).encode("latin-1") in p.stderr, (it, data)


# ---------------------------------------------------------------------------------- the engine's device path
@pytest.fixture(scope="module")
def sim():
    from tests import simlib
    return simlib.sim_library()


def convert(ctx, data, rc, on_gpu):
    cap = (2 if rc else 1) * len(data) + 64
    if on_gpu:
        import torch
        src = torch.frombuffer(bytearray(data + b"\0" * 16), dtype=torch.uint8).to("cuda:0")
        dst = torch.zeros(cap, dtype=torch.uint8, device="cuda:0")
        n, ns = ctx.fastx_convert(src.data_ptr(), len(data), rc, dst.data_ptr(), cap)
        torch.cuda.synchronize()
        return bytes(dst[:n].cpu().numpy()), ns
    src = np.frombuffer(data + b"\0" * 16, dtype=np.uint8).copy()
    dst = np.zeros(cap, dtype=np.uint8)
    n, ns = ctx.fastx_convert(src.ctypes.data, len(data), rc, dst.ctypes.data, cap)
    return bytes(dst[:n]), ns


def check_engine(lib, oracle_mod, on_gpu):
    with engine.Context(0, 0, lib) as ctx:
        for c in FIX:
            raw = raw_bytes(c["name"])
            for mode, rc in (("plain", False), ("revcomp", True)):
                exp = c[mode]
                if exp["exit"] == 0:
                    out, ns = convert(ctx, raw, rc, on_gpu)
                    assert (md5(out), len(out), ns) == (exp["md5"], exp["size"], exp["n_strings"]), (c["name"], mode)
                else:
                    with pytest.raises(engine.NotDNA) as e:
                        convert(ctx, raw, rc, on_gpu)
                    assert e.value.code == -86 and str(e.value).split("grlbwt error -86: ")[1] in exp["stderr"], (c["name"], str(e.value))
        # random inputs against the oracle, damaged ones included (those take the record-by-record walk)
        rng = np.random.default_rng(5)
        for it in range(200):
            data = rand_fastx(rng)
            for rc in (False, True):
                try:
                    got = convert(ctx, data, rc, on_gpu)
                except engine.NotDNA as e:
                    with pytest.raises(oracle_mod.NotDNA) as o:
                        oracle_mod.fastx2plain(data, rc)
                    assert ("(invalid symbol:%s)" % o.value.args[0]) in str(e), (it, data)
                    continue
                assert got == oracle_mod.fastx2plain(data, rc), (it, rc, data)
        with pytest.raises(engine.GrlbwtError):
            convert(ctx, b"ACGT\nACGT\n", False, on_gpu)            # not FASTA/Q


def file_cases(lib, oracle_mod, tmp_path):
    """grlbwt_text_load_fastx: file (gzip or not) -> text in HBM -> the BWT of the converted collection."""
    for name, rc in (("fa_wrapped60.fa", False), ("fq_gz.fq.gz", False), ("fq_gz_two_members.fq.gz", True), ("fa_gz.fa.gz", True), ("fq_regular.fq", True)):
        path = os.path.join(GOLD, "fastx", name)
        assert engine.fastx_probe(path, lib) == (True, name.endswith(".gz"))
        text, ns = oracle_mod.fastx2plain(raw_bytes(name), rc)
        with engine.Context(0, 0, lib) as ctx:
            assert ctx.load_fastx(path, rc) == ns
            st = ctx.stats()
            assert st["n_strings"] == ns and st["n_syms"] == len(text)
            ctx.build()
            assert ctx.result_bytes() == oracle_mod.rl_bwt(text, 1)
    # bytes behind the last gzip member that do not start another one (zero padding of blocked / tape files, stray bytes)
    # are ignored, as gzread -- which the reference's converter reads through -- does; a member split over two reads still counts
    gz = open(os.path.join(GOLD, "fastx", "fq_gz.fq.gz"), "rb").read()
    text, ns = oracle_mod.fastx2plain(raw_bytes("fq_gz.fq.gz"), False)
    for tail in (b"\0" * 700, b"\x1f", b"junk behind the member"):
        padded = tmp_path / "padded.fq.gz"
        padded.write_bytes(gz + tail)
        with engine.Context(0, 0, lib) as ctx:
            assert ctx.load_fastx(str(padded), False) == ns
            assert ctx.stats()["n_syms"] == len(text)
    plain = os.path.join(GOLD, "test_byte_alphabet.txt")
    assert engine.fastx_probe(plain, lib) == (False, False)
    fake = tmp_path / "not_really.gz"                     # the extension alone does not make a gzip file (check_gzip: + magic number)
    fake.write_bytes(b">x\nACGT\n")
    assert engine.fastx_probe(str(fake), lib) == (True, False)
    with engine.Context(0, 0, lib) as ctx:
        with pytest.raises(engine.NotDNA):
            ctx.load_fastx(os.path.join(GOLD, "fastx", "fq_with_N.fq"), True)
        with pytest.raises(engine.GrlbwtError):
            ctx.load_fastx(str(tmp_path / "missing.fa"))


def test_engine_logic_on_the_stand_in(sim, oracle_mod):
    check_engine(sim, oracle_mod, False)


def test_file_loader_on_the_stand_in(sim, oracle_mod, tmp_path):
    file_cases(sim, oracle_mod, tmp_path)


@pytest.mark.gpu
def test_fastx_hip(oracle_mod):
    import __graft_entry__ as g
    check_engine(g.build_hip(), oracle_mod, True)


@pytest.mark.gpu
def test_fastx_files_hip(oracle_mod, tmp_path):
    import __graft_entry__ as g
    file_cases(g.build_hip(), oracle_mod, tmp_path)


@pytest.mark.gpu
def test_fastx_large_reads_round_trip(oracle_mod, tmp_path):
    """A FASTQ of 400,000 x 150 bp reads (gzip), with reverse complements: converted on the device, compared with the reads."""
    import gzip
    import torch
    import __graft_entry__ as g
    from grlbwt_amd import workloads
    reads, L = 400000, 150
    text = workloads.sampled_reads(reads, L, 3000000, seed=8)                 # reads separated by '\\n'
    rows = text.reshape(reads, L + 1)[:, :L]
    qual = np.full((reads, L), ord("I"), dtype=np.uint8)
    rec = np.concatenate([np.tile(np.frombuffer(b"@r\n", dtype=np.uint8), (reads, 1)), rows, np.full((reads, 1), 10, np.uint8),
                          np.tile(np.frombuffer(b"+\n", dtype=np.uint8), (reads, 1)), qual, np.full((reads, 1), 10, np.uint8)], axis=1)
    path = tmp_path / "reads.fq.gz"
    with gzip.open(path, "wb", compresslevel=1) as f:
        f.write(rec.tobytes())
    comp = np.zeros(256, dtype=np.uint8)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    both = np.concatenate([rows, np.full((reads, 1), 10, np.uint8), comp[rows[:, ::-1]], np.full((reads, 1), 10, np.uint8)], axis=1)
    with engine.Context(0, 0, g.build_hip()) as ctx:
        assert ctx.load_fastx(str(path), True) == 2 * reads
        assert ctx.stats()["n_syms"] == both.size
        ctx.build()
        nb, _ = ctx.result_size()
        out = torch.zeros(both.size, dtype=torch.uint8, device="cuda:0")
        n = ctx.invert_image(ctx.result_device_ptr(), nb, 1, out.data_ptr(), out.numel())
    assert n == both.size and np.array_equal(out.cpu().numpy(), both.reshape(-1))

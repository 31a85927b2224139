"""CLI contract of the reference (main.cpp:43-154): flags, naming, exit codes."""
import hashlib
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def cli():
    import __graft_entry__ as g
    g.build_hip()
    return g.build_cli()


def run(cli, *args, cwd=None):
    p = subprocess.run([cli, *args], cwd=cwd, capture_output=True, text=True)
    return p.returncode, p.stdout, p.stderr


def test_cli_argument_contract(cli, tmp_path):
    f = os.path.join(GOLD, "test_2bytes_alphabet.txt")
    assert run(cli)[0] == 106                                   # TEXT is required
    assert run(cli, "-v")[0] == 106                             # -v alone still needs TEXT [probe in SURVEY 8b]
    rc, out, _ = run(cli, f, "-v")
    assert rc == 0 and out.strip() == "v1.0.1 alpha"
    assert run(cli, str(tmp_path / "missing.txt"))[0] == 105
    assert run(cli, f, "-a", "3")[0] == 105
    assert run(cli, f, "-b", "6")[0] == 105
    assert run(cli, f, "-f", "1.5")[0] == 105
    assert run(cli, f, "-T", str(tmp_path / "nodir"))[0] == 105
    assert run(cli, f, "--gpus", "0")[0] == 105
    rc, out, _ = run(cli, "--help")
    assert rc == 0 and "--gpus" in out and "--rev-comp" in out
    assert run(cli, f, "-a", "2", "-R")[0] == 105                 # reverse complements need a FASTA/Q input
    assert run(cli, os.path.join(GOLD, "fastx", "fq_regular.fq"), "--fastx", "-a", "2")[0] == 105
    assert run(cli, f, "--fastx")[0] == 105                       # --fastx on a file that is not FASTA/Q
    assert "--fastx" in out


@pytest.mark.gpu
def test_cli_end_to_end(cli, tmp_path):
    tab = json.load(open(os.path.join(GOLD, "golden_table.json")))
    # default naming: basename(TEXT).rl_bwt in the CWD
    rc, out, err = run(cli, os.path.join(GOLD, "test_byte_alphabet.txt"), "-t", "4", "-b", "2", cwd=str(tmp_path))
    assert rc == 0, err
    blob = open(tmp_path / "test_byte_alphabet.rl_bwt", "rb").read()
    assert hashlib.md5(blob).hexdigest() == tab["test_byte_alphabet.txt"]["md5"]
    assert "Parsing round 8" in out and "The resulting BCR BWT was stored in" in out
    # the reference's stage labels (exact_par_phase.cpp:380,410,111,124,428,452; exact_ind_phase.cpp:121,141,270) and its
    # report_time wording (utils.h:109-126): harnesses grep them
    for label in ("Reading the file", "Computing the dictionary of LMS phrases", "Compacting the dictionary",
                  "Sorting the dictionary and constructing the preliminary BWT", "Compressing the dictionary",
                  "Assigning metasymbols to the LMS phrases", "Creating the parse of the text", "Inferring the BWT",
                  "Computing the deepest recursive BWT", "Inducing the BWT for parse 8", "Computing the number of induced symbols",
                  "Performing the induction from the previous BWT", "Assembling the new BWT", "Elapsed time ("):
        assert label in out, label
    assert "grlbwt-timing: read+upload" in out
    # -o with an extension: the last extension is replaced (main.cpp:112-113)
    rc, out, err = run(cli, os.path.join(GOLD, "test_2bytes_alphabet.txt"), "-a", "2", "-o", str(tmp_path / "x.y"))
    assert rc == 0, err
    blob = open(tmp_path / "x.rl_bwt", "rb").read()
    assert hashlib.md5(blob).hexdigest() == tab["test_2bytes_alphabet.txt"]["md5"]
    # ill-formed input: message on stdout, exit 1
    bad = tmp_path / "bad.txt"
    bad.write_bytes(b"AC\nGT")
    rc, out, _ = run(cli, str(bad), cwd=str(tmp_path))
    assert rc == 1 and "Error: the file is ill formed" in out


@pytest.mark.gpu
def test_cli_collection_level_mode_over_rccl(cli, tmp_path):
    """`grlbwt --gpus N`: one process per GPU, record shards cut by the parent, the library's own RCCL transport
    (grlbwt_rccl_comm_create) and grlbwt_dist_build.  This box has one GPU: the path runs with one rank
    (GRLBWT_CLI_FORCE_RCCL=1), and a request for two GPUs must end on every rank with an error instead of hanging."""
    tab = json.load(open(os.path.join(GOLD, "golden_table.json")))
    env = dict(os.environ, GRLBWT_CLI_FORCE_RCCL="1", GRLBWT_A2A_SELF_VIA_COMM="1")
    p = subprocess.run([cli, os.path.join(GOLD, "test_byte_alphabet.txt"), "-o", str(tmp_path / "one")], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert hashlib.md5(open(tmp_path / "one.rl_bwt", "rb").read()).hexdigest() == tab["test_byte_alphabet.txt"]["md5"]
    assert "Parsing round 8" in p.stdout and "on 1 GPUs (record shards, RCCL)" in p.stdout and "grlbwt-timing:" in p.stdout
    p = subprocess.run([cli, os.path.join(GOLD, "test_2bytes_alphabet.txt"), "-a", "2", "-o", str(tmp_path / "two")], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert hashlib.md5(open(tmp_path / "two.rl_bwt", "rb").read()).hexdigest() == tab["test_2bytes_alphabet.txt"]["md5"]
    bad = tmp_path / "bad.txt"
    bad.write_bytes(b"AC\nGT")
    p = subprocess.run([cli, str(bad)], capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=600)
    assert p.returncode == 1 and "Error: the file is ill formed" in p.stdout
    import torch
    if torch.cuda.device_count() == 1:
        p = subprocess.run([cli, os.path.join(GOLD, "test_byte_alphabet.txt"), "--gpus", "2", "-o", str(tmp_path / "x")], capture_output=True, text=True,
                           timeout=600)
        assert p.returncode == 3 and "could not load its shard" in p.stderr


@pytest.mark.gpu
def test_cli_fasta_fastq_inputs(cli, tmp_path, oracle_mod):
    """FASTA/FASTQ (gzip) inputs from the command line, with and without -R: the .rl_bwt of the converted collection."""
    import zlib
    for name, flags in (("fa_wrapped60.fa", ["--fastx"]), ("fq_gz.fq.gz", ["-R"]), ("fa_gz.fa.gz", ["--rev-comp"])):
        raw = open(os.path.join(GOLD, "fastx", name), "rb").read()
        if raw[:2] == b"\x1f\x8b":
            raw = zlib.decompress(raw, 31)
        text, _ = oracle_mod.fastx2plain(raw, flags != ["--fastx"])
        p = subprocess.run([cli, os.path.join(GOLD, "fastx", name), "-o", str(tmp_path / "fx")] + flags, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0 and "The input is in FASTA/Q format" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
        assert open(tmp_path / "fx.rl_bwt", "rb").read() == oracle_mod.rl_bwt(text, 1)
    p = subprocess.run([cli, os.path.join(GOLD, "fastx", "fq_with_N.fq"), "-R", "-o", str(tmp_path / "n")], capture_output=True, text=True, timeout=600)
    assert p.returncode == 1 and "The input seems not to be DNA (invalid symbol:N)" in p.stderr


@pytest.mark.gpu
def test_cli_plain_input_that_starts_like_fasta(cli, tmp_path, oracle_mod):
    """The reference has its FASTA/Q branch switched off (main.cpp:117-136) and takes ANY file as one-string-per-line cells.
    A plain collection whose first byte is '>' or '@' must therefore give the BWT of exactly its bytes; conversion is opt-in."""
    for first in (b">", b"@"):
        data = first + b" quoted line\n>> deeper\n@handle says ACGT\nplain\n"
        p = tmp_path / "mail.txt"
        p.write_bytes(data)
        r = subprocess.run([cli, str(p), "-o", str(tmp_path / "mail")], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        assert "The input is in FASTA/Q format" not in r.stdout and "pass --fastx" in r.stderr
        assert open(tmp_path / "mail.rl_bwt", "rb").read() == oracle_mod.rl_bwt(data, 1)

"""Pins the CPU oracle (oracle/grlbwt_oracle.c) before anything trusts it.

Golden vectors: the reference has no test suite; its builder (main.cpp) needs SDSL-lite and
cannot be rebuilt in this image, its .rl_bwt consumer programs can (oracle/Makefile target `ref`).  The oracle is
pinned by: the table of reference outputs SURVEY.md section 8c recorded (md5, size, header, run count, byte-exact tiny
cases -- this file); what the reference's own reader and re-writer make of the oracle's bytes
(tests/test_consumers.py, tests/golden/ref_consumers.json); the textbook definition (naive sorter) and LF inversion.
"""
import hashlib
import os
import zlib

import numpy as np
import pytest

from tests import bcr_check as bc

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
REF_BYTE = os.path.join(GOLD, "test_byte_alphabet.txt")   # the reference's own test_data file (2.96 MB)


def test_golden_tiny_newline(oracle_mod):
    # "\n" -> 18 bytes: 01 00x7 01 00x7 0a 01
    out = oracle_mod.rl_bwt(b"\n")
    assert out == bytes([1] + [0] * 7 + [1] + [0] * 7 + [0x0A, 1])


def test_golden_tiny_empty_strings(oracle_mod):
    # "A\n\nA\n" -> records 41 01 / 0a 01 / 41 01 / 0a 02
    out = oracle_mod.rl_bwt(b"A\n\nA\n")
    assert len(out) == 24
    assert out[16:] == bytes([0x41, 1, 0x0A, 1, 0x41, 1, 0x0A, 2])


def test_golden_2bytes_alphabet(oracle_mod):
    # size 4,016, md5 0c7d563d..., header (2,2), 1,000 runs
    data = open(os.path.join(GOLD, "test_2bytes_alphabet.txt"), "rb").read()
    out = oracle_mod.rl_bwt(data, 2)
    assert len(out) == 4016
    assert hashlib.md5(out).hexdigest() == "0c7d563d770aefc6feff28123d5e3e73"
    sb, fb, sym, ln = bc.parse_rl_bwt(out)
    assert (sb, fb, len(sym)) == (2, 2, 1000)
    assert out == bc.naive_rl_bwt(data, 2)


def test_golden_byte_alphabet(oracle_mod):
    # size 7,164,992, md5 5e825a2f..., header (1,3), 1,791,244 runs, max run 74
    data = open(REF_BYTE, "rb").read()
    r = oracle_mod.OracleResult(data, 1)
    out = r.rl_bwt
    assert len(out) == 7164992
    assert hashlib.md5(out).hexdigest() == "5e825a2f76a038ad8d27af75d74dadfe"
    sb, fb, sym, ln = bc.parse_rl_bwt(out)
    assert (sb, fb, len(sym), int(ln.max()), int(ln.sum())) == (1, 3, 1791244, 74, 2956004)
    assert bc.runs_are_maximal(sym)
    assert r.n_rounds == 8                      # "8 parsing rounds"
    # LF inversion gives back the collection in input order
    cells = np.frombuffer(data, dtype=np.uint8)
    strings, sep = bc.split_strings(cells)
    rec = bc.lf_invert(sym, ln, sep)
    assert len(rec) == len(strings)
    for a, b in zip(rec, strings):
        assert np.array_equal(a.astype(np.uint8), b)


@pytest.mark.parametrize("mx,w,sb_expect", [
    (251, 1, 1), (252, 1, 2), (255, 1, 2),          # SURVEY.md A.8 verified header edges
    (65531, 2, 2), (65532, 2, 3), (65535, 2, 3),
    (2 ** 32 - 1, 4, 5),
])
def test_header_edges(oracle_mod, mx, w, sb_expect):
    dt = {1: np.uint8, 2: np.uint16, 4: np.uint32}[w]
    # the reference assumes a dense alphabet (allocates sigma-sized arrays), keep the 4-byte case tiny
    cells = np.array([3, mx, 5, 1, mx, 1], dtype=dt)
    if w == 4:
        pytest.skip("oracle follows the reference's dense-alphabet assumption; 2^32 symbols is a 4 GiB table")
    out = oracle_mod.rl_bwt(cells.tobytes(), w)
    sb, fb, sym, ln = bc.parse_rl_bwt(out)
    assert sb == sb_expect and fb == 1
    assert out == bc.naive_rl_bwt(cells.tobytes(), w)


def test_ill_formed(oracle_mod):
    with pytest.raises(oracle_mod.IllFormed):
        oracle_mod.rl_bwt(b"AC\nGT")           # last cell is not the separator
    with pytest.raises(oracle_mod.IllFormed):
        oracle_mod.rl_bwt(b"B\nA\x01B\n")      # separator is not the smallest symbol


def _rand_collection(rng, kind):
    if kind == "dna":
        n_str = int(rng.integers(1, 40))
        parts = []
        for _ in range(n_str):
            L = int(rng.integers(0, 60))
            parts.append(bytes(rng.choice(list(b"ACGT"), size=L).astype(np.uint8)) + b"\n")
        return b"".join(parts), 1
    if kind == "binary":
        n_str = int(rng.integers(1, 12))
        parts = [bytes(rng.choice(list(b"ab"), size=int(rng.integers(0, 80))).astype(np.uint8)) + b"\n"
                 for _ in range(n_str)]
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
    if kind == "u16":
        n_str = int(rng.integers(1, 10))
        cells = []
        for _ in range(n_str):
            L = int(rng.integers(0, 40))
            cells += [int(x) for x in rng.integers(1, 300, size=L)] + [0]
        return np.array(cells, dtype=np.uint16).tobytes(), 2
    if kind == "u32":
        n_str = int(rng.integers(1, 8))
        cells = []
        for _ in range(n_str):
            L = int(rng.integers(0, 30))
            cells += [int(x) for x in rng.integers(8, 2000, size=L)] + [7]
        return np.array(cells, dtype=np.uint32).tobytes(), 4
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["dna", "binary", "repeat", "dups", "u16", "u32"])
def test_oracle_matches_definition_fuzz(oracle_mod, kind):
    rng = np.random.default_rng(zlib.crc32(kind.encode()))
    for _ in range(60):
        data, w = _rand_collection(rng, kind)
        assert oracle_mod.rl_bwt(data, w) == bc.naive_rl_bwt(data, w), (kind, data)


def test_oracle_u64_cells(oracle_mod):
    cells = np.array([9, 4, 9, 2, 5, 2, 2, 9, 9, 4, 2], dtype=np.uint64)
    assert oracle_mod.rl_bwt(cells.tobytes(), 8) == bc.naive_rl_bwt(cells.tobytes(), 8)

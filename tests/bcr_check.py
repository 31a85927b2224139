"""Independent checkers for BCR-BWT outputs (own code, pure python/numpy).

* naive_bcr_runs : textbook definition (SURVEY.md note N2) by sorting all suffixes
                   -- only for small inputs.
* parse_rl_bwt   : decode the .rl_bwt container (include/bwt_io.h:377-382 format).
* lf_invert      : invert a BCR BWT by LF walks (what scripts/reverse_bwt.cpp does
                   with an FM-index) and return the strings in input order.
"""
import numpy as np


def split_strings(cells):
    cells = np.asarray(cells)
    sep = int(cells[-1])
    ends = np.flatnonzero(cells == sep)
    out, st = [], 0
    for e in ends:
        out.append(cells[st:e + 1])
        st = e + 1
    return out, sep


def bitlen(v):
    return int(v).bit_length()


def header_widths(cells, cell_bytes):
    """sb from max_sym+4, fb from max byte frequency (-a 1) or n."""
    cells = np.asarray(cells)
    mx = int(cells.max())
    if cell_bytes == 1:
        F = int(np.bincount(cells.astype(np.int64), minlength=256).max())
    else:
        F = len(cells)
    sb = (bitlen(mx + 4) + 7) // 8
    fb = (bitlen(F) + 7) // 8
    return sb, fb


def naive_bcr_symbols(cells):
    """BCR BWT as a python list of symbols: suffixes of all strings sorted, the
    i-th string's terminator ordered by i; a whole-string suffix is preceded by
    its own terminator."""
    strings, sep = split_strings(cells)
    items = []
    for i, s in enumerate(strings):
        s = [int(x) for x in s]
        body = s[:-1]
        for j in range(len(s)):
            key = tuple((c, 0) for c in body[j:]) + ((-1, i),)
            left = s[j - 1] if j > 0 else sep
            items.append((key, left))
    items.sort(key=lambda t: t[0])
    return [l for _, l in items]


def rle(symbols):
    runs = []
    for c in symbols:
        if runs and runs[-1][0] == c:
            runs[-1][1] += 1
        else:
            runs.append([c, 1])
    return [(a, b) for a, b in runs]


def encode_rl_bwt(runs, sb, fb):
    out = bytearray()
    out += int(sb).to_bytes(8, "little") + int(fb).to_bytes(8, "little")
    for s, l in runs:
        out += int(s).to_bytes(sb, "little") + int(l).to_bytes(fb, "little")
    return bytes(out)


def naive_rl_bwt(data, cell_bytes=1):
    dt = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[cell_bytes]
    cells = np.frombuffer(bytes(data), dtype=dt)
    sb, fb = header_widths(cells, cell_bytes)
    return encode_rl_bwt(rle(naive_bcr_symbols(cells)), sb, fb)


def parse_rl_bwt(blob):
    """-> (sb, fb, sym[uint64], len[uint64])"""
    sb = int.from_bytes(blob[0:8], "little")
    fb = int.from_bytes(blob[8:16], "little")
    body = np.frombuffer(blob, dtype=np.uint8, offset=16)
    rec = sb + fb
    assert len(body) % rec == 0
    body = body.reshape(-1, rec).astype(np.uint64)
    sym = np.zeros(len(body), dtype=np.uint64)
    ln = np.zeros(len(body), dtype=np.uint64)
    for i in range(sb):
        sym |= body[:, i] << np.uint64(8 * i)
    for i in range(fb):
        ln |= body[:, sb + i] << np.uint64(8 * i)
    return sb, fb, sym, ln


def runs_are_maximal(sym):
    return len(sym) < 2 or bool(np.all(sym[1:] != sym[:-1]))


def lf_invert(sym, ln, sep):
    """Invert the BCR BWT given as runs; returns list of numpy arrays (strings with
    terminator) in input order."""
    bwt = np.repeat(np.asarray(sym, dtype=np.uint64), np.asarray(ln, dtype=np.int64))
    n = len(bwt)
    order = np.argsort(bwt, kind="stable")
    lf = np.empty(n, dtype=np.int64)
    lf[order] = np.arange(n, dtype=np.int64)
    k = int(np.count_nonzero(bwt == sep))
    rows = np.arange(k, dtype=np.int64)
    active = np.ones(k, dtype=bool)
    cols = []
    while active.any():
        c = bwt[rows]
        active = active & (c != sep)
        cols.append(np.where(active, c, np.uint64(sep)))
        rows = np.where(active, lf[rows], rows)
    out = []
    if cols:
        mat = np.stack(cols, axis=1)      # k x steps, reversed strings padded with sep
    else:
        mat = np.zeros((k, 0), dtype=np.uint64)
    for i in range(k):
        row = mat[i]
        stop = np.flatnonzero(row == sep)
        m = int(stop[0]) if len(stop) else len(row)
        s = row[:m][::-1]
        out.append(np.concatenate([s, np.array([sep], dtype=np.uint64)]))
    return out

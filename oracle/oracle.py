"""TEST INFRASTRUCTURE -- ctypes front end of oracle/grlbwt_oracle.c.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product (grlbwt_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle.so")
CLI = os.path.join(_HERE, "_build", "oracle_cli")

ORACLE_OK = 0
ORACLE_ERR_ILLFORMED = -84


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    # (make decides what is stale: the library has several sources)
    subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []), stdout=subprocess.DEVNULL)
    return _LIB


REF_DIR = os.path.join(_HERE, "_ref")
REF_PROGS = ("bwt_stats", "grl2plain", "grlbwt2rle", "split_runs", "fastx2plain", "ref_stats")
REFERENCE_ROOT = "/root/reference"


def build_ref(force=False):
    """Compile the reference's own .rl_bwt consumer programs from /root/reference/scripts (oracle/Makefile target
    `ref`) into oracle/_ref/.  Only possible where the reference tree exists (the build container); elsewhere the
    prebuilt binaries that travelled with the repository are used.  Returns the directory or None."""
    have = all(os.path.exists(os.path.join(REF_DIR, p)) for p in REF_PROGS)
    if os.path.isdir(os.path.join(REFERENCE_ROOT, "scripts")) and (force or not have):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
        have = True
    return REF_DIR if have else None


def ref_prog(name):
    """Path of a built reference program (oracle/_ref/<name>) or None."""
    p = os.path.join(REF_DIR, name)
    return p if os.path.exists(p) else None


_lib = None


def _load():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.oracle_run.restype = C.c_void_p
        L.oracle_run.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int]
        L.oracle_status.argtypes = [C.c_void_p]
        L.oracle_out_size.restype = C.c_uint64
        L.oracle_out_size.argtypes = [C.c_void_p]
        L.oracle_out_bytes.restype = C.POINTER(C.c_uint8)
        L.oracle_out_bytes.argtypes = [C.c_void_p]
        L.oracle_n_rounds.argtypes = [C.c_void_p]
        L.oracle_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.oracle_round_counters.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
        for name in ("oracle_level_text", "oracle_level_bwt", "oracle_level_prebwt"):
            f = getattr(L, name)
            f.restype = C.c_uint64
            f.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        L.oracle_level_grammar.restype = C.c_uint64
        L.oracle_level_grammar.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                           C.POINTER(C.c_void_p)]
        L.oracle_free.argtypes = [C.c_void_p]
        L.oracle_fastx2plain.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_uint8, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
                                         C.POINTER(C.c_uint64), C.POINTER(C.c_uint8)]
        _lib = L
    return _lib


def _arr(ptr, n, dtype):
    if n == 0 or not ptr.value:
        return np.zeros(0, dtype=dtype)
    ct = {np.uint64: C.c_uint64, np.uint8: C.c_uint8}[dtype]
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(n,)).copy()


class IllFormed(Exception):
    pass


class OracleResult:
    """Outcome of one oracle run (final .rl_bwt image + optional per-level trace)."""

    def __init__(self, data, cell_bytes=1, trace=False):
        L = _load()
        buf = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8))
        n = len(buf) // cell_bytes
        self._h = L.oracle_run(buf.ctypes.data_as(C.c_void_p), n, cell_bytes, 1 if trace else 0)
        st = L.oracle_status(self._h)
        if st == ORACLE_ERR_ILLFORMED:
            L.oracle_free(self._h)
            self._h = None
            raise IllFormed("Error: the file is ill formed")
        if st != ORACLE_OK:
            L.oracle_free(self._h)
            self._h = None
            raise ValueError("oracle status %d" % st)
        self.trace = trace
        self.n_rounds = L.oracle_n_rounds(self._h)
        sz = L.oracle_out_size(self._h)
        self.rl_bwt = bytes(np.ctypeslib.as_array(L.oracle_out_bytes(self._h), shape=(sz,)))
        st8 = (C.c_uint64 * 8)()
        L.oracle_stats(self._h, st8)
        keys = ["n_strings", "n_syms", "min_sym", "max_sym", "max_sym_freq", "longest", "sb", "fb"]
        self.stats = dict(zip(keys, [int(x) for x in st8]))

    def counters(self, rnd):
        c = (C.c_uint64 * 6)()
        _load().oracle_round_counters(self._h, rnd, c)
        return dict(zip(["n_in", "D", "S", "M", "parse_size", "sigma"], [int(x) for x in c]))

    def _two(self, fn, level, t2=np.uint64):
        a, b = C.c_void_p(), C.c_void_p()
        n = getattr(_load(), fn)(self._h, level, C.byref(a), C.byref(b))
        return _arr(a, n, np.uint64), _arr(b, n, t2)

    def level_text(self, level):
        """(sym[], rep[]) of the text at `level` (level 0 = input cells)."""
        return self._two("oracle_level_text", level, np.uint8)

    def level_bwt(self, level):
        return self._two("oracle_level_bwt", level)

    def level_prebwt(self, level):
        return self._two("oracle_level_prebwt", level)

    def level_grammar(self, level):
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        n = _load().oracle_level_grammar(self._h, level, C.byref(a), C.byref(b), C.byref(c))
        return _arr(a, n, np.uint64), _arr(b, n, np.uint64), _arr(c, n, np.uint8)

    def close(self):
        if self._h:
            _load().oracle_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rl_bwt(data, cell_bytes=1):
    """Bytes of the .rl_bwt file the reference would write for `data`."""
    r = OracleResult(data, cell_bytes)
    out = r.rl_bwt
    r.close()
    return out


class NotDNA(Exception):
    """The reference's "The input seems not to be DNA (invalid symbol:X)", exit(1)."""


def fastx2plain(data, rev_comp=False):
    """(plain text bytes, n_strings) of a decompressed FASTA/FASTQ buffer, as the reference's fastx2plain_format."""
    L = _load()
    data = bytes(data)
    cap = 2 * len(data) + 16
    out = (C.c_uint8 * cap)()
    n_out, n_str, bad = C.c_uint64(0), C.c_uint64(0), C.c_uint8(0)
    rc = L.oracle_fastx2plain(data, len(data), 1 if rev_comp else 0, 10, out, cap, C.byref(n_out), C.byref(n_str), C.byref(bad))
    if rc == 1:
        raise NotDNA(chr(bad.value))
    if rc != 0:
        raise RuntimeError("oracle_fastx2plain failed: %d" % rc)
    return bytes(out[:n_out.value]), n_str.value

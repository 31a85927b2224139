"""Host-side mirror of the reference's driver interface over the C-ABI library.

Mirrors include/grl_bwt.hpp:23-79 (grl_bwt_algo = par_phase + ind_phase + write
.rl_bwt) with the same stage names.  Everything runs through
libgrlbwt_hip.so (include/grlbwt_hip.h); there is no CPU path: if the library
is missing or no HIP device is usable this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "csrc", "libgrlbwt_hip.so")

FLAG_KEEP_LEVELS = 1
FLAG_SYNC_DEBUG = 2
FLAG_FORCE_IDX64 = 4
FLAG_CLASSIC_POOL = 8

OK = 0
EILLFORMED = -84
ENOTDNA = -86

# every symbol include/grlbwt_hip.h declares
ABI_SYMBOLS = [
    "grlbwt_abi_version", "grlbwt_backend_name", "grlbwt_strerror", "grlbwt_last_error", "grlbwt_ctx_create", "grlbwt_ctx_destroy",
    "grlbwt_ctx_set_stream", "grlbwt_text_upload", "grlbwt_text_load_file", "grlbwt_text_load_file_range", "grlbwt_rccl_unique_id", "grlbwt_rccl_comm_create",
    "grlbwt_rccl_comm_destroy", "grlbwt_fastx_probe", "grlbwt_text_load_fastx", "grlbwt_fastx_convert_device", "grlbwt_text_attach_device", "grlbwt_get_stats",
    "grlbwt_parse_round", "grlbwt_parse_phase", "grlbwt_round_info_get", "grlbwt_induce_first",
    "grlbwt_induce_level", "grlbwt_induce_phase", "grlbwt_level_info_get", "grlbwt_build", "grlbwt_result_size",
    "grlbwt_result_device_ptr", "grlbwt_result_download", "grlbwt_result_write_file", "grlbwt_result_part", "grlbwt_result_write_part", "grlbwt_level_text_size",
    "grlbwt_level_text_download", "grlbwt_level_bwt_size", "grlbwt_level_bwt_download", "grlbwt_get_counters",
    "grlbwt_selftest", "grlbwt_profile_enable", "grlbwt_profile_dump", "grlbwt_dist_build", "grlbwt_memory_usage", "grlbwt_invert_image", "grlbwt_invert_image_tails",
    "grlbwt_image_plain", "grlbwt_image_rle", "grlbwt_image_stats_get", "grlbwt_image_split_runs",
    "grlbwt_level_grammar_size", "grlbwt_level_grammar_download",
]


class ImageStats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("n_runs", "sigma", "text_size", "min_run", "max_run", "fit1", "fit2", "fit3")] + \
               [("runs_of", C.c_uint64 * 256), ("freq_of", C.c_uint64 * 256), ("deciles", C.c_uint64 * 9), ("non_maximal", C.c_uint64)]


class SplitInfo(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("runs_before", "runs_after", "overflow_splits", "block_splits", "n_syms", "n_blocks", "out_bytes")]


class Stats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("n_strings", "n_syms", "min_sym", "max_sym", "max_sym_freq", "sb", "fb")]


class RoundInfo(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("n_in", "n_phrases", "dict_syms", "n_metasyms", "parse_size", "sigma",
                                          "max_phrase_len", "sort_iters")]


class LevelInfo(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("n", "n_runs", "runs_next", "induced_cells", "prebwt_runs", "segments",
                                          "atoms", "chain_steps", "merged_cells")]


class Counters(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("t_stats", "t_classify", "t_hash", "t_dict_sort", "t_dict_groups", "t_emit",
                                          "t_ind_expand", "t_ind_split", "t_ind_assemble", "t_finish")] + \
               [(k, C.c_uint64) for k in ("bytes_classify_hash", "bytes_emit", "bytes_induce_scatter",
                                          "bytes_induce_assemble", "idx_bytes")]


def _as_dict(s):
    return {k: getattr(s, k) for k, _ in s._fields_}


class GrlbwtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("grlbwt error %d: %s" % (code, msg))
        self.code = code


class NotDNA(GrlbwtError):
    """The reference's "The input seems not to be DNA (invalid symbol:X)", exit(1) (fastx_handler.cpp:30-33)."""


FASTX_REVCOMP = 1


def fastx_probe(path, lib=None):
    """(is_fastx, is_gz) as the reference's is_fastx / check_gzip decide them (host only: no device is touched)."""
    L = load_library(lib)
    a, b = C.c_int(0), C.c_int(0)
    rc = L.grlbwt_fastx_probe(os.fsencode(path), C.byref(a), C.byref(b))
    if rc != 0:
        raise GrlbwtError(rc, "cannot probe " + str(path))
    return bool(a.value), bool(b.value)


class IllFormedInput(GrlbwtError):
    """The reference prints "Error: the file is ill formed" and exits 1 (utils.cpp:177-180)."""


_libs = {}
_standin_paths = set()


def _test_allow_standin(path):
    """TESTS ONLY: accept the serial stand-in library at exactly this path (tests/hostsim).  There is deliberately no
    environment switch: nothing outside a test's own code can point the product at a CPU path."""
    _standin_paths.add(os.path.abspath(path))


def load_library(path=None, allow_test_standin=False):
    """dlopen the C-ABI library; fails loudly when it has not been built.  Only the HIP build is accepted:
    the serial stand-in of tests/hostsim identifies itself and is refused unless a test explicitly allows it."""
    path = path or os.environ.get("GRLBWT_HIP_LIB", DEFAULT_LIB)
    allow_test_standin = allow_test_standin or os.path.abspath(path) in _standin_paths
    if path in _libs:
        L = _libs[path]
        if L.grlbwt_backend_name() != b"hip-gfx950" and not allow_test_standin:
            raise RuntimeError("%s is not the HIP library (backend %r); the product has no CPU path" % (path, L.grlbwt_backend_name()))
        return L
    if not os.path.exists(path):
        raise RuntimeError("HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)" % path)
    if not allow_test_standin:
        # PyTorch ships its own copy of the HIP runtime.  If torch is imported AFTER this library has pulled in
        # /opt/rocm's copy, the process ends up with two runtimes and torch reports "No HIP GPUs are available";
        # importing torch first makes both sides bind to one.  (Processes that never use torch are unaffected.)
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(path)
    vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int
    L.grlbwt_abi_version.restype = i32
    L.grlbwt_backend_name.restype = C.c_char_p
    if L.grlbwt_backend_name() != b"hip-gfx950" and not allow_test_standin:
        raise RuntimeError("%s is not the HIP library (backend %r); the product has no CPU path" % (path, L.grlbwt_backend_name()))
    L.grlbwt_strerror.restype = C.c_char_p
    L.grlbwt_strerror.argtypes = [i32]
    L.grlbwt_last_error.restype = C.c_char_p
    L.grlbwt_last_error.argtypes = [vp]
    L.grlbwt_ctx_create.argtypes = [i32, C.c_uint32, C.POINTER(vp)]
    L.grlbwt_ctx_destroy.argtypes = [vp]
    L.grlbwt_ctx_destroy.restype = None
    L.grlbwt_ctx_set_stream.argtypes = [vp, vp]
    L.grlbwt_text_upload.argtypes = [vp, vp, u64, i32]
    L.grlbwt_text_load_file.argtypes = [vp, C.c_char_p, i32]
    L.grlbwt_text_attach_device.argtypes = [vp, vp, u64, i32]
    L.grlbwt_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.grlbwt_parse_round.argtypes = [vp, C.POINTER(RoundInfo), C.POINTER(i32)]
    L.grlbwt_parse_phase.argtypes = [vp, C.POINTER(i32)]
    L.grlbwt_round_info_get.argtypes = [vp, i32, C.POINTER(RoundInfo)]
    L.grlbwt_induce_first.argtypes = [vp]
    L.grlbwt_induce_level.argtypes = [vp, C.POINTER(i32), C.POINTER(LevelInfo)]
    L.grlbwt_induce_phase.argtypes = [vp]
    L.grlbwt_level_info_get.argtypes = [vp, i32, C.POINTER(LevelInfo)]
    L.grlbwt_build.argtypes = [vp]
    L.grlbwt_result_size.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    L.grlbwt_result_device_ptr.argtypes = [vp, C.POINTER(vp)]
    L.grlbwt_result_download.argtypes = [vp, vp, u64]
    L.grlbwt_result_write_file.argtypes = [vp, C.c_char_p]
    L.grlbwt_result_part.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    L.grlbwt_result_write_part.argtypes = [vp, C.c_char_p]
    L.grlbwt_level_text_size.argtypes = [vp, i32, C.POINTER(u64)]
    L.grlbwt_level_text_download.argtypes = [vp, i32, vp]
    L.grlbwt_level_bwt_size.argtypes = [vp, i32, C.POINTER(u64)]
    L.grlbwt_level_bwt_download.argtypes = [vp, i32, vp, vp]
    L.grlbwt_level_grammar_size.argtypes = [vp, i32, C.POINTER(u64), C.POINTER(u64)]
    L.grlbwt_level_grammar_download.argtypes = [vp, i32, vp, vp, vp, vp, vp]
    L.grlbwt_get_counters.argtypes = [vp, C.POINTER(Counters)]
    L.grlbwt_selftest.argtypes = [vp, u64, u64]
    L.grlbwt_memory_usage.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    L.grlbwt_invert_image.argtypes = [vp, vp, u64, i32, vp, u64, C.POINTER(u64)]
    L.grlbwt_invert_image_tails.argtypes = [vp, vp, u64, i32, u64, vp, u64, C.POINTER(u64), C.POINTER(u64)]
    L.grlbwt_image_plain.argtypes = [vp, vp, u64, vp, u64, i32, C.POINTER(u64)]
    L.grlbwt_image_rle.argtypes = [vp, vp, u64, vp, vp, u64, C.POINTER(u64)]
    L.grlbwt_image_stats_get.argtypes = [vp, vp, u64, C.POINTER(ImageStats)]
    L.grlbwt_image_split_runs.argtypes = [vp, vp, u64, i32, u64, vp, u64, C.POINTER(SplitInfo)]
    L.grlbwt_profile_enable.argtypes = [vp, i32]
    L.grlbwt_profile_dump.argtypes = [vp, C.c_char_p, u64]
    _libs[path] = L
    return L


class Context:
    """One engine context per GPU (grlbwt_ctx)."""

    def __init__(self, device=0, flags=0, lib=None, _test_standin=False):
        self.L = load_library(lib, allow_test_standin=_test_standin)
        h = C.c_void_p()
        rc = self.L.grlbwt_ctx_create(device, flags, C.byref(h))
        if rc != OK:
            raise GrlbwtError(rc, self.L.grlbwt_strerror(rc).decode())
        self._h = h
        self._keep = None

    def _ck(self, rc):
        if rc != OK:
            msg = self.L.grlbwt_last_error(self._h).decode() or self.L.grlbwt_strerror(rc).decode()
            if rc == EILLFORMED:
                raise IllFormedInput(rc, msg)
            if rc == ENOTDNA:
                raise NotDNA(rc, msg)
            raise GrlbwtError(rc, msg)

    def close(self):
        if getattr(self, "_h", None):
            self.L.grlbwt_ctx_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- input ----------------------------------------------------------
    def set_stream(self, hip_stream):
        self._ck(self.L.grlbwt_ctx_set_stream(self._h, C.c_void_p(hip_stream)))

    def upload(self, data, cell_bytes=1):
        """collection_stats + first-round input from a host buffer (bytes / numpy)."""
        import numpy as np
        buf = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if isinstance(data, (bytes, bytearray, memoryview))
                                   else np.asarray(data).view(np.uint8).reshape(-1))
        self._keep = buf
        self._ck(self.L.grlbwt_text_upload(self._h, buf.ctypes.data_as(C.c_void_p), len(buf) // cell_bytes, cell_bytes))

    def load_file(self, path, cell_bytes=1):
        """collection_stats + first-round input straight from a file of raw cells (pinned, overlapped upload)."""
        self._keep = None
        self._ck(self.L.grlbwt_text_load_file(self._h, os.fsencode(path), cell_bytes))

    def load_fastx(self, path, rev_comp=False):
        """FASTA/FASTQ (optionally gzip) -> the one-string-per-line text in HBM (fastx2plain_format of the reference);
        returns the number of strings."""
        self._keep = None
        n = C.c_uint64(0)
        self._ck(self.L.grlbwt_text_load_fastx(self._h, os.fsencode(path), FASTX_REVCOMP if rev_comp else 0, C.byref(n)))
        return n.value

    def fastx_convert(self, src_ptr, n_in, rev_comp, dst_ptr, capacity):
        """The conversion alone, device buffer to device buffer: (bytes written, n_strings)."""
        n_out, n_str = C.c_uint64(0), C.c_uint64(0)
        self._ck(self.L.grlbwt_fastx_convert_device(self._h, C.c_void_p(src_ptr), n_in, FASTX_REVCOMP if rev_comp else 0, C.c_void_p(dst_ptr),
                                                    capacity, C.byref(n_out), C.byref(n_str)))
        return n_out.value, n_str.value

    def attach_device(self, dev_ptr, n_cells, cell_bytes=1, keepalive=None):
        """Use cells already resident in HBM (e.g. a torch tensor's data_ptr())."""
        self._keep = keepalive
        self._ck(self.L.grlbwt_text_attach_device(self._h, C.c_void_p(dev_ptr), n_cells, cell_bytes))

    def stats(self):
        s = Stats()
        self._ck(self.L.grlbwt_get_stats(self._h, C.byref(s)))
        return _as_dict(s)

    # ---- phases (names follow the reference) -----------------------------
    def parse_round(self):
        info, done = RoundInfo(), C.c_int()
        self._ck(self.L.grlbwt_parse_round(self._h, C.byref(info), C.byref(done)))
        return _as_dict(info), bool(done.value)

    def par_phase(self):
        n = C.c_int()
        self._ck(self.L.grlbwt_parse_phase(self._h, C.byref(n)))
        return n.value

    def round_info(self, r):
        info = RoundInfo()
        self._ck(self.L.grlbwt_round_info_get(self._h, r, C.byref(info)))
        return _as_dict(info)

    def parse2bwt(self):
        self._ck(self.L.grlbwt_induce_first(self._h))

    def infer_lvl_bwt(self):
        lvl, info = C.c_int(), LevelInfo()
        self._ck(self.L.grlbwt_induce_level(self._h, C.byref(lvl), C.byref(info)))
        return lvl.value, _as_dict(info)

    def ind_phase(self):
        self._ck(self.L.grlbwt_induce_phase(self._h))

    def level_info(self, lvl):
        info = LevelInfo()
        self._ck(self.L.grlbwt_level_info_get(self._h, lvl, C.byref(info)))
        return _as_dict(info)

    def build(self):
        """grl_bwt_algo: par_phase + ind_phase on the loaded text."""
        self._ck(self.L.grlbwt_build(self._h))

    # ---- output -----------------------------------------------------------
    def result_size(self):
        nb, nr = C.c_uint64(), C.c_uint64()
        self._ck(self.L.grlbwt_result_size(self._h, C.byref(nb), C.byref(nr)))
        return nb.value, nr.value

    def result_device_ptr(self):
        p = C.c_void_p()
        self._ck(self.L.grlbwt_result_device_ptr(self._h, C.byref(p)))
        return p.value

    def result_part(self):
        """(offset, bytes) of what this context holds of the image: all of it, or this rank's part (collection-level build
        with dist.COMM_KEEP_PARTS)."""
        off, nb = C.c_uint64(), C.c_uint64()
        self._ck(self.L.grlbwt_result_part(self._h, C.byref(off), C.byref(nb)))
        return off.value, nb.value

    def result_bytes(self):
        """The image bytes this context holds (result_part)."""
        _, nb = self.result_part()
        out = (C.c_uint8 * max(nb, 1))()
        self._ck(self.L.grlbwt_result_download(self._h, out, nb))
        return bytes(out)[:nb]

    def write_file(self, path):
        self._ck(self.L.grlbwt_result_write_file(self._h, os.fsencode(path)))

    def write_part(self, path):
        """This context's part at its offset of `path` (created if missing, never truncated)."""
        self._ck(self.L.grlbwt_result_write_part(self._h, os.fsencode(path)))

    # ---- inspection --------------------------------------------------------
    def level_text(self, lvl):
        import numpy as np
        n = C.c_uint64()
        self._ck(self.L.grlbwt_level_text_size(self._h, lvl, C.byref(n)))
        out = np.zeros(n.value, dtype=np.uint64)
        self._ck(self.L.grlbwt_level_text_download(self._h, lvl, out.ctypes.data_as(C.c_void_p)))
        return out

    def level_bwt(self, lvl):
        import numpy as np
        n = C.c_uint64()
        self._ck(self.L.grlbwt_level_bwt_size(self._h, lvl, C.byref(n)))
        s = np.zeros(n.value, dtype=np.uint64)
        l = np.zeros(n.value, dtype=np.uint64)
        self._ck(self.L.grlbwt_level_bwt_download(self._h, lvl, s.ctypes.data_as(C.c_void_p), l.ctypes.data_as(C.c_void_p)))
        return s, l

    def level_grammar(self, lvl):
        """(g0, g1, has_hocc, prebwt_sym, prebwt_len) of a level that has been parsed and not yet induced."""
        import numpy as np
        m, p = C.c_uint64(), C.c_uint64()
        self._ck(self.L.grlbwt_level_grammar_size(self._h, lvl, C.byref(m), C.byref(p)))
        g0 = np.zeros(m.value, dtype=np.uint64); g1 = np.zeros(m.value, dtype=np.uint64); hh = np.zeros(m.value, dtype=np.uint8)
        ps = np.zeros(p.value, dtype=np.uint64); pl = np.zeros(p.value, dtype=np.uint64)
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)
        self._ck(self.L.grlbwt_level_grammar_download(self._h, lvl, ptr(g0), ptr(g1), ptr(hh), ptr(ps), ptr(pl)))
        return g0, g1, hh, ps, pl

    def counters(self):
        c = Counters()
        self._ck(self.L.grlbwt_get_counters(self._h, C.byref(c)))
        return _as_dict(c)

    def invert_image(self, dev_image_ptr, image_bytes, cell_bytes, dev_out_ptr, capacity_cells):
        """reverse_bwt on the device: .rl_bwt image (device) -> the collection's cells (device); returns #cells."""
        n = C.c_uint64()
        self._ck(self.L.grlbwt_invert_image(self._h, C.c_void_p(dev_image_ptr), image_bytes, cell_bytes,
                                            C.c_void_p(dev_out_ptr), capacity_cells, C.byref(n)))
        return n.value

    def invert_image_tails(self, dev_image_ptr, image_bytes, cell_bytes, tail_cells, dev_out_ptr, capacity_cells):
        """The last `tail_cells` cells of every string (slot i of the output: string i's end, right-aligned); returns
        (strings, cells written)."""
        k, n = C.c_uint64(), C.c_uint64()
        self._ck(self.L.grlbwt_invert_image_tails(self._h, C.c_void_p(dev_image_ptr), image_bytes, cell_bytes, tail_cells,
                                                  C.c_void_p(dev_out_ptr), capacity_cells, C.byref(k), C.byref(n)))
        return k.value, n.value

    def image_plain(self, dev_image_ptr, image_bytes, dev_out_ptr, capacity, null_char=-1):
        """grl2plain on the device (scripts/grl2plain.cpp): plain BWT bytes; returns their number."""
        n = C.c_uint64()
        self._ck(self.L.grlbwt_image_plain(self._h, C.c_void_p(dev_image_ptr), image_bytes, C.c_void_p(dev_out_ptr), capacity,
                                           null_char, C.byref(n)))
        return n.value

    def image_rle(self, dev_image_ptr, image_bytes, dev_syms_ptr, dev_lens_ptr, capacity_runs):
        """grlbwt2rle on the device (scripts/grlbwt2rle.cpp): uint8 symbols + uint32 lengths; returns #runs."""
        n = C.c_uint64()
        self._ck(self.L.grlbwt_image_rle(self._h, C.c_void_p(dev_image_ptr), image_bytes, C.c_void_p(dev_syms_ptr),
                                         C.c_void_p(dev_lens_ptr), capacity_runs, C.byref(n)))
        return n.value

    def image_stats(self, dev_image_ptr, image_bytes):
        """bwt_stats on the device (scripts/bwt_stats.cpp)."""
        st = ImageStats()
        self._ck(self.L.grlbwt_image_stats_get(self._h, C.c_void_p(dev_image_ptr), image_bytes, C.byref(st)))
        d = {k: int(getattr(st, k)) for k in ("n_runs", "sigma", "text_size", "min_run", "max_run", "fit1", "fit2", "fit3")}
        d["runs_of"] = list(st.runs_of); d["freq_of"] = list(st.freq_of); d["deciles"] = list(st.deciles)
        d["non_maximal"] = int(st.non_maximal)
        return d

    def image_split_runs(self, dev_image_ptr, image_bytes, bits, block_size, dev_out_ptr, capacity_bytes):
        """split_runs on the device (scripts/split_runs.cpp): the re-encoded image lands in dev_out; returns the counters."""
        si = SplitInfo()
        self._ck(self.L.grlbwt_image_split_runs(self._h, C.c_void_p(dev_image_ptr), image_bytes, bits, block_size,
                                                C.c_void_p(dev_out_ptr), capacity_bytes, C.byref(si)))
        return _as_dict(si)

    def memory_usage(self):
        a, b = C.c_uint64(), C.c_uint64()
        self._ck(self.L.grlbwt_memory_usage(self._h, C.byref(a), C.byref(b)))
        return {"peak_live_bytes": a.value, "reserved_bytes": b.value}

    def profile_enable(self, on=True):
        self._ck(self.L.grlbwt_profile_enable(self._h, 1 if on else 0))

    def profile(self):
        """{kernel name: (launches, total_ms, stated algorithmic bytes)} measured with HIP events on the engine's stream."""
        buf = C.create_string_buffer(1 << 17)
        self._ck(self.L.grlbwt_profile_dump(self._h, buf, len(buf)))
        out = {}
        for line in buf.value.decode().splitlines():
            name, cnt, ms, nbytes = line.rsplit(" ", 3)
            out[name] = (int(cnt), float(ms), int(nbytes))
        return out

    def selftest(self, n=100000, seed=1):
        return self.L.grlbwt_selftest(self._h, n, seed)


def grl_bwt_algo(data, cell_bytes=1, device=0, flags=0, lib=None):
    """Bytes of the .rl_bwt the reference would write for `data` (grl_bwt.hpp:23-79)."""
    with Context(device, flags, lib) as ctx:
        ctx.upload(data, cell_bytes)
        ctx.build()
        return ctx.result_bytes()

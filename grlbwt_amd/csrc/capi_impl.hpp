// capi_impl.hpp -- extern "C" layer of include/grlbwt_hip.h over the two index-width
// instantiations of the engine (grl32 / grl64).  Included by engine_hip.hip.
#include "../../include/grlbwt_hip.h"

#include <fcntl.h>
#include <zlib.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>


struct grlbwt_ctx {
    uint32_t flags = 0;
    int device = 0;
    std::unique_ptr<grl32::Engine> e32;
    std::unique_ptr<grl64::Engine> e64;
    std::string err;
};

namespace {

template <class Fn>
int guarded(grlbwt_ctx *ctx, Fn fn) {
    try {
        fn();
        return GRLBWT_OK;
    } catch (const prim::Error &e) {
        if (ctx) ctx->err = e.what();
        return e.code;
    } catch (const std::bad_alloc &) {
        if (ctx) ctx->err = "host allocation failed";
        return GRLBWT_ENOMEM;
    } catch (const std::exception &e) {
        if (ctx) ctx->err = e.what();
        return GRLBWT_EINTERNAL;
    }
}

template <class E>
void fill_round(const E &e, int r, grlbwt_round_info *o) {
    const auto &I = e.levels[r].info;
    o->n_in = I.n_in; o->n_phrases = I.D; o->dict_syms = I.S; o->n_metasyms = I.M; o->parse_size = I.parse_size;
    o->sigma = I.sigma; o->max_phrase_len = I.max_phrase_len; o->sort_iters = I.sort_iters;
}
template <class E>
void fill_stats(const E &e, grlbwt_stats *out) {
    const auto &s = e.stats;
    out->n_strings = s.n_strings; out->n_syms = s.n_syms; out->min_sym = s.min_sym; out->max_sym = s.max_sym;
    out->max_sym_freq = s.max_sym_freq; out->sb = s.sb; out->fb = s.fb;
}
template <class E>
void fill_level(const E &e, int l, grlbwt_level_info *o) {
    const auto &I = e.linfo[l];
    o->n = I.n; o->n_runs = I.R; o->runs_next = I.R_next; o->induced_cells = I.E; o->prebwt_runs = I.P;
    o->segments = I.G; o->atoms = I.A; o->chain_steps = I.Esteps; o->merged_cells = I.Emerged;
}
template <class E>
void fill_counters(const E &e, grlbwt_counters *o) {
    memset(o, 0, sizeof(*o));
    const auto &t = e.tm;
    o->t_stats = t.stats; o->t_classify = t.classify; o->t_hash = t.hash; o->t_dict_sort = t.dict_sort;
    o->t_dict_groups = t.dict_groups; o->t_emit = t.emit; o->t_ind_expand = t.ind_expand; o->t_ind_split = t.ind_sort;
    o->t_ind_assemble = t.ind_assemble; o->t_finish = t.finish;
    const uint64_t ib = sizeof(typename std::remove_reference<decltype(e.bwt.len.p[0])>::type);
    o->idx_bytes = ib;
    for (size_t r = 0; r < e.levels.size(); r++) {
        const auto &I = e.levels[r].info;
        o->bytes_classify_hash += I.n_in * (r == 0 ? (uint64_t)e.cell_bytes : 4ull);
        o->bytes_emit += I.parse_size * 4ull;
    }
    for (size_t l = 0; l + 1 < e.linfo.size(); l++) {
        const auto &I = e.linfo[l];
        o->bytes_induce_scatter += I.R_next * (4 + ib) + I.R_next * 4 + I.E * (4 + ib);
        o->bytes_induce_assemble += (I.P + I.E + I.R_next + I.R) * (4 + ib);
    }
}

template <class E>
void text_download(const E &e, int level, uint64_t *out) {
    const auto &b = e.kept_texts[level - 1];
    std::vector<uint32_t> h = b.to_host(b.n);
    for (uint64_t i = 0; i < b.n; i++) out[i] = (uint64_t)(h[i] >> 1);   // (rank<<2|rep<<1|T) -> (rank<<1|rep)
}
template <class E>
void bwt_download(const E &e, int level, uint64_t *sym, uint64_t *len) {
    const auto &b = e.kept_bwts[level];
    auto hs = b.sym.to_host(b.R);
    auto hl = b.len.to_host(b.R);
    for (uint64_t i = 0; i < b.R; i++) { sym[i] = hs[i]; len[i] = hl[i]; }
}

template <class E>
void grammar_download(const E &e, int level, uint64_t *g0, uint64_t *g1, uint8_t *hh, uint64_t *ps, uint64_t *pl) {
    const auto &L = e.levels[level];
    const uint64_t M = L.M, P = L.prebwt.R;
    auto a = L.g0.to_host(M), b = L.g1.to_host(M);
    auto h = L.has_hocc.to_host(M);
    for (uint64_t i = 0; i < M; i++) { if (g0) g0[i] = a[i]; if (g1) g1[i] = b[i]; if (hh) hh[i] = h[i]; }
    auto s = L.prebwt.sym.to_host(P);
    auto l = L.prebwt.len.to_host(P);
    for (uint64_t i = 0; i < P; i++) { if (ps) ps[i] = s[i]; if (pl) pl[i] = l[i]; }
}
#define ENG(ctx, expr) ((ctx)->e32 ? (ctx)->e32->expr : (ctx)->e64->expr)
#define HAS_ENG(ctx) ((ctx) && ((ctx)->e32 || (ctx)->e64))

void load(grlbwt_ctx *ctx, const void *cells, uint64_t n, int w, bool host) {
    ctx->e32.reset();
    ctx->e64.reset();
    bool big = (n >= 0xFFFFFF00ull) || (ctx->flags & GRLBWT_FLAG_FORCE_IDX64);
    bool keep = ctx->flags & GRLBWT_FLAG_KEEP_LEVELS;
    if (big) {
        std::unique_ptr<grl64::Engine> e(new grl64::Engine());
        e->keep_texts = keep;
        if (host) e->upload_text(cells, n, w); else e->load_text(cells, n, w);
        ctx->e64 = std::move(e);          // only a successfully loaded text leaves an engine behind
    } else {
        std::unique_ptr<grl32::Engine> e(new grl32::Engine());
        e->keep_texts = keep;
        if (host) e->upload_text(cells, n, w); else e->load_text(cells, n, w);
        ctx->e32 = std::move(e);
    }
}

// ---- file in / file out: pinned staging buffers, reader/writer threads, copies overlapped with the I/O ---------
// pread/pwrite of [off, off+len) split over `nthr` threads (page-cache copies are memory-bound per thread)
bool par_io(int fd, char *buf, uint64_t off, uint64_t len, bool write, int nthr) {
    auto one = [&](uint64_t a, uint64_t b, bool *ok) {
        while (a < b) {
            ssize_t r = write ? pwrite(fd, buf + (a - off), b - a, (off_t)a) : pread(fd, buf + (a - off), b - a, (off_t)a);
            if (r <= 0) { *ok = false; return; }
            a += (uint64_t)r;
        }
    };
    if (nthr < 1) nthr = 1;
    if (len < ((uint64_t)4 << 20)) nthr = 1;
    if (nthr == 1) { bool ok1 = true; one(off, off + len, &ok1); return ok1; }      // (no thread of its own)
    std::vector<std::thread> th;
    std::vector<char> oks(nthr, 1);
    const uint64_t part = (len + nthr - 1) / nthr;
    for (int t = 0; t < nthr; t++) {
        uint64_t a = off + (uint64_t)t * part, b = a + part < off + len ? a + part : off + len;
        if (a >= b) break;
        th.emplace_back(one, a, b, (bool *)&oks[t]);
    }
    for (auto &x : th) x.join();
    for (char c : oks) if (!c) return false;
    return true;
}
constexpr uint64_t kIoChunk = (uint64_t)64 << 20;
constexpr int kIoBufs = 3;
// reader / writer threads per chunk: page-cache copies run at about 2 GB/s per thread (GRLBWT_IO_THREADS overrides)
int io_threads() {
    static const int n = [] {
        if (const char *e = getenv("GRLBWT_IO_THREADS")) { int v = atoi(e); if (v >= 1 && v <= 64) return v; }
        const unsigned hc = std::thread::hardware_concurrency();
        return (int)std::min<unsigned>(std::max<unsigned>(hc / 2, 4u), 16u);      // (pread from the page cache: 99 GB/s with 8 threads, 118 with 16, 97 with 32 on a 256-core host)
    }();
    return n;
}

// file -> HBM: chunk k+1 is read from the file while chunk k travels over PCIe; for byte cells the histogram of
// collection_stats is taken from every chunk on the device as soon as it has landed (no second pass over the text)
// Reader threads that live for one load: a chunk is cut into 4 MiB pieces which the threads take one by one (pread from the
// page cache: 14 GB/s with one thread, 100-118 GB/s with 8-16 on the GPU box, tools/io_probe.cpp).  (Eight new threads per
// 64 MiB chunk, joined before the next chunk, read the 10 GB input at 43 GB/s.)
struct ReadPool {
    static constexpr uint64_t kPiece = (uint64_t)4 << 20;
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    int fd = -1; char *buf = nullptr; uint64_t off = 0, len = 0;       // the chunk being read
    uint64_t next = 0, done = 0, pieces = 0, generation = 0;
    bool stop = false, ok = true;
    explicit ReadPool(int n) {
        try { for (int i = 0; i < n; i++) th.emplace_back([this] { run(); }); } catch (const std::system_error &) {}      // (as many as there are; none: read() reads in line)
    }
    ~ReadPool() {
        { std::lock_guard<std::mutex> g(mu); stop = true; }
        cv_work.notify_all();
        for (auto &t : th) t.join();
    }
    void run() {
        uint64_t seen = 0;
        for (;;) {
            uint64_t p, a, b; int f; char *dst; uint64_t base;
            {
                std::unique_lock<std::mutex> g(mu);
                cv_work.wait(g, [&] { return stop || (generation != seen && next < pieces) ; });
                if (stop) return;
                if (next >= pieces) { seen = generation; continue; }
                p = next++;
                f = fd; dst = buf; base = off;
                a = p * kPiece; b = a + kPiece < len ? a + kPiece : len;
            }
            bool good = true;
            for (uint64_t x = a; x < b && good;) {
                const ssize_t r = pread(f, dst + x, b - x, (off_t)(base + x));
                if (r <= 0) good = false; else x += (uint64_t)r;
            }
            {
                std::lock_guard<std::mutex> g(mu);
                if (!good) ok = false;
                if (++done == pieces) cv_done.notify_all();
            }
        }
    }
    // bytes [off_, off_ + len_) of the file into buf_; returns when they are all there
    bool read(int fd_, char *buf_, uint64_t off_, uint64_t len_) {
        if (th.empty()) return par_io(fd_, buf_, off_, len_, false, 1);
        std::unique_lock<std::mutex> g(mu);
        fd = fd_; buf = buf_; off = off_; len = len_;
        pieces = (len_ + kPiece - 1) / kPiece; next = 0; done = 0; generation++;
        if (pieces == 0) return ok;
        cv_work.notify_all();
        cv_done.wait(g, [&] { return done == pieces; });
        return ok;
    }
};
// GRLBWT_IO_TRACE=1: where the loader and the image writer spend their time (stderr)
inline bool io_trace() { static const bool on = getenv("GRLBWT_IO_TRACE") != nullptr; return on; }
inline double io_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <class E>
void load_file_into(E &e, int fd, uint64_t base, uint64_t bytes, int w) {      // bytes [base, base + bytes) of the file
    const double t_a = io_now();
    e.own0.alloc(bytes + 16);
    const double t_b = io_now();
    double t_read = 0, t_wait = 0;
    char *bufs[kIoBufs] = {nullptr, nullptr, nullptr};
    prim::Fence fences[kIoBufs];
    uint64_t *d_hist = nullptr;
    auto cleanup = [&] {
        for (int k = 0; k < kIoBufs; k++) { prim::fence_destroy(fences[k]); prim::pinned_free(bufs[k]); bufs[k] = nullptr; }
        if (d_hist) prim::dev_free(d_hist);
        d_hist = nullptr;
    };
    try {
        const uint64_t chunk = bytes < kIoChunk ? (bytes + 15) / 16 * 16 : kIoChunk;
        // (the pinned buffers are made as the ring reaches them: pinning 3 x 64 MiB takes 60 ms, and the second and third are not
        // needed before the first chunk is on its way)
        if (w == 1) { d_hist = (uint64_t *)prim::dev_alloc(256 * 8); prim::dev_memset(d_hist, 0, 256 * 8); }
        int k = 0;
        ReadPool pool(bytes > ReadPool::kPiece ? io_threads() : 0);
        for (uint64_t off = 0; off < bytes; off += chunk, k = (k + 1) % kIoBufs) {
            const uint64_t len = bytes - off < chunk ? bytes - off : chunk;
            const double t0 = io_now();
            if (!bufs[k]) bufs[k] = (char *)prim::pinned_alloc(chunk ? chunk : 16);
            prim::fence_wait(fences[k]);                                // the copy that last used this buffer is done
            const double t1 = io_now();
            if (!pool.read(fd, bufs[k], base + off, len)) throw prim::Error(GRLBWT_EINVAL, "cannot read the input file");
            t_wait += t1 - t0; t_read += io_now() - t1;
            prim::h2d_async(e.own0.p + off, bufs[k], len);
            if (d_hist) prim::byte_histogram_accumulate(e.own0.p + off, len, d_hist);
            prim::fence_record(fences[k]);
        }
        uint64_t hist[256];
        const double t_c = io_now();
        if (d_hist) prim::d2h(hist, d_hist, sizeof hist); else prim::sync();
        const double t_d = io_now();
        cleanup();
        if (io_trace()) fprintf(stderr, "[grlbwt] load: device buffer %.3f s, pinned buffers + loop %.3f s (reading %.3f s, waiting for copies %.3f s), drain %.3f s, "
                                        "free %.3f s, %d reader threads\n", t_b - t_a, t_c - t_b, t_read, t_wait, t_d - t_c, io_now() - t_d, io_threads());
        e.load_text(e.own0.p, bytes / (uint64_t)w, w, w == 1 ? hist : nullptr);
    } catch (...) {
        try { prim::sync(); } catch (...) {}
        cleanup();
        throw;
    }
}
void load_file(grlbwt_ctx *ctx, const char *path, int w, uint64_t base = 0, uint64_t range_bytes = ~0ull) {
    ctx->e32.reset();
    ctx->e64.reset();
    if (!(w == 1 || w == 2 || w == 4 || w == 8)) throw prim::Error(GRLBWT_EINVAL, "bad cell width");
    int fd = open(path, O_RDONLY);
    if (fd < 0) throw prim::Error(GRLBWT_EINVAL, std::string("cannot open ") + path);
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); throw prim::Error(GRLBWT_EINVAL, std::string("cannot stat ") + path); }
    uint64_t bytes = (uint64_t)st.st_size;
    try {
        if (bytes == 0 || bytes % (uint64_t)w) throw prim::Error(GRLBWT_EILLFORMED, "Error: the file is ill formed");
        if (range_bytes != ~0ull) {               // a record shard of the file
            if (base % (uint64_t)w || range_bytes % (uint64_t)w || base > bytes || range_bytes > bytes - base)
                throw prim::Error(GRLBWT_EINVAL, "file range outside the file or not on cell boundaries");
            if (range_bytes == 0) throw prim::Error(GRLBWT_EILLFORMED, "Error: the file is ill formed");
            bytes = range_bytes;
        } else base = 0;
        const uint64_t n = bytes / (uint64_t)w;
        bool big = (n >= 0xFFFFFF00ull) || (ctx->flags & GRLBWT_FLAG_FORCE_IDX64);
        bool keep = ctx->flags & GRLBWT_FLAG_KEEP_LEVELS;
        if (big) {
            std::unique_ptr<grl64::Engine> e(new grl64::Engine());
            e->keep_texts = keep;
            load_file_into(*e, fd, base, bytes, w);
            ctx->e64 = std::move(e);
        } else {
            std::unique_ptr<grl32::Engine> e(new grl32::Engine());
            e->keep_texts = keep;
            load_file_into(*e, fd, base, bytes, w);
            ctx->e32 = std::move(e);
        }
    } catch (...) { close(fd); throw; }
    close(fd);
}
// ---- f3: FASTA/FASTQ files (optionally gzip) ------------------------------------------------------------------------
// gzip iff the file starts with 1F 8B (gzopen's rule, which the reference's converter relies on: fastx_handler.cpp:10)
bool file_magic(const char *path, unsigned char out[2]) {
    int fd = open(path, O_RDONLY);
    if (fd < 0) return false;
    out[0] = out[1] = 0;
    ssize_t r = pread(fd, out, 2, 0);
    close(fd);
    return r >= 1;
}
struct RawText { grl64::DBuf<uint8_t> buf; uint64_t n = 0; };
// the (decompressed) bytes of the file in device memory; plain files go through the pinned reader of load_file_into, gzip
// members are inflated on the host (zlib, one stream: the format has no parallel entry points) chunk by chunk into pinned
// buffers that are copied while the next chunk inflates
void read_fastx_raw(const char *path, RawText &R) {
    int fd = open(path, O_RDONLY);
    if (fd < 0) throw prim::Error(GRLBWT_EINVAL, std::string("cannot open ") + path);
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); throw prim::Error(GRLBWT_EINVAL, std::string("cannot stat ") + path); }
    const uint64_t bytes = (uint64_t)st.st_size;
    unsigned char mg[2] = {0, 0};
    if (bytes >= 2 && pread(fd, mg, 2, 0) != 2) { close(fd); throw prim::Error(GRLBWT_EINVAL, "cannot read the input file"); }
    const bool gz = mg[0] == 0x1F && mg[1] == 0x8B;
    constexpr int NB = 2;
    char *bufs[NB] = {nullptr, nullptr};
    prim::Fence fences[NB];
    std::vector<unsigned char> cin;
    z_stream zs;
    bool z_open = false;
    auto cleanup = [&] {
        for (int k = 0; k < NB; k++) { prim::fence_destroy(fences[k]); prim::pinned_free(bufs[k]); bufs[k] = nullptr; }
        if (z_open) inflateEnd(&zs);
        z_open = false;
        close(fd);
    };
    try {
        const uint64_t chunk = kIoChunk;
        for (int k = 0; k < NB; k++) bufs[k] = (char *)prim::pinned_alloc(chunk);
        if (!gz) {
            R.buf.alloc(bytes + 16);
            int k = 0;
            for (uint64_t off = 0; off < bytes; off += chunk, k = (k + 1) % NB) {
                const uint64_t len = bytes - off < chunk ? bytes - off : chunk;
                prim::fence_wait(fences[k]);
                if (!par_io(fd, bufs[k], off, len, false, io_threads())) throw prim::Error(GRLBWT_EINVAL, "cannot read the input file");
                prim::h2d_async(R.buf.p + off, bufs[k], len);
                prim::fence_record(fences[k]);
            }
            R.n = bytes;
        } else {
            uint64_t cap = bytes * 5 + (1 << 20);
            R.buf.alloc(cap + 16);
            memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, 15 + 32) != Z_OK) throw prim::Error(GRLBWT_EINTERNAL, "zlib: inflateInit2 failed");
            z_open = true;
            cin.resize((size_t)8 << 20);
            uint64_t in_off = 0, out_n = 0;
            int k = 0;
            prim::fence_wait(fences[k]);
            zs.next_out = (Bytef *)bufs[k]; zs.avail_out = (uInt)chunk;
            auto flush = [&](bool final) {
                const uint64_t len = chunk - zs.avail_out;
                if (len) {
                    if (out_n + len > cap) {                         // grow the device buffer (copy on the engine's stream)
                        const uint64_t ncap = (out_n + len) * 3 / 2 + (1 << 20);
                        grl64::DBuf<uint8_t> nb(ncap + 16);
                        prim::d2d(nb.p, R.buf.p, out_n);
                        R.buf = std::move(nb);
                        cap = ncap;
                    }
                    prim::h2d_async(R.buf.p + out_n, bufs[k], len);
                    prim::fence_record(fences[k]);
                    out_n += len;
                }
                if (!final) {
                    k = (k + 1) % NB;
                    prim::fence_wait(fences[k]);
                    zs.next_out = (Bytef *)bufs[k]; zs.avail_out = (uInt)chunk;
                }
            };
            bool done = false;
            while (!done) {
                if (zs.avail_in == 0) {
                    if (in_off >= bytes) break;
                    const uint64_t len = bytes - in_off < cin.size() ? bytes - in_off : cin.size();
                    if (!par_io(fd, (char *)cin.data(), in_off, len, false, 1)) throw prim::Error(GRLBWT_EINVAL, "cannot read the input file");
                    in_off += len;
                    zs.next_in = cin.data(); zs.avail_in = (uInt)len;
                }
                const int r = inflate(&zs, Z_NO_FLUSH);
                if (r == Z_STREAM_END) {
                    // another member follows only if the bytes behind this one start with the gzip magic; anything else
                    // (zero padding of blocked / tape files, stray bytes) is ignored, as gzread does
                    if (zs.avail_in < 2 && in_off < bytes) {          // the two bytes may straddle a read: pull more behind the leftover
                        const size_t keep = zs.avail_in;
                        if (keep) memmove(cin.data(), zs.next_in, keep);
                        const uint64_t len = bytes - in_off < cin.size() - keep ? bytes - in_off : cin.size() - keep;
                        if (!par_io(fd, (char *)cin.data() + keep, in_off, len, false, 1)) throw prim::Error(GRLBWT_EINVAL, "cannot read the input file");
                        in_off += len;
                        zs.next_in = cin.data(); zs.avail_in = (uInt)(keep + len);
                    }
                    if (zs.avail_in >= 2 && zs.next_in[0] == 0x1F && zs.next_in[1] == 0x8B) {
                        if (inflateReset(&zs) != Z_OK) throw prim::Error(GRLBWT_EINTERNAL, "zlib: inflateReset failed");
                    } else done = true;
                } else if (r != Z_OK && r != Z_BUF_ERROR) {
                    throw prim::Error(GRLBWT_EINVAL, std::string("the gzip stream is damaged (zlib: ") + (zs.msg ? zs.msg : "error") + ")");
                } else if (r == Z_BUF_ERROR && zs.avail_in == 0 && in_off >= bytes) done = true;       // truncated file: take what there is (gzread does)
                if (zs.avail_out == 0) flush(false);
            }
            flush(true);
            R.n = out_n;
        }
        prim::sync();
        cleanup();
    } catch (...) {
        try { prim::sync(); } catch (...) {}
        cleanup();
        throw;
    }
}
void load_fastx(grlbwt_ctx *ctx, const char *path, uint32_t fx_flags, uint64_t *n_strings) {
    ctx->e32.reset();
    ctx->e64.reset();
    RawText raw;
    read_fastx_raw(path, raw);
    const bool rc = (fx_flags & GRLBWT_FASTX_REVCOMP) != 0;
    const uint64_t cap = (rc ? 2 : 1) * raw.n + 16;
    grl64::DBuf<uint8_t> text(cap + 16);
    grl64::Engine::FastxInfo info = grl64::Engine::fastx_to_text(raw.buf.p, raw.n, rc, text.p, cap);
    raw.buf.release();
    if (n_strings) *n_strings = info.n_strings;
    if (info.n_out == 0) throw prim::Error(GRLBWT_EILLFORMED, "Error: the file is ill formed");      // no record at all
    const uint64_t n = info.n_out;
    bool big = (n >= 0xFFFFFF00ull) || (ctx->flags & GRLBWT_FLAG_FORCE_IDX64);
    bool keep = ctx->flags & GRLBWT_FLAG_KEEP_LEVELS;
    if (big) {
        std::unique_ptr<grl64::Engine> e(new grl64::Engine());
        e->keep_texts = keep;
        e->own0 = std::move(text);
        e->load_text(e->own0.p, n, 1);
        ctx->e64 = std::move(e);
    } else {
        std::unique_ptr<grl32::Engine> e(new grl32::Engine());
        e->keep_texts = keep;
        e->own0.alloc(n + 16);                                        // (the two engines have their own buffer types)
        prim::d2d(e->own0.p, text.p, n);
        text.release();
        e->load_text(e->own0.p, n, 1);
        ctx->e32 = std::move(e);
    }
}
// HBM image -> file.  A buffered write to ONE file runs at the rate of one thread copying into the page cache, whatever the
// number of threads (the inode's lock; tools/io_probe.cpp on the GPU box, 8 GiB to /tmp: 1 thread 10.9 GB/s, 4-16 threads with
// pwrite on disjoint ranges 9.3-9.7, 64 threads 3.8, preallocated with posix_fallocate 11.1, O_DIRECT 7.1 -- and 64 GB/s into
// EIGHT files; a shared mapping filled by 16 threads was 4x slower: page faults).  So: the blocks are preallocated, ONE writer
// thread lives for the whole image and writes the chunks in order as their copies land in a ring of pinned buffers; the device
// copies (50+ GB/s) stay ahead of it.  (Before: a writer thread per 64 MiB chunk that started eight more: 0.97-1.17 s for the
// 8.3 GB image of the 10 GB build.)
// (part = true: the nb bytes go to offset file_off of `path` itself, created if needed and never truncated -- one of N ranks
// writing one file, grlbwt_result_write_part; publishing the complete file is the caller's business)
void write_image(const uint8_t *dev_image, uint64_t nb, const char *path, bool part = false, uint64_t file_off = 0, uint64_t image_total = 0) {
    // The image goes to <path>.tmp~<pid> and is renamed over the target once it is complete and closed (the reference renames
    // bwt_lev_0 to the output name, grl_bwt.hpp:77): an existing output stays intact until then, and a run that is killed or fails
    // leaves at most the temporary behind -- removed on every error path here.  What replacing an existing output costs is the
    // release of its cached pages inside rename() (~0.6 s for 8.3 GB).
    const double t_a = io_now();
    const std::string tmp = part ? std::string(path) : std::string(path) + ".tmp~" + std::to_string((long)getpid());
    int fd = open(tmp.c_str(), part ? (O_WRONLY | O_CREAT) : (O_WRONLY | O_CREAT | O_TRUNC), 0644);
    if (fd < 0) throw prim::Error(GRLBWT_EINVAL, std::string("cannot open ") + tmp);
    constexpr int NBUF = 4;
    char *bufs[NBUF] = {nullptr, nullptr, nullptr, nullptr};
    prim::Fence fences[NBUF];
    struct Job { uint64_t off, len; };
    Job jobs[NBUF];
    std::mutex mu;
    std::condition_variable cv;
    int filled = 0, written = 0;              // chunks handed to the writer / chunks it has finished (ring positions = count % NBUF)
    bool stop = false;
    std::atomic<bool> ok(true);
    std::thread writer;                       // outside the try block: a failing copy must not unwind past a joinable thread
    double t_wait_copy = 0, t_pwrite = 0;
    auto finish_writer = [&] {
        { std::lock_guard<std::mutex> g(mu); stop = true; }
        cv.notify_all();
        if (writer.joinable()) writer.join();
    };
    try {
        const uint64_t chunk = nb < kIoChunk ? nb : kIoChunk;
        if (!part && nb && posix_fallocate(fd, 0, (off_t)nb) != 0 && ftruncate(fd, (off_t)nb) != 0) ok = false;
        const double t_b = io_now();
        auto body = [&] {
            prim::thread_attach();
            for (;;) {
                int k;
                Job j;
                {
                    std::unique_lock<std::mutex> g(mu);
                    cv.wait(g, [&] { return stop || written < filled; });
                    if (written >= filled) return;                   // (stop, nothing left)
                    k = written % NBUF;
                    j = jobs[k];
                }
                const double t0 = io_now();
                try { prim::fence_wait(fences[k]); } catch (...) { ok = false; }
                const double t1 = io_now();
                if (ok && !par_io(fd, bufs[k], file_off + j.off, j.len, true, 1)) ok = false;
                t_wait_copy += t1 - t0; t_pwrite += io_now() - t1;
                { std::lock_guard<std::mutex> g(mu); written++; }
                cv.notify_all();
            }
        };
        try { writer = std::thread(body); } catch (const std::system_error &) { /* no thread to be had: the chunks are written below */ }
        for (uint64_t off = 0; off < nb && ok; off += chunk) {
            const uint64_t len = nb - off < chunk ? nb - off : chunk;
            int k;
            {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&] { return filled - written < NBUF; });         // a free buffer
                k = filled % NBUF;
            }
            if (!bufs[k]) bufs[k] = (char *)prim::pinned_alloc(chunk ? chunk : 16);      // (made as the ring reaches them: the writer is busy by then)
            prim::d2h_async(bufs[k], dev_image + off, len);
            prim::fence_record(fences[k]);
            if (writer.joinable()) {
                { std::lock_guard<std::mutex> g(mu); jobs[k] = Job{off, len}; filled++; }
                cv.notify_all();
            } else {                                                          // (no writer thread: in line)
                prim::fence_wait(fences[k]);
                if (!par_io(fd, bufs[k], file_off + off, len, true, 1)) ok = false;
            }
        }
        {   // everything handed over: wait for the writer to drain
            std::unique_lock<std::mutex> g(mu);
            cv.wait(g, [&] { return written >= filled; });
        }
        finish_writer();
        if (io_trace()) fprintf(stderr, "[grlbwt] write: open + buffers + preallocation %.3f s, chunks %.3f s (writer: waiting for copies %.3f s, writing %.3f s)\n",
                                t_b - t_a, io_now() - t_b, t_wait_copy, t_pwrite);
    } catch (...) {
        finish_writer();
        try { prim::sync(); } catch (...) {}
        for (int k = 0; k < NBUF; k++) { prim::fence_destroy(fences[k]); prim::pinned_free(bufs[k]); }
        close(fd);
        if (!part) unlink(tmp.c_str());
        throw;
    }
    for (int k = 0; k < NBUF; k++) { prim::fence_destroy(fences[k]); prim::pinned_free(bufs[k]); }
    // (a part: the rank whose part ends the image gives the file its size -- a longer file that was there before, a caller that
    // reuses a path, keeps no stale tail)
    if (ok && part && image_total && file_off + nb == image_total && ftruncate(fd, (off_t)image_total) != 0) ok = false;
    if (close(fd) != 0) ok = false;
    if (ok && !part && rename(tmp.c_str(), path) != 0) ok = false;
    if (!ok) { if (!part) unlink(tmp.c_str()); throw prim::Error(GRLBWT_EINVAL, std::string("short write to ") + path); }
}

// ---- primitive self-test (device vs host loops) -----------------------------
uint64_t sm64(uint64_t &s) {
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
struct OddPred {
    const uint32_t *v;
    GRL_DEV bool operator()(uint64_t i) const { return (v[i] % 3u) == 1u; }
};
struct U32In {
    const uint32_t *v;
    GRL_DEV uint64_t operator()(uint64_t i) const { return (uint64_t)(v[i] & 1023u); }
};
struct U32Raw {
    const uint32_t *v;
    GRL_DEV uint32_t operator()(uint64_t i) const { return v[i]; }
};

template <class A, class B>
struct PairOfIn {
    const uint32_t *v;
    GRL_DEV prim::Pair<A, B> operator()(uint64_t i) const { return prim::Pair<A, B>((A)(v[i] & 1u), (B)(v[i] & 4095u)); }
};
template <class A, class B>
int test_pair_scan(uint64_t n, const std::vector<uint32_t> &h, const uint32_t *d) {
    typedef prim::Pair<A, B> P;
    grl32::DBuf<P> o(n + 1);
    P tot = prim::exclusive_scan<P>(n, PairOfIn<A, B>{d}, o.p, true, "selftest.pairscan");
    std::vector<P> ho = o.to_host(n + 1);
    A a = 0; B b = 0;
    for (uint64_t i = 0; i < n; i++) {
        if (ho[i].a != a || ho[i].b != b) return 1;
        a += (A)(h[i] & 1u); b += (B)(h[i] & 4095u);
    }
    if (ho[n].a != a || ho[n].b != b || tot.a != a || tot.b != b) return 2;
    return 0;
}

int test_sort_keys(uint64_t n, uint64_t seed, int bits) {       // keys-only sort on the low `bits` bits: stability through the high bits
    std::vector<uint64_t> hk(n);
    uint64_t s = seed;
    uint64_t mask = (1ull << bits) - 1;
    for (uint64_t i = 0; i < n; i++) hk[i] = (sm64(s) & mask) | (i << bits);       // original index rides above the sorted bits
    grl32::DBuf<uint64_t> ka(n), kb(n);
    prim::h2d(ka.p, hk.data(), n * 8);
    int res = prim::sort_keys<uint64_t>(ka.p, kb.p, n, 0, bits, "selftest.sort_keys");
    std::vector<uint64_t> ok = (res ? kb : ka).to_host(n);
    for (uint64_t i = 0; i < n; i++) {
        uint64_t idx = ok[i] >> bits;
        if (idx >= n || hk[idx] != ok[i]) return 1;
        if (i > 0 && (ok[i - 1] & mask) > (ok[i] & mask)) return 2;
        if (i > 0 && (ok[i - 1] & mask) == (ok[i] & mask) && (ok[i - 1] >> bits) >= idx) return 3;
    }
    return 0;
}

template <class K, class V>
int test_sort(uint64_t n, uint64_t seed, int bits) {
    std::vector<K> hk(n);
    std::vector<V> hv(n);
    uint64_t s = seed;
    K mask = bits >= (int)(8 * sizeof(K)) ? ~K(0) : (K)((K(1) << bits) - 1);
    for (uint64_t i = 0; i < n; i++) { hk[i] = (K)sm64(s) & mask; hv[i] = (V)i; }
    grl32::DBuf<K> ka(n), kb(n);
    grl32::DBuf<V> va(n), vb(n);
    prim::h2d(ka.p, hk.data(), n * sizeof(K));
    prim::h2d(va.p, hv.data(), n * sizeof(V));
    int res = prim::sort_pairs<K, V>(ka.p, va.p, kb.p, vb.p, n, 0, bits, "selftest.sort");
    std::vector<K> ok = (res ? kb : ka).to_host(n);
    std::vector<V> ov = (res ? vb : va).to_host(n);
    for (uint64_t i = 0; i < n; i++) {
        if (ok[i] != hk[(uint64_t)ov[i]]) return 1;                          // pair integrity
        if (i > 0 && ok[i - 1] > ok[i]) return 2;                            // order
        if (i > 0 && ok[i - 1] == ok[i] && ov[i - 1] >= ov[i]) return 3;     // stability
    }
    return 0;
}

struct SelfValid {
    GRL_DEV bool operator()(uint64_t hi) const { return (hi >> 61) != 0; }
};
// RecSort (forward, backward) + rec_dedupe against host containers
int test_part(uint64_t n, uint64_t seed, int pbits) {
    while (pbits < 20 && (n >> pbits) > 3000) pbits++;          // partitions that fit the LDS table (as the engine sizes them)
    std::vector<uint64_t> hk(n), hh(n);
    uint64_t s = seed;
    for (uint64_t i = 0; i < n; i++) {
        const uint64_t r = sm64(s);
        const uint64_t id = r % (n / 3 + 1);                                          // ~3 records per distinct value
        if (r % 11 == 0) { hk[i] = i; hh[i] = 0; }                                    // invalid record
        else { hh[i] = (id % 5) | (3ull << 61); hk[i] = (id * 0x9E3779B97F4A7C15ull) ^ (hh[i] * 0xD6E8FEB86659FD93ull); }
    }
    grl32::DBuf<uint64_t> ka(n), ha(n), kb(n), hb(n), dkey(n), dhi(n);
    grl32::DBuf<uint32_t> wa(n), wb(n), wc(n), lid(n), dcnt(n);
    prim::h2d(ka.p, hk.data(), n * 8);
    prim::h2d(ha.p, hh.data(), n * 8);
    prim::RecSort ps;
    const int res = ps.forward(ka.p, ha.p, kb.p, hb.p, n, pbits, "selftest.part");
    std::vector<uint64_t> sk = (res ? kb : ka).to_host(n), sh = (res ? hb : ha).to_host(n);
    {   // a permutation of the input pairs, grouped by partition, stable inside a partition
        std::map<std::pair<uint64_t, uint64_t>, int64_t> bal;
        for (uint64_t i = 0; i < n; i++) bal[{hk[i], hh[i]}]++;
        for (uint64_t j = 0; j < n; j++) {
            if (j && prim::RecSort::part_of(sk[j - 1], pbits) > prim::RecSort::part_of(sk[j], pbits)) return 1;
            if (--bal[{sk[j], sh[j]}] < 0) return 2;                                  // keys and values moved together
        }
    }
    // backward: an element per sorted record returns to the record's original place
    std::vector<uint32_t> hw(n);
    for (uint64_t j = 0; j < n; j++) hw[j] = (uint32_t)(sk[j] * 31 + sh[j]);
    prim::h2d(wa.p, hw.data(), n * 4);
    ps.backward(wa.p, wb.p, wc.p, "selftest.part_back");
    std::vector<uint32_t> ho = wc.to_host(n);
    for (uint64_t i = 0; i < n; i++) if (ho[i] != (uint32_t)(hk[i] * 31 + hh[i])) return 3;
    // per-partition de-duplication
    const uint64_t nparts = (uint64_t)1 << pbits;
    grl32::DBuf<uint64_t> pstart(nparts + 1);
    grl32::DBuf<uint32_t> pcount(nparts), ovf(1);
    ovf.zero();
    prim::for_each(nparts + 1, prim::RecBoundsFn{res ? kb.p : ka.p, n, pbits, nparts, pstart.p}, "selftest.bounds");
    prim::rec_dedupe(nparts, pstart.p, res ? kb.p : ka.p, res ? hb.p : ha.p, SelfValid{}, lid.p, pcount.p, dkey.p, dhi.p, dcnt.p, ovf.p, "selftest.dedupe");
    if (ovf.get(0)) return 4;
    std::vector<uint64_t> hp = pstart.to_host(nparts + 1), hdk = dkey.to_host(n), hdh = dhi.to_host(n);
    std::vector<uint32_t> hl = lid.to_host(n), hc = pcount.to_host(nparts), hdc = dcnt.to_host(n);
    if (hp[0] != 0 || hp[nparts] != n) return 10;
    for (uint64_t p = 0; p < nparts; p++) {
        std::map<std::pair<uint64_t, uint64_t>, uint32_t> seen;
        for (uint64_t i = hp[p]; i < hp[p + 1]; i++) {
            if (prim::RecSort::part_of(sk[i], pbits) != p) return 11;
            if ((sh[i] >> 61) == 0) { if (hl[i] != prim::kNoId) return 5; continue; }
            seen[{sk[i], sh[i]}]++;
            if (hl[i] >= hc[p]) return 6;
            if (hdk[hp[p] + hl[i]] != sk[i] || hdh[hp[p] + hl[i]] != sh[i]) return 7;    // the local id names the record's value
        }
        if (seen.size() != hc[p]) return 8;
        for (uint32_t j = 0; j < hc[p]; j++) if (seen[{hdk[hp[p] + j], hdh[hp[p] + j]}] != hdc[hp[p] + j]) return 9;
    }
    return 0;
}

int selftest(uint64_t n, uint64_t seed) {
    if (n < 2) n = 2;
    std::vector<uint32_t> h(n);
    uint64_t s = seed;
    for (uint64_t i = 0; i < n; i++) h[i] = (uint32_t)sm64(s);
    grl32::DBuf<uint32_t> d(n);
    prim::h2d(d.p, h.data(), n * 4);
    // 1: exclusive scan (u64) with total
    {
        grl32::DBuf<uint64_t> o(n + 1);
        uint64_t tot = prim::exclusive_scan<uint64_t>(n, U32In{d.p}, o.p, true, "selftest.scan");
        auto ho = o.to_host(n + 1);
        uint64_t acc = 0;
        for (uint64_t i = 0; i < n; i++) { if (ho[i] != acc) return -1; acc += h[i] & 1023u; }
        if (ho[n] != acc || tot != acc) return -2;
    }
    // 2: exclusive scan (u32), in place over a pointer input
    {
        std::vector<uint32_t> small(n);
        for (uint64_t i = 0; i < n; i++) small[i] = h[i] & 7u;
        grl32::DBuf<uint32_t> o(n);
        prim::h2d(o.p, small.data(), n * 4);
        uint32_t tot = prim::exclusive_scan<uint32_t>(n, prim::PtrIn<uint32_t>{o.p}, o.p, false, "selftest.scan32");
        auto ho = o.to_host(n);
        uint32_t acc = 0;
        for (uint64_t i = 0; i < n; i++) { if (ho[i] != acc) return -3; acc += small[i]; }
        if (tot != acc) return -4;
    }
    // 3: reductions
    {
        uint64_t sum = 0; uint32_t mn = ~0u, mx = 0;
        for (uint64_t i = 0; i < n; i++) { sum += h[i] & 1023u; if (h[i] < mn) mn = h[i]; if (h[i] > mx) mx = h[i]; }
        if (prim::reduce_sum<uint64_t>(n, U32In{d.p}) != sum) return -5;
        if (prim::reduce_min<uint32_t>(n, U32Raw{d.p}) != mn) return -6;
        if (prim::reduce_max<uint32_t>(n, U32Raw{d.p}) != mx) return -7;
    }
    // 4: ballot bit-vector
    {
        uint64_t nw = (n + 63) / 64;
        grl32::DBuf<uint64_t> w(nw);
        prim::bitvector_from_pred(n, OddPred{d.p}, w.p, "selftest.bits");
        auto hw = w.to_host(nw);
        for (uint64_t i = 0; i < n; i++) {
            bool b = (hw[i >> 6] >> (i & 63)) & 1ull;
            if (b != ((h[i] % 3u) == 1u)) return -8;
        }
        if (n % 64) if (hw[nw - 1] >> (n % 64)) return -9;
    }
    // 5: byte histogram
    {
        uint64_t hist[256], ref[256] = {0};
        const uint8_t *hb = (const uint8_t *)h.data();
        uint64_t nb = n * 4 - 3;     // odd length: exercises the tail path
        for (uint64_t i = 0; i < nb; i++) ref[hb[i]]++;
        prim::byte_histogram((const uint8_t *)d.p, nb, hist);
        for (int i = 0; i < 256; i++) if (hist[i] != ref[i]) return -10;
    }
    // 6: stable radix sort, all key/value widths used by the engine
    { int r = test_sort<uint32_t, uint32_t>(n, seed + 1, 19); if (r) return -20 - r; }
    { int r = test_sort<uint64_t, uint32_t>(n, seed + 2, 45); if (r) return -30 - r; }
    { int r = test_sort<uint32_t, uint64_t>(n, seed + 3, 8); if (r) return -40 - r; }
    { int r = test_sort<uint64_t, uint32_t>(n, seed + 4, 3); if (r) return -50 - r; }   // heavy duplicates
    { int r = test_sort<uint64_t, uint64_t>(n, seed + 5, 33); if (r) return -60 - r; }
    { int r = test_sort_keys(n, seed + 6, 21); if (r) return -90 - r; }
    // digit plans with 9- and 10-bit digits (GRLBWT_SORT_DIGIT): 54 = 6 x 9, 51 = 9,9,9,8,8,8, 18 = 9,9, 27 = 9,9,9, 20 = 10,10
    { int r = test_sort<uint64_t, uint32_t>(n, seed + 7, 54); if (r) return -100 - r; }
    { int r = test_sort<uint64_t, uint32_t>(n, seed + 8, 51); if (r) return -110 - r; }
    { int r = test_sort_keys(n, seed + 9, 18); if (r) return -120 - r; }
    { int r = test_sort<uint32_t, uint64_t>(n, seed + 10, 27); if (r) return -130 - r; }
    { int r = test_sort<uint32_t, uint32_t>(n, seed + 11, 20); if (r) return -140 - r; }
    // 6b: partition sort that can be undone + per-partition de-duplication (the phrase naming of the levels above 0)
    { int r = test_part(n, seed + 12, 6); if (r) return -150 - r; }
    { int r = test_part(n, seed + 13, 11); if (r) return -160 - r; }
    { int r = test_part(n, seed + 14, 18); if (r) return -170 - r; }
    // 7: fused pair scans (8- and 16-byte elements: the 16-byte result stores and the LDS staging of the scan)
    { int r = test_pair_scan<uint32_t, uint32_t>(n, h, d.p); if (r) return -70 - r; }
    { int r = test_pair_scan<uint64_t, uint64_t>(n, h, d.p); if (r) return -80 - r; }
    return 0;
}

}   // namespace

extern "C" {

int grlbwt_abi_version(void) { return GRLBWT_ABI_VERSION; }
const char *grlbwt_backend_name(void) { return prim::kIsDevice ? "hip-gfx950" : "serial-test-standin"; }

const char *grlbwt_strerror(int code) {
    switch (code) {
        case GRLBWT_OK: return "ok";
        case GRLBWT_EINVAL: return "invalid argument or call out of order";
        case GRLBWT_EDEVICE: return "HIP device/runtime error";
        case GRLBWT_ENOMEM: return "out of memory";
        case GRLBWT_EILLFORMED: return "Error: the file is ill formed";
        case GRLBWT_ERANGE: return "input beyond the supported range";
        case GRLBWT_ENOSPC: return "phrase table overflow";
        case GRLBWT_EINTERNAL: return "internal consistency check failed";
        case GRLBWT_ENOTDNA: return "The input seems not to be DNA";
        default: return "unknown error";
    }
}
const char *grlbwt_last_error(const grlbwt_ctx *ctx) { return ctx ? ctx->err.c_str() : ""; }

// The GRLBWT_* environment switches choose between forms of one computation (tests force most of them and compare the image
// with the oracle: none changes the output) -- but several change what a run COSTS by integer factors (GRLBWT_NOPOOL,
// GRLBWT_NO_PART, GRLBWT_DIST_REPLICATED_*).  A run that has any of them set says so, once per process, on stderr.
// (GRLBWT_QUIET_ENV=1 silences the note: the test suites set switches on purpose.)
static void warn_env_switches_once() {
    static bool done = false;
    if (done) return;
    done = true;
    if (getenv("GRLBWT_QUIET_ENV")) return;
    extern char **environ;
    std::string names;
    int n = 0;
    for (char **e = environ; e && *e; e++) {
        if (strncmp(*e, "GRLBWT_", 7) != 0) continue;
        const char *eq = strchr(*e, '=');
        std::string name(*e, eq ? (size_t)(eq - *e) : strlen(*e));
        // the bench's / tests' own bookkeeping variables are not switches of the library
        if (name.rfind("GRLBWT_BENCH_", 0) == 0 || name == "GRLBWT_HIP_LIB" || name == "GRLBWT_E2E_TMP" || name == "GRLBWT_SIM_LIB") continue;
        names += (n++ ? ", " : "") + name;
    }
    if (n) fprintf(stderr, "[grlbwt] note: %d GRLBWT_* switch%s set in the environment (%s): the image is the same, time and memory of this run may not be\n",
                   n, n == 1 ? "" : "es", names.c_str());
}

int grlbwt_ctx_create(int device_id, uint32_t flags, grlbwt_ctx **out) {
    if (!out) return GRLBWT_EINVAL;
    *out = nullptr;
    warn_env_switches_once();
    grlbwt_ctx *c = new (std::nothrow) grlbwt_ctx();
    if (!c) return GRLBWT_ENOMEM;
    c->flags = flags;
    c->device = device_id;
    int rc = guarded(c, [&] {
        prim::init(device_id);            // refuses a second device while contexts are alive (one GPU per process)
        if (flags & GRLBWT_FLAG_CLASSIC_POOL) prim::pool_classic();
        if (flags & GRLBWT_FLAG_SYNC_DEBUG) prim::rt().sync_each_launch = true;
        prim::rt().live_ctx++;
    });
    if (rc != GRLBWT_OK) { delete c; return rc; }
    *out = c;
    return GRLBWT_OK;
}
void grlbwt_ctx_destroy(grlbwt_ctx *ctx) {
    if (!ctx) return;
    try {
        ctx->e32.reset(); ctx->e64.reset(); prim::sync(); prim::pool_trim();
        if (--prim::rt().live_ctx <= 0) {      // process-wide debug settings and a borrowed stream end with the last context
            prim::rt().live_ctx = 0;
            prim::rt().sync_each_launch = false;
            prim::set_stream(nullptr);
        }
    } catch (...) {}
    delete ctx;
}
int grlbwt_ctx_set_stream(grlbwt_ctx *ctx, void *hip_stream) {
    if (!ctx) return GRLBWT_EINVAL;
    return guarded(ctx, [&] { prim::sync(); prim::set_stream(hip_stream); });   // work queued on the old stream finishes first
}

int grlbwt_text_upload(grlbwt_ctx *ctx, const void *host_cells, uint64_t n_cells, int cell_bytes) {
    if (!ctx || !host_cells) return GRLBWT_EINVAL;
    return guarded(ctx, [&] { load(ctx, host_cells, n_cells, cell_bytes, true); });
}
int grlbwt_text_load_file(grlbwt_ctx *ctx, const char *path, int cell_bytes) {
    if (!ctx || !path) return GRLBWT_EINVAL;
    return guarded(ctx, [&] { load_file(ctx, path, cell_bytes); });
}
int grlbwt_text_load_file_range(grlbwt_ctx *ctx, const char *path, uint64_t offset_bytes, uint64_t n_bytes, int cell_bytes) {
    if (!ctx || !path) return GRLBWT_EINVAL;
    return guarded(ctx, [&] { load_file(ctx, path, cell_bytes, offset_bytes, n_bytes); });
}
int grlbwt_fastx_probe(const char *path, int *is_fastx, int *is_gz) {
    if (!path) return GRLBWT_EINVAL;
    unsigned char mg[2];
    if (!file_magic(path, mg)) return GRLBWT_EINVAL;
    // check_gzip (external/cdt/lib/utils.cpp:53-60): extension ".gz" AND the magic number
    const std::string p(path);
    const bool gz = p.size() >= 3 && p.compare(p.size() - 3, 3, ".gz") == 0 && mg[0] == 0x1F && mg[1] == 0x8B;
    unsigned char first = mg[0];
    if (gz) {                                                        // is_fastx (utils.cpp:13-30): first DEcompressed byte
        gzFile z = gzopen(path, "rb");
        if (!z) return GRLBWT_EINVAL;
        first = 0;
        gzread(z, &first, 1);
        gzclose(z);
    }
    if (is_gz) *is_gz = gz ? 1 : 0;
    if (is_fastx) *is_fastx = (first == '>' || first == '@') ? 1 : 0;
    return GRLBWT_OK;
}
int grlbwt_fastx_convert_device(grlbwt_ctx *ctx, const void *dev_in, uint64_t n_in, uint32_t fx_flags, void *dev_out, uint64_t capacity,
                                uint64_t *n_out, uint64_t *n_strings) {
    if (!ctx || !dev_in || !dev_out) return GRLBWT_EINVAL;
    return guarded(ctx, [&] {
        grl64::Engine::FastxInfo info = grl64::Engine::fastx_to_text((const uint8_t *)dev_in, n_in, (fx_flags & GRLBWT_FASTX_REVCOMP) != 0,
                                                                     (uint8_t *)dev_out, capacity);
        if (n_out) *n_out = info.n_out;
        if (n_strings) *n_strings = info.n_strings;
    });
}
int grlbwt_text_load_fastx(grlbwt_ctx *ctx, const char *path, uint32_t fx_flags, uint64_t *n_strings) {
    if (!ctx || !path) return GRLBWT_EINVAL;
    return guarded(ctx, [&] { load_fastx(ctx, path, fx_flags, n_strings); });
}
int grlbwt_text_attach_device(grlbwt_ctx *ctx, const void *dev_cells, uint64_t n_cells, int cell_bytes) {
    if (!ctx || !dev_cells || ((uintptr_t)dev_cells & 15)) return GRLBWT_EINVAL;
    return guarded(ctx, [&] { load(ctx, dev_cells, n_cells, cell_bytes, false); });
}
int grlbwt_get_stats(const grlbwt_ctx *ctx, grlbwt_stats *out) {
    if (!HAS_ENG(ctx) || !out) return GRLBWT_EINVAL;
    if (ctx->e32) fill_stats(*ctx->e32, out); else fill_stats(*ctx->e64, out);
    return GRLBWT_OK;
}

int grlbwt_parse_round(grlbwt_ctx *ctx, grlbwt_round_info *info, int *done) {
    if (!HAS_ENG(ctx)) return GRLBWT_EINVAL;
    return guarded(ctx, [&] {
        bool d = ENG(ctx, parse_round());
        if (done) *done = d ? 1 : 0;
        if (info) { int r = (int)ENG(ctx, levels.size()) - 1; if (ctx->e32) fill_round(*ctx->e32, r, info); else fill_round(*ctx->e64, r, info); }
    });
}
int grlbwt_parse_phase(grlbwt_ctx *ctx, int *n_rounds) {
    if (!HAS_ENG(ctx)) return GRLBWT_EINVAL;
    return guarded(ctx, [&] { int r = ENG(ctx, parse_phase()); if (n_rounds) *n_rounds = r; });
}
int grlbwt_round_info_get(const grlbwt_ctx *ctx, int round, grlbwt_round_info *info) {
    if (!HAS_ENG(ctx) || !info || round < 0 || round >= (int)ENG(ctx, levels.size())) return GRLBWT_EINVAL;
    if (ctx->e32) fill_round(*ctx->e32, round, info); else fill_round(*ctx->e64, round, info);
    return GRLBWT_OK;
}

int grlbwt_induce_first(grlbwt_ctx *ctx) {
    if (!HAS_ENG(ctx)) return GRLBWT_EINVAL;
    return guarded(ctx, [&] { ENG(ctx, first_bwt()); });
}
int grlbwt_induce_level(grlbwt_ctx *ctx, int *level, grlbwt_level_info *info) {
    if (!HAS_ENG(ctx)) return GRLBWT_EINVAL;
    return guarded(ctx, [&] {
        ENG(ctx, induce_level());
        int l = ENG(ctx, bwt_level);
        if (level) *level = l;
        if (info) { if (ctx->e32) fill_level(*ctx->e32, l, info); else fill_level(*ctx->e64, l, info); }
        if (l == 0) ENG(ctx, finish());
    });
}
int grlbwt_induce_phase(grlbwt_ctx *ctx) {
    if (!HAS_ENG(ctx)) return GRLBWT_EINVAL;
    return guarded(ctx, [&] { ENG(ctx, induce_phase()); ENG(ctx, finish()); });
}
int grlbwt_level_info_get(const grlbwt_ctx *ctx, int level, grlbwt_level_info *info) {
    if (!HAS_ENG(ctx) || !info || level < 0 || level >= (int)ENG(ctx, linfo.size())) return GRLBWT_EINVAL;
    if (ctx->e32) fill_level(*ctx->e32, level, info); else fill_level(*ctx->e64, level, info);
    return GRLBWT_OK;
}
int grlbwt_build(grlbwt_ctx *ctx) {
    if (!HAS_ENG(ctx)) return GRLBWT_EINVAL;
    return guarded(ctx, [&] { ENG(ctx, run_all()); });
}

int grlbwt_result_size(const grlbwt_ctx *ctx, uint64_t *image_bytes, uint64_t *n_runs) {
    if (!HAS_ENG(ctx) || ENG(ctx, image_bytes) == 0) return GRLBWT_EINVAL;
    if (image_bytes) *image_bytes = ENG(ctx, image_bytes);
    if (n_runs) *n_runs = ENG(ctx, image_runs);
    return GRLBWT_OK;
}
int grlbwt_result_device_ptr(const grlbwt_ctx *ctx, const void **dev_ptr) {
    if (!HAS_ENG(ctx) || !dev_ptr || ENG(ctx, image_bytes) == 0) return GRLBWT_EINVAL;
    *dev_ptr = ENG(ctx, image.p);
    return GRLBWT_OK;
}
int grlbwt_result_download(const grlbwt_ctx *ctx, void *host_out, uint64_t capacity) {
    if (!HAS_ENG(ctx) || !host_out || ENG(ctx, image_bytes) == 0 || capacity < ENG(ctx, image_part_bytes)) return GRLBWT_EINVAL;
    return guarded(const_cast<grlbwt_ctx *>(ctx), [&] { if (ENG(ctx, image_part_bytes)) prim::d2h(host_out, ENG(ctx, image.p), ENG(ctx, image_part_bytes)); });
}
int grlbwt_result_write_file(const grlbwt_ctx *ctx, const char *path) {
    if (!HAS_ENG(ctx) || !path || ENG(ctx, image_bytes) == 0) return GRLBWT_EINVAL;
    if (ENG(ctx, image_part_bytes) != ENG(ctx, image_bytes)) return GRLBWT_EINVAL;      // (this context holds a part: grlbwt_result_write_part)
    return guarded(const_cast<grlbwt_ctx *>(ctx), [&] { write_image(ENG(ctx, image.p), ENG(ctx, image_bytes), path); });
}
int grlbwt_result_part(const grlbwt_ctx *ctx, uint64_t *offset, uint64_t *bytes) {
    if (!HAS_ENG(ctx) || ENG(ctx, image_bytes) == 0) return GRLBWT_EINVAL;
    if (offset) *offset = ENG(ctx, image_part_off);
    if (bytes) *bytes = ENG(ctx, image_part_bytes);
    return GRLBWT_OK;
}
int grlbwt_result_write_part(const grlbwt_ctx *ctx, const char *path) {
    if (!HAS_ENG(ctx) || !path || ENG(ctx, image_bytes) == 0) return GRLBWT_EINVAL;
    return guarded(const_cast<grlbwt_ctx *>(ctx), [&] { write_image(ENG(ctx, image.p), ENG(ctx, image_part_bytes), path, true, ENG(ctx, image_part_off), ENG(ctx, image_bytes)); });
}

int grlbwt_level_text_size(const grlbwt_ctx *ctx, int level, uint64_t *n_cells) {
    if (!HAS_ENG(ctx) || !n_cells || level < 1 || level > (int)ENG(ctx, kept_texts.size())) return GRLBWT_EINVAL;
    *n_cells = ENG(ctx, kept_texts[level - 1].n);
    return GRLBWT_OK;
}
int grlbwt_level_text_download(const grlbwt_ctx *ctx, int level, uint64_t *cells_out) {
    if (!HAS_ENG(ctx) || !cells_out || level < 1 || level > (int)ENG(ctx, kept_texts.size())) return GRLBWT_EINVAL;
    return guarded(const_cast<grlbwt_ctx *>(ctx), [&] {
        if (ctx->e32) text_download(*ctx->e32, level, cells_out); else text_download(*ctx->e64, level, cells_out);
    });
}
int grlbwt_level_bwt_size(const grlbwt_ctx *ctx, int level, uint64_t *n_runs) {
    if (!HAS_ENG(ctx) || !n_runs || level < 0 || level >= (int)ENG(ctx, kept_bwts.size())) return GRLBWT_EINVAL;
    *n_runs = ENG(ctx, kept_bwts[level].R);
    return GRLBWT_OK;
}
int grlbwt_level_bwt_download(const grlbwt_ctx *ctx, int level, uint64_t *sym_out, uint64_t *len_out) {
    if (!HAS_ENG(ctx) || !sym_out || !len_out || level < 0 || level >= (int)ENG(ctx, kept_bwts.size())) return GRLBWT_EINVAL;
    return guarded(const_cast<grlbwt_ctx *>(ctx), [&] {
        if (ctx->e32) bwt_download(*ctx->e32, level, sym_out, len_out); else bwt_download(*ctx->e64, level, sym_out, len_out);
    });
}

int grlbwt_level_grammar_size(const grlbwt_ctx *ctx, int level, uint64_t *n_metasyms, uint64_t *prebwt_runs) {
    if (!HAS_ENG(ctx) || level < 0 || level >= (int)ENG(ctx, levels.size())) return GRLBWT_EINVAL;
    if (ENG(ctx, levels[level].g0.p) == nullptr) return GRLBWT_EINVAL;           // already consumed by the induction of this level
    if (n_metasyms) *n_metasyms = ENG(ctx, levels[level].M);
    if (prebwt_runs) *prebwt_runs = ENG(ctx, levels[level].prebwt.R);
    return GRLBWT_OK;
}
int grlbwt_level_grammar_download(const grlbwt_ctx *ctx, int level, uint64_t *g0, uint64_t *g1, uint8_t *has_hocc,
                                  uint64_t *prebwt_sym, uint64_t *prebwt_len) {
    if (grlbwt_level_grammar_size(ctx, level, nullptr, nullptr) != GRLBWT_OK) return GRLBWT_EINVAL;
    return guarded(const_cast<grlbwt_ctx *>(ctx), [&] {
        if (ctx->e32) grammar_download(*ctx->e32, level, g0, g1, has_hocc, prebwt_sym, prebwt_len);
        else grammar_download(*ctx->e64, level, g0, g1, has_hocc, prebwt_sym, prebwt_len);
    });
}

int grlbwt_get_counters(const grlbwt_ctx *ctx, grlbwt_counters *out) {
    if (!HAS_ENG(ctx) || !out) return GRLBWT_EINVAL;
    try { prim::sync(); } catch (...) { return GRLBWT_EDEVICE; }   // folds the stage clocks still in flight
    if (ctx->e32) fill_counters(*ctx->e32, out); else fill_counters(*ctx->e64, out);
    return GRLBWT_OK;
}

int grlbwt_invert_image_tails(grlbwt_ctx *ctx, const void *dev_image, uint64_t image_bytes, int cell_bytes, uint64_t tail_cells,
                              void *dev_out, uint64_t capacity_cells, uint64_t *n_strings_out, uint64_t *n_cells_out) {
    if (!ctx || !dev_image || !dev_out || tail_cells == 0) return GRLBWT_EINVAL;
    return guarded(ctx, [&] {
        const uint64_t total = grl64::Engine::image_total_symbols(dev_image, image_bytes);
        const bool big = total >= 0xFFFFFF00ull || (ctx->flags & GRLBWT_FLAG_FORCE_IDX64);
        uint64_t k = 0;
        const uint64_t n = big ? grl64::Engine::invert_image_tails(dev_image, image_bytes, cell_bytes, tail_cells, dev_out, capacity_cells, &k)
                               : grl32::Engine::invert_image_tails(dev_image, image_bytes, cell_bytes, tail_cells, dev_out, capacity_cells, &k);
        if (n_strings_out) *n_strings_out = k;
        if (n_cells_out) *n_cells_out = n;
    });
}

int grlbwt_invert_image(grlbwt_ctx *ctx, const void *dev_image, uint64_t image_bytes, int cell_bytes,
                        void *dev_text_out, uint64_t capacity_cells, uint64_t *n_cells_out) {
    if (!ctx || !dev_image || !dev_text_out) return GRLBWT_EINVAL;
    return guarded(ctx, [&] {
        // the index width follows the number of symbols the image DESCRIBES (summed in 64 bits), not the buffer sizes:
        // a small image can describe >= 2^32 symbols, and a 32-bit scan of its run lengths would wrap
        const uint64_t total = grl64::Engine::image_total_symbols(dev_image, image_bytes);
        if (total > capacity_cells) throw prim::Error(GRLBWT_EINVAL, "inversion: output buffer too small");
        bool big = total >= 0xFFFFFF00ull || (ctx->flags & GRLBWT_FLAG_FORCE_IDX64);
        uint64_t n = big ? grl64::Engine::invert_image(dev_image, image_bytes, cell_bytes, dev_text_out, capacity_cells, total)
                         : grl32::Engine::invert_image(dev_image, image_bytes, cell_bytes, dev_text_out, capacity_cells, total);
        if (n_cells_out) *n_cells_out = n;
    });
}


int grlbwt_image_plain(grlbwt_ctx *ctx, const void *dev_image, uint64_t image_bytes, void *dev_out_u8,
                       uint64_t capacity, int null_char, uint64_t *n_out) {
    if (!ctx || !dev_image || !dev_out_u8 || null_char > 255) return GRLBWT_EINVAL;
    return guarded(ctx, [&] {
        const uint64_t total = grl64::Engine::image_total_symbols(dev_image, image_bytes);
        if (total > capacity) throw prim::Error(GRLBWT_EINVAL, "grl2plain: output buffer too small");
        bool big = total >= 0xFFFFFF00ull || (ctx->flags & GRLBWT_FLAG_FORCE_IDX64);
        uint64_t n = big ? grl64::Engine::image_plain(dev_image, image_bytes, (uint8_t *)dev_out_u8, capacity, null_char)
                         : grl32::Engine::image_plain(dev_image, image_bytes, (uint8_t *)dev_out_u8, capacity, null_char);
        if (n_out) *n_out = n;
    });
}
int grlbwt_image_rle(grlbwt_ctx *ctx, const void *dev_image, uint64_t image_bytes, void *dev_syms_u8, void *dev_lens_u32,
                     uint64_t capacity_runs, uint64_t *n_runs_out) {
    if (!ctx || !dev_image || !dev_syms_u8 || !dev_lens_u32) return GRLBWT_EINVAL;
    return guarded(ctx, [&] {
        uint64_t r = grl64::Engine::image_rle(dev_image, image_bytes, (uint8_t *)dev_syms_u8, (uint32_t *)dev_lens_u32, capacity_runs);
        if (n_runs_out) *n_runs_out = r;
    });
}
int grlbwt_image_stats_get(grlbwt_ctx *ctx, const void *dev_image, uint64_t image_bytes, grlbwt_image_stats *out) {
    if (!ctx || !dev_image || !out) return GRLBWT_EINVAL;
    return guarded(ctx, [&] {
        grl64::Engine::ImageStats st;
        grl64::Engine::image_stats(dev_image, image_bytes, st);
        out->n_runs = st.n_runs; out->sigma = st.sigma; out->text_size = st.text_size; out->min_run = st.min_run; out->max_run = st.max_run;
        out->fit1 = st.fit1; out->fit2 = st.fit2; out->fit3 = st.fit3;
        for (int c = 0; c < 256; c++) { out->runs_of[c] = st.runs_of[c]; out->freq_of[c] = st.freq_of[c]; }
        for (int i = 0; i < 9; i++) out->deciles[i] = st.deciles[i];
        out->non_maximal = st.non_maximal;
    });
}

int grlbwt_image_split_runs(grlbwt_ctx *ctx, const void *dev_image, uint64_t image_bytes, int bits, uint64_t block_size,
                            void *dev_out, uint64_t capacity_bytes, grlbwt_split_info *info) {
    if (!ctx || !dev_image || !dev_out) return GRLBWT_EINVAL;
    return guarded(ctx, [&] {
        const uint64_t total = grl64::Engine::image_total_symbols(dev_image, image_bytes);
        bool big = total >= 0xFFFFFF00ull || (ctx->flags & GRLBWT_FLAG_FORCE_IDX64);
        uint64_t v[6];
        if (big) {
            auto si = grl64::Engine::image_split_runs(dev_image, image_bytes, bits, block_size, (uint8_t *)dev_out, capacity_bytes);
            v[0] = si.runs_before; v[1] = si.runs_after; v[2] = si.overflow_splits; v[3] = si.block_splits; v[4] = si.n_syms; v[5] = si.out_bytes;
        } else {
            auto si = grl32::Engine::image_split_runs(dev_image, image_bytes, bits, block_size, (uint8_t *)dev_out, capacity_bytes);
            v[0] = si.runs_before; v[1] = si.runs_after; v[2] = si.overflow_splits; v[3] = si.block_splits; v[4] = si.n_syms; v[5] = si.out_bytes;
        }
        if (info) {
            info->runs_before = v[0]; info->runs_after = v[1]; info->overflow_splits = v[2]; info->block_splits = v[3];
            info->n_syms = v[4]; info->out_bytes = v[5];
            info->n_blocks = block_size ? (v[4] > 0 ? 1 + (v[4] - 1) / block_size : 0) : 0;
        }
    });
}

int grlbwt_memory_usage(const grlbwt_ctx *ctx, uint64_t *peak_live_bytes, uint64_t *reserved_bytes) {
    if (!ctx) return GRLBWT_EINVAL;
    if (peak_live_bytes) *peak_live_bytes = prim::pool_peak_bytes();
    if (reserved_bytes) *reserved_bytes = prim::pool_reserved_bytes();
    return GRLBWT_OK;
}

int grlbwt_dist_build(grlbwt_ctx *ctx, const grlbwt_comm *comm) {
    if (!HAS_ENG(ctx) || !comm || !comm->allgather || !comm->alltoallv || comm->size < 1 || comm->rank < 0 || comm->rank >= comm->size) return GRLBWT_EINVAL;
    return guarded(ctx, [&] {
        if (ctx->e32) {
            grl32::Engine::Comm C;
            C.rank = comm->rank; C.size = comm->size; C.user = comm->user; C.ag = comm->allgather; C.a2a = comm->alltoallv;
            C.stream_ordered = (comm->flags & GRLBWT_COMM_STREAM_ORDERED) != 0;
            C.keep_parts = (comm->flags & GRLBWT_COMM_KEEP_PARTS) != 0;
            ctx->e32->dist_build(C);
        } else {
            grl64::Engine::Comm C;
            C.rank = comm->rank; C.size = comm->size; C.user = comm->user; C.ag = comm->allgather; C.a2a = comm->alltoallv;
            C.stream_ordered = (comm->flags & GRLBWT_COMM_STREAM_ORDERED) != 0;
            C.keep_parts = (comm->flags & GRLBWT_COMM_KEEP_PARTS) != 0;
            ctx->e64->dist_build(C);
        }
    });
}

// ---- grlbwt_comm over RCCL, inside the library ---------------------------------------------------------------------
#ifdef GRLBWT_PRIM_HIP
}   // extern "C"
#include <dlfcn.h>
#include <rccl/rccl.h>      // types and prototypes only: librccl (573 MB) is loaded on first use, not linked
namespace {
struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
    bool load() {
        if (lib) return true;
        // a librccl already mapped into the process (torch brings its own) is found by its soname
        for (const char *name : {"librccl.so.1", "librccl.so"}) { lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
        if (!lib) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
        auto sym = [&](const char *n) { void *p = dlsym(lib, n); if (!p) err = std::string("librccl lacks ") + n; return p; };
        GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        AllGather = (decltype(AllGather))sym("ncclAllGather");
        Send = (decltype(Send))sym("ncclSend");
        Recv = (decltype(Recv))sym("ncclRecv");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        if (!err.empty()) { dlclose(lib); lib = nullptr; return false; }
        return true;
    }
};
RcclApi &rccl() { static RcclApi a; return a; }
struct RcclState { ncclComm_t comm = nullptr; int rank = 0, size = 1; };
// stream-ordered callbacks: everything is enqueued on the engine's stream and nobody waits on the host
int rccl_allgather(void *user, const void *send, void *recv, uint64_t bytes) {
    RcclState *S = (RcclState *)user;
    return rccl().AllGather(send, recv, (size_t)bytes, ncclUint8, S->comm, prim::rt().stream) == ncclSuccess ? 0 : 1;
}
int rccl_alltoallv(void *user, const void *send, const uint64_t *sb, const uint64_t *so, void *recv, const uint64_t *rb, const uint64_t *ro) {
    RcclState *S = (RcclState *)user;
    RcclApi &A = rccl();
    bool ok = A.GroupStart() == ncclSuccess;
    for (int g = 0; g < S->size && ok; g++) {        // one grouped send/recv per peer: direct xGMI writes, no staging
        if (sb[g]) ok = A.Send((const char *)send + so[g], (size_t)sb[g], ncclUint8, g, S->comm, prim::rt().stream) == ncclSuccess;
        if (ok && rb[g]) ok = A.Recv((char *)recv + ro[g], (size_t)rb[g], ncclUint8, g, S->comm, prim::rt().stream) == ncclSuccess;
    }
    return (A.GroupEnd() == ncclSuccess && ok) ? 0 : 1;
}
}   // namespace
extern "C" {
int grlbwt_rccl_unique_id(void *id128) {
    if (!id128) return GRLBWT_EINVAL;
    if (!rccl().load()) return GRLBWT_EDEVICE;
    ncclUniqueId id;
    if (rccl().GetUniqueId(&id) != ncclSuccess) return GRLBWT_EINTERNAL;
    static_assert(sizeof id == GRLBWT_RCCL_ID_BYTES, "ncclUniqueId size");
    memcpy(id128, &id, sizeof id);
    return GRLBWT_OK;
}
int grlbwt_rccl_comm_create(grlbwt_ctx *ctx, const void *id128, int rank, int size, grlbwt_comm *comm) {
    if (!ctx || !id128 || !comm || size < 1 || rank < 0 || rank >= size) return GRLBWT_EINVAL;
    return guarded(ctx, [&] {
        if (!rccl().load()) throw prim::Error(GRLBWT_EDEVICE, rccl().err);
        ncclUniqueId id;
        memcpy(&id, id128, sizeof id);
        std::unique_ptr<RcclState> S(new RcclState());
        S->rank = rank; S->size = size;
        ncclResult_t r = rccl().CommInitRank(&S->comm, size, id, rank);          // collective over all ranks; uses the current device
        if (r != ncclSuccess) throw prim::Error(GRLBWT_EINTERNAL, std::string("ncclCommInitRank: ") + rccl().GetErrorString(r));
        comm->rank = rank; comm->size = size; comm->user = S.release();
        comm->allgather = rccl_allgather; comm->alltoallv = rccl_alltoallv;
        comm->flags = GRLBWT_COMM_STREAM_ORDERED;
    });
}
int grlbwt_rccl_comm_destroy(grlbwt_comm *comm) {
    if (!comm || comm->allgather != rccl_allgather || !comm->user) return GRLBWT_EINVAL;
    RcclState *S = (RcclState *)comm->user;
    try { prim::sync(); } catch (...) {}
    if (S->comm) rccl().CommDestroy(S->comm);
    delete S;
    comm->user = nullptr; comm->allgather = nullptr; comm->alltoallv = nullptr;
    return GRLBWT_OK;
}
#else   // the tests' serial stand-in has no device and no RCCL
int grlbwt_rccl_unique_id(void *) { return GRLBWT_EDEVICE; }
int grlbwt_rccl_comm_create(grlbwt_ctx *, const void *, int, int, grlbwt_comm *) { return GRLBWT_EDEVICE; }
int grlbwt_rccl_comm_destroy(grlbwt_comm *) { return GRLBWT_EDEVICE; }
#endif

int grlbwt_profile_enable(grlbwt_ctx *ctx, int on) {
    if (!ctx) return GRLBWT_EINVAL;
    return guarded(ctx, [&] {
        prim::sync();
        prim::rt().prof.clear();
        prim::rt().profile = on != 0;
    });
}
int grlbwt_profile_dump(grlbwt_ctx *ctx, char *buf, uint64_t capacity) {
    if (!ctx || !buf || capacity == 0) return GRLBWT_EINVAL;
    return guarded(ctx, [&] {
        prim::sync();
        std::string s;
        for (const auto &kv : prim::rt().prof) {
            char line[256];
            snprintf(line, sizeof line, "%s %llu %.6f %llu\n", kv.first.c_str(), (unsigned long long)kv.second.launches, kv.second.ms,
                     (unsigned long long)kv.second.bytes);
            s += line;
        }
        size_t n = s.size() < capacity - 1 ? s.size() : (size_t)capacity - 1;
        memcpy(buf, s.data(), n);
        buf[n] = 0;
    });
}

int grlbwt_selftest(grlbwt_ctx *ctx, uint64_t n, uint64_t seed) {
    if (!ctx) return GRLBWT_EINVAL;
    int res = 0;
    int rc = guarded(ctx, [&] { res = selftest(n, seed); });
    return rc != GRLBWT_OK ? rc - 1000 : res;
}

}   // extern "C"

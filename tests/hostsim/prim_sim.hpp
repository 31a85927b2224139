// prim_sim.hpp -- TEST INFRASTRUCTURE ONLY.
//
// Serial host stand-in for grlbwt_amd/csrc/prim_hip.hpp with the same API, so
// that the engine's per-element kernel bodies (engine_impl.hpp functors) and
// its host orchestration can be exercised by the CPU test-suite in a container
// without a GPU.  It is compiled only into tests/hostsim/_build/libgrlbwt_sim.so,
// which nothing in the product (grlbwt_amd/, the C-ABI library, the CLI) loads:
// the product has no CPU path and fails loudly without the HIP library.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <stdexcept>
#include <string>
#include <vector>
#include <map>
#include <utility>

#define GRL_HD inline
#define GRL_DEV inline

namespace prim {

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

static constexpr bool kIsDevice = false;

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

struct Runtime {
    void *stream = nullptr;
    int device = 0;
    int live_ctx = 0;
    int num_cus = 1;
    bool sync_each_launch = false;
    bool profile = false;
    int tag = -1;
    char phase = 0;
    struct ProfAcc { u64 launches = 0; double ms = 0; u64 bytes = 0; };
    std::map<std::string, ProfAcc> prof;
};
inline Runtime &rt() {
    static Runtime r;
    return r;
}
inline void init(int) {}
inline void set_stream(void *) {}
inline void sync() {}

inline void *dev_alloc(size_t bytes) {
    void *p = std::malloc(bytes ? bytes : 16);
    if (!p) throw Error(-12, "malloc");
    return p;
}
inline void dev_free(void *p) { std::free(p); }
inline void dev_shrink(void *, size_t) {}
inline const char *dev_env(const char *) { return nullptr; }      // (experiment switches: development builds of the device library only)
inline const char *test_env(const char *name) { return getenv(name); }      // (switches of the CPU test suites: read here, never by the device library)
inline void h2d(void *d, const void *h, size_t n) { if (n) std::memcpy(d, h, n); }
inline void d2h(void *h, const void *d, size_t n) { if (n) std::memcpy(h, d, n); }
inline void d2d(void *dst, const void *src, size_t n) { if (n) std::memmove(dst, src, n); }
inline void *pinned_alloc(size_t n) { void *p = std::malloc(n ? n : 16); if (!p) throw Error(-12, "malloc"); return p; }
inline void pinned_free(void *p) { std::free(p); }
inline void h2d_async(void *d, const void *h, size_t n) { if (n) std::memcpy(d, h, n); }
inline void d2h_async(void *h, const void *d, size_t n) { if (n) std::memcpy(h, d, n); }
struct Fence { bool armed = false; };
inline void fence_record(Fence &f) { f.armed = true; }
inline void fence_wait(Fence &f) { f.armed = false; }
inline void thread_attach() {}
inline void fence_destroy(Fence &) {}
inline void dev_memset(void *d, int v, size_t n) { if (n) std::memset(d, v, n); }

inline u32 atomic_add(u32 *p, u32 v) { u32 o = *p; *p += v; return o; }
inline u64 atomic_add(u64 *p, u64 v) { u64 o = *p; *p += v; return o; }
inline void atomic_or(u64 *p, u64 v) { *p |= v; }
inline void wave_or_words(u64 *words, bool has, u64 w, u64 m) { if (has) words[w] |= m; }      // (the device form: one atomic per wave and word)
inline void wave_word_store(u64 *words, u64 i, bool has) { if (has) words[i >> 6] |= 1ull << (i & 63); }   // (the device form: one store per wave)
inline u32 atomic_min(u32 *p, u32 v) { u32 o = *p; if (v < o) *p = v; return o; }
inline u32 atomic_max(u32 *p, u32 v) { u32 o = *p; if (v > o) *p = v; return o; }
inline u64 atomic_min(u64 *p, u64 v) { u64 o = *p; if (v < o) *p = v; return o; }
inline u64 atomic_max(u64 *p, u64 v) { u64 o = *p; if (v > o) *p = v; return o; }
inline u64 atomic_cas(u64 *p, u64 expect, u64 desired) { u64 o = *p; if (o == expect) *p = desired; return o; }
inline u64 load_relaxed(const u64 *p) { return *p; }
inline u32 load_relaxed(const u32 *p) { return *p; }

template <class F>
inline void for_each(u64 n, F f, const char * = "") {
    for (u64 i = 0; i < n; i++) f(i);
}
template <class IDX, class F>
inline void for_each_set_bit(u64 nbits, const u64 *words, const IDX *wordbase, F f, const char * = "") {      // f(p, ord) for every set bit p, in order
    for (u64 w = 0; w < (nbits + 63) / 64; w++) {
        u64 ord = (u64)wordbase[w];
        for (u64 x = words[w]; x; x &= x - 1) f(w * 64 + (u64)__builtin_ctzll(x), ord++);
    }
}
template <class F>
inline void bitvector_from_pred(u64 n, F pred, u64 *words, const char * = "") {
    u64 nw = (n + 63) / 64;
    for (u64 w = 0; w < nw; w++) {
        u64 m = 0;
        for (u64 b = 0; b < 64 && w * 64 + b < n; b++)
            if (pred(w * 64 + b)) m |= (1ull << b);
        words[w] = m;
    }
}
template <class cell_t, class OPS, class F>
inline void start_bitvector(u64 n, const cell_t *, OPS, F pred, u64 *words, const char * = "") { bitvector_from_pred(n, pred, words); }
static constexpr u32 kNoBucket = 0xFFFFFFFFu;
static constexpr u32 kDeferBucket = 0xFFFFFFFEu;
static constexpr u32 kClaimBit = 0x80000000u;
template <class F>
inline u32 agg_take_claim(const F &f, u32 s, u64 item) {       // the claim protocol of prim_hip.hpp, serially
    if constexpr (F::kClaims) {
        if (f.claim_bits && s != kNoBucket && s != kDeferBucket) {
            const u64 cp = f.claim_pos(item);
            if (s & kClaimBit) f.claim_bits[cp >> 6] |= 1ull << (cp & 63);
            s &= ~kClaimBit;
        }
    }
    return s;
}
static constexpr u32 kGiantBucket = 0xFFFFFFFDu;
template <class F, class A>
inline void for_each_giant(u64 n_items, const u64 *items, F f, A add, const char * = "") {       // the giant-item kernel of prim_hip.hpp, serially
    for (u64 it = 0; it < n_items; it++) {
        const u64 item = items[it];
        u64 p, ee;
        f.giant_bounds(item, p, ee);
        u64 acc = 0x9E3779B97F4A7C15ull;
        for (int k = 0; k < 64; k++) acc = F::giant_mix(acc, f.giant_piece(p, ee, k));
        u32 s = f.process_giant(item, acc, ee);
        if (s != kNoBucket && s != kGiantBucket) {
            if constexpr (F::kClaims) {
                if (f.claim_bits) {
                    const u64 cp = f.claim_pos(item);
                    if (s & kClaimBit) f.claim_bits[cp >> 6] |= 1ull << (cp & 63);
                    s &= ~kClaimBit;
                }
            }
            add(s, 1u);
        }
    }
}
template <class F>
inline auto agg_first_seen(const F &f, u32 s, u64 item, int) -> decltype(f.first_seen(s, item), void()) { f.first_seen(s, item); }
template <class F>
inline void agg_first_seen(const F &, u32, u64, long) {}
// prim_hip.hpp name_stream, serially: every item named by the functor's streaming methods, the others marked in defer_bits
template <class F, class A>
inline void name_stream(u64 n, F f, A add, u64 *defer_bits, const char * = "") {
    std::vector<u64> items;
    for (u64 i = 0; i < n; i++) if (f.is_start(i)) items.push_back(i);
    const u64 ord0 = n ? (u64)f.ordinal_base(0) : 0;
    for (u64 k = 0; k < items.size(); k++) {
        const u64 p = items[k];
        const u64 next = (k + 1 < items.size() && k % 100 != 99) ? items[k + 1] : f.next_item(p);
        u32 s = kDeferBucket;
        if (p + 8 <= n) s = f.stream_name(p, f.stream_load(p), next);
        if (s == kDeferBucket) defer_bits[p >> 6] |= 1ull << (p & 63);
        else { f.stream_store(ord0 + k, s); add(s, 1u); f.first_seen(s, p); }
    }
}
template <class F>
constexpr auto agg_is_stream(int) -> decltype(F::kStream) { return F::kStream; }
template <class F>
constexpr bool agg_is_stream(long) { return false; }
template <class F, class A>
inline void for_each_agg(u64 n, F f, A add, bool, const char * = "") {
    if constexpr (F::kBatch > 1 && agg_is_stream<F>(0)) {
        // the streaming batched form (prim_hip.hpp k_for_each_agg, STREAM): the k-th work item has ordinal ordinal_base(0) + k and
        // knows the item behind it; the last one of a "ring" looks its successor up (every 100th item here, to take that path too)
        std::vector<u64> items;
        for (u64 i = 0; i < n; i++) if (f.is_start(i)) items.push_back(i);
        const u64 ord0 = n ? (u64)f.ordinal_base(0) : 0;
        for (u64 k0 = 0; k0 < items.size(); k0 += F::kBatch) {
            u64 item[F::kBatch], next[F::kBatch], ord[F::kBatch];
            bool valid[F::kBatch];
            u32 slot[F::kBatch];
            for (int j = 0; j < F::kBatch; j++) {
                const u64 k = k0 + (u64)j;
                valid[j] = k < items.size();
                item[j] = valid[j] ? items[k] : 0;
                next[j] = 0; ord[j] = ord0 + k;
                if (valid[j]) next[j] = (k + 1 < items.size() && k % 100 != 99) ? items[k + 1] : f.next_item(item[j]);
            }
            f.process_batch_stream(item, valid, slot, next, ord);
            for (int j = 0; j < F::kBatch && k0 + (u64)j < items.size(); j++) {
                slot[j] = agg_take_claim(f, slot[j], item[j]);
                if (slot[j] == kDeferBucket) slot[j] = agg_take_claim(f, f.process(item[j]), item[j]);
                if (slot[j] != kNoBucket) { add(slot[j], 1u); agg_first_seen(f, slot[j], item[j], 0); }
            }
        }
    } else
    if constexpr (F::kBatch > 1) {       // the functor's batched form (what the HIP kernel calls), kBatch work items at a time
        u64 item[F::kBatch];
        bool valid[F::kBatch];
        u32 slot[F::kBatch];
        int k = 0;
        auto flush = [&] {
            for (int j = k; j < F::kBatch; j++) { item[j] = 0; valid[j] = false; }
            f.process_batch(item, valid, slot);
            for (int j = 0; j < k; j++) {
                slot[j] = agg_take_claim(f, slot[j], item[j]);
                if (slot[j] == kDeferBucket) slot[j] = agg_take_claim(f, f.process(item[j]), item[j]);       // (the HIP kernel queues these up)
                if (slot[j] != kNoBucket) { add(slot[j], 1u); agg_first_seen(f, slot[j], item[j], 0); }
            }
            k = 0;
        };
        for (u64 i = 0; i < n; i++)
            if (f.is_start(i)) { item[k] = i; valid[k] = true; if (++k == F::kBatch) flush(); }
        if (k) flush();
    } else {
        for (u64 i = 0; i < n; i++) { u32 s = agg_take_claim(f, f(i), i); if (s != kNoBucket) { add(s, 1u); agg_first_seen(f, s, i, 0); } }
    }
}
// stage clocks: host wall time here (the HIP runtime uses event pairs on its stream)
#include <time.h>
inline double sim_now_s() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
inline std::vector<double> &sim_stage_stack() { static std::vector<double> s; return s; }
inline void stage_begin() { sim_stage_stack().push_back(sim_now_s()); }
inline void stage_end(double *acc) {
    auto &s = sim_stage_stack();
    if (s.empty()) return;
    if (acc) *acc += sim_now_s() - s.back();
    s.pop_back();
}
inline void stages_drop() { sim_stage_stack().clear(); }
template <class T, class F>
inline void exclusive_scan_nosync(u64 n, F in, T *out, bool store_total_at_n = false, const char * = "") {
    T acc = 0;
    for (u64 i = 0; i < n; i++) { T v = (T)in(i); out[i] = acc; acc += v; }
    if (store_total_at_n) out[n] = acc;
}
inline void pool_trim() {}
inline void pool_classic() {}
inline void pool_reserve(size_t) {}
inline u64 pool_stage_begin() { return 0; }
inline void pool_stage_end(u64, const void *) {}
inline u64 pool_peak_bytes() { return 0; }
inline u64 pool_reserved_bytes() { return 0; }
inline u64 mem_available() { return ~0ull >> 1; }
template <class T, class F>
inline T reduce_sum(u64 n, F f, const char * = "") { T r = 0; for (u64 i = 0; i < n; i++) r += (T)f(i); return r; }
template <class T, class F>
inline T reduce_min(u64 n, F f, const char * = "") { T r = ~T(0); for (u64 i = 0; i < n; i++) { T v = (T)f(i); if (v < r) r = v; } return r; }
template <class T, class F>
inline T reduce_max(u64 n, F f, const char * = "") { T r = 0; for (u64 i = 0; i < n; i++) { T v = (T)f(i); if (v > r) r = v; } return r; }

template <class A, class B>
struct Pair {
    A a; B b;
    Pair() = default;
    Pair(int) : a(0), b(0) {}
    Pair(A a_, B b_) : a(a_), b(b_) {}
    Pair &operator+=(const Pair &o) { a += o.a; b += o.b; return *this; }
    Pair operator+(const Pair &o) const { return Pair(a + o.a, b + o.b); }
    Pair operator-(const Pair &o) const { return Pair(a - o.a, b - o.b); }
};
template <class T>
struct PtrIn {
    const T *p;
    T operator()(u64 i) const { return p[i]; }
};
template <class T, class F>
inline T exclusive_scan(u64 n, F in, T *out, bool store_total_at_n = false, const char * = "") {
    T acc = 0;
    for (u64 i = 0; i < n; i++) { T v = (T)in(i); out[i] = acc; acc += v; }
    if (store_total_at_n) out[n] = acc;
    return acc;
}
template <class T, class F, class E>
inline T exclusive_scan_emit(u64 n, F in, E emit, const char * = "") {
    T acc = 0;
    for (u64 i = 0; i < n; i++) { T v = (T)in(i); emit(i, acc, v); acc += v; }
    return acc;
}
template <class T, class F, class E>
inline void exclusive_scan_emit_nosync(u64 n, F in, E emit, const char * = "") { (void)exclusive_scan_emit<T, F, E>(n, in, emit); }
inline void byte_histogram_accumulate(const u8 *p, u64 n, u64 *d_hist) { for (u64 i = 0; i < n; i++) d_hist[p[i]]++; }
inline void byte_histogram(const u8 *p, u64 n, u64 *hist_host) {
    std::memset(hist_host, 0, 256 * 8);
    for (u64 i = 0; i < n; i++) hist_host[p[i]]++;
}
template <class K, class V, int SITE = 0>
inline int sort_pairs(K *keys_a, V *vals_a, K *keys_b, V *vals_b, u64 n, int begin_bit, int end_bit,
                      const char * = "") {
    if (n == 0 || end_bit <= begin_bit) return 0;
    int bits = end_bit - begin_bit;
    K mask = bits >= (int)(8 * sizeof(K)) ? ~K(0) : (K)(((K(1) << bits) - 1));
    std::vector<u64> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](u64 a, u64 b) {
        return ((keys_a[a] >> begin_bit) & mask) < ((keys_a[b] >> begin_bit) & mask);
    });
    for (u64 i = 0; i < n; i++) { keys_b[i] = keys_a[idx[i]]; vals_b[i] = vals_a[idx[i]]; }
    // mimic the ping-pong parity of the HIP version
    int passes = (bits + 7) / 8;
    if (passes % 2 == 0) {
        std::memcpy(keys_a, keys_b, n * sizeof(K));
        std::memcpy(vals_a, vals_b, n * sizeof(V));
        return 0;
    }
    return 1;
}

template <class K, int SITE = 0>
inline int sort_keys(K *keys_a, K *keys_b, u64 n, int begin_bit, int end_bit, const char * = "") {
    if (n == 0 || end_bit <= begin_bit) return 0;
    int bits = end_bit - begin_bit;
    K mask = bits >= (int)(8 * sizeof(K)) ? ~K(0) : (K)(((K(1) << bits) - 1));
    std::vector<K> v(keys_a, keys_a + n);
    std::stable_sort(v.begin(), v.end(), [&](K a, K b) { return ((a >> begin_bit) & mask) < ((b >> begin_bit) & mask); });
    int passes = (bits + 7) / 8;
    K *dst = (passes % 2 == 0) ? keys_a : keys_b;
    std::memcpy(dst, v.data(), n * sizeof(K));
    return passes % 2 == 0 ? 0 : 1;
}

// expand + multi-split (serial): same contract as the HIP version
struct XsPlan {
    bool ok = false;
    u64 n = 0, E = 0;
    u32 maxc = 0;
    int bits = 0, db = 8;
    void release() {}
};
template <class GEN, class F>
inline void xs_walk(const GEN &gen, u64 i, bool with_bits, F put) {
    const u32 cur = gen.start(i);
    u64 rec = gen.node(cur);
    const u64 ib = with_bits ? gen.item_bits(i) : 0;
    if (gen.owns(rec)) put(gen.key_own(cur, ib));
    while (gen.more(rec)) {
        const u32 nx = gen.next(rec);
        put(gen.key_step(rec, nx, ib));
        rec = gen.node(nx);
    }
    if (with_bits) gen.finish(i, rec);
}
template <class GEN>
inline u64 expand_count(u64 n, GEN gen, int bits, XsPlan &plan, const char * = "") {
    plan = XsPlan();
    plan.n = n; plan.bits = bits;
    for (u64 i = 0; i < n; i++) {
        u32 c = 0;
        xs_walk(gen, i, false, [&](u64) { c++; });
        plan.E += c;
        if (c > plan.maxc) plan.maxc = c;
    }
    // the stand-in has no staging limit; GRLBWT_SIM_XS_MAXC lets a test push items over the HIP limit's fallback branch
    const char *lim = getenv("GRLBWT_SIM_XS_MAXC");
    plan.ok = plan.maxc <= (lim ? (u32)atoi(lim) : 32u);
    return plan.E;
}
template <class GEN, class K = u64>
inline int expand_sort(GEN gen, XsPlan &plan, K *buf_a, K *buf_b, const char * = "") {
    if (!plan.ok) throw Error(-71, "expand_sort: plan not usable");
    u64 e = 0;
    for (u64 i = 0; i < plan.n; i++) xs_walk(gen, i, true, [&](u64 k) { buf_a[e++] = (K)k; });
    if (e != plan.E) throw Error(-71, "expand_sort: generator produced a different number of keys");
    return sort_keys<K>(buf_a, buf_b, plan.E, 0, plan.bits);
}


// partition sort that can be undone + per-partition de-duplication (serial forms of the primitives of prim_hip.hpp)
struct alignas(16) U128 {
    u64 lo, hi;
    U128() = default;
    U128(int) : lo(0), hi(0) {}
    U128(u64 l, u64 h) : lo(l), hi(h) {}
    bool operator==(const U128 &o) const { return lo == o.lo && hi == o.hi; }
};
static constexpr u32 kNoId = 0xFFFFFFFFu;
// prim_hip.hpp RecSort, serially: records (key, hi) grouped by the top bits of key x kMixMul, stable; and the way back
static constexpr u64 kMixMul = 0x9E3779B97F4A7C15ull;
struct RecSort {
    u64 n = 0;
    int bits = 0;
    std::vector<u64> order;          // order[j] = original position of the record at sorted position j
    static u64 part_of(u64 key, int bits_) { return bits_ ? (key * kMixMul) >> (64 - bits_) : 0; }
    int forward(u64 *key_a, u64 *hi_a, u64 *key_b, u64 *hi_b, u64 n_, int bits_, const char * = "") {
        n = n_; bits = bits_ > 0 ? bits_ : 0;
        order.resize(n);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](u64 a, u64 b) { return part_of(key_a[a], bits) < part_of(key_a[b], bits); });
        for (u64 j = 0; j < n; j++) { key_b[j] = key_a[order[j]]; hi_b[j] = hi_a[order[j]]; }
        return 1;
    }
    template <class W>
    void backward(W *in, W *, W *out, const char * = "") const { for (u64 j = 0; j < n; j++) out[order[j]] = in[j]; }
    void release() { order.clear(); n = 0; bits = 0; }
};
struct RecBoundsFn {
    const u64 *skey; u64 n; int bits; u64 nparts; u64 *pstart;
    void operator()(u64 p) const {
        u64 lo = 0, hi = n;
        if (p == nparts) lo = n;
        else while (lo < hi) { const u64 mid = (lo + hi) >> 1; if (RecSort::part_of(skey[mid], bits) < p) lo = mid + 1; else hi = mid; }
        pstart[p] = lo;
    }
};
template <class VALID>
inline void rec_dedupe(u64 nparts, const u64 *pstart, const u64 *skey, const u64 *shi, VALID valid, u32 *lid, u32 *pcount, u64 *dkey, u64 *dhi, u32 *dcnt,
                       u32 *overflow, const char * = "") {
    // (GRLBWT_SIM_PD_LIMIT: the tests make partitions "overflow" so that the caller's fallback runs)
    const char *lim = getenv("GRLBWT_SIM_PD_LIMIT");
    const u32 limit = lim ? (u32)atoi(lim) : 6000u;
    for (u64 p = 0; p < nparts; p++) {
        const u64 a = pstart[p], b = pstart[p + 1];
        u32 d = 0;
        for (u64 i = a; i < b; i++) {
            if (!valid(shi[i])) { lid[i] = kNoId; continue; }
            u32 j = 0;
            while (j < d && !(dkey[a + j] == skey[i] && dhi[a + j] == shi[i])) j++;
            if (j == d) { dkey[a + d] = skey[i]; dhi[a + d] = shi[i]; dcnt[a + d] = 0; d++; }
            dcnt[a + j]++;
            lid[i] = j;
        }
        pcount[p] = d;
        if (d > limit) *overflow = 1;
    }
}

template <class F>
inline void pack_records(u64 n, F f, u32 rec, u8 *out, const char * = "") {       // out[i * rec ..] = the low rec bytes of f(i)
    for (u64 i = 0; i < n; i++) { const u64 v = f(i); for (u32 b = 0; b < rec; b++) out[i * rec + b] = (u8)(v >> (8 * b)); }
}
// stream merge (serial form of prim_hip.hpp's: segments in output order -> maximal runs)
template <class IDX>
struct SmPlan {
    u64 G = 0;
    u64 take_total = 0, len_total = 0, heads = 0, atoms = 0;
    void release() {}
};
template <class SEG, class IDX, class PUT>
inline void sm_walk(u64 G, const SEG &seg, SmPlan<IDX> &plan, PUT put) {
    u64 x = 0, L = 0, heads = 0, atoms = 0;
    u32 prev = 0xFFFFFFFFu;
    for (u64 g = 0; g < G; g++) {
        u32 sym; IDX len; bool take;
        seg.fetch(seg.locate(g), sym, len, take);
        if (take) {
            const u64 k0 = seg.erank(x + 1) - 1, k1 = seg.erank(x + (u64)len);
            for (u64 k = k0; k < k1; k++) {
                const u32 s = seg.esym(k);
                atoms++;
                if (k > k0 || s != prev) { put(heads, s, k == k0 ? L : L + (seg.epos(k) - x)); heads++; }
                prev = s;
            }
            x += (u64)len;
        } else {
            atoms++;
            if (sym != prev) { put(heads, sym, L); heads++; }
            prev = sym;
        }
        L += (u64)len;
    }
    plan.take_total = x; plan.len_total = L; plan.heads = heads; plan.atoms = atoms;
}
template <class SEG, class IDX>
inline void stream_merge_count(u64 G, SEG seg, SmPlan<IDX> &plan, const char * = "", bool = false) {
    plan = SmPlan<IDX>();
    plan.G = G;
    sm_walk(G, seg, plan, [](u64, u32, u64) {});
}
// the one-walk form (prim_hip.hpp stream_merge_onepass): heads written into arrays of out_cap entries; false = "gave up", the caller
// takes count + emit (GRLBWT_SIM_ONEPASS_GIVES_UP: the tests take that branch)
static constexpr u32 kSmInline = 8;
template <class SEG, class IDX>
inline bool stream_merge_onepass(u64 G, SEG seg, SmPlan<IDX> &plan, u32 *osym, IDX *ostart, u64 out_cap, u64 queue_cap, const char * = "", bool = false) {
    plan = SmPlan<IDX>();
    plan.G = G;
    if (getenv("GRLBWT_SIM_ONEPASS_GIVES_UP")) return false;
    // (serially: the bound the caller computed must hold for every head the walk writes, and for the queued segments)
    u64 wide = 0;
    {
        u64 x = 0;
        for (u64 g = 0; g < G; g++) {
            u32 sym; IDX len; bool take;
            seg.fetch(seg.locate(g), sym, len, take);
            if (take) { const u64 k0 = seg.erank(x + 1) - 1, k1 = seg.erank(x + (u64)len); if (k1 - 1 - k0 > kSmInline) wide++; x += (u64)len; }
        }
    }
    if (wide > queue_cap) throw Error(-71, "stream_merge_onepass: the caller's queue bound does not hold");
    sm_walk(G, seg, plan, [&](u64 r, u32 s, u64 at) {
        if (r >= out_cap) throw Error(-71, "stream_merge_onepass: the caller's bound on the runs does not hold");
        osym[r] = s; ostart[r] = (IDX)at;
    });
    return true;
}
template <class SEG, class IDX>
inline void stream_merge_emit(SEG seg, SmPlan<IDX> &plan, u32 *osym, IDX *ostart, const char * = "") {
    SmPlan<IDX> again = plan;
    sm_walk(plan.G, seg, again, [&](u64 r, u32 s, u64 at) { osym[r] = s; ostart[r] = (IDX)at; });
    if (again.heads != plan.heads) throw Error(-71, "stream_merge_emit: the second walk found a different number of runs");
}

}   // namespace prim

// prim_hip.hpp -- device primitives for gfx950 (MI355X), hand-written HIP.
//
// Everything the engine (engine_impl.hpp) needs from the device besides its own
// kernels: memory, grid-stride element kernels, wave64-ballot bit-vector
// construction, reductions, a reduce-then-scan exclusive scan, an LDS byte
// histogram and a stable LSD radix sort whose in-tile ranking is done with
// wave64 ballots (match-by-bit) and LDS per-wave digit counters.
//
// wave = 64 lanes everywhere (CDNA4); block = 256 threads = 4 waves, one per SIMD.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <stdexcept>
#include <string>
#include <vector>
#include <map>
#include <utility>
#include <type_traits>

#define GRL_HD __host__ __device__ __forceinline__
#define GRL_DEV __device__ __forceinline__

namespace prim {

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

static constexpr bool kIsDevice = true;
#define GRLBWT_PRIM_HIP 1        // (capi_impl.hpp: the in-library RCCL transport exists in this build only)

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

#define GRL_HIP_CHECK(expr)                                                                  \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            throw prim::Error(-5, std::string(#expr) + ": " + hipGetErrorString(e_) +        \
                                      " at " + __FILE__ + ":" + std::to_string(__LINE__));  \
    } while (0)

// Experiment switches (tile shapes, alternative kernels measured against each other) exist in DEVELOPMENT builds only:
// tools/build_dev.sh compiles the library with -DGRLBWT_DEV_SWITCHES into tools/_build/ (bench.py takes it through
// GRLBWT_HIP_LIB).  The product library reads none of them -- dev_env() is a constant there and the names do not reach the binary.
inline const char *dev_env(const char *name) {
#ifdef GRLBWT_DEV_SWITCHES
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}
// Switches that only the CPU test suites set -- older forms of the collection-level flow kept as references for the multi-rank
// tests, limits lowered so that small inputs take a branch -- are read by the serial test stand-in of this header alone
// (tests/hostsim/prim_sim.hpp returns getenv there).  The device library never looks at them.
inline const char *test_env(const char *) { return nullptr; }

// ---------------------------------------------------------------- runtime state
struct Runtime {
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr; // the stream init() created (set_stream may point `stream` at a caller's)
    int device = -1;                  // device the stream and the slabs belong to (-1: not initialised)
    int live_ctx = 0;                 // contexts alive in this process (they share stream and allocator)
    int num_cus = 256;
    u64 bytes_allocated = 0, peak_bytes = 0;
    bool sync_each_launch = false;   // debug: catch faults at the launch site
    // per-kernel timing with HIP events on the engine's stream (bench.py roofline leg)
    bool trace = false;              // GRLBWT_TRACE=1: print every launch and synchronise after it
    bool profile = false;
    int tag = -1;                    // appended to profile names as "#<phase><tag>" (the engine sets it to the level)
    char phase = 0;                  // 'p' parsing round, 'i' induction level (both use the level number as tag)
    struct Prof { std::string name; hipEvent_t a, b; u64 bytes; };
    std::vector<Prof> pending;
    // stage clocks: event pairs on the engine's stream, folded into *acc at the next host synchronisation
    struct Stage { hipEvent_t a, b; double *acc; bool closed; };
    std::vector<Stage> stages;
    std::vector<hipEvent_t> spare_events;
    struct ProfAcc { u64 launches = 0; double ms = 0; u64 bytes = 0; };
    std::map<std::string, ProfAcc> prof;              // name -> launches, total ms, stated algorithmic bytes
};
inline Runtime &rt() {
    static Runtime r;
    return r;
}

inline void pool_trim();
// Stream and slab allocator are process-wide and belong to ONE device.  A context for another device is refused while
// contexts are alive; with none alive the runtime moves over (old stream drained, idle slabs returned).
inline void init(int device) {
    Runtime &R = rt();
    if (R.device >= 0 && R.device != device) {
        if (R.live_ctx > 0)
            throw Error(-22, "this process already runs device " + std::to_string(R.device) + ": one GPU per process (asked for device " +
                                 std::to_string(device) + ")");
        if (R.stream) (void)hipStreamSynchronize(R.stream);
        pool_trim();
        if (R.own_stream) { (void)hipStreamDestroy(R.own_stream); R.own_stream = nullptr; }
        R.stream = nullptr;
    }
    GRL_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t p;
    GRL_HIP_CHECK(hipGetDeviceProperties(&p, device));
    R.device = device;
    if (const char *t = getenv("GRLBWT_TRACE")) { R.trace = t[0] == '1'; R.sync_each_launch = R.trace; }
    R.num_cus = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    if (!R.stream) {
        if (!R.own_stream) GRL_HIP_CHECK(hipStreamCreateWithFlags(&R.own_stream, hipStreamNonBlocking));
        R.stream = R.own_stream;
    }
}
inline void set_stream(void *s) { rt().stream = s ? (hipStream_t)s : rt().own_stream; }
inline void prof_collect();
inline void stages_collect();
inline void sync() {
    GRL_HIP_CHECK(hipStreamSynchronize(rt().stream));
    if (!rt().stages.empty()) stages_collect();
    if (rt().profile) {
        // (the last launch name is read before prof_collect() empties the pending list)
        static const bool by_site = getenv("GRLBWT_SYNC_SITES") != nullptr;      // which launch sites the host waited behind
        std::string last = by_site && !rt().pending.empty() ? rt().pending.back().name : std::string();
        prof_collect();
        rt().prof["@host_sync"].launches += 1;    // how often the host waited for the stream (no kernel of that name)
        if (by_site) rt().prof["@sync_after:" + (last.empty() ? std::string("(nothing launched since the last one)") : last)].launches += 1;
    }
}
inline void prof_begin(const std::string &name, u64 algo_bytes = 0) {
    if (rt().trace) { fprintf(stderr, "[grlbwt] launch %s\n", name.c_str()); fflush(stderr); }
    if (!rt().profile) return;
    Runtime::Prof p;
    p.name = rt().tag >= 0 ? name + "#" + (rt().phase ? std::string(1, rt().phase) : std::string()) + std::to_string(rt().tag) : name;
    p.bytes = algo_bytes;
    GRL_HIP_CHECK(hipEventCreate(&p.a));
    GRL_HIP_CHECK(hipEventCreate(&p.b));
    GRL_HIP_CHECK(hipEventRecord(p.a, rt().stream));
    rt().pending.push_back(p);
}
inline void prof_end() {
    if (!rt().profile) return;
    GRL_HIP_CHECK(hipEventRecord(rt().pending.back().b, rt().stream));
}
// fold finished event pairs into the per-name table (call after a stream sync)
inline void prof_collect() {
    for (auto &p : rt().pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            auto &e = rt().prof[p.name];
            e.launches += 1;
            e.ms += ms;
            e.bytes += p.bytes;
        }
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    rt().pending.clear();
}
// Stage clocks without draining the stream: stage_begin/stage_end record events; the elapsed device time is added
// to *acc the next time the host synchronises anyway (every readback does).  stages_drop() forgets open clocks
// (their accumulators are about to be destroyed).
inline hipEvent_t stage_event() {
    Runtime &R = rt();
    if (!R.spare_events.empty()) { hipEvent_t e = R.spare_events.back(); R.spare_events.pop_back(); return e; }
    hipEvent_t e;
    GRL_HIP_CHECK(hipEventCreate(&e));
    return e;
}
inline void stage_begin() {
    Runtime::Stage st{stage_event(), stage_event(), nullptr, false};
    GRL_HIP_CHECK(hipEventRecord(st.a, rt().stream));
    rt().stages.push_back(st);
}
inline void stage_end(double *acc) {     // closes the innermost open stage (stages nest like scopes)
    Runtime &R = rt();
    for (size_t k = R.stages.size(); k-- > 0;)
        if (!R.stages[k].closed) {
            (void)hipEventRecord(R.stages[k].b, R.stream);
            R.stages[k].acc = acc;
            R.stages[k].closed = true;
            return;
        }
}
inline void stages_collect() {
    Runtime &R = rt();
    std::vector<Runtime::Stage> keep;
    for (auto &st : R.stages) {
        if (!st.closed) { keep.push_back(st); continue; }
        float ms = 0;
        if (hipEventElapsedTime(&ms, st.a, st.b) == hipSuccess && st.acc) *st.acc += (double)ms * 1e-3;
        R.spare_events.push_back(st.a);
        R.spare_events.push_back(st.b);
    }
    R.stages.swap(keep);
}
inline void stages_drop() {
    Runtime &R = rt();
    for (auto &st : R.stages) { R.spare_events.push_back(st.a); R.spare_events.push_back(st.b); }
    R.stages.clear();
}
inline void after_launch(const char *name) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) throw Error(-5, std::string("launch ") + name + ": " + hipGetErrorString(e));
    if (rt().sync_each_launch) {
        e = hipStreamSynchronize(rt().stream);
        if (e != hipSuccess) throw Error(-5, std::string("kernel ") + name + ": " + hipGetErrorString(e));
    }
}

// Slab allocator.  The engine allocates and frees hundreds of scratch arrays per build, from
// bytes to tens of GB; hipMalloc/hipFree cost tens of microseconds for small blocks and tens of
// milliseconds for multi-GB ones, and synchronise the device.  Device memory is therefore taken
// from the runtime in a few large slabs (1 GiB, doubling) and carved up here: first-fit over an
// offset-ordered free list with coalescing (the engine's allocation pattern is close to a stack,
// so fragmentation stays low).  Everything is stream-ordered on ONE stream: a block freed by the
// host right after its last kernel was enqueued may be handed to a later launch on that stream.
//
// The preferred slab is an ARENA: one virtual address range as large as the device's memory, reserved once
// (hipMemAddressReserve costs nothing) and backed with physical memory in 1 GiB steps as the high-water mark rises
// (hipMemCreate + hipMemMap + hipMemSetAccess per 1 GiB piece: ~0.4 ms on a warm device).  One contiguous slab never fragments across
// slabs and costs only what a build touches -- a single hipMalloc of > ~130 GB was measured at 6 s on this device
// (tools/alloc_probe.hip: 128 GB 0.000 s, 200 GB 6.18 s), which is what a whole-device reservation used to cost a
// one-shot run.  hipMalloc slabs (1 GiB, doubling) remain as the fallback when the virtual-memory calls are not
// available and for the multi-process RCCL path (GRLBWT_FLAG_CLASSIC_POOL), whose buffers are handed to the
// communication library.
struct Slab {
    char *base = nullptr;
    size_t size = 0;                          // bytes usable (arena: bytes mapped so far)
    std::map<size_t, size_t> free_list;      // offset -> length
    bool arena = false;
    size_t va_size = 0;                       // arena: reserved address range
    std::vector<std::pair<hipMemGenericAllocationHandle_t, size_t>> chunks;   // arena: physical pieces, in address order
};
struct Pool {
    std::vector<Slab> slabs;
    std::map<void *, std::pair<int, size_t>> live;   // ptr -> (slab, length)
    size_t live_bytes = 0, peak_bytes = 0, slab_bytes = 0, next_slab = (size_t)1 << 30;
    int arena_state = 0;                      // 0 not tried, 1 in use, -1 unavailable / switched off
    bool classic = false;                     // hipMalloc slabs only (GRLBWT_FLAG_CLASSIC_POOL)
};
inline Pool &pool() {
    static Pool p;
    return p;
}
// per-stage high-water marks (GRLBWT_MEM_TRACE=1): the stage is identified by the address of its timer slot
inline u64 pool_stage_begin() {
    Pool &P = pool();
    u64 old = P.peak_bytes;
    P.peak_bytes = P.live_bytes;
    return old;
}
inline void pool_stage_end(u64 old_peak, const void *stage_name) {
    Pool &P = pool();
    static const bool trace = getenv("GRLBWT_MEM_TRACE") != nullptr;
    if (trace) fprintf(stderr, "[grlbwt] stage %-12s %c%d: peak live %.2f GB (live at end %.2f GB)\n", (const char *)stage_name, rt().phase ? rt().phase : '-', rt().tag,
                       P.peak_bytes / 1e9, P.live_bytes / 1e9);
    if (old_peak > P.peak_bytes) P.peak_bytes = old_peak;
}
inline void pool_classic() { pool().classic = true; }    // hipMalloc slabs only (takes effect when no arena is live)
inline u64 pool_peak_bytes() { return pool().peak_bytes; }
inline u64 pool_reserved_bytes() { return pool().slab_bytes; }
// device memory a caller could still get from dev_alloc: what the runtime reports free + what the slabs hold unused
inline u64 mem_available() {
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); fr = 0; }
    const Pool &P = pool();
    return (u64)fr + (u64)(P.slab_bytes > P.live_bytes ? P.slab_bytes - P.live_bytes : 0);
}
inline bool pool_disabled() {
    static int d = getenv("GRLBWT_NOPOOL") ? 1 : 0;
    return d != 0;
}
inline void pool_trim() {                    // give every fully free hipMalloc slab back to the runtime
    Pool &P = pool();
    // The arena stays mapped for the life of the process and is reused by the next context.  Returning its backing
    // (hipMemUnmap + hipMemRelease, with or without freeing the address range) and mapping new memory later was
    // measured UNSAFE on this stack: the third context of a process read zeros where its kernels had just written
    // (tests/test_consumers.py under -m gpu; the same tests pass with hipMalloc slabs).  Nothing is ever unmapped.
    for (auto &sl : P.slabs) {
        if (sl.arena) continue;
        if (sl.base && sl.free_list.size() == 1 && sl.free_list.begin()->second == sl.size) {
            (void)hipFree(sl.base);
            P.slab_bytes -= sl.size;
            sl.base = nullptr; sl.size = 0; sl.free_list.clear();
        }
    }
    bool any = false;
    for (auto &sl : P.slabs) any = any || sl.base;
    if (!any) { P.slabs.clear(); P.next_slab = (size_t)1 << 30; }
}
// ---- arena: reserve once, back on demand
inline bool arena_create(Pool &P) {
    if (P.arena_state != 0) return P.arena_state > 0;
    if (P.classic || getenv("GRLBWT_POOL_CLASSIC")) return false;
    P.arena_state = -1;
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess || tot == 0) { (void)hipGetLastError(); return false; }
    const size_t gib = (size_t)1 << 30;
    size_t va = (tot + gib - 1) / gib * gib;
    void *base = nullptr;
    if (hipMemAddressReserve(&base, va, gib, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
    Slab sl;
    sl.base = (char *)base; sl.size = 0; sl.arena = true; sl.va_size = va;
    P.slabs.insert(P.slabs.begin(), sl);
    for (auto &kv : P.live) kv.second.first += 1;
    P.arena_state = 1;
    return true;
}
// back at least `more` further bytes of the arena (slab 0); false when the device has no memory left for it
struct PoolTrace {                            // GRLBWT_POOL_TRACE=1: what backing the arena cost this process, printed at exit
    bool on = getenv("GRLBWT_POOL_TRACE") != nullptr;
    double grow_s = 0, malloc_s = 0;
    u64 grows = 0, grow_fail = 0, mallocs = 0, malloc_bytes = 0;
    ~PoolTrace() {
        if (on) fprintf(stderr, "[grlbwt] pool: %llu arena steps in %.3f s (%llu refused), %llu hipMalloc slabs (%.2f GB) in %.3f s\n",
                        (unsigned long long)grows, grow_s, (unsigned long long)grow_fail, (unsigned long long)mallocs, malloc_bytes / 1e9, malloc_s);
    }
};
inline PoolTrace &pool_trace() { static PoolTrace t; return t; }
inline bool arena_grow_impl(Pool &P, size_t more);
inline bool arena_grow(Pool &P, size_t more) {
    PoolTrace &T = pool_trace();
    if (!T.on) return arena_grow_impl(P, more);
    const auto t0 = std::chrono::steady_clock::now();
    const bool ok = arena_grow_impl(P, more);
    T.grow_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    T.grows++; if (!ok) T.grow_fail++;
    return ok;
}
inline bool arena_grow_impl(Pool &P, size_t more) {
    Slab &sl = P.slabs[0];
    const size_t gib = (size_t)1 << 30;
    size_t want = (more + gib - 1) / gib * gib;
    if (sl.size + want > sl.va_size) {
        if (pool_trace().on) fprintf(stderr, "[grlbwt] pool: arena of %.2f GB cannot take %.2f GB more (address range %.2f GB)\n", sl.size / 1e9, want / 1e9, sl.va_size / 1e9);
        return false;
    }
    // Pieces of exactly 1 GiB, each with its own handle: hipMemSetAccess refuses ("invalid argument") a piece mapped behind a
    // piece of another size in most combinations (tools/arena_probe.hip, profiles/r03/arena_probe.txt: 10 GiB then 1 GiB,
    // 4 then 1, 1-2-4, 4-4-2 all fail; 104 x 1 GiB works) -- with steps of "whatever the request needs" the arena stopped
    // growing after a first multi-GiB step and every later request became a hipMalloc slab (145-177 GB held for a 10 GB build).
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = rt().device;
    hipMemAccessDesc ad = {};
    ad.location = prop.location;
    ad.flags = hipMemAccessFlagsProtReadWrite;
    size_t added = 0;
    bool ok = true;
    while (added < want && ok) {
        char *at = sl.base + sl.size + added;
        hipMemGenericAllocationHandle_t h;
        hipError_t e = hipMemCreate(&h, gib, &prop, 0);
        const char *what = "hipMemCreate";
        if (e == hipSuccess) {
            e = hipMemMap(at, gib, 0, h, 0);
            what = "hipMemMap";
            if (e == hipSuccess) {
                e = hipMemSetAccess(at, gib, &ad, 1);
                what = "hipMemSetAccess";
                if (e != hipSuccess) (void)hipMemUnmap(at, gib);
            }
            if (e != hipSuccess) (void)hipMemRelease(h);
        }
        if (e != hipSuccess) {
            if (pool_trace().on) fprintf(stderr, "[grlbwt] pool: %s(1 GiB) at arena size %.2f GB: %s\n", what, (sl.size + added) / 1e9, hipGetErrorString(e));
            (void)hipGetLastError();
            ok = false;
        } else {
            sl.chunks.emplace_back(h, gib);
            added += gib;
        }
    }
    if (added) {
        // the new range joins the free list (coalescing with a free tail)
        size_t off = sl.size, len = added;
        if (!sl.free_list.empty()) {
            auto last = std::prev(sl.free_list.end());
            if (last->first + last->second == off) { off = last->first; len += last->second; sl.free_list.erase(last); }
        }
        sl.free_list.emplace(off, len);
        sl.size += added;
        P.slab_bytes += added;
    }
    return ok;
}
inline void *slab_take(Pool &P, int si, size_t need) {
    Slab &sl = P.slabs[si];
    for (auto it = sl.free_list.begin(); it != sl.free_list.end(); ++it) {
        if (it->second >= need) {
            size_t off = it->first, len = it->second;
            sl.free_list.erase(it);
            if (len > need) sl.free_list.emplace(off + need, len - need);
            void *p = sl.base + off;
            P.live[p] = {si, need};
            P.live_bytes += need;
            if (P.live_bytes > P.peak_bytes) P.peak_bytes = P.live_bytes;
            return p;
        }
    }
    return nullptr;
}
// Make sure one slab of at least `bytes` exists (capped to what the device has free): a build of a
// multi-GB input then lives in ONE slab, so large blocks never fail for cross-slab fragmentation.
inline void pool_reserve(size_t bytes) {
    if (pool_disabled()) return;
    Pool &P = pool();
    if (arena_create(P)) return;                     // the arena grows with the build: nothing to reserve
    size_t largest = 0;
    for (auto &sl : P.slabs) if (sl.base && sl.size > largest) largest = sl.size;
    if (largest >= bytes) return;
    size_t fr = 0, tot = 0;
    (void)hipMemGetInfo(&fr, &tot);
    // (24 GB stay with the host framework -- its own tensors, a second context; a cap at 60 % of the free memory was
    // measured to cost 3.4x on the 10 GB build: it then outgrew the slab and paid for further hipMallocs in every step)
    size_t cap = fr > ((size_t)64 << 30) ? fr - ((size_t)24 << 30) : (fr > ((size_t)4 << 30) ? fr - ((size_t)2 << 30) : fr / 2);
    if (bytes > cap) bytes = cap;
    if (bytes > ((size_t)128 << 30)) bytes = (size_t)128 << 30;     // hipMalloc is instant up to 128 GB and takes seconds to minutes above (tools/alloc_probe.hip)
    if (bytes < ((size_t)1 << 30) || bytes <= largest) return;      // the doubling schedule covers small builds; a second, smaller
                                                                    // reservation next to the first helps nobody (and cost a hipFree +
                                                                    // hipMalloc of tens of GB in EVERY build when it was tried)
    pool_trim();                                     // give idle slabs back first
    void *base = nullptr;
    if (hipMalloc(&base, bytes) != hipSuccess) { (void)hipGetLastError(); return; }
    Slab sl;
    sl.base = (char *)base; sl.size = bytes; sl.free_list.emplace(0, bytes);
    P.slabs.insert(P.slabs.begin(), sl);             // first-fit looks here first
    // slab indices of live blocks shift by one
    for (auto &kv : P.live) kv.second.first += 1;
    P.slab_bytes += bytes;
}
inline void *dev_alloc(size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (pool_disabled()) {
        void *q = nullptr;
        hipError_t e0 = hipMalloc(&q, bytes);
        if (e0 != hipSuccess) throw Error(-12, "hipMalloc: " + std::string(hipGetErrorString(e0)));
        return q;
    }
    Pool &P = pool();
    size_t need = (bytes + 255) & ~(size_t)255;
    for (int si = 0; si < (int)P.slabs.size(); si++)
        if (P.slabs[si].base) if (void *p = slab_take(P, si, need)) return p;
    if (arena_create(P)) {
        // back as much more of the arena as the request needs beyond its free tail
        Slab &ar = P.slabs[0];
        size_t tail = 0;
        if (!ar.free_list.empty()) { auto last = std::prev(ar.free_list.end()); if (last->first + last->second == ar.size) tail = last->second; }
        if (arena_grow(P, need - tail)) if (void *p = slab_take(P, 0, need)) return p;
        // no memory left for the arena: fall through to the runtime's own allocator, which reports the failure
    }
    // new slab: at least the request, otherwise the doubling schedule, never more than what is free
    size_t fr = 0, tot = 0;
    (void)hipMemGetInfo(&fr, &tot);
    size_t want = P.next_slab > need ? P.next_slab : need + need / 8;
    size_t cap = fr > ((size_t)1 << 30) ? fr - ((size_t)512 << 20) : fr;
    if (want > cap) want = cap;
    if (want < need) want = need;
    void *base = nullptr;
    const auto t_m0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(&base, want);
    if (pool_trace().on) {
        pool_trace().malloc_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_m0).count();
        pool_trace().mallocs++; pool_trace().malloc_bytes += want;
    }
    if (e != hipSuccess && want > need) { (void)hipGetLastError(); want = need; e = hipMalloc(&base, want); }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)hipStreamSynchronize(rt().stream);
        pool_trim();
        e = hipMalloc(&base, want);
    }
    if (e != hipSuccess) throw Error(-12, "hipMalloc(" + std::to_string(want) + "): " + hipGetErrorString(e));
    Slab sl;
    sl.base = (char *)base; sl.size = want; sl.free_list.emplace(0, want);
    P.slabs.push_back(sl);
    P.slab_bytes += want;
    if (P.next_slab < ((size_t)64 << 30)) P.next_slab *= 2;
    void *p = slab_take(P, (int)P.slabs.size() - 1, need);
    if (!p) throw Error(-12, "slab allocator: internal error");
    return p;
}
inline void dev_free(void *p) {
    if (!p) return;
    Pool &P = pool();
    auto it = P.live.find(p);
    if (it == P.live.end()) { (void)hipFree(p); return; }
    int si = it->second.first;
    size_t len = it->second.second;
    P.live.erase(it);
    P.live_bytes -= len;
    Slab &sl = P.slabs[si];
    size_t off = (size_t)((char *)p - sl.base);
    auto nx = sl.free_list.lower_bound(off);
    if (nx != sl.free_list.end() && off + len == nx->first) { len += nx->second; nx = sl.free_list.erase(nx); }
    if (nx != sl.free_list.begin()) {
        auto pv = std::prev(nx);
        if (pv->first + pv->second == off) { pv->second += len; return; }
    }
    sl.free_list.emplace(off, len);
}
// Keep the first `bytes` of a block and give the rest back (an array allocated for an upper bound, once its size is known).
// Stream-ordered like dev_free: the tail may go to a later launch of the same stream at once.
inline void dev_shrink(void *p, size_t bytes) {
    if (!p || pool_disabled()) return;
    Pool &P = pool();
    auto it = P.live.find(p);
    if (it == P.live.end()) return;
    if (bytes == 0) bytes = 16;
    const size_t need = (bytes + 255) & ~(size_t)255, len = it->second.second;
    if (need >= len) return;
    it->second.second = need;
    P.live_bytes -= len - need;
    Slab &sl = P.slabs[it->second.first];
    size_t off = (size_t)((char *)p - sl.base) + need, tail = len - need;
    auto nx = sl.free_list.lower_bound(off);
    if (nx != sl.free_list.end() && off + tail == nx->first) { tail += nx->second; sl.free_list.erase(nx); }
    sl.free_list.emplace(off, tail);
}
inline void h2d(void *d, const void *h, size_t n) {
    // small uploads (counts, offsets tables, headers) go through a ring of pinned staging slots and do NOT wait: the
    // caller's buffer is free as soon as this returns and the copy is ordered on the stream; the ring is drained
    // before a slot is reused
    constexpr size_t kSlot = 4096, kSlots = 64;
    static char *ring = nullptr;
    static size_t next = 0;
    if (n && n <= kSlot) {
        if (!ring) GRL_HIP_CHECK(hipHostMalloc((void **)&ring, kSlot * kSlots, hipHostMallocDefault));
        if (next == kSlots) { GRL_HIP_CHECK(hipStreamSynchronize(rt().stream)); next = 0; }
        char *slot = ring + kSlot * next++;
        std::memcpy(slot, h, n);
        GRL_HIP_CHECK(hipMemcpyAsync(d, slot, n, hipMemcpyHostToDevice, rt().stream));
        return;
    }
    if (n) GRL_HIP_CHECK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, rt().stream));
    sync();
}
inline void d2h(void *h, const void *d, size_t n) {
    // scalars and small tables (scan totals, flags, per-rank counts) come back through a pinned bounce buffer:
    // a copy to pageable memory is staged and synchronised by the runtime and costs tens of microseconds more
    static void *pinned = nullptr;
    constexpr size_t kPinned = 1 << 16;
    if (n && n <= kPinned) {
        if (!pinned) GRL_HIP_CHECK(hipHostMalloc(&pinned, kPinned, hipHostMallocDefault));
        GRL_HIP_CHECK(hipMemcpyAsync(pinned, d, n, hipMemcpyDeviceToHost, rt().stream));
        sync();
        std::memcpy(h, pinned, n);
        return;
    }
    if (n) GRL_HIP_CHECK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, rt().stream));
    sync();
}
// Host totals of scans and reductions: the kernel stores them straight into a pinned, device-visible page and the host
// reads them after the stream synchronisation -- no copy operation in between (a copy engine transfer of 8 bytes behind
// every scan cost 20-30 us, ~250 times per build).  One page is enough: every user synchronises before it returns.
constexpr size_t kResultPage = 1 << 16;
inline void *result_page() {
    static void *page = nullptr;
    if (!page) GRL_HIP_CHECK(hipHostMalloc(&page, kResultPage, hipHostMallocMapped));
    return page;
}
// pinned staging + stream-ordered copies with completion fences (the file reader / image writer of the C-ABI)
inline void *pinned_alloc(size_t n) { void *p = nullptr; GRL_HIP_CHECK(hipHostMalloc(&p, n, hipHostMallocDefault)); return p; }
inline void pinned_free(void *p) { if (p) (void)hipHostFree(p); }
inline void h2d_async(void *d, const void *h, size_t n) { if (n) GRL_HIP_CHECK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, rt().stream)); }
inline void d2h_async(void *h, const void *d, size_t n) { if (n) GRL_HIP_CHECK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, rt().stream)); }
struct Fence { hipEvent_t e = nullptr; bool armed = false; };
inline void fence_record(Fence &f) {
    if (!f.e) GRL_HIP_CHECK(hipEventCreateWithFlags(&f.e, hipEventDisableTiming));
    GRL_HIP_CHECK(hipEventRecord(f.e, rt().stream));
    f.armed = true;
}
inline void fence_wait(Fence &f) { if (f.armed) { GRL_HIP_CHECK(hipEventSynchronize(f.e)); f.armed = false; } }
inline void thread_attach() { if (rt().device >= 0) (void)hipSetDevice(rt().device); }      // a helper thread that is going to wait for fences
inline void fence_destroy(Fence &f) { if (f.e) (void)hipEventDestroy(f.e); f.e = nullptr; f.armed = false; }
inline void d2d(void *dst, const void *src, size_t n) {
    if (n) GRL_HIP_CHECK(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, rt().stream));
}
inline void dev_memset(void *d, int v, size_t n) {
    if (n) GRL_HIP_CHECK(hipMemsetAsync(d, v, n, rt().stream));
}

// -------------------------------------------------------------------- atomics
GRL_DEV u32 atomic_add(u32 *p, u32 v) { return atomicAdd(p, v); }
GRL_DEV u64 atomic_add(u64 *p, u64 v) {
    return (u64)atomicAdd(reinterpret_cast<unsigned long long *>(p), (unsigned long long)v);
}
GRL_DEV void atomic_or(u64 *p, u64 v) {
    atomicOr(reinterpret_cast<unsigned long long *>(p), (unsigned long long)v);
}
GRL_DEV u32 atomic_min(u32 *p, u32 v) { return atomicMin(p, v); }
GRL_DEV u32 atomic_max(u32 *p, u32 v) { return atomicMax(p, v); }
GRL_DEV u64 atomic_min(u64 *p, u64 v) { return (u64)atomicMin(reinterpret_cast<unsigned long long *>(p), (unsigned long long)v); }
GRL_DEV u64 atomic_max(u64 *p, u64 v) { return (u64)atomicMax(reinterpret_cast<unsigned long long *>(p), (unsigned long long)v); }
GRL_DEV u64 atomic_cas(u64 *p, u64 expect, u64 desired) {
    return (u64)atomicCAS(reinterpret_cast<unsigned long long *>(p), (unsigned long long)expect,
                          (unsigned long long)desired);
}
GRL_DEV u32 load_relaxed(const u32 *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
GRL_DEV u64 load_relaxed(const u64 *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------ launch geometry
static constexpr int kBlock = 256;
inline unsigned grid_for(u64 n, u64 per_block) {
    u64 blocks = (n + per_block - 1) / per_block;
    u64 cap = (u64)rt().num_cus * 8;   // >= 2048 workgroups on 256 CUs, grid-stride beyond
    if (blocks > cap) blocks = cap;
    if (blocks == 0) blocks = 1;
    return (unsigned)blocks;
}

// --------------------------------------------------------------------- for_each
template <class F>
__global__ void __launch_bounds__(kBlock) k_for_each(u64 n, F f) {
    u64 stride = (u64)gridDim.x * kBlock;
    for (u64 i = (u64)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) f(i);
}
// (grid of a grid-stride launch: what the CUs hold at once, when that is less than the 8 workgroups per CU of grid_for -- a functor
// that takes more than 64 vector registers runs 5 or fewer workgroups per CU, and with 8 per CU launched the last 3 start when the
// first 5 are done and run at low occupancy.  GRLBWT_FOR_EACH_GRID=fixed keeps 8 per CU; =2x launches twice the resident number.)
template <class F>
inline unsigned grid_for_each(u64 n) {
    static const int mode = dev_env("GRLBWT_FOR_EACH_GRID") ? (dev_env("GRLBWT_FOR_EACH_GRID")[0] == 'f' ? 0 : 2) : 1;
    static const u64 per_cu = [] {
        int occ = 0;
        if (mode == 0 || hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_for_each<F>, kBlock, 0) != hipSuccess || occ < 1) { (void)hipGetLastError(); return (u64)8; }
        return (u64)(occ >= 8 ? 8 : occ * mode);
    }();
    u64 blocks = (n + kBlock - 1) / kBlock;
    const u64 cap = (u64)rt().num_cus * per_cu;
    if (blocks > cap) blocks = cap;
    if (blocks == 0) blocks = 1;
    return (unsigned)blocks;
}
template <class F>
inline void for_each(u64 n, F f, const char *name = "for_each") {
    if (n == 0) return;
    prof_begin(name);
    hipLaunchKernelGGL(k_for_each<F>, dim3(grid_for_each<F>(n)), dim3(kBlock), 0, rt().stream, n, f);
    prof_end();
    after_launch(name);
}

// --------------------------------------------------- a functor over the set bits of a bit-vector
// f(p, ord) for every set bit p of words[] (ord = the number of set bits in front of p; wordbase[w] = set bits in front of word w).
// A wave walks a contiguous span of words, compacts each word's set bits into an LDS ring (the word IS the ballot of its 64
// positions) and runs f on 64 of them at a time: every lane busy, consecutive lanes at consecutive ordinals.
// (One lane per POSITION with the work behind an `if (bit set)` leaves two lanes in three idle where a third of the positions are
// phrase starts -- the record pass of the phrase naming: 14.8 ms for the 2.9 G positions of level 1 of the 10 GB build.)
template <class IDX, class F>
__global__ void __launch_bounds__(kBlock) k_for_each_set_bit(u64 nwords, const u64 *words, const IDX *wordbase, u32 span, F f) {
    constexpr u32 QCAP = 128;
    __shared__ u32 s_queue[kBlock / 64][QCAP];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    volatile u32 *queue = s_queue[w];
    const u64 w0 = ((u64)blockIdx.x * (kBlock / 64) + (u64)w) * (u64)span;
    if (w0 < nwords) {                               // (wave-uniform; the kernel has no workgroup barrier)
        const u64 w1 = w0 + span < nwords ? w0 + span : nwords;
        const u64 p0 = w0 * 64;
        u64 ord = (u64)wordbase[w0];
        u32 qh = 0, qt = 0;                          // ring positions (items are offsets from p0: a span is below 2^32 positions)
        for (u64 x = w0; x < w1; x++) {
            const u64 word = words[x];
            if ((word >> lane) & 1ull) queue[(qt + (u32)__popcll(word & ((1ull << lane) - 1ull))) & (QCAP - 1)] = (u32)((x - w0) * 64) + (u32)lane;
            qt += (u32)__popcll(word);
            if (qt - qh >= 64u) {
                f(p0 + (u64)queue[(qh + (u32)lane) & (QCAP - 1)], ord + (u64)lane);
                qh += 64u; ord += 64u;
            }
        }
        if ((u32)lane < qt - qh) f(p0 + (u64)queue[(qh + (u32)lane) & (QCAP - 1)], ord + (u64)lane);
    }
}
template <class IDX, class F>
inline void for_each_set_bit(u64 nbits, const u64 *words, const IDX *wordbase, F f, const char *name = "for_each_set_bit") {
    const u64 nwords = (nbits + 63) / 64;
    if (nwords == 0) return;
    // spans of 64 words (4096 positions) per wave -- shorter when that leaves the CUs short of waves
    u32 span = 64;
    while (span > 4 && (nwords + span - 1) / span < (u64)rt().num_cus * 32) span >>= 1;
    const u64 waves = (nwords + span - 1) / span;
    prof_begin(name);
    hipLaunchKernelGGL((k_for_each_set_bit<IDX, F>), dim3((unsigned)((waves + kBlock / 64 - 1) / (kBlock / 64))), dim3(kBlock), 0, rt().stream, nwords, words, wordbase, span, f);
    prof_end();
    after_launch(name);
}

// --------------------------------------------------- bit-vector from predicate
// words[i/64] bit (i%64) = pred(i); one wave64 ballot per word, lane 0 stores.
template <class F>
__global__ void __launch_bounds__(kBlock) k_bitvector(u64 n_padded, u64 n, F pred, u64 *words) {
    u64 stride = (u64)gridDim.x * kBlock;
    for (u64 i = (u64)blockIdx.x * kBlock + threadIdx.x; i < n_padded; i += stride) {
        bool b = (i < n) ? pred(i) : false;
        unsigned long long m = __ballot(b);
        if ((threadIdx.x & 63) == 0) words[i >> 6] = m;
    }
}
template <class F>
inline void bitvector_from_pred(u64 n, F pred, u64 *words, const char *name = "bitvector") {
    if (n == 0) return;
    u64 n_padded = (n + 63) & ~u64(63);
    prof_begin(name);
    hipLaunchKernelGGL(k_bitvector<F>, dim3(grid_for(n_padded, kBlock)), dim3(kBlock), 0, rt().stream, n_padded, n,
                       pred, words);
    prof_end();
    after_launch(name);
}

// Phrase-start bit-vector of a text level (LMS breaks + string starts), one wave per 64 positions.
//   start(p) = p == 0 | T(p-1) | ( sym(p-1) > sym(p) & rep(p-1) & rep(p) & S(p) )
//   S(p)     = !T(p) & ( sym(p+1) > sym(p)  |  sym(p+1) == sym(p) & S(p+1) )            (type of p is S)
// S runs right-to-left through runs of equal symbols.  On the 64-bit masks G = !T & (next > own) and P = !T & (next ==
// own) that recurrence is the carry chain of an addition once the bits are reversed: carry_out(k) = G'(k) | P'(k) &
// carry_in(k) is exactly what (G'|P') + G' produces, so one 64-bit add resolves every run inside the word; only a run
// that reaches the last position of the word needs a look at the following cells (a wave-uniform walk).
// `ops` supplies sym()/rep()/isT() of a cell; `pred` is the per-position definition (used by the serial test stand-in
// of this header, and here for nothing: kept so that both take the same arguments).
// (One cell per lane and a ballot per predicate ran at 93 % of the VALU issue rate -- 50 vector + 25 scalar instructions
// per 64 cells, 13.5 ms for 10 GB of bytes.  Now a lane takes C consecutive cells (8 bytes, or 4 cells of 2/4 bytes, or 2 of
// 8) out of LDS, evaluates the four predicates on them in registers, and the C-bit results are put together into bytes
// (shuffles) and bytes into words (LDS); one lane per word then does the carry-chain step.)
template <class cell_t, class OPS>
__global__ void __launch_bounds__(kBlock) k_start_bits(u64 n, const cell_t *t, OPS ops, u64 *words) {
    // A tile of kBlock x 16 bytes is read with 16-byte loads and staged in LDS with one cell of halo on either side.
    constexpr int CPL = 16 / (int)sizeof(cell_t);      // cells per lane per load
    constexpr int TILE = kBlock * CPL;                 // cells per tile (a multiple of 64)
    constexpr int C = sizeof(cell_t) == 1 ? 8 : (sizeof(cell_t) == 8 ? 2 : 4);    // cells per lane per step
    constexpr int WPS = C;                             // words per wave step (64 * C cells)
    constexpr int STEP = 64 * C;
    struct alignas(16) Vec { cell_t v[CPL]; };
    __shared__ __attribute__((aligned(16))) cell_t s_raw[TILE + 2 * CPL];   // tile at [CPL, CPL+TILE): 16-byte aligned
    __shared__ __attribute__((aligned(8))) u8 s_m[kBlock / 64][4][8 * C];   // per wave: the four masks of a step, one byte per 8 cells
    cell_t *s_c = s_raw + CPL;                          // s_c[-1] = t[base-1], s_c[TILE] = t[base+TILE]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u64 ntiles = (n + TILE - 1) / TILE;
    const bool aligned = ((uintptr_t)t & 15) == 0;
    for (u64 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const u64 base = tile * TILE;
        const u64 q0 = base + (u64)threadIdx.x * CPL;
        if (aligned && q0 + CPL <= n) {
            *reinterpret_cast<Vec *>(s_c + threadIdx.x * CPL) = *reinterpret_cast<const Vec *>(t + q0);
        } else {
#pragma unroll
            for (int k = 0; k < CPL; k++) s_c[threadIdx.x * CPL + k] = (q0 + k < n) ? t[q0 + k] : cell_t(0);
        }
        if (threadIdx.x == 0) s_c[-1] = base > 0 ? t[base - 1] : cell_t(0);
        if (threadIdx.x == 64) s_c[TILE] = (base + TILE < n) ? t[base + TILE] : cell_t(0);
        __syncthreads();
        for (int st = wave; st < TILE / STEP; st += kBlock / 64) {
            const int o = st * STEP + lane * C;            // my C cells: s_c[o .. o+C)
            const u64 p0 = base + (u64)o;
            cell_t c[C + 2];
#pragma unroll
            for (int j = 0; j < C + 2; j++) c[j] = s_c[o - 1 + j];     // c[0] = the cell in front, c[C+1] = the cell behind
            u32 mg = 0, mp = 0, mf = 0, mc = 0;
#pragma unroll
            for (int j = 0; j < C; j++) {
                const u64 p = p0 + (u64)j;
                const bool in = p < n, has_nx = p + 1 < n;
                const cell_t cc = c[j + 1], nx = c[j + 2], pv = c[j];
                const u32 s = ops.sym(cc), sn = ops.sym(nx), sp = ops.sym(pv);
                const bool T = ops.isT(cc);
                if (in && has_nx && !T && sn > s) mg |= 1u << j;
                if (in && has_nx && !T && sn == s) mp |= 1u << j;
                if (in && (p == 0 || ops.isT(pv))) mf |= 1u << j;
                if (in && p > 0 && sp > s && ops.rep(pv) && ops.rep(cc)) mc |= 1u << j;
            }
            // C-bit pieces -> bytes (neighbouring lanes), bytes -> LDS
#pragma unroll
            for (int k = C; k < 8; k *= 2) {
                mg |= (u32)__shfl_down((int)mg, k / C) << k;
                mp |= (u32)__shfl_down((int)mp, k / C) << k;
                mf |= (u32)__shfl_down((int)mf, k / C) << k;
                mc |= (u32)__shfl_down((int)mc, k / C) << k;
            }
            if (lane % (8 / C) == 0) {
                const int bi = lane / (8 / C);
                s_m[wave][0][bi] = (u8)mg; s_m[wave][1][bi] = (u8)mp; s_m[wave][2][bi] = (u8)mf; s_m[wave][3][bi] = (u8)mc;
            }
            // (the same wave reads what it wrote: LDS operations of a wave complete in order)
            __builtin_amdgcn_wave_barrier();
            if (lane < WPS) {
                const int wv = st * WPS + lane;                         // word of the tile
                const unsigned long long G = *reinterpret_cast<const u64 *>(&s_m[wave][0][lane * 8]);
                const unsigned long long P = *reinterpret_cast<const u64 *>(&s_m[wave][1][lane * 8]);
                const unsigned long long first = *reinterpret_cast<const u64 *>(&s_m[wave][2][lane * 8]);
                const unsigned long long cand = *reinterpret_cast<const u64 *>(&s_m[wave][3][lane * 8]);
                // carry into the word: type of the position after the word, within the run of the word's last symbol.
                // Only the word that holds the START of that run can need it (a candidate is the first cell of its run);
                // the words inside a long run must not walk it again (that would be quadratic in the run length).
                unsigned long long cin = 0;
                const int trailing = (~P == 0ull) ? 64 : __builtin_clzll(~P);    // length of the run of set bits ending at bit 63
                const int rstart = 64 - trailing;                    // first position of that run inside the word
                if ((P >> 63) && ((cand >> rstart) & 1ull)) {
                    const u32 sl = ops.sym(s_c[wv * 64 + 63]);
                    u64 q = base + (u64)wv * 64 + 64;                // t[q] continues the run (P bit 63), q < n
                    bool walking = true;
                    {   // long runs (an N gap of millions of cells behind a larger symbol): 4 x 8 bytes per step while they repeat the
                        // run's cell -- the walk below takes two dependent loads per cell: 1 s for 5 M cells
                        constexpr u64 per = 8 / sizeof(cell_t);
                        const cell_t rc = t[q];
                        u64 pat = 0;
                        for (u64 x = 0; x < per; x++) pat = sizeof(cell_t) == 8 ? (u64)rc : ((pat << (8 * sizeof(cell_t) % 64)) | (u64)rc);
                        bool fast = !ops.isT(rc);
                        while (fast && q + 4 * per + 1 < n) {
                            u64 d = 0;
#pragma unroll
                            for (int x = 0; x < 4; x++) { u64 v; __builtin_memcpy(&v, t + q + 1 + (u64)x * per, 8); d |= v ^ pat; }
                            if (d) fast = false; else q += 4 * per;
                        }
                    }
                    while (walking) {
                        const cell_t cq = t[q];
                        if (ops.isT(cq) || q + 1 >= n) walking = false;            // the run reaches the string end: type L
                        else {
                            const u32 sq = ops.sym(t[q + 1]);
                            if (sq != sl) { cin = sq > sl ? 1 : 0; walking = false; }
                            else q++;
                        }
                    }
                }
                const unsigned long long Gr = __brevll(G), Pr = __brevll(P);
                const unsigned long long a = Gr | Pr, b = Gr;
                const unsigned long long s1 = a + b;
                const unsigned long long sum = s1 + cin;
                const unsigned long long ovf = ((s1 < a) || (sum < s1)) ? 1ull : 0ull;
                const unsigned long long cinto = sum ^ a ^ b;        // carry into every bit
                const unsigned long long Sr = (cinto >> 1) | (ovf << 63);   // carry out of every bit
                const unsigned long long S = __brevll(Sr);
                if ((base + (u64)wv * 64) < n) words[(base >> 6) + wv] = first | (cand & S);
            }
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
    }
}
template <class cell_t, class OPS, class F>
inline void start_bitvector(u64 n, const cell_t *t, OPS ops, F /*pred*/, u64 *words, const char *name = "start_bits") {
    if (n == 0) return;
    u64 nwords = (n + 63) >> 6;
    prof_begin(name);
    (void)nwords;
    u64 ntiles = (n + (u64)kBlock * (16 / sizeof(cell_t)) - 1) / ((u64)kBlock * (16 / sizeof(cell_t)));
    u64 cap = (u64)rt().num_cus * 32;
    hipLaunchKernelGGL((k_start_bits<cell_t, OPS>), dim3((unsigned)(ntiles < cap ? ntiles : cap)), dim3(kBlock), 0, rt().stream, n, t, ops, words);
    prof_end();
    after_launch(name);
}

// OR of (word, mask) contributions of the lanes of a wave into words[]: ONE atomic per distinct word instead of one per lane
// (the lanes of a wave hold consecutive elements, whose marks fall into a handful of words)
GRL_DEV void wave_or_words(u64 *words, bool has, u64 w, u64 m) {
    const int lane = threadIdx.x & 63;
    unsigned long long pending = __ballot(has);
    while (pending) {
        const int leader = __ffsll((long long)pending) - 1;
        const u64 wl = (u64)__shfl((unsigned long long)w, leader);
        const bool same = has && w == wl;
        unsigned long long mm = same ? (unsigned long long)m : 0ull;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mm |= __shfl_xor(mm, off);
        if (lane == leader) atomicOr(reinterpret_cast<unsigned long long *>(&words[wl]), mm);
        pending &= ~__ballot(same);
    }
}

// words[i / 64] = the wave's `has` flags, for a functor of prim::for_each / for_each_agg-free launches whose lanes hold the 64
// CONSECUTIVE elements i - (i % 64) .. of one word (k_for_each: element = blockIdx * 256 + threadIdx + k * stride, strides are
// multiples of 256) -- a plain store by one lane, no atomic, nothing when no lane has a flag (words[] starts zeroed).  Call it
// from converged code (every lane that holds an element).
GRL_DEV void wave_word_store(u64 *words, u64 i, bool has) {
    const unsigned long long b = __ballot(has);
    if (b && (threadIdx.x & 63) == (unsigned)(__ffsll((long long)b) - 1)) words[i >> 6] = (u64)b;
}

// ------------------------------------------ for_each with LDS count aggregation
// f(i) returns a bucket id (u32) or kNoBucket; every returned id must be counted once
// in a global table through add(id, count).  Same-address global atomics serialise at
// the memory side, so each workgroup first aggregates (id -> count) in a 2048-entry LDS
// open-addressing cache over its CONTIGUOUS chunk of the index space and flushes one
// atomic per distinct id at the end; ids that do not fit the cache go straight to add().
static constexpr u32 kNoBucket = 0xFFFFFFFFu;
// f.is_start(i): cheap test (is i a work item?);  f.process(i): the expensive, divergent part.
// The measured kernel was instruction-issue bound by divergence (30 % of the lanes carry a phrase
// at DNA phrase lengths; 577 SALU + 266 VALU instructions per wave iteration, mostly exec-mask
// bookkeeping), so every workgroup first COMPACTS the work items of a 2048-position chunk into an
// LDS queue (wave ballot + one LDS atomic per wave) and then runs process() with all lanes busy.  (With the queue the
// counters read 69-88 % of the wave cycles waiting on memory at 6-7 waves per SIMD: the rest is gather latency.)
static constexpr int kAggChunk = 2048;
static constexpr u32 kDeferBucket = 0xFFFFFFFEu;   // process_batch: "take this item through process() later"
// CLAIMS (functors with F::kClaims and a non-null f.claim_bits): a bucket id with bit 31 set tells that THIS work item created
// its bucket (the lane's CAS claimed the table slot).  The kernel strips the bit and sets bit (item) of f.claim_bits --
// the lanes of a wave hold neighbouring items, so the wave ORs its marks together and issues one atomic per word.  The set bits
// are the distinct buckets, each once: the caller compacts from them instead of scanning a sparse table.  (Bucket ids < 2^31.)
static constexpr u32 kClaimBit = 0x80000000u;
// GIANT work items (a phrase of 10^8 cells is too long for one lane: 4 s).  The functor of a hashing pass lists them
// (positions, in f.giant_list) instead of processing them; for_each_giant then takes one WAVE per listed item: lane 0 finds
// the item's bounds, every lane hashes one of 64 pieces, the piece hashes are mixed in lane order and lane 0 finishes with the
// result (slot lookup / insertion, claim mark, count).  A kernel of its own: with the same steps inside the hashing kernel the
// level-0 pass of the 10 GB build went from 46 to 142 ms (registers), although it never meets such a phrase.
static constexpr u32 kGiantBucket = 0xFFFFFFFDu;
template <class F, class A>
__global__ void __launch_bounds__(64) k_giant(u64 n_items, const u64 *items, F f, A add) {
    const int lane = threadIdx.x & 63;
    for (u64 it = blockIdx.x; it < n_items; it += gridDim.x) {
        const u64 item = items[it];
        u64 p = 0, ee = 0;
        if (lane == 0) f.giant_bounds(item, p, ee);
        p = (u64)__shfl((unsigned long long)p, 0);
        ee = (u64)__shfl((unsigned long long)ee, 0);
        const u64 mine = f.giant_piece(p, ee, lane);
        u64 acc = 0x9E3779B97F4A7C15ull;
        for (int k = 0; k < 64; k++) acc = F::giant_mix(acc, (u64)__shfl((unsigned long long)mine, k));
        if (lane == 0) {
            u32 s = f.process_giant(item, acc, ee);
            if (s != kNoBucket && s != kGiantBucket) {
                if constexpr (F::kClaims) {
                    if (f.claim_bits) {
                        const u64 cp = f.claim_pos(item);
                        if (s & kClaimBit) atomicOr(reinterpret_cast<unsigned long long *>(&f.claim_bits[cp >> 6]), 1ull << (cp & 63));
                        s &= ~kClaimBit;
                    }
                }
                add(s, 1u);
            }
        }
    }
}
template <class F, class A>
inline void for_each_giant(u64 n_items, const u64 *items, F f, A add, const char *name = "giant_items") {
    if (n_items == 0) return;
    prof_begin(name);
    hipLaunchKernelGGL((k_giant<F, A>), dim3((unsigned)(n_items < 65536 ? n_items : 65536)), dim3(64), 0, rt().stream, n_items, items, f, add);
    prof_end();
    after_launch(name);
}
template <class F>
GRL_DEV u32 agg_take_claim(const F &f, u32 s, u64 item, bool valid) {
    if constexpr (F::kClaims) {
        if (f.claim_bits) {                                      // (uniform)
            const bool real = valid && s != kNoBucket && s != kDeferBucket;
            const u64 cp = f.claim_pos(item);                  // (a functor over a sample of the positions maps its virtual index)
            wave_or_words(f.claim_bits, real && (s & kClaimBit), cp >> 6, 1ull << (cp & 63));
            if (real) s &= ~kClaimBit;
        }
    }
    return s;
}
// f.first_seen(bucket, item), for functors that have one: called for (at least) one item of every bucket a workgroup counts --
// when the bucket enters the workgroup's LDS count cache, or for every item that goes past the cache
template <class F>
GRL_DEV auto agg_first_seen(const F &f, u32 s, u64 item, int) -> decltype(f.first_seen(s, item), void()) { f.first_seen(s, item); }
template <class F>
GRL_DEV void agg_first_seen(const F &, u32, u64, long) {}
template <class F>
constexpr auto agg_is_stream(int) -> decltype(F::kStream) { return F::kStream; }
template <class F>
constexpr bool agg_is_stream(long) { return false; }
template <int SLOTS, bool AGG, class F, class A>
__global__ void __launch_bounds__(kBlock) k_for_each_agg(u64 n, u64 per_block, F f, A add) {
    __shared__ u32 c_key[AGG ? SLOTS : 1];
    __shared__ u32 c_cnt[AGG ? SLOTS : 1];
    // every wave compacts and processes its own quarter of the chunk: no workgroup barrier inside the loop, the
    // waves drift apart freely and hide each other's gather latency
    constexpr int kWaveChunk = kAggChunk / (kBlock / 64);
    constexpr bool BATCH = F::kBatch > 1;
    // STREAMING batched functors (F::kStream): a wave takes a CONTIGUOUS quarter of the block's positions, so the work items it
    // queues are consecutive work items of the whole index space -- the k-th one has ordinal f.ordinal_base(first position) + k,
    // and the item behind it in the queue is the next work item.  The functor gets both (process_batch_stream) and needs neither
    // the rank structure nor a window of the flag bits per item: three of its five loads per item (level 0 of the 10 GB build).
    constexpr bool STREAM = agg_is_stream<F>(0);
    // batched functors: the work items queue up ACROSS chunks in a ring until a full batch (64 * kBatch items) is there,
    // and the items the batch code hands back (kDeferBucket: the rare long cases) queue up in a second ring until 64 of
    // them can take the generic path together -- without this, almost every wave ran the divergent generic code for a few
    // of its lanes in every batch (level 0 of DNA: 5 % long phrases, 96 % of the waves affected, 485 VALU instructions per
    // phrase measured against ~100 in the batch code proper).
    // (the ring takes the < 64 * kBatch items a batch run leaves plus the starts of HALF a wave chunk: 255 + 256 < 512.  With
    // 1024 entries per wave the kernel held 40 KB of LDS and ran 4 workgroups per CU; at 32 KB it runs the 5 its registers allow.)
    constexpr u32 QCAP = BATCH ? 512u : (u32)kWaveChunk, DCAP = BATCH ? 512u : 1u;
    __shared__ u32 s_queue[kBlock / 64][QCAP];
    __shared__ u32 s_defer[kBlock / 64][DCAP];
    if (AGG) {
        for (int i = threadIdx.x; i < SLOTS; i += kBlock) { c_key[i] = kNoBucket; c_cnt[i] = 0; }
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    volatile u32 *queue = s_queue[w];
    volatile u32 *defer = s_defer[w];
    u64 start = (u64)blockIdx.x * per_block;
    u64 end = start + per_block < n ? start + per_block : n;
    auto count = [&](u32 s, u64 item) {
        if (s != kNoBucket) {                 // (nested, no early return: see the compiler note in engine_impl.hpp)
            if (AGG) {
                u32 h = (s * 2654435761u) >> (32 - __builtin_ctz(SLOTS));
                bool done = false;
                for (int p = 0; p < 4 && !done; p++) {
                    u32 old = atomicCAS(&c_key[h], kNoBucket, s);
                    if (old == kNoBucket || old == s) { atomicAdd(&c_cnt[h], 1u); done = true; if (old == kNoBucket) agg_first_seen(f, s, item, 0); }
                    else h = (h + 1) & (SLOTS - 1);
                }
                if (!done) { add(s, 1u); agg_first_seen(f, s, item, 0); }
            } else { add(s, 1u); agg_first_seen(f, s, item, 0); }
        }
    };
    if constexpr (BATCH) {
        u32 qh = 0, qt = 0, dh = 0, dt = 0;            // ring heads / tails (items are offsets from `start`, < 2^32)
        const u64 wspan = per_block / (kBlock / 64);   // (per_block is a multiple of kAggChunk: wave spans start on multiples of 64)
        const u64 wstart = start + (u64)w * wspan < end ? start + (u64)w * wspan : end;
        const u64 wend = wstart + wspan < end ? wstart + wspan : end;
        u64 ord0 = 0;
        if constexpr (STREAM) ord0 = wstart < wend ? f.ordinal_base(wstart) : 0;
        auto run_deferred = [&](bool all) {
            while (dt - dh >= 64u || (all && dt != dh)) {
                const u32 k = dh + (u32)lane;
                const bool v = k < dt;                  // (u32 counters never wrap: a block has < 2^32 positions)
                u32 s = kNoBucket;
                const u64 it = start + (v ? defer[k & (DCAP - 1)] : 0u);
                if (v) s = f.process(it);
                s = agg_take_claim(f, s, it, v);
                if (v) count(s, it);
                dh = dt - dh >= 64u ? dh + 64u : dt;
            }
        };
        auto run_batches = [&](bool all) {
            while (qt - qh >= 64u * F::kBatch || (all && qt != qh)) {
                u64 item[F::kBatch];
                bool valid[F::kBatch];
                u32 slot[F::kBatch];
#pragma unroll
                for (int j = 0; j < F::kBatch; j++) {
                    const u32 k = qh + (u32)j * 64 + lane;
                    valid[j] = k < qt;
                    item[j] = start + (valid[j] ? queue[k & (QCAP - 1)] : 0u);
                }
                if constexpr (STREAM) {
                    u64 next[F::kBatch], ord[F::kBatch];
#pragma unroll
                    for (int j = 0; j < F::kBatch; j++) {
                        const u32 k = qh + (u32)j * 64 + lane;
                        const bool have = k + 1u < qt;                       // (the item behind mine is in the ring)
                        next[j] = start + (have ? queue[(k + 1u) & (QCAP - 1)] : 0u);
                        if (valid[j] && !have) next[j] = f.next_item(item[j]);   // (the last one queued so far: looked up)
                        ord[j] = ord0 + (u64)k;
                    }
                    f.process_batch_stream(item, valid, slot, next, ord);
                } else
                f.process_batch(item, valid, slot);
#pragma unroll
                for (int j = 0; j < F::kBatch; j++) {
                    slot[j] = agg_take_claim(f, slot[j], item[j], valid[j]);
                    const bool df = valid[j] && slot[j] == kDeferBucket;
                    const unsigned long long m = __ballot(df);
                    if (df) defer[(dt + (u32)__popcll(m & ((1ull << lane) - 1ull))) & (DCAP - 1)] = (u32)(item[j] - start);
                    dt += (u32)__popcll(m);
                    if (valid[j] && !df) count(slot[j], item[j]);
                }
                qh = qt - qh >= 64u * F::kBatch ? qh + 64u * F::kBatch : qt;
                run_deferred(false);
            }
        };
        static_assert(!BATCH || 64 * F::kBatch + kWaveChunk / 2 <= (int)QCAP, "the item ring is too small for this batch width");
        for (u64 cbase = STREAM ? wstart : start; cbase < (STREAM ? wend : end); cbase += STREAM ? (u64)kWaveChunk : (u64)kAggChunk) {
            const u64 base = STREAM ? cbase : cbase + (u64)w * kWaveChunk;
            const u64 lim = STREAM ? wend : end;
#pragma nounroll
            for (int half = 0; half < 2; half++) {      // (a loop, not two copies of the batch code: the kernel is large already)
#pragma unroll
                for (int k = half * (kWaveChunk / 128); k < (half + 1) * (kWaveChunk / 128); k++) {
                    u64 i = base + (u64)k * 64 + lane;
                    bool st = (i < lim) && f.is_start(i);
                    unsigned long long m = __ballot(st);
                    if (st) queue[(qt + (u32)__popcll(m & ((1ull << lane) - 1ull))) & (QCAP - 1)] = (u32)(i - start);
                    qt += (u32)__popcll(m);
                }
                run_batches(false);   // leaves fewer than 64 * kBatch items: the next half chunk's <= kWaveChunk / 2 starts still fit the ring
            }
        }
        run_batches(true);
        run_deferred(true);
    } else {
        for (u64 cbase = start; cbase < end; cbase += kAggChunk) {
            const u64 base = cbase + (u64)w * kWaveChunk;
            u32 qn = 0;
#pragma unroll
            for (int k = 0; k < kWaveChunk / 64; k++) {
                u64 i = base + (u64)k * 64 + lane;
                bool st = (i < end) && f.is_start(i);
                unsigned long long m = __ballot(st);
                if (st) queue[qn + (u32)__popcll(m & ((1ull << lane) - 1ull))] = (u32)(i - base);
                qn += (u32)__popcll(m);
            }
            for (u32 q0 = 0; q0 < qn; q0 += 64) {         // (uniform trip count: the claim marks are wave-cooperative)
                const u32 q = q0 + (u32)lane;
                const bool v = q < qn;
                const u64 it = base + (v ? queue[q] : 0u);
                u32 s = kNoBucket;
                if (v) s = f.process(it);
                s = agg_take_claim(f, s, it, v);
                if (v) count(s, it);
            }
        }
    }
    if (AGG) {
        __syncthreads();
        for (int i = threadIdx.x; i < SLOTS; i += kBlock) {
            u32 c = c_cnt[i];
            if (c) add(c_key[i], c);
        }
    }
}
// ------------------------------------------ name_stream: the streaming form alone, for functors that NAME most of their work
// items by a computation on the item's own cells (level 0 of a byte text with a direct index: engine_impl.hpp, HashInsertFn).
// The same queue as k_for_each_agg's STREAM form -- a wave takes a contiguous quarter of the block's positions, the k-th item it
// queues has ordinal f.ordinal_base(first position) + k, the queue entry behind it is the next item -- and nothing else: an item
// the functor cannot name (f.stream_name returns kDeferBucket) only leaves its bit in defer_bits[] and goes through the general
// code in a launch of its own afterwards.  k_for_each_agg carries that general code inline (58 KB of instructions, 106 scalar
// registers with uniforms spilled to vector lanes, 276 vector + 206 scalar instructions per 64 items); this kernel does not.
//   f.is_start(i), f.ordinal_base(p), f.next_item(p), f.stream_load(p) -> u64 (p + 8 <= n), f.stream_name(p, cells, next) -> bucket,
//   f.stream_store(ordinal, bucket), f.first_seen(bucket, p); add(bucket, count) as in for_each_agg.
template <int SLOTS, class F, class A>
__global__ void __launch_bounds__(kBlock) k_name_stream(u64 n, u64 per_block, F f, A add, u64 *defer_bits) {
    __shared__ u32 c_key[SLOTS];
    __shared__ u32 c_cnt[SLOTS];
    constexpr int NW = kBlock / 64, B = 4;
    constexpr u32 QCAP = 512;
    __shared__ u32 s_queue[NW][QCAP];
    for (int i = threadIdx.x; i < SLOTS; i += kBlock) { c_key[i] = kNoBucket; c_cnt[i] = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    volatile u32 *queue = s_queue[w];
    const u64 start = (u64)blockIdx.x * per_block;
    const u64 end = start + per_block < n ? start + per_block : n;
    const u64 wspan = per_block / NW;                // (per_block is a multiple of kAggChunk: wave spans start on multiples of 64)
    const u64 wstart = start + (u64)w * wspan < end ? start + (u64)w * wspan : end;
    const u64 wend = wstart + wspan < end ? wstart + wspan : end;
    const u64 ord0 = wstart < wend ? f.ordinal_base(wstart) : 0;
    u32 qh = 0, qt = 0;                              // (items are offsets from wstart: a wave's span is below 2^32 positions)
    auto run = [&](bool all) {
        while (qt - qh >= 64u * B || (all && qt != qh)) {
            u32 off[B], nx[B];
            u64 cells[B];
            bool valid[B], can[B];
#pragma unroll
            for (int j = 0; j < B; j++) {
                const u32 k = qh + (u32)j * 64 + lane;
                valid[j] = k < qt;
                off[j] = valid[j] ? queue[k & (QCAP - 1)] : 0u;
                const bool have = k + 1u < qt;
                nx[j] = have ? queue[(k + 1u) & (QCAP - 1)] : 0xFFFFFFFFu;
                can[j] = valid[j] && wstart + (u64)off[j] + 8 <= n;
                cells[j] = can[j] ? f.stream_load(wstart + (u64)off[j]) : 0ull;
            }
#pragma unroll
            for (int j = 0; j < B; j++) {
                const u32 k = qh + (u32)j * 64 + lane;
                const u64 p = wstart + (u64)off[j];
                u64 next = wstart + (u64)nx[j];
                if (valid[j] && nx[j] == 0xFFFFFFFFu) next = f.next_item(p);      // (the last item queued so far)
                u32 s = kDeferBucket;
                if (can[j]) s = f.stream_name(p, cells[j], next);
                if (valid[j] && s == kDeferBucket) atomicOr(reinterpret_cast<unsigned long long *>(&defer_bits[p >> 6]), 1ull << (p & 63));
                if (valid[j] && s != kDeferBucket) {
                    f.stream_store(ord0 + (u64)k, s);
                    u32 h = (s * 2654435761u) >> (32 - __builtin_ctz(SLOTS));
                    bool done = false;
                    for (int q = 0; q < 4 && !done; q++) {
                        const u32 old = atomicCAS(&c_key[h], kNoBucket, s);
                        if (old == kNoBucket || old == s) { atomicAdd(&c_cnt[h], 1u); done = true; if (old == kNoBucket) f.first_seen(s, p); }
                        else h = (h + 1) & (SLOTS - 1);
                    }
                    if (!done) { add(s, 1u); f.first_seen(s, p); }
                }
            }
            qh = qt - qh >= 64u * B ? qh + 64u * B : qt;
        }
    };
    for (u64 cbase = wstart; cbase < wend; cbase += 256) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u64 i = cbase + (u64)k * 64 + lane;
            const bool st = (i < wend) && f.is_start(i);
            const unsigned long long m = __ballot(st);
            if (st) queue[(qt + (u32)__popcll(m & ((1ull << lane) - 1ull))) & (QCAP - 1)] = (u32)(i - wstart);
            qt += (u32)__popcll(m);
        }
        run(false);            // leaves fewer than 256 items: the next 256 positions' starts still fit the ring
    }
    run(true);
    __syncthreads();
    for (int i = threadIdx.x; i < SLOTS; i += kBlock) {
        const u32 c = c_cnt[i];
        if (c) add(c_key[i], c);
    }
}
// Workgroups per CU for the kernels that cut the index space into ONE contiguous span per workgroup (k_for_each_agg,
// k_name_stream): a multiple of what a CU holds at once.  With 8 per CU -- the figure these launches had until round 5 -- and 5 or 6
// resident, the second wave of workgroups ran at half occupancy: the level-0 naming kernel took 37.2 ms at 8 per CU, 29.6 at 6,
// 29.3 at 12 (tools/_build sweep, 10 GB build).  (GRLBWT_SPAN_BLOCKS_PER_CU overrides.)
template <class K>
inline u64 span_blocks_per_cu(K kernel, int threads) {
    static const u64 forced = dev_env("GRLBWT_SPAN_BLOCKS_PER_CU") ? (u64)atoll(dev_env("GRLBWT_SPAN_BLOCKS_PER_CU")) : 0;
    if (forced) return forced;
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, threads, 0) != hipSuccess || occ < 1) { (void)hipGetLastError(); return 8; }
    return (u64)occ * 2;
}
template <class F, class A>
inline void name_stream(u64 n, F f, A add, u64 *defer_bits, const char *name = "name_stream") {
    if (n == 0) return;
    static const u64 per_cu = span_blocks_per_cu(k_name_stream<2048, F, A>, kBlock);
    u64 blocks = (u64)rt().num_cus * per_cu;
    u64 per_block = ((n + blocks - 1) / blocks + kAggChunk - 1) / kAggChunk * kAggChunk;
    if (per_block / (kBlock / 64) >= 0xFFFFFF00ull) throw Error(-75, "name_stream: a wave's span has >= 2^32 positions");
    blocks = (n + per_block - 1) / per_block;
    prof_begin(name);
    hipLaunchKernelGGL((k_name_stream<2048, F, A>), dim3((unsigned)blocks), dim3(kBlock), 0, rt().stream, n, per_block, f, add, defer_bits);
    prof_end();
    after_launch(name);
}
template <class F, class A>
struct NoAggFn {
    F f; A add;
    GRL_DEV void operator()(u64 i) const {
        u32 s = f(i);
        if (s != kNoBucket) {
            if constexpr (F::kClaims) {
                if (f.claim_bits) {
                    const u64 cp = f.claim_pos(i);
                    if (s & kClaimBit) atomicOr(reinterpret_cast<unsigned long long *>(&f.claim_bits[cp >> 6]), 1ull << (cp & 63));
                    s &= ~kClaimBit;
                }
            }
            add(s, 1u);
            agg_first_seen(f, s, i, 0);
        }
    }
};
template <class F, class A>
inline void for_each_agg(u64 n, F f, A add, bool aggregate, const char *name = "for_each_agg") {
    if (n == 0) return;
    if (dev_env("GRLBWT_NOAGG")) { for_each(n, NoAggFn<F, A>{f, add}, name); return; }
    // (measured on the 10 GB build: the list passes of the levels above 0 -- no LDS count cache -- 11.0 -> 8.7 ms with twice the
    // resident workgroups per CU instead of 8; the level-0 kernel with the cache 40.4 -> 43.2 ms: it keeps 8)
    static const u64 per_cu_plain = span_blocks_per_cu(k_for_each_agg<2048, false, F, A>, kBlock);
    u64 blocks = (u64)rt().num_cus * (aggregate ? (u64)8 : per_cu_plain);
    u64 per_block = ((n + blocks - 1) / blocks + kAggChunk - 1) / kAggChunk * kAggChunk;
    blocks = (n + per_block - 1) / per_block;
    prof_begin(name);
    if (aggregate) hipLaunchKernelGGL((k_for_each_agg<2048, true, F, A>), dim3((unsigned)blocks), dim3(kBlock), 0, rt().stream, n, per_block, f, add);
    else hipLaunchKernelGGL((k_for_each_agg<2048, false, F, A>), dim3((unsigned)blocks), dim3(kBlock), 0, rt().stream, n, per_block, f, add);
    prof_end();
    after_launch(name);
}

// -------------------------------------------------------------------- reduce
enum class Op { Sum, Min, Max };
template <class T, Op OP>
GRL_HD T op_identity() {
    if constexpr (OP == Op::Min) return ~T(0);
    else return T(0);
}
template <class T, Op OP>
GRL_HD T op_apply(T a, T b) {
    if constexpr (OP == Op::Sum) return a + b;
    else if constexpr (OP == Op::Min) return a < b ? a : b;
    else return a > b ? a : b;
}
// two-component value for fused scans (e.g. run heads + run lengths in one pass)
template <class A, class B>
struct Pair {
    A a; B b;
    Pair() = default;                     // trivial: usable in __shared__ arrays
    GRL_HD Pair(int) : a(0), b(0) {}
    GRL_HD Pair(A a_, B b_) : a(a_), b(b_) {}
    GRL_HD Pair &operator+=(const Pair &o) { a += o.a; b += o.b; return *this; }
    GRL_HD Pair operator+(const Pair &o) const { return Pair(a + o.a, b + o.b); }
    GRL_HD Pair operator-(const Pair &o) const { return Pair(a - o.a, b - o.b); }
};
template <class T>
GRL_DEV T shfl_down_any(T v, int off) { return __shfl_down(v, off, 64); }
template <class T>
GRL_DEV T shfl_up_any(T v, int off) { return __shfl_up(v, off, 64); }
template <class A, class B>
GRL_DEV Pair<A, B> shfl_down_any(Pair<A, B> v, int off) { return Pair<A, B>(__shfl_down(v.a, off, 64), __shfl_down(v.b, off, 64)); }
template <class A, class B>
GRL_DEV Pair<A, B> shfl_up_any(Pair<A, B> v, int off) { return Pair<A, B>(__shfl_up(v.a, off, 64), __shfl_up(v.b, off, 64)); }

template <class T, Op OP>
GRL_DEV T wave_reduce(T v) {
    for (int off = 32; off > 0; off >>= 1) {
        T o = shfl_down_any(v, off);
        v = op_apply<T, OP>(v, o);
    }
    return v;
}
template <class T, Op OP, class F>
__global__ void __launch_bounds__(kBlock) k_reduce(u64 n, F f, T *partials) {
    __shared__ T s_w[4];
    T acc = op_identity<T, OP>();
    u64 stride = (u64)gridDim.x * kBlock;
    for (u64 i = (u64)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) acc = op_apply<T, OP>(acc, (T)f(i));
    acc = wave_reduce<T, OP>(acc);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        T r = s_w[0];
        for (int w = 1; w < 4; w++) r = op_apply<T, OP>(r, s_w[w]);
        partials[blockIdx.x] = r;
    }
}
template <class T, Op OP, class F>
inline T reduce(u64 n, F f, const char *name = "reduce") {
    if (n == 0) return op_identity<T, OP>();
    unsigned g = grid_for(n, kBlock * 8);
    const bool direct = sizeof(T) * g <= kResultPage;          // the partials go straight to the host page
    T *d = direct ? (T *)result_page() : (T *)dev_alloc(sizeof(T) * g);
    prof_begin(name);
    hipLaunchKernelGGL((k_reduce<T, OP, F>), dim3(g), dim3(kBlock), 0, rt().stream, n, f, d);
    prof_end();
    after_launch(name);
    std::vector<T> h(g);
    if (direct) { sync(); std::memcpy(h.data(), d, sizeof(T) * g); }
    else { d2h(h.data(), d, sizeof(T) * g); dev_free(d); }
    T r = op_identity<T, OP>();
    for (unsigned i = 0; i < g; i++) r = op_apply<T, OP>(r, h[i]);
    return r;
}
template <class T, class F>
inline T reduce_sum(u64 n, F f, const char *name = "reduce_sum") { return reduce<T, Op::Sum, F>(n, f, name); }
template <class T, class F>
inline T reduce_min(u64 n, F f, const char *name = "reduce_min") { return reduce<T, Op::Min, F>(n, f, name); }
template <class T, class F>
inline T reduce_max(u64 n, F f, const char *name = "reduce_max") { return reduce<T, Op::Max, F>(n, f, name); }

// ------------------------------------------------------------ exclusive scan
// reduce-then-scan: tile sums -> (recursive) scan of tile sums -> per-tile scan.
static constexpr int kScanItems = 8;
static constexpr int kScanTile = kBlock * kScanItems;

template <class T>
GRL_DEV T wave_incl_scan(T v) {
    int lane = threadIdx.x & 63;
    for (int off = 1; off < 64; off <<= 1) {
        T o = shfl_up_any(v, off);
        if (lane >= off) v += o;
    }
    return v;
}
// exclusive prefix of `v` over the 256 threads of the block; *block_total = sum
template <class T>
GRL_DEV T block_excl_scan(T v, T *s_w /*[4]*/, T *block_total) {
    T incl = wave_incl_scan<T>(v);
    int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 63) s_w[w] = incl;
    __syncthreads();
    T base = 0, tot = 0;
    for (int k = 0; k < 4; k++) {
        T x = s_w[k];
        if (k < w) base += x;
        tot += x;
    }
    *block_total = tot;
    return base + incl - v;
}

template <class T, class F>
__global__ void __launch_bounds__(kBlock) k_scan_tile_sums(u64 n, F in, T *tile_sums) {
    __shared__ T s_w[4];
    u64 base = (u64)blockIdx.x * kScanTile + (u64)threadIdx.x;     // striped: a sum does not care about the order
    T acc = 0;
#pragma unroll
    for (int j = 0; j < kScanItems; j++) {
        u64 i = base + (u64)j * kBlock;
        if (i < n) acc += (T)in(i);
    }
    acc = wave_reduce<T, Op::Sum>(acc);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
// tile_offsets == nullptr: single tile, offset 0.  The thread holding element n-1 stores the grand
// total to total_a / total_b when they are non-null (no separate copy kernels for scalars).
struct NoEmit { static constexpr bool kWaveEmit = false; };
template <class T, class F, class E = NoEmit>
__global__ void __launch_bounds__(kBlock) k_scan_tiles(u64 n, F in, const T *tile_offsets, T *out, T *total_a, T *total_b, E emit = E()) {
    __shared__ T s_w[4];
    __shared__ T s_x[kScanTile + kScanTile / 32];      // skewed by one slot per 32: the blocked reads spread over the banks
    const u64 tile_base = (u64)blockIdx.x * kScanTile;
    u64 base = tile_base + (u64)threadIdx.x * kScanItems;
    // the input is read striped (neighbouring lanes, neighbouring elements) and turned into the blocked arrangement
    // the scan wants through LDS; blocked global loads are kScanItems instructions of 64 addresses 32+ bytes apart
    T vs[std::is_same<E, NoEmit>::value ? 1 : kScanItems];     // striped copies, kept for the fused consumer
#pragma unroll
    for (int j = 0; j < kScanItems; j++) {
        u32 k = (u32)j * kBlock + threadIdx.x;
        u64 i = tile_base + k;
        T x = (i < n) ? (T)in(i) : T(0);
        s_x[k + (k >> 5)] = x;
        if constexpr (!std::is_same<E, NoEmit>::value) vs[j] = x;
    }
    __syncthreads();
    T v[kScanItems];
    T acc = 0;
#pragma unroll
    for (int j = 0; j < kScanItems; j++) {
        u32 k = threadIdx.x * kScanItems + j;
        v[j] = s_x[k + (k >> 5)];
        acc += v[j];
    }
    T tot;
    T ex = block_excl_scan<T>(acc, s_w, &tot) + (tile_offsets ? tile_offsets[blockIdx.x] : T(0));
    // A lane's kScanItems results are contiguous: full tiles store them as 16-byte pieces (scalar stores would be
    // kScanItems store instructions per lane, each hitting 64 different 32-byte-apart addresses).
    constexpr int kBytes = (int)sizeof(T) * kScanItems;
    const bool full = (u64)(blockIdx.x + 1) * kScanTile < n;      // uniform; the tile holding n-1 takes the scalar path
    if constexpr (!std::is_same<E, NoEmit>::value) {
        // fused consumer: emit(i, exclusive prefix, own value) instead of materialising the prefix array.  The
        // prefixes go back through LDS into the striped arrangement, so that neighbouring lanes emit neighbouring
        // elements (a compaction then writes neighbouring addresses).
        __syncthreads();                       // everybody has read its blocked inputs
#pragma unroll
        for (int j = 0; j < kScanItems; j++) {
            u32 k = threadIdx.x * kScanItems + j;
            s_x[k + (k >> 5)] = ex;
            ex = ex + v[j];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kScanItems; j++) {
            u32 k = (u32)j * kBlock + threadIdx.x;
            u64 i = tile_base + k;
            if constexpr (E::kWaveEmit) {
                // wave-cooperative consumer: called by ALL lanes of the wave (valid tells whether i is an element), so that
                // it may use ballots / shuffles over the 64 consecutive elements the wave holds
                const T e = s_x[k + (k >> 5)];
                if (i == n - 1) {
                    T nx = e + vs[j];
                    if (total_a) *total_a = nx;
                    if (total_b) *total_b = nx;
                }
                emit.wave(i, e, vs[j], i < n);
            } else if (i < n) {
                T e = s_x[k + (k >> 5)];
                if (i == n - 1) {
                    T nx = e + vs[j];
                    if (total_a) *total_a = nx;
                    if (total_b) *total_b = nx;
                }
                emit(i, e, vs[j]);
            }
        }
        return;
    }
    if (kBytes % 16 == 0 && full && ((uintptr_t)out & 15) == 0) {
        union { T t[kScanItems]; uint4 q[kBytes / 16 > 0 ? kBytes / 16 : 1]; } u;
#pragma unroll
        for (int j = 0; j < kScanItems; j++) { u.t[j] = ex; ex = ex + v[j]; }
        uint4 *dst = reinterpret_cast<uint4 *>(out + base);
#pragma unroll
        for (int k = 0; k < kBytes / 16; k++) dst[k] = u.q[k];
        return;
    }
#pragma unroll
    for (int j = 0; j < kScanItems; j++) {
        u64 i = base + j;
        T nx = ex + v[j];
        if (i < n) {
            if (i == n - 1) {
                if (total_a) *total_a = nx;
                if (total_b) *total_b = nx;
            }
            out[i] = ex;
        }
        ex = nx;
    }
}
template <class T>
struct PtrIn {
    const T *p;
    GRL_HD T operator()(u64 i) const { return p[i]; }
};

// Device-only exclusive scan: out[i] = sum_{j<i} in(j); the grand total is written to total_a and
// total_b when non-null (device memory; total_a may be out + n).  No host synchronisation and no
// scalar copies: everything is stream-ordered kernels.  `out` may alias the array `in` reads (each
// tile is read before it is written, tiles are disjoint) -- but then total_a/total_b must not
// alias an element `in` still reads.
template <class T, class F>
inline void exclusive_scan_async(u64 n, F in, T *out, T *total_a, T *total_b = nullptr, const char *name = "scan") {
    if (n == 0) {
        if (total_a) dev_memset(total_a, 0, sizeof(T));
        if (total_b) dev_memset(total_b, 0, sizeof(T));
        return;
    }
    u64 tiles = (n + kScanTile - 1) / kScanTile;
    if (tiles == 1) {
        prof_begin(name);
        hipLaunchKernelGGL((k_scan_tiles<T, F>), dim3(1), dim3(kBlock), 0, rt().stream, n, in, (const T *)nullptr, out, total_a, total_b);
        prof_end();
        after_launch(name);
        return;
    }
    T *sums = (T *)dev_alloc(sizeof(T) * (tiles + 1));
    prof_begin(name);
    hipLaunchKernelGGL((k_scan_tile_sums<T, F>), dim3((unsigned)tiles), dim3(kBlock), 0, rt().stream, n, in, sums);
    prof_end();
    after_launch(name);
    exclusive_scan_async<T, PtrIn<T>>(tiles, PtrIn<T>{sums}, sums, nullptr, nullptr, name);
    prof_begin(name);
    hipLaunchKernelGGL((k_scan_tiles<T, F>), dim3((unsigned)tiles), dim3(kBlock), 0, rt().stream, n, in, (const T *)sums, out, total_a, total_b);
    prof_end();
    after_launch(name);
    dev_free(sums);      // stream-ordered reuse (pool)
}

// (A single-pass form -- tiles ordered by a ticket counter, sums and inclusive prefixes published in self-validating
// 64-bit words, wave-wide look-back -- was built and measured on the 10 GB build: pass C's TAKE-prefix scan 53 vs 51 ms,
// the run merges 64 vs 48 ms.  With 2048-element tiles the descriptor traffic and the publication chain across the
// 8 XCDs cost more than the second read of the inputs saves; reduce-then-scan stays.)
// Scan fused with its consumer: emit(i, sum_{j<i} in(j), in(i)) is called for every i instead of storing the prefix
// array (stream compaction and "scan then scatter at heads" patterns).  Returns the grand total (host value, one sync).
template <class T, class F, class E>
inline void exclusive_scan_emit_async(u64 n, F in, E emit, T *tot /*device-visible, may be null*/, const char *name = "scan");
template <class T, class F, class E>
inline T exclusive_scan_emit(u64 n, F in, E emit, const char *name = "scan") {
    if (n == 0) return T(0);
    T *tot = (T *)result_page();
    exclusive_scan_emit_async<T, F, E>(n, in, emit, tot, name);
    sync();
    T total;
    std::memcpy(&total, tot, sizeof(T));
    return total;
}
// the same without the host total (no synchronisation)
template <class T, class F, class E>
inline void exclusive_scan_emit_nosync(u64 n, F in, E emit, const char *name = "scan") { exclusive_scan_emit_async<T, F, E>(n, in, emit, (T *)nullptr, name); }
template <class T, class F, class E>
inline void exclusive_scan_emit_async(u64 n, F in, E emit, T *tot, const char *name) {
    if (n == 0) return;
    u64 tiles = (n + kScanTile - 1) / kScanTile;
    if (tiles == 1) {
        prof_begin(name);
        hipLaunchKernelGGL((k_scan_tiles<T, F, E>), dim3(1), dim3(kBlock), 0, rt().stream, n, in, (const T *)nullptr, (T *)nullptr, tot, (T *)nullptr, emit);
        prof_end();
        after_launch(name);
    } else {
        T *sums = (T *)dev_alloc(sizeof(T) * (tiles + 1));
        prof_begin(name);
        hipLaunchKernelGGL((k_scan_tile_sums<T, F>), dim3((unsigned)tiles), dim3(kBlock), 0, rt().stream, n, in, sums);
        prof_end();
        after_launch(name);
        exclusive_scan_async<T, PtrIn<T>>(tiles, PtrIn<T>{sums}, sums, nullptr, nullptr, name);
        prof_begin(name);
        hipLaunchKernelGGL((k_scan_tiles<T, F, E>), dim3((unsigned)tiles), dim3(kBlock), 0, rt().stream, n, in, (const T *)sums, (T *)nullptr, tot, (T *)nullptr, emit);
        prof_end();
        after_launch(name);
        dev_free(sums);
    }
}

// out[i] = sum_{j<i} in(j) for i in [0,n); returns the grand total (host value, one sync).
// If store_total_at_n, also stores the total at out[n].
template <class T, class F>
inline T exclusive_scan(u64 n, F in, T *out, bool store_total_at_n = false, const char *name = "scan") {
    if (n == 0) {                               // (nothing is launched: the total is zero)
        if (store_total_at_n) dev_memset(out, 0, sizeof(T));
        return T(0);
    }
    T *tot = (T *)result_page();
    exclusive_scan_async<T, F>(n, in, out, tot, store_total_at_n ? out + n : nullptr, name);
    sync();
    T total;
    std::memcpy(&total, tot, sizeof(T));
    return total;
}

// same without the host total (no synchronisation)
template <class T, class F>
inline void exclusive_scan_nosync(u64 n, F in, T *out, bool store_total_at_n = false, const char *name = "scan") {
    exclusive_scan_async<T, F>(n, in, out, nullptr, store_total_at_n ? out + n : nullptr, name);
}

// ------------------------------------------------------------- byte histogram
// Small alphabets (DNA: five byte values) send all 64 lanes of a wave to a handful of LDS addresses: every wave keeps
// kHistCopies copies of the table, a lane takes copy (lane mod kHistCopies), and the copies sit 257 words apart so that the
// same bin of different copies falls into different banks (one table per wave ran at 0.9 TB/s on 10 GB of reads).
static constexpr int kHistCopies = 8;
__global__ void __launch_bounds__(kBlock) k_byte_hist(const u8 *p, u64 n, u64 *hist) {
    __shared__ u32 s_h[4][kHistCopies][257];
    for (int i = threadIdx.x; i < 4 * kHistCopies * 257; i += kBlock) (&s_h[0][0][0])[i] = 0;
    __syncthreads();
    u32 *h = s_h[threadIdx.x >> 6][threadIdx.x & (kHistCopies - 1)];
    u64 stride = (u64)gridDim.x * kBlock * 16;
    for (u64 i = ((u64)blockIdx.x * kBlock + threadIdx.x) * 16; i < n; i += stride) {
        if (i + 16 <= n) {
            uint4 v = *reinterpret_cast<const uint4 *>(p + i);   // 16 B per lane, base is 16-B aligned
            u32 ws[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                atomicAdd(&h[ws[k] & 255], 1u);
                atomicAdd(&h[(ws[k] >> 8) & 255], 1u);
                atomicAdd(&h[(ws[k] >> 16) & 255], 1u);
                atomicAdd(&h[ws[k] >> 24], 1u);
            }
        } else {
            for (u64 j = i; j < n; j++) atomicAdd(&h[p[j]], 1u);
        }
    }
    __syncthreads();
    u32 t = 0;
#pragma unroll
    for (int w = 0; w < 4; w++)
#pragma unroll
        for (int c = 0; c < kHistCopies; c++) t += s_h[w][c][threadIdx.x];
    if (t) atomicAdd(reinterpret_cast<unsigned long long *>(&hist[threadIdx.x]), (unsigned long long)t);
}
// adds the byte frequencies of p[0..n) to the device table d_hist[256] (stream-ordered, no synchronisation)
inline void byte_histogram_accumulate(const u8 *p, u64 n, u64 *d_hist) {
    if (!n) return;
    prof_begin("byte_hist");
    hipLaunchKernelGGL(k_byte_hist, dim3(grid_for(n, kBlock * 16 * 4)), dim3(kBlock), 0, rt().stream, p, n, d_hist);
    prof_end();
    after_launch("byte_hist");
}
// hist_host[256] = byte frequencies of p[0..n) (p must be 16-byte aligned)
inline void byte_histogram(const u8 *p, u64 n, u64 *hist_host) {
    u64 *d = (u64 *)dev_alloc(256 * 8);
    dev_memset(d, 0, 256 * 8);
    if (n) {
        prof_begin("byte_hist");
        hipLaunchKernelGGL(k_byte_hist, dim3(grid_for(n, kBlock * 16 * 4)), dim3(kBlock), 0, rt().stream, p, n, d);
        prof_end();
        after_launch("byte_hist");
    }
    d2h(hist_host, d, 256 * 8);
    dev_free(d);
}

// ------------------------------------------------------------------ radix sort
// Stable LSD radix sort of (key, value) pairs, 8 bits per pass.
//   pass = k_rs_hist (per-tile digit histogram in LDS)  -> counts[tile][digit]
//        + k_rs_chunk_sums / scan of the chunk table / k_rs_tile_offsets -> offsets[tile][digit] (digit-major order)
//        + k_rs_scatter (in-tile stable ranking with wave64 ballots, tile permuted in LDS, linear write-out)
// (Measured on MI355X, 101 MB reads workload, all scatter launches of one build: direct per-key stores with one
// barrier set per 256 keys 7.6 ms, per 1024 keys 5.9 ms, per 2048 keys 5.6 ms -- bound by the rate of 4-8 byte
// store requests; the LDS-staged form below 3.8 ms.  One wave per 1024-key tile without workgroup barriers was
// 1.3x slower than the first of those.)
static constexpr int kRsItems = 16;                 // keys per lane (24 and 32 measured slower: registers, LDS)
static constexpr int kRsTile = kBlock * kRsItems;   // 4096 keys per workgroup of 256 threads (the plain sorts below 2^22 keys: sort_pairs)

// The lanes of the wave whose digit equals mine (valid lanes only): how many of them sit below me, and how many there are.
// The mismatch mask is the OR over the digit bits of ballot ^ (my bit ? ~0 : 0): two vector instructions per bit and half.
// (BATCH: all bit spreads, then all ballots, then the folds.  A vector compare that writes a scalar pair and the vector
// instruction that reads it need two instructions between them or the compiler pads with s_nop: 199 of them in the histogram
// kernel, 30 in batch order.  The ranking kernels keep the bit-by-bit order: there the compiler finds that schedule itself,
// and written in batch order it hoists the ballots of all sixteen rows -- scalar registers spilled to vector lanes, 250 bytes
// of scratch per lane.)
template <int DB, bool BATCH = false>
GRL_DEV void wave_match(u32 d, bool valid, u32 &below, u32 &count) {
    u32 mlo = 0, mhi = 0;
    if constexpr (BATCH) {
        u32 e[DB];
        unsigned long long m[DB];
#pragma unroll
        for (int b = 0; b < DB; b++) asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(e[b]) : "v"(d), "n"(b));
#pragma unroll
        for (int b = 0; b < DB; b++) m[b] = __ballot(e[b] != 0u);
#pragma unroll
        for (int b = 0; b < DB; b++) {
            mlo |= (u32)m[b] ^ e[b];
            mhi |= (u32)(m[b] >> 32) ^ e[b];
        }
    } else {
#pragma unroll
        for (int b = 0; b < DB; b++) {
            u32 e;                                                    // my bit b, spread over the word (0 or ~0): ONE instruction
            asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(e) : "v"(d), "n"(b));     // (the compiler turns the builtin into shift + shift)
            const unsigned long long m = __ballot(e != 0u);
            mlo |= (u32)m ^ e;
            mhi |= (u32)(m >> 32) ^ e;
        }
    }
    const unsigned long long vm = __ballot(valid);
    const u32 plo = (u32)vm & ~mlo, phi = (u32)(vm >> 32) & ~mhi;
    below = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
    count = (u32)__popc(plo) + (u32)__popc(phi);
}
// Per-tile digit counts.  Per row of 64 keys the wave finds the lanes of equal digit (wave_match) and the FIRST lane of every
// digit adds the digit's count to the wave's LDS table: one conflict-free LDS add per distinct digit, so the cost does not
// depend on how skewed the digits are (one LDS atomic per key serialises on a 5-symbol alphabet), any digit width.
// (Rounds 1-4: lane l owned the bins l, l+64, l+128, l+192 and built their member masks from six ballots AND-ed per lane plus
// two more -- ~64 vector instructions per row against ~45 here; digits above 8 bits took one LDS atomic per key.)
// (SITE: a tag that only names the instantiation -- the induction's bucket split gets kernels of its own in profiler
// output, apart from the other keys-only 64-bit sorts; TB = threads per workgroup: the tile is TB x kRsItems keys, 4096 at
// 256 threads, 8192 at 512; hist and scatter of a pass agree)
// (MIX: the digit is taken from key x kMixMul instead of the key -- a record sort that groups by a hash of the key without
// a hash array of its own: RecSort)
static constexpr u64 kMixMul = 0x9E3779B97F4A7C15ull;
template <bool MIX, class K>
GRL_DEV u32 rs_digit(K k, int shift, u32 dmask) {
    if constexpr (MIX) return (u32)(((u64)k * kMixMul) >> shift) & dmask;
    else return (u32)(k >> shift) & dmask;
}
template <class K, int SITE = 0, int DB = 8, int TB = kBlock, bool MIX = false>
__global__ void __launch_bounds__(TB) k_rs_hist(const K *keys, u64 n, int shift, u32 dmask, u32 *counts, u32 tiles) {
    constexpr int TILE = TB * kRsItems, NB = 1 << DB;
    __shared__ u32 s_h[TB / 64][NB];
    for (int i = threadIdx.x; i < (TB / 64) * NB; i += TB) (&s_h[0][0])[i] = 0;
    u32 *h = s_h[threadIdx.x >> 6];
    const u64 base = (u64)blockIdx.x * TILE;
    K k[kRsItems];
    const bool full = base + TILE <= n;
    constexpr int per = 16 / (int)sizeof(K);            // keys per 16-byte load (a count does not care which lane sees a key)
    struct alignas(16) Vec { K v[per]; };
    const bool vec = full && ((uintptr_t)keys & 15) == 0;
    if (vec) {
#pragma unroll
        for (int j = 0; j < kRsItems / per; j++) {
            Vec x = *reinterpret_cast<const Vec *>(keys + base + ((u64)j * TB + threadIdx.x) * per);
#pragma unroll
            for (int e = 0; e < per; e++) k[j * per + e] = x.v[e];
        }
    } else {
#pragma unroll
        for (int r = 0; r < kRsItems; r++) {
            u64 i = base + (u64)r * TB + threadIdx.x;
            k[r] = (i < n) ? keys[i] : K(0);
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kRsItems; r++) {
        const u64 i = base + (u64)r * TB + threadIdx.x;      // only meaningful on the guarded path
        const u32 d = rs_digit<MIX>(k[r], shift, dmask);
        const bool valid = vec || i < n;
        u32 below, count;
        wave_match<DB, true>(d, valid, below, count);
        if (valid && below == 0u) (void)__hip_atomic_fetch_add(&h[d], count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < NB; d += TB) {
        u32 tot = 0;
#pragma unroll
        for (int q = 0; q < TB / 64; q++) tot += s_h[q][d];
        counts[(u64)blockIdx.x * NB + d] = tot;         // tile-major: one coalesced row per tile
    }
}

// In-wave stable ranking of ROWS rows of 64 digits (the shared body of k_rs_scatter, k_xs_scatter and k_rs_unscatter):
// idx[q] = number of keys of the same digit in front of key (row q, this lane) among this wave's rows 0..q, `cnt` = the
// wave's own NB running counters in LDS (zeroed by the caller, barrier behind the zeroing).  Row q of a lane is valid when
// t0 + 64 q < limit; rows at and behind `nrows` (wave-uniform) hold nothing.
// Three phases instead of one loop (round 5).  The one-loop form read and rewrote the wave's counter through LDS and
// broadcast the old value with a shuffle ROW BY ROW -- sixteen dependent LDS round trips per wave, each behind the row's
// eight ballots.  Here (A) the match masks of all rows are computed first (wave_match: pure vector/scalar work, rows independent), (B) every row issues ONE plain LDS read of its digit's counter by all its lanes (equal addresses broadcast) and
// ONE no-return LDS add by the first lane of every digit -- the add does not depend on the read, LDS operations of one
// wave execute in issue order, so the 2 x ROWS operations go out back to back and are waited for once -- and (C) adds.
// (dig(q): the digit of row q -- an array read, or recomputed from the key where registers are short)
// (PHASED = false: row by row -- match, LDS read, LDS add, sum -- with nothing kept per row but idx: for k_xs_scatter, whose
// four interleaved chain walks leave no registers for a second per-row array (phased there: 22.3 -> 26.1 ms at level 0 of the
// 10 GB build, spills in the ranking loop); one LDS round trip per row instead of the three of rounds 1-4)
// (CT = u16: the counters of a wide digit at 16 waves per workgroup -- 16 x 512 words do not fit beside a 16384-key tile, halves do:
// a wave holds at most 1024 keys.  The halves are read and added to through the 32-bit word that holds them -- one address for the
// load and the add, so their order is the program's.)
template <class CT>
GRL_DEV u32 rank_cnt_load(CT *cnt, u32 d) {
    if constexpr (sizeof(CT) == 4) return __hip_atomic_load(&cnt[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    else return (__hip_atomic_load(reinterpret_cast<u32 *>(cnt) + (d >> 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) >> (16u * (d & 1u))) & 0xFFFFu;
}
template <class CT>
GRL_DEV void rank_cnt_add(CT *cnt, u32 d, u32 v) {
    if constexpr (sizeof(CT) == 4) (void)__hip_atomic_fetch_add(&cnt[d], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    else (void)__hip_atomic_fetch_add(reinterpret_cast<u32 *>(cnt) + (d >> 1), v << (16u * (d & 1u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
template <int DB, int ROWS, bool PHASED = true, class CT = u32, class DIG>
GRL_DEV void wave_rank(DIG dig, u32 t0, u32 limit, u32 nrows, CT *cnt, u32 (&idx)[ROWS]) {
    if constexpr (!PHASED) {
#pragma unroll
        for (int q = 0; q < ROWS; q++) {
            idx[q] = 0;
            if ((u32)q < nrows) {
                const bool valid = t0 + 64u * (u32)q < limit;
                const u32 d = dig(q);
                u32 mlo = 0, mhi = 0;
#pragma unroll
                for (int b = 0; b < DB; b++) {
                    u32 e;
                    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(e) : "v"(d), "n"(b));
                    const unsigned long long m = __ballot(e != 0u);
                    mlo |= (u32)m ^ e;
                    mhi |= (u32)(m >> 32) ^ e;
                }
                const unsigned long long vm = __ballot(valid);
                const u32 plo = (u32)vm & ~mlo, phi = (u32)(vm >> 32) & ~mhi;
                const u32 below = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
                if (valid) {
                    const u32 old = rank_cnt_load<CT>(cnt, d);
                    if (below == 0u) rank_cnt_add<CT>(cnt, d, (u32)__popc(plo) + (u32)__popc(phi));
                    idx[q] = old + below;
                }
            }
        }
        return;
    }
    u32 info[ROWS];                       // keys of my digit in front of me in the row | all of them << 8 | valid << 16
#pragma unroll
    for (int q = 0; q < ROWS; q++) {
        info[q] = 0;
        if ((u32)q < nrows) {
            const bool valid = t0 + 64u * (u32)q < limit;
            // (wave_match spelled out, bit by bit: through the function the compiler schedules the sixteen rows as ONE block --
            // ballots of all rows live at once, scalar registers spilled to vector lanes, 192 vector registers; here the
            // select on `valid` below leaves a branch per row and the rows stay apart: 100 registers, no spills)
            const u32 d = dig(q);
            u32 mlo = 0, mhi = 0;
#pragma unroll
            for (int b = 0; b < DB; b++) {
                u32 e;
                asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(e) : "v"(d), "n"(b));
                const unsigned long long m = __ballot(e != 0u);
                mlo |= (u32)m ^ e;
                mhi |= (u32)(m >> 32) ^ e;
            }
            const unsigned long long vm = __ballot(valid);
            const u32 plo = (u32)vm & ~mlo, phi = (u32)(vm >> 32) & ~mhi;
            const u32 below = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
            const u32 count = (u32)__popc(plo) + (u32)__popc(phi);
            info[q] = valid ? (below | count << 8 | 1u << 16) : 0u;
        }
    }
#pragma unroll
    for (int q = 0; q < ROWS; q++) {
        idx[q] = 0;
        if ((u32)q < nrows) {
            if (info[q] >> 16) {
                const u32 d = dig(q);
                idx[q] = rank_cnt_load<CT>(cnt, d);
                if ((info[q] & 0xFFu) == 0u) rank_cnt_add<CT>(cnt, d, (info[q] >> 8) & 0xFFu);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < ROWS; q++) idx[q] += info[q] & 0xFFu;
}

// Global write positions from the tile-major counts.  The order of a stable pass is digit-major (all tiles of digit
// 0, then digit 1, ...); storing counts/offsets in that order makes every tile's row 256 accesses `tiles` words apart
// (measured: the histogram kernel ran at the rate of its scattered 4-byte stores, 10 G/s, not of its key reads).
// Instead: column sums over chunks of kRsChunk tiles -> digit-major scan of the small chunk table -> running column
// prefix inside each chunk, all with coalesced rows.
static constexpr int kRsChunk = 32;
template <int NB>
__global__ void __launch_bounds__(kBlock) k_rs_chunk_sums(const u32 *counts, u32 tiles, u32 *chunk_sums) {
    u32 t0 = blockIdx.x * kRsChunk, t1 = t0 + kRsChunk < tiles ? t0 + kRsChunk : tiles;
    for (int d = threadIdx.x; d < NB; d += kBlock) {
        u32 acc = 0;
#pragma unroll 8
        for (u32 t = t0; t < t1; t++) acc += counts[(u64)t * NB + d];
        chunk_sums[(u64)blockIdx.x * NB + d] = acc;
    }
}
struct RsChunkIn {       // chunk table read in digit-major order
    const u32 *c; u32 chunks; u32 nb;
    GRL_HD u64 operator()(u64 i) const { u64 d = i / chunks, k = i % chunks; return (u64)c[k * nb + d]; }
};
template <int NB>
__global__ void __launch_bounds__(kBlock) k_rs_tile_offsets(const u32 *counts, const u64 *chunk_off /*[NB][chunks]*/, u32 tiles,
                                                            u32 chunks, u64 *offsets /*[tiles][NB]*/) {
    u32 t0 = blockIdx.x * kRsChunk, t1 = t0 + kRsChunk < tiles ? t0 + kRsChunk : tiles;
    for (int d = threadIdx.x; d < NB; d += kBlock) {
        u64 run = chunk_off[(u64)d * chunks + blockIdx.x];
#pragma unroll 8
        for (u32 t = t0; t < t1; t++) {
            u32 c = counts[(u64)t * NB + d];
            offsets[(u64)t * NB + d] = run;
            run += c;
        }
    }
}
// offsets[tile][digit] (exclusive, digit-major order) from counts[tile][digit]; *total = number of counted keys (optional)
template <int NB>
inline void rs_offsets(const u32 *counts, u32 tiles, u32 *chunk_sums, u64 *chunk_off, u64 *offsets, u64 *total, const char *name) {
    u32 chunks = (tiles + kRsChunk - 1) / kRsChunk;
    hipLaunchKernelGGL(k_rs_chunk_sums<NB>, dim3(chunks), dim3(kBlock), 0, rt().stream, counts, tiles, chunk_sums);
    after_launch(name);
    exclusive_scan_async<u64, RsChunkIn>((u64)NB * chunks, RsChunkIn{chunk_sums, chunks, (u32)NB}, chunk_off, total, nullptr, name);
    hipLaunchKernelGGL(k_rs_tile_offsets<NB>, dim3(chunks), dim3(kBlock), 0, rt().stream, counts, chunk_off, tiles, chunks, offsets);
    after_launch(name);
}

// The tile (kBlock x kRsKeys keys, wave w owns a contiguous quarter) is ranked with per-wave running digit
// counters (no workgroup barrier inside the ranking loop), permuted into digit order in LDS, and written out
// linearly, so that neighbouring lanes store to neighbouring addresses of the same digit run.  Values are loaded
// once the keys' registers are free; those loads overlap the key write-out.
static constexpr int kRsKeys = kRsTile / kBlock;
struct NoVal { unsigned char unused; };      // keys-only sort: no value arrays are read or written
// (DB = digit bits, 8-10: thread t owns the BPT = 2^DB / 256 neighbouring bins [t * BPT, (t + 1) * BPT) in the offset phase)
// (second launch bound: 4 waves per SIMD = 4 / 2 / 1 workgroups of 256 / 512 / 1024 threads per CU, what the LDS tile leaves room for)
// (Values moved directly instead of staged in LDS -- 16384-record tiles for 16-byte values -- were measured slower in round 5:
// the 16-byte stores of a wave then go to 64 different lines; round 6 split the record into two staged words instead: RecSort.)
// (rank_out: where every key went INSIDE its tile's sorted order, in input order -- 2 bytes per key: the way back of a RecSort pass
// reads it instead of ranking again)
template <class K, class V, int SITE = 0, int DB = 8, int TB = kBlock, bool MIX = false>
__global__ void __launch_bounds__(TB)
    k_rs_scatter(const K *keys_in, const V *vals_in, K *keys_out, V *vals_out, u64 n, int shift, u32 dmask,
                        const u64 *offsets /*[tiles][NB] exclusive*/, u32 tiles, u16 *rank_out = nullptr) {
    constexpr int NB = 1 << DB, BPT = NB / TB > 0 ? NB / TB : 1, NW = TB / 64, TILE = TB * kRsItems;
    constexpr bool STAGED = !std::is_same<V, NoVal>::value;
    constexpr int EB = (STAGED && sizeof(V) > sizeof(K)) ? sizeof(V) : sizeof(K);
    typedef typename std::conditional<(NW * NB > 4096), u16, u32>::type CT;      // (per-wave counters: halves when the words would not fit, see wave_rank)
    __shared__ __attribute__((aligned(16))) unsigned char s_buf[TILE * EB];
    __shared__ __attribute__((aligned(4))) CT s_cnt[NW][NB];      // per wave: running count, then exclusive base, of each digit
    __shared__ u64 s_gbase[NB];       // global position of tile-local index 0 of the digit's run (may wrap; mod 2^64)
    __shared__ u32 s_wsum[NW];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < NW * NB; i += TB) (&s_cnt[0][0])[i] = 0;
    const u64 base = (u64)blockIdx.x * TILE;
    const u64 left = n - base;
    const u32 tile_n = left < (u64)TILE ? (u32)left : (u32)TILE;
    const u32 wbase = (u32)w * (64 * kRsKeys);
    K key[kRsKeys];
    V val[STAGED ? kRsKeys : 1];
    u32 idx[kRsKeys];
#pragma unroll
    for (int q = 0; q < kRsKeys; q++) {
        u32 t = wbase + q * 64 + lane;
        bool valid = t < tile_n;
        key[q] = valid ? keys_in[base + t] : K(0);
    }
    __syncthreads();
    u32 dig[kRsKeys];
#pragma unroll
    for (int q = 0; q < kRsKeys; q++) dig[q] = rs_digit<MIX>(key[q], shift, dmask);
    wave_rank<DB, kRsKeys, true, CT>([&](int q) { return dig[q]; }, wbase + (u32)lane, tile_n, (u32)kRsKeys, &s_cnt[w][0], idx);
    __syncthreads();
    {   // thread t, bins [t*BPT, (t+1)*BPT): wave bases, tile-local digit starts, global bases (threads behind the last bin idle)
        u32 cw[BPT][NW], tt[BPT], sum = 0;
        const bool has = (int)threadIdx.x * BPT < NB;
#pragma unroll
        for (int e = 0; e < BPT; e++) {
            const int d = threadIdx.x * BPT + e;
            tt[e] = 0;
#pragma unroll
            for (int k = 0; k < NW; k++) { cw[e][k] = has ? (u32)s_cnt[k][d] : 0u; tt[e] += cw[e][k]; }
            sum += tt[e];
        }
        u32 incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            u32 o = (u32)__shfl_up((int)incl, off);
            if (lane >= off) incl += o;
        }
        if (lane == 63) s_wsum[w] = incl;
        __syncthreads();
        u32 start = incl - sum;
        for (int k = 0; k < w; k++) start += s_wsum[k];
        if (has) {
#pragma unroll
            for (int e = 0; e < BPT; e++) {
                const int d = threadIdx.x * BPT + e;
                u32 run = start;
#pragma unroll
                for (int k = 0; k < NW; k++) { s_cnt[k][d] = (CT)run; run += cw[e][k]; }
                s_gbase[d] = offsets[(u64)blockIdx.x * NB + d] - (u64)start;
                start += tt[e];
            }
        }
    }
    __syncthreads();
    K *kb = (K *)s_buf;
#pragma unroll
    for (int q = 0; q < kRsKeys; q++) {
        u32 t = wbase + q * 64 + lane;
        if (t < tile_n) {
            idx[q] += (u32)s_cnt[w][dig[q]];
            kb[idx[q]] = key[q];
        }
    }
    if constexpr (STAGED) {
#pragma unroll
        for (int q = 0; q < kRsKeys; q++) {
            u32 t = wbase + q * 64 + lane;
            val[q] = t < tile_n ? vals_in[base + t] : V(0);
        }
    }
    __syncthreads();
    u32 dpack[(kRsKeys + 1) / 2];            // the digit of every key this thread writes out (16 bits each): the values follow them
#pragma unroll
    for (int j = 0; j < (kRsKeys + 1) / 2; j++) dpack[j] = 0;
#pragma unroll
    for (int j = 0; j < kRsKeys; j++) {
        u32 t = (u32)j * TB + threadIdx.x;
        if (t < tile_n) {
            K k = kb[t];
            u32 d = rs_digit<MIX>(k, shift, dmask);
            dpack[j >> 1] |= d << (16 * (j & 1));
            keys_out[s_gbase[d] + t] = k;
        }
    }
    if (rank_out) {          // (behind the keys' write-out: in front of it these stores cost the level-1 passes of the 10 GB build 1.5-2 ms)
#pragma unroll
        for (int q = 0; q < kRsKeys; q++) { const u32 t = wbase + q * 64 + lane; if (t < tile_n) rank_out[base + t] = (u16)idx[q]; }
    }
    if constexpr (STAGED) {
        __syncthreads();
        V *vb = (V *)s_buf;
#pragma unroll
        for (int q = 0; q < kRsKeys; q++) {
            u32 t = wbase + q * 64 + lane;
            if (t < tile_n) vb[idx[q]] = val[q];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kRsKeys; j++) {
            u32 t = (u32)j * TB + threadIdx.x;
            if (t < tile_n) {
                u32 d = (dpack[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
                vals_out[s_gbase[d] + t] = vb[t];
            }
        }
    }
}

// Digit plan of an LSD sort over `bits` key bits: the fewest passes with digits of at most rs_max_digit() bits, widths as equal
// as possible (at 9: 51 bits = 9,9,9,8,8,8; 54 = 6 x 9; 18 = 9,9).  GRLBWT_SORT_DIGIT=8|9|10 sets the widest digit; the default
// stays 8: measured on the 10 GB build (profiles/r03), 9- and 10-bit digits save a pass over the 51-56-bit suffix keys and the
// 18-19-bit bucket residues but every pass gets slower by more than that -- a 4096-key tile then leaves 8 or 4 keys per bin,
// i.e. 32-64-byte runs of keys and 16-32-byte runs of values at the write front: suffix_sort0 at level 2, 7 passes x 5.5 ms
// vs 6 x 7.2 ms; whole build 1005 / 1010 / 1011 ms at 8 / 9 / 10 bits.
inline int rs_max_digit() {
    static const int d = [] { const char *e = dev_env("GRLBWT_SORT_DIGIT"); int v = e ? atoi(e) : 8; return v < 8 ? 8 : (v > 10 ? 10 : v); }();
    return d;
}
// (a caller may widen the digits of ONE sort: expand_sort takes 9-bit digits for a bucket split whose pass count that lowers)
inline int &rs_digit_override() { static int d = 0; return d; }
inline int rs_plan(int bits, int *widths /*[>= bits/8 + 1]*/) {
    const int maxd = rs_digit_override() ? rs_digit_override() : rs_max_digit();
    const int passes = (bits + maxd - 1) / maxd;
    if (bits <= 8) { widths[0] = bits; return 1; }
    const int lo = bits / passes, extra = bits % passes;
    for (int p = 0; p < passes; p++) widths[p] = lo + (p < extra ? 1 : 0);
    return passes;
}
inline int sort_keys_fwd(u64 *a, u64 *b, u64 n, int begin_bit, int end_bit, const char *name);
inline int sort_keys_fwd(u32 *a, u32 *b, u64 n, int begin_bit, int end_bit, const char *name);
// ------------------------------------------------------- expand + multi-split
// Stable multi-split of GENERATED keys.  Item i (0 <= i < n) walks a chain through a table of packed node records and
// drops one u64 key per step; the result is the sequence of all keys ordered by key bits [0, bits), stable with respect
// to (item, step).  (Induction pass B: item = run of BWT_{r+1}, node = metasymbol, key = hocc cell, sort bits = bucket.)
// The generator GEN describes the walk:
//     u32  start(i)            first node of item i            u64  node(u)        packed record of node u
//     bool owns(rec)           the start node drops a key      bool more(rec)      the chain goes on behind this node
//     u32  next(rec)           the node behind it              u64  item_bits(i)   key bits shared by all keys of item i
//     u64  key_own(u, ib)      key dropped for the start node  u64  key_step(rec, b, ib)  key dropped when stepping from rec to b
//     void finish(i, rec)      side effect at the end of the walk (rec = record of the last node)
// and the low bits of a key are its sort bits (key_own(u, 0) / key_step(rec, b, 0) suffice for counting).
//
// The generation is FUSED with the first radix pass: the keys are never written in generation order.
//   k_xs_count    walks every item once: per-tile digit histogram of the first digit (LDS atomics on per-wave copies),
//                 the number of keys of every item (one byte), the largest such number
//   rs_offsets    digit-major exclusive offsets per (tile, digit); their grand total is the number of keys E
//   k_xs_scatter  per walk of 1024 items (4 per thread, their 4 chains walked INTERLEAVED: the walk is a chain of
//                 dependent gathers, and 4 independent chains per lane is what hides their latency at the occupancy
//                 the LDS staging leaves): key counts -> block scan -> keys straight into an LDS window in sequence
//                 order; when the next walk no longer fits the window -> the in-tile stable ranking of k_rs_scatter
//                 (wave64 ballots, per-wave running counters, in-place LDS permutation) -> linear write-out at the
//                 tile's running digit offsets
// Traffic per key: 8 B written by the fused pass (+ the items' own arrays), then 8 B hist + 16 B scatter per further
// pass -- against 8 B expand + 24 B per pass for expand-then-sort.  The first digit takes 9 bits when that saves a pass.
// (IPT items per thread and walk, TI items per tile -- a multiple of 256 x IPT.  SEVERAL walks fill one ranking window (round 6): a walk
// of 1024 runs drops ~1600 keys at level 0 of the 10 GB build, so ranking after every walk filled 40 % of the window and paid the five
// barriers of a ranking round per 1600 keys.  Measured on the 10 GB build: levels 1-3 13.9 / 6.7 / 3.1 -> 11.9 / 5.7 / 2.6 ms, level 0
// 18.8 -> 19.2-20 ms -- and 16.2 ms at four workgroups per CU, which the kernel's registers allow now that nothing spills (153-161 of
// 170 for 8-byte cells, 137 -> 128 for 4-byte ones; rounds 3-5: 168 + ~100 bytes of scratch).  Walks of 768 items (IPT 3, six per
// tile) fill the window better at levels 1-2 and were no faster: 19.4 / 13.1 / 5.6 ms.)
static constexpr int kXsWin = 4096;                       // keys staged and ranked per round (16 rows of 64 per wave)
struct XsPlan {
    bool ok = false;         // false: the fused path does not apply (an item with more than 32 keys); E is still valid
    u64 n = 0, E = 0;
    u32 maxc = 0, tiles = 0;
    int bits = 0, db = 8;    // sort bits, bits of the first digit
    int ipt = 4, ti = 4096;  // items per thread and walk, items per tile (count and scatter agree)
    int md = 0;              // widest digit of the passes behind the fused one (0: the default)
    u8 *cnt8 = nullptr; u32 *counts = nullptr; u64 *offsets = nullptr;
    void release() {
        if (cnt8) dev_free(cnt8);
        if (counts) dev_free(counts);
        if (offsets) dev_free(offsets);
        cnt8 = nullptr; counts = nullptr; offsets = nullptr;
    }
};
// workgroup -> tile so that every XCD (workgroups are dealt to the 8 XCDs round-robin) takes a CONTIGUOUS range of
// tiles: neighbouring tiles write neighbouring pieces of every digit's run, and the partially written lines then meet
// in one L2 instead of two
GRL_DEV u32 xcd_tile(u32 b, u32 g) {
    const u32 per = g >> 3, rem = g & 7u, x = b & 7u, k = b >> 3;
    return x * per + (x < rem ? x : rem) + k;
}
template <class GEN, int DB, int TI = 4096>
__global__ void __launch_bounds__(kBlock) k_xs_count(u64 n, GEN gen, u32 dmask, u8 *cnt8, u32 *counts, u32 *scal /*[0] max keys per item*/) {
    constexpr int kXsTileItems = TI;
    constexpr int NB = 1 << DB;
    __shared__ u32 s_h[kBlock / 64][NB];
    __shared__ u32 s_max;
    for (int i = threadIdx.x; i < (kBlock / 64) * NB; i += kBlock) (&s_h[0][0])[i] = 0;
    if (threadIdx.x == 0) s_max = 0;
    __syncthreads();
    u32 *h = s_h[threadIdx.x >> 6];
    const u64 base = (u64)blockIdx.x * kXsTileItems;
    u32 mx = 0;
    // (four chains per lane walked together, as in k_xs_scatter: every step of the walk is four independent gathers -- one chain per
    // lane left the walk at the latency of its dependent loads whatever the occupancy)
    constexpr int CH = 4;
    for (int b = 0; b < kXsTileItems / (kBlock * CH); b++) {
        u64 item[CH], rec[CH];
        u32 cur[CH], c[CH];
        bool act[CH];
#pragma unroll
        for (int j = 0; j < CH; j++) {
            item[j] = base + ((u64)b * CH + (u64)j) * kBlock + threadIdx.x;
            act[j] = item[j] < n;
            cur[j] = gen.start(act[j] ? item[j] : 0);
            c[j] = 0;
        }
#pragma unroll
        for (int j = 0; j < CH; j++) rec[j] = gen.node(cur[j]);
#pragma unroll
        for (int j = 0; j < CH; j++) if (act[j] && gen.owns(rec[j])) { atomicAdd(&h[(u32)gen.key_own(cur[j], 0) & dmask], 1u); c[j]++; }
        for (;;) {
            bool m[CH], any = false;
            u32 nx[CH];
#pragma unroll
            for (int j = 0; j < CH; j++) { m[j] = act[j] && gen.more(rec[j]); any = any || m[j]; nx[j] = m[j] ? gen.next(rec[j]) : 0u; }
            if (!any) break;
            u64 nrec[CH];
#pragma unroll
            for (int j = 0; j < CH; j++) nrec[j] = gen.node(nx[j]);
#pragma unroll
            for (int j = 0; j < CH; j++) {
                if (m[j]) {
                    atomicAdd(&h[(u32)gen.key_step(rec[j], nx[j], 0) & dmask], 1u);
                    c[j]++;
                    rec[j] = nrec[j];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < CH; j++) {
            if (act[j]) { cnt8[item[j]] = (u8)(c[j] < 255u ? c[j] : 255u); mx = c[j] > mx ? c[j] : mx; }
        }
    }
    mx = wave_reduce<u32, Op::Max>(mx);
    if ((threadIdx.x & 63) == 0 && mx) atomicMax(&s_max, mx);
    __syncthreads();
    for (int d = threadIdx.x; d < NB; d += kBlock) {
        u32 t = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; w++) t += s_h[w][d];
        counts[(u64)blockIdx.x * NB + d] = t;
    }
    if (threadIdx.x == 0 && s_max) atomicMax(&scal[0], s_max);
}
// (MINB = workgroups per CU the register budget is cut for: see expand_sort)
// (K = u64, or u32 when a whole key fits 32 bits: half the bytes in LDS, in the write-out and in every later pass)
template <class GEN, int DB, int MINB = 3, class K = u64, int IPT = 4, int TI = 4096>
__global__ void __launch_bounds__(kBlock, MINB) k_xs_scatter(u64 n, GEN gen, u32 dmask, const u8 *cnt8, const u64 *offsets /*[tiles][NB]*/, K *out,
                                                       int xcd_aware) {
    constexpr int NB = 1 << DB, BPT = NB / kBlock > 0 ? NB / kBlock : 1;     // bins per thread (contiguous)
    constexpr int ROWS = kXsWin / kBlock;                                   // 16 rows of 64 keys per wave and round
    constexpr int kXsBatch = kBlock * IPT, kWalks = TI / kXsBatch;
    static_assert(TI % kXsBatch == 0 && IPT <= 4, "tile = whole walks; the packed count scan holds four 16-bit fields");
    __shared__ __attribute__((aligned(16))) K s_cells[kXsWin];
    __shared__ u32 s_cnt[kBlock / 64][NB];
    __shared__ u64 s_goff[NB];       // running global offset of every digit for this tile
    __shared__ u64 s_gbase[NB];      // global position of round-local index 0 of the digit's run (mod 2^64)
    __shared__ u64 s_w[4];
    __shared__ u32 s_wsum[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const u32 tile = xcd_aware ? xcd_tile(blockIdx.x, gridDim.x) : blockIdx.x;
    for (int d = threadIdx.x; d < NB; d += kBlock) s_goff[d] = offsets[(u64)tile * NB + d];
    const u64 base = (u64)tile * TI;
    // A ROUND = walks until the window is full, then one ranking + write-out.  Walk b covers the items [i0, i0 + 256 IPT) of the tile;
    // its keys, in sequence order, have the numbers [0, tot); a walk takes the keys [wlo, wlo + room) of them, room = what the window
    // still holds -- the whole walk almost always; a walk of more than a window's keys (an item may drop 32) goes in pieces, its
    // chains walked once per piece.  All of this is uniform over the workgroup.
    int b = 0;                       // the walk to take keys from next
    u32 wlo = 0;                     // keys of walk b already taken
    u32 fill = 0;                    // keys in the window
    bool have = false;               // counts and positions of walk b are in registers
    u64 item[IPT];
    u32 c[IPT], o[IPT], tot = 0;
    for (;;) {
        bool full = false;
        while (!full) {
            if (!have) {
                const u64 i0 = base + (u64)b * kXsBatch;
                if (b >= kWalks || i0 >= n) break;                             // uniform: the tile's items are used up
                // ---- key counts of my items (slab j holds items i0 + j*256 ..): one packed scan gives every item its position
                // in the walk's key sequence (16 bits per slab: a slab has at most 256 * 32 keys)
                u64 packed = 0;
#pragma unroll
                for (int j = 0; j < IPT; j++) {
                    item[j] = i0 + (u64)j * kBlock + threadIdx.x;
                    c[j] = item[j] < n ? (u32)cnt8[item[j]] : 0u;
                    packed |= (u64)c[j] << (16 * j);
                }
                u64 ptot;
                const u64 pex = block_excl_scan<u64>(packed, s_w, &ptot);
                tot = 0;
#pragma unroll
                for (int j = 0; j < IPT; j++) {
                    o[j] = tot + (u32)((pex >> (16 * j)) & 0xFFFFu);
                    tot += (u32)((ptot >> (16 * j)) & 0xFFFFu);
                }
                have = true;
                wlo = 0;
            }
            const u32 rest = tot - wlo, room = (u32)kXsWin - fill;
            if (rest > room && fill > 0) full = true;                         // (rank what is there first: the walk then starts an empty window)
            else {
                // ---- walk: the chains of a lane advance together; every step is IPT independent gathers.  Key number q of the
                // walk goes to window slot fill + q - wlo when that lies in [fill, kXsWin).
                bool act[IPT];
                u32 cur[IPT], pos[IPT];
                u64 rec[IPT], ib[IPT];
#pragma unroll
                for (int j = 0; j < IPT; j++) {
                    act[j] = item[j] < n && (c[j] ? (o[j] < wlo + room && o[j] + c[j] > wlo) : wlo == 0);
                    cur[j] = gen.start(act[j] ? item[j] : 0);
                }
#pragma unroll
                for (int j = 0; j < IPT; j++) rec[j] = gen.node(cur[j]);
#pragma unroll
                for (int j = 0; j < IPT; j++) {
                    pos[j] = o[j] - wlo;                                   // may wrap below zero: the window test below is unsigned
                    ib[j] = act[j] ? gen.item_bits(item[j]) : 0;
                    if (act[j] && gen.owns(rec[j])) {
                        if (pos[j] < room) s_cells[fill + pos[j]] = (K)gen.key_own(cur[j], ib[j]);
                        pos[j]++;
                    }
                }
                for (;;) {
                    bool m[IPT], any = false;
                    u32 nx[IPT];
#pragma unroll
                    for (int j = 0; j < IPT; j++) { m[j] = act[j] && gen.more(rec[j]); any = any || m[j]; nx[j] = m[j] ? gen.next(rec[j]) : 0u; }
                    if (!any) break;
                    u64 nrec[IPT];
#pragma unroll
                    for (int j = 0; j < IPT; j++) nrec[j] = gen.node(nx[j]);       // unconditional: IPT loads in flight per lane
#pragma unroll
                    for (int j = 0; j < IPT; j++) {
                        if (m[j]) {
                            if (pos[j] < room) s_cells[fill + pos[j]] = (K)gen.key_step(rec[j], nx[j], ib[j]);
                            pos[j]++;
                            rec[j] = nrec[j];
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < IPT; j++) if (act[j]) gen.finish(item[j], rec[j]);
                const u32 took = rest < room ? rest : room;
                fill += took;
                wlo += took;
                if (wlo >= tot) { have = false; b++; }                       // the walk is done (also a walk without keys: its side effects are)
                else full = true;                                            // a walk larger than the window: the next piece after the ranking
            }
        }
        if (fill == 0) break;                                                // uniform: nothing left
        __syncthreads();
        // ---- rank the window's keys by the digit and write them out (the body of k_rs_scatter)
        const u32 hn = fill;
        fill = 0;
        const u32 rpw = (u32)__builtin_amdgcn_readfirstlane((int)(((hn + 63) / 64 + 3) / 4));   // rows per wave (uniform, in an SGPR): every wave takes a contiguous share
        for (int d = threadIdx.x; d < (kBlock / 64) * NB; d += kBlock) (&s_cnt[0][0])[d] = 0;
        K key[ROWS];
        u32 idx[ROWS];
#pragma unroll
        for (int q = 0; q < ROWS; q++) {
            const u32 t = ((u32)w * rpw + q) * 64 + lane;
            key[q] = ((u32)q < rpw && t < hn) ? s_cells[t] : K(0);
        }
        __syncthreads();
        // (rows behind rpw hold nothing: skipped by a workgroup-uniform branch)
        wave_rank<DB, ROWS, false>([&](int q) { return (u32)key[q] & dmask; }, (u32)w * rpw * 64u + (u32)lane, hn, rpw, &s_cnt[w][0], idx);
        __syncthreads();
        {   // thread t owns bins [t*BPT, (t+1)*BPT): wave bases, round-local digit starts, global bases
            u32 cw[BPT][4], tt[BPT], sum = 0;
#pragma unroll
            for (int e = 0; e < BPT; e++) {
                const int d = threadIdx.x * BPT + e;
                tt[e] = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) { cw[e][k] = d < NB ? s_cnt[k][d] : 0u; tt[e] += cw[e][k]; }
                sum += tt[e];
            }
            u32 incl = sum;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                u32 v = (u32)__shfl_up((int)incl, off);
                if (lane >= off) incl += v;
            }
            if (lane == 63) s_wsum[w] = incl;
            __syncthreads();
            u32 start = incl - sum;
            for (int k = 0; k < w; k++) start += s_wsum[k];
#pragma unroll
            for (int e = 0; e < BPT; e++) {
                const int d = threadIdx.x * BPT + e;
                if (d < NB) {
                    s_cnt[0][d] = start;
                    s_cnt[1][d] = start + cw[e][0];
                    s_cnt[2][d] = start + cw[e][0] + cw[e][1];
                    s_cnt[3][d] = start + cw[e][0] + cw[e][1] + cw[e][2];
                    const u64 g = s_goff[d];
                    s_gbase[d] = g - (u64)start;
                    s_goff[d] = g + tt[e];
                    start += tt[e];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < ROWS; q++) {
            const u32 t = ((u32)w * rpw + q) * 64 + lane;
            if ((u32)q < rpw && t < hn) s_cells[idx[q] + s_cnt[w][(u32)key[q] & dmask]] = key[q];   // in place: the round's keys are in registers
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < ROWS; j++) {
            const u32 t = (u32)j * kBlock + threadIdx.x;
            if (t < hn) {
                const K k = s_cells[t];
                out[s_gbase[(u32)k & dmask] + t] = k;
            }
        }
        __syncthreads();
    }
}
// Phase 1: count.  Returns the number of keys; plan.ok tells whether expand_sort may follow (else the caller expands
// by other means; the plan's buffers are released either way by plan.release()).
template <class GEN>
inline u64 expand_count(u64 n, GEN gen, int bits, XsPlan &plan, const char *name = "expand") {
    plan = XsPlan();
    plan.n = n; plan.bits = bits;
    if (n == 0) { plan.ok = true; return 0; }
    // 9-bit first digit when that saves a pass over all keys (the later passes take 8 bits each)
    {
        const int md = rs_max_digit();
        plan.db = (bits > 8 && (bits - 9 + md - 1) / md < (bits - 8 + md - 1) / md) ? 9 : 8;
        // 9-bit digits all the way when THAT saves a pass (27 bucket bits: 9 + 9 + 9 instead of 8 + 7 + 6 + 6 -- a 9-bit pass costs
        // ~30 % more than an 8-bit one, a pass saved is a pass saved: 14.1 -> 11.3 ms for the passes of level 1 of the 10 GB build)
        const int passes_a = 1 + (bits > plan.db ? (bits - plan.db + md - 1) / md : 0);
        const int passes_b = bits > 9 ? 1 + (bits - 9 + 8) / 9 : 99;
        if (md == 8 && passes_b < passes_a && !dev_env("GRLBWT_XS_DIGIT8")) { plan.db = 9; plan.md = 9; }
    }
    const int NB = 1 << plan.db;
    const u32 dmask = bits >= plan.db ? (u32)NB - 1u : (1u << bits) - 1u;
    plan.tiles = (u32)((n + (u64)plan.ti - 1) / (u64)plan.ti);
    plan.cnt8 = (u8 *)dev_alloc(n);
    plan.counts = (u32 *)dev_alloc((u64)NB * plan.tiles * sizeof(u32));
    plan.offsets = (u64 *)dev_alloc((u64)NB * plan.tiles * sizeof(u64));
    u32 chunks = (plan.tiles + kRsChunk - 1) / kRsChunk;
    u32 *chunk_sums = (u32 *)dev_alloc((u64)NB * chunks * sizeof(u32));
    u64 *chunk_off = (u64 *)dev_alloc((u64)NB * chunks * sizeof(u64));
    u64 *scal = (u64 *)dev_alloc(16);
    dev_memset(scal, 0, 16);
    prof_begin(std::string(name) + ".xcount");
    if (plan.db == 9) hipLaunchKernelGGL((k_xs_count<GEN, 9>), dim3(plan.tiles), dim3(kBlock), 0, rt().stream, n, gen, dmask, plan.cnt8, plan.counts, (u32 *)scal);
    else hipLaunchKernelGGL((k_xs_count<GEN, 8>), dim3(plan.tiles), dim3(kBlock), 0, rt().stream, n, gen, dmask, plan.cnt8, plan.counts, (u32 *)scal);
    prof_end();
    after_launch(name);
    if (plan.db == 9) rs_offsets<512>(plan.counts, plan.tiles, chunk_sums, chunk_off, plan.offsets, scal + 1, name);
    else rs_offsets<256>(plan.counts, plan.tiles, chunk_sums, chunk_off, plan.offsets, scal + 1, name);
    u64 h[2];
    d2h(h, scal, 16);
    dev_free(chunk_sums); dev_free(chunk_off); dev_free(scal);
    plan.maxc = (u32)h[0];
    plan.E = h[1];
    // (GRLBWT_XS_MAXC: the tests lower the limit so that ordinary inputs take the caller's unfused branch)
    const char *lim = getenv("GRLBWT_XS_MAXC");
    plan.ok = plan.maxc <= (lim ? (u32)atoi(lim) : 32u);
    return plan.E;
}
// Phase 2: generate + first pass into buf_a, remaining passes ping-pong; returns 0 if the result is in buf_a, 1 if in buf_b
template <class GEN, class K = u64>
inline int expand_sort(GEN gen, XsPlan &plan, K *buf_a, K *buf_b, const char *name = "expand") {
    if (!plan.ok) throw Error(-71, "expand_sort: plan not usable");
    if (plan.n == 0) return 0;                 // (with E == 0 the walk still runs: cells() has side effects)
    const int NB = 1 << plan.db;
    const u32 dmask = plan.bits >= plan.db ? (u32)NB - 1u : (1u << plan.bits) - 1u;
    // XCD-contiguous tile ranges were measured 7 % SLOWER for this kernel on the 10 GB build (32.8 vs 30.6 ms at level 0,
    // 16.9 vs 14.8 at level 1): the round-robin deal already lets the 8 L2s share every digit's write front; opt-in only
    static const int xcd_aware = dev_env("GRLBWT_XCD_MAP") ? 1 : 0;      // (development builds)
    prof_begin(std::string(name) + ".xscatter", plan.E * sizeof(K));
    // (three workgroups per CU: at two the kernel keeps everything in registers, at three it spills ~100 bytes per lane and is the
    // faster one -- it hides its gathers with waves, not with registers: round 3)
    // Workgroups per CU the register budget is cut for.  8-byte cells: three -- the LDS window allows no more.  4-byte cells (level 0 of
    // a DNA collection): FOUR (round 6: 137 -> 128 registers, nothing spilled; level 0 of the 10 GB build 19.2 -> 16.2 ms) -- the kernel
    // hides its chain gathers with waves, not with registers (round 3: two per CU, everything in registers, was the slower form).
    constexpr int kOcc = sizeof(K) == 4 ? 4 : 3;
    if (plan.db == 9) hipLaunchKernelGGL((k_xs_scatter<GEN, 9, kOcc, K>), dim3(plan.tiles), dim3(kBlock), 0, rt().stream, plan.n, gen, dmask, plan.cnt8, plan.offsets, buf_a, xcd_aware);
    else hipLaunchKernelGGL((k_xs_scatter<GEN, 8, kOcc, K>), dim3(plan.tiles), dim3(kBlock), 0, rt().stream, plan.n, gen, dmask, plan.cnt8, plan.offsets, buf_a, xcd_aware);
    prof_end();
    after_launch(name);
    if (plan.bits <= plan.db || plan.E == 0) return 0;
    struct Widen { int old; Widen(int d) : old(rs_digit_override()) { if (d) rs_digit_override() = d; } ~Widen() { rs_digit_override() = old; } } widen(plan.md);
    return sort_keys_fwd(buf_a, buf_b, plan.E, plan.db, plan.bits, name);
}

template <class K, class V, int SITE, int DB, int TB = kBlock, bool MIX = false>
inline void rs_pass(const K *kin, const V *vin, K *kout, V *vout, u64 n, int shift, u32 dmask, u32 tiles, u32 *counts, u64 *offsets,
                    u32 *chunk_sums, u64 *chunk_off, const char *name, u16 *rank_out = nullptr) {
    prof_begin(std::string(name) + ".hist", n * sizeof(K));
    hipLaunchKernelGGL((k_rs_hist<K, SITE, DB, TB, MIX>), dim3(tiles), dim3(TB), 0, rt().stream, kin, n, shift, dmask, counts, tiles);
    prof_end();
    after_launch(name);
    rs_offsets<(1 << DB)>(counts, tiles, chunk_sums, chunk_off, offsets, nullptr, name);
    prof_begin(std::string(name) + ".scatter", n * (sizeof(K) + (std::is_same<V, NoVal>::value ? 0 : sizeof(V))) * 2);   // pairs read once + written once
    hipLaunchKernelGGL((k_rs_scatter<K, V, SITE, DB, TB, MIX>), dim3(tiles), dim3(TB), 0, rt().stream, kin, vin, kout, vout, n, shift, dmask, offsets, tiles, rank_out);
    prof_end();
    after_launch(name);
}
// Threads per workgroup of the plain sorts (GRLBWT_RS_THREADS=256|512; 0 = by record size).  512 threads = tiles of 8192 keys:
// a tile leaves 32 instead of 16 keys per bin on average, i.e. runs of 256 instead of 128 bytes at the write front of a pass
// over 8-byte keys -- the passes run at the rate of their scattered writes (tools/sortbench.hip: uniform digits 2.2-2.5 TB/s,
// skewed digits 4.5 TB/s with the same kernel).  Records above 12 bytes keep 4096-key tiles (LDS).
inline int rs_threads_override() {
    static const int v = [] { const char *e = dev_env("GRLBWT_RS_THREADS"); return e ? atoi(e) : 0; }();
    return v;
}
inline int rs_threads_override_xs() {      // (SITE 1: the passes behind the fused expansion of the induction)
    static const int v = [] { const char *e = dev_env("GRLBWT_RS_THREADS_XS"); return e ? atoi(e) : 0; }();
    return v;
}
template <class K, class V, int SITE, int TB>
inline int sort_pairs_tb(K *keys_a, V *vals_a, K *keys_b, V *vals_b, u64 n, int begin_bit, int end_bit, const char *name) {
    int widths[16];
    const int passes = rs_plan(end_bit - begin_bit, widths);
    int maxw = 8;
    for (int p = 0; p < passes; p++) if (widths[p] > maxw) maxw = widths[p];
    const u64 NBmax = (u64)1 << maxw;
    constexpr u64 TILE = (u64)TB * kRsItems;
    u32 tiles = (u32)((n + TILE - 1) / TILE);
    u32 *counts = (u32 *)dev_alloc(NBmax * tiles * sizeof(u32));
    u64 *offsets = (u64 *)dev_alloc(NBmax * tiles * sizeof(u64));
    u32 chunks = (tiles + kRsChunk - 1) / kRsChunk;
    u32 *chunk_sums = (u32 *)dev_alloc(NBmax * chunks * sizeof(u32));
    u64 *chunk_off = (u64 *)dev_alloc(NBmax * chunks * sizeof(u64));
    int cur = 0, shift = begin_bit;
    for (int p = 0; p < passes; p++) {
        // a digit narrower than its kernel's table: key bits at and above end_bit (a payload riding in the key) must not
        // take part in the order
        const int wd = widths[p];
        const u32 dmask = (1u << wd) - 1u;
        K *kin = cur ? keys_b : keys_a;
        V *vin = cur ? vals_b : vals_a;
        K *kout = cur ? keys_a : keys_b;
        V *vout = cur ? vals_a : vals_b;
        if (wd <= 8) rs_pass<K, V, SITE, 8, TB>(kin, vin, kout, vout, n, shift, dmask, tiles, counts, offsets, chunk_sums, chunk_off, name);
        else if (wd == 9) rs_pass<K, V, SITE, 9, TB>(kin, vin, kout, vout, n, shift, dmask, tiles, counts, offsets, chunk_sums, chunk_off, name);      // (1024 threads: counters in 16-bit halves)
        else if constexpr (TB <= 512) rs_pass<K, V, SITE, 10, TB>(kin, vin, kout, vout, n, shift, dmask, tiles, counts, offsets, chunk_sums, chunk_off, name);
        else throw Error(-71, "sort_pairs: digit too wide for 1024-thread workgroups");      // (the per-wave counters of a 10-bit digit do not fit beside the tile)
        shift += wd;
        cur ^= 1;
    }
    dev_free(counts);      // stream-ordered reuse: no host synchronisation needed
    dev_free(offsets);
    dev_free(chunk_sums);
    dev_free(chunk_off);
    return cur;
}
// Sorts n pairs by key bits [begin_bit, end_bit).  Buffers a/b ping-pong; returns
// 0 if the result is in (keys_a, vals_a), 1 if in (keys_b, vals_b).
template <class K, class V, int SITE = 0>
inline int sort_pairs(K *keys_a, V *vals_a, K *keys_b, V *vals_b, u64 n, int begin_bit, int end_bit,
                      const char *name = "radix_sort") {
    if (n == 0 || end_bit <= begin_bit) return 0;
    constexpr int rec = (int)sizeof(K) + (std::is_same<V, NoVal>::value ? 0 : (int)sizeof(V));
    if constexpr (rec <= 12) {
        const int ov = SITE == 1 ? rs_threads_override_xs() : rs_threads_override();
        // (large tiles want many of them: 16384-key tiles from 2^24 keys on -- 1024 tiles, four per CU --, 8192-key tiles from 2^22)
        const int tb = ov ? ov : (SITE == 1 ? 256 : (n >= ((u64)1 << 24) ? 1024 : 512));
        if (n >= (u64)1 << 22 || (ov && n >= (u64)1 << 20)) {
            // 9-bit digits on the 16384-key tiles where they save a pass (round 6: the per-wave counters of a 9-bit digit fit beside the
            // tile as 16-bit halves; 32 instead of 64 keys per bin at the write front cost a pass ~15 %, a pass of seven saved is ~14 % --
            // measured on the 10 GB build: suffix_sort0 of level 2, 54 key bits, 7 x 3.10 -> 6 x 3.16 ms; level 1, 51 bits, 14.5 -> 13.5 ms;
            // the dictionary stage 322 -> 317 ms.  Not on the smaller tiles: 8 or 16 keys per bin, rounds 3 and 5.)
            struct Widen { int old; Widen(int d) : old(rs_digit_override()) { if (d) rs_digit_override() = d; } ~Widen() { rs_digit_override() = old; } };
            const int nbits = end_bit - begin_bit;
            Widen widen((tb >= 1024 && SITE == 0 && !rs_digit_override() && rs_max_digit() == 8 && (nbits + 8) / 9 < (nbits + 7) / 8) ? 9 : 0);
            int widths[16], maxw = 0;
            const int passes = rs_plan(end_bit - begin_bit, widths);
            for (int p = 0; p < passes; p++) if (widths[p] > maxw) maxw = widths[p];
            if (tb >= 1024 && maxw <= 9) return sort_pairs_tb<K, V, SITE, 1024>(keys_a, vals_a, keys_b, vals_b, n, begin_bit, end_bit, name);
            if (tb >= 512) return sort_pairs_tb<K, V, SITE, 512>(keys_a, vals_a, keys_b, vals_b, n, begin_bit, end_bit, name);
        }
    }
    return sort_pairs_tb<K, V, SITE, kBlock>(keys_a, vals_a, keys_b, vals_b, n, begin_bit, end_bit, name);
}

// keys only: returns 0 if the result is in keys_a, 1 if in keys_b
template <class K, int SITE = 0>
inline int sort_keys(K *keys_a, K *keys_b, u64 n, int begin_bit, int end_bit, const char *name = "radix_sort") {
    return sort_pairs<K, NoVal, SITE>(keys_a, (NoVal *)nullptr, keys_b, (NoVal *)nullptr, n, begin_bit, end_bit, name);
}
inline int sort_keys_fwd(u64 *a, u64 *b, u64 n, int begin_bit, int end_bit, const char *name) {
    return sort_keys<u64, 1>(a, b, n, begin_bit, end_bit, name);      // SITE 1: the passes behind the fused expand + first pass
}
inline int sort_keys_fwd(u32 *a, u32 *b, u64 n, int begin_bit, int end_bit, const char *name) {
    return sort_keys<u32, 1>(a, b, n, begin_bit, end_bit, name);
}


// ------------------------------------------------------- partition sort that can be undone + per-partition de-duplication
// (phrase naming of the levels above 0: every phrase occurrence becomes a 128-bit record, the records are grouped by a hash
// prefix into partitions small enough for an LDS table, de-duplicated and counted there, and the values of the phrases travel
// back to text order through the same passes in reverse -- no table in HBM, no per-occurrence atomics or random accesses)
struct alignas(16) U128 {
    u64 lo, hi;
    U128() = default;
    GRL_HD U128(int) : lo(0), hi(0) {}
    GRL_HD U128(u64 l, u64 h) : lo(l), hi(h) {}
    GRL_HD bool operator==(const U128 &o) const { return lo == o.lo && hi == o.hi; }
};
static constexpr u32 kNoId = 0xFFFFFFFFu;

// One pass of RecSort::backward: src holds one element per record in the OUTPUT order of the forward pass, dst gets them in its
// INPUT order.  The forward pass left every record's place inside its tile's sorted order (rank_in, 2 bytes); the tile's digit
// runs follow from the pass's offsets (a run's length = the next tile's offset of the same digit - mine).  So: the digit of every
// output slot by one prefix-maximum over the run starts, the elements read run by run -- neighbouring lanes, neighbouring
// addresses -- into LDS in the tile's sorted order, and every record picks its own by its stored rank.  No ranking on the way back.
// (Rounds 3-5 recomputed the ballot ranking of the forward pass from its input keys -- and round 6 first from stored digits --:
// the pass then costs per RECORD what the forward pass costs, 7 ms per pass over the 964 M records of level 1 of the 10 GB build
// for 10 bytes per record.  Storing where every element went and gathering through that -- 4 bytes per element and pass -- was
// measured slower than re-ranking in round 3; two bytes of tile-local rank + the offsets the pass has anyway carry the same.)
template <class W, int DB, int TB>
__global__ void __launch_bounds__(TB)
    k_rs_unscatter(const u16 *rank_in, const W *src, W *dst, u64 n, const u64 *offsets /*[tiles][NB]*/, u32 tiles) {
    constexpr int NB = 1 << DB, TILE = TB * kRsItems, NW = TB / 64;
    static_assert(NB <= TB, "one thread per digit");
    __shared__ u16 s_d[TILE];                    // digit of every output slot of the tile
    __shared__ W s_v[TILE];                      // the elements in the tile's sorted order
    __shared__ u64 s_gbase[NB];                  // global position of slot 0 of the digit's run, minus the run's first slot
    __shared__ u32 s_wsum[NW];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const u32 tile = blockIdx.x;
    const u64 base = (u64)tile * TILE;
    const u64 left = n - base;
    const u32 tile_n = left < (u64)TILE ? (u32)left : (u32)TILE;
    // (my records' ranks first: the kernel's workgroup is alone on its CU and waits out every trip to memory -- this one is back
    // when the elements are in LDS)
    u16 rk[kRsItems];
#pragma unroll
    for (int j = 0; j < kRsItems; j++) { const u32 t = (u32)j * TB + threadIdx.x; rk[j] = t < tile_n ? rank_in[base + t] : (u16)0; }
    for (u32 t = threadIdx.x; t < (u32)TILE; t += TB) s_d[t] = 0;
    // ---- run lengths and starts of my digits
    u32 cnt = 0;
    u64 g0 = 0;
    if ((int)threadIdx.x < NB) {
        const u32 d = threadIdx.x;
        g0 = offsets[(u64)tile * NB + d];
        const u64 nx = tile + 1 < tiles ? offsets[(u64)(tile + 1) * NB + d] : (d + 1 < (u32)NB ? offsets[d + 1] : n);      // (digit-major: behind the last tile of d comes tile 0 of d + 1)
        cnt = (u32)(nx - g0);
    }
    u32 incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        u32 o = (u32)__shfl_up((int)incl, off);
        if (lane >= off) incl += o;
    }
    if (lane == 63) s_wsum[w] = incl;
    __syncthreads();
    u32 start = incl - cnt;
    for (int k = 0; k < w; k++) start += s_wsum[k];
    if ((int)threadIdx.x < NB) {
        s_gbase[threadIdx.x] = g0 - (u64)start;
        if (cnt) s_d[start] = (u16)threadIdx.x;         // the run's first slot names its digit (slot 0: digit of the first non-empty run, possibly 0)
    }
    __syncthreads();
    // ---- digit of every slot: the last run start at or in front of it (digits ascend with the slots: a prefix maximum)
    {
        const u32 t0 = threadIdx.x * kRsItems;
        u32 dloc[kRsItems], m = 0;
#pragma unroll
        for (int j = 0; j < kRsItems; j++) { const u32 h = s_d[t0 + j]; m = h > m ? h : m; dloc[j] = m; }
        u32 pm = m;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            u32 o = (u32)__shfl_up((int)pm, off);
            if (lane >= off) pm = o > pm ? o : pm;
        }
        __syncthreads();                                 // (everybody has read its heads; s_wsum is free again)
        if (lane == 63) s_wsum[w] = pm;
        __syncthreads();
        u32 before = (u32)__shfl_up((int)pm, 1);
        if (lane == 0) before = 0;
        for (int k = 0; k < w; k++) before = s_wsum[k] > before ? s_wsum[k] : before;
#pragma unroll
        for (int j = 0; j < kRsItems; j++) s_d[t0 + j] = (u16)(dloc[j] > before ? dloc[j] : before);
    }
    __syncthreads();
    // ---- the elements, run by run, into the tile's sorted order
#pragma unroll
    for (int j = 0; j < kRsItems; j++) {
        const u32 t = (u32)j * TB + threadIdx.x;
        if (t < tile_n) s_v[t] = src[s_gbase[s_d[t]] + t];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kRsItems; j++) {
        const u32 t = (u32)j * TB + threadIdx.x;
        if (t < tile_n) dst[base + t] = s_v[rk[j]];
    }
}

// Records of two 64-bit words (key, hi) grouped into 2^bits partitions by the top bits of key x kMixMul -- a hash of the key that
// needs no array of its own -- and the way back: one element per record from sorted order to the original order.
// (Phrase naming of the levels above 0.  Rounds 3-5 sorted (32-bit hash, 128-bit record) pairs on tiles of 4096 records, the 16-byte
// values being what limited the tile: 20 bytes per record and pass at 0.32-0.34 of the HBM peak, 30.8 ms for the two passes over the
// 964 M records of level 1 of the 10 GB build.  A record split into two words sorts like the (key, position) pairs of the suffix
// sort -- 16384-record tiles, both words staged through the same 128 KB of LDS one after the other -- and the way back reads the
// 2-byte digits the forward pass left instead of the keys.)
// The caller owns the four record buffers (forward tells which pair holds the result); the sort owns the digits and tile offsets of
// its passes until release().
struct RecSort {
    u64 n = 0;
    int bits = 0, passes = 0, shifts[4] = {0, 0, 0, 0}, widths[4] = {0, 0, 0, 0};
    u32 tiles = 0;
    u16 *rnk[4] = {nullptr, nullptr, nullptr, nullptr};       // every record's place inside its tile's sorted order, by input position of pass p
    u64 *offs[4] = {nullptr, nullptr, nullptr, nullptr};
    static constexpr int kThreads = 1024;
    GRL_HD static u64 part_of(u64 key, int bits_) { return bits_ ? (key * kMixMul) >> (64 - bits_) : 0; }
    // returns 0 if the sorted records are in (key_a, hi_a), 1 if in (key_b, hi_b)
    int forward(u64 *key_a, u64 *hi_a, u64 *key_b, u64 *hi_b, u64 n_, int bits_, const char *name = "rec_sort") {
        release();
        n = n_; bits = bits_;
        if (n == 0 || bits <= 0) { bits = 0; return 0; }
        if (bits > 27) throw Error(-22, "RecSort: more than 2^27 partitions");
        // digits of at most 9 bits (the 16 waves' counters of a 10-bit digit do not fit beside the tile): two passes up to 18 bits
        passes = (bits + 8) / 9;
        for (int p = 0, sh = 64 - bits; p < passes; p++) {
            widths[p] = bits / passes + (p < bits % passes ? 1 : 0);
            shifts[p] = sh;
            sh += widths[p];
        }
        const u64 tile = (u64)kThreads * kRsItems;
        tiles = (u32)((n + tile - 1) / tile);
        u32 *counts = (u32 *)dev_alloc((u64)512 * tiles * sizeof(u32));
        u32 chunks = (tiles + kRsChunk - 1) / kRsChunk;
        u32 *chunk_sums = (u32 *)dev_alloc((u64)512 * chunks * sizeof(u32));
        u64 *chunk_off = (u64 *)dev_alloc((u64)512 * chunks * sizeof(u64));
        int cur = 0;
        for (int p = 0; p < passes; p++) {
            const int db = widths[p] <= 8 ? 8 : 9;
            rnk[p] = (u16 *)dev_alloc(n * sizeof(u16));
            offs[p] = (u64 *)dev_alloc(((u64)1 << db) * tiles * sizeof(u64));
            const u32 dmask = (1u << widths[p]) - 1u;
            u64 *kin = cur ? key_b : key_a, *kout = cur ? key_a : key_b, *vin = cur ? hi_b : hi_a, *vout = cur ? hi_a : hi_b;
            if (db == 8) rs_pass<u64, u64, 2, 8, kThreads, true>(kin, vin, kout, vout, n, shifts[p], dmask, tiles, counts, offs[p], chunk_sums, chunk_off, name, rnk[p]);
            else rs_pass<u64, u64, 2, 9, kThreads, true>(kin, vin, kout, vout, n, shifts[p], dmask, tiles, counts, offs[p], chunk_sums, chunk_off, name, rnk[p]);
            cur ^= 1;
        }
        dev_free(counts); dev_free(chunk_sums); dev_free(chunk_off);
        return cur;
    }
    // in[j] belongs to the record at sorted position j; out[i] = the element of the record that was at position i originally.
    // tmp: scratch of n elements (used with more than one pass); `in` is overwritten when passes > 2.
    template <class W>
    void backward(W *in, W *tmp, W *out, const char *name = "rec_sort.back") const {
        if (n == 0) return;
        if (passes == 0) { d2d(out, in, n * sizeof(W)); return; }
        W *src = in;
        for (int p = passes - 1; p >= 0; p--) {
            W *dst = (p == 0) ? out : ((src == tmp) ? in : tmp);
            prof_begin(name, n * (sizeof(u16) + 2 * sizeof(W)));
            if (widths[p] <= 8) hipLaunchKernelGGL((k_rs_unscatter<W, 8, kThreads>), dim3(tiles), dim3(kThreads), 0, rt().stream, (const u16 *)rnk[p], src, dst, n, offs[p], tiles);
            else hipLaunchKernelGGL((k_rs_unscatter<W, 9, kThreads>), dim3(tiles), dim3(kThreads), 0, rt().stream, (const u16 *)rnk[p], src, dst, n, offs[p], tiles);
            prof_end();
            after_launch(name);
            src = dst;
        }
    }
    void release() {
        for (int p = 0; p < 4; p++) {
            if (rnk[p]) dev_free(rnk[p]);
            if (offs[p]) dev_free(offs[p]);
            rnk[p] = nullptr; offs[p] = nullptr;
        }
        passes = 0; n = 0; bits = 0;
    }
    RecSort() {}
    RecSort(const RecSort &) = delete;
    RecSort &operator=(const RecSort &) = delete;
    ~RecSort() { release(); }
};
// lane p in [0, 2^bits]: first record of partition p in the sorted key array
struct RecBoundsFn {
    const u64 *skey; u64 n; int bits; u64 nparts; u64 *pstart;
    GRL_DEV void operator()(u64 p) const {
        u64 lo = 0, hi = n;
        if (p == nparts) lo = n;
        else while (lo < hi) { const u64 mid = (lo + hi) >> 1; if (RecSort::part_of(skey[mid], bits) < p) lo = mid + 1; else hi = mid; }
        pstart[p] = lo;
    }
};

// Per-partition de-duplication of (key, hi) records in LDS.  Partition p = records [pstart[p], pstart[p + 1]).  One workgroup
// of 1024 threads per partition: an open-addressing table of kPdSlots entries -- five bits of hi | index of the entry's first
// record, claimed by ONE compare-and-swap, with that record's KEY beside it and a counter.  A probe compares the key in LDS and
// reads the representative's hi only when that word has more in it than the entry's five bits say (symbols above bit 64); while
// the key of a fresh entry is not there yet -- its owner writes it right behind the swap -- the probe reads the representative's
// key from memory instead: nobody waits for anybody.
// (Rounds 3-5 kept a hash tag per entry and confirmed every tag match against the representative's 16 bytes: one random 16-byte
// read per record -- two 8-byte ones with the record in two arrays: 15 and 19.4 ms for the 964 M records of level 1 of the 10 GB
// build.)
// Records with valid(hi) == false take no part (lid = kNoId).  Outputs: lid[i] = dense local id of record i's value among the
// distinct values of its partition; pcount[p]; the j-th distinct value of p and its count at dkey / dhi / dcnt[pstart[p] + j].
// *overflow is set when a partition does not fit (more distinct values than the table takes, or more than 2^26 records): the
// caller falls back to another method.
static constexpr int kPdSlots = 8192, kPdThreads = 1024;
// what an entry keeps of hi: the top four bits (length, ends-a-string) and a flag "nothing else is set"
GRL_DEV u32 pd_hi_bits(u64 hi) { return (u32)(hi >> 60) | (((hi << 4) == 0ull) ? 16u : 0u); }
template <class VALID>
__global__ void __launch_bounds__(kPdThreads) k_rec_dedupe(const u64 *pstart, const u64 *skey, const u64 *shi, VALID valid, u32 *lid, u32 *pcount,
                                                           u64 *dkey, u64 *dhi, u32 *dcnt, u32 *overflow) {
    __shared__ u32 s_idx[kPdSlots];                     // hi bits << 27 | index of the representative + 1   (0: empty); later the entry's dense number
    __shared__ unsigned long long s_key[kPdSlots];      // its key (0: not written yet; the keys 0 and 1 are never kept here)
    __shared__ u32 s_cnt[kPdSlots];
    __shared__ u32 s_w[kPdThreads / 64];
    __shared__ u32 s_fail;
    const u64 a = pstart[blockIdx.x], b = pstart[blockIdx.x + 1];
    for (int i = threadIdx.x; i < kPdSlots; i += kPdThreads) { s_idx[i] = 0u; s_key[i] = 0ull; s_cnt[i] = 0u; }
    if (threadIdx.x == 0) s_fail = (b - a >= (1ull << 26)) ? 1u : 0u;
    __syncthreads();
    const bool usable = s_fail == 0;
    // (a thread's first kPdMine records keep the slot they found in registers until the slots have their dense numbers: one
    // store of lid per record instead of a store, a load and a store -- and one round trip to memory less in a kernel whose
    // workgroup, the only one on its CU, waits out every one of them)
    constexpr int kPdMine = 8;
    u32 myres[kPdMine];
#pragma unroll
    for (int k = 0; k < kPdMine; k++) myres[k] = kNoId;
    auto insert = [&](u64 i, u64 vk, u64 vh) -> u32 {
        u32 res = kNoId;
        if (valid(vh)) {
            u64 g = (vk ^ (vh * 0x9E3779B97F4A7C15ull)) * 0xD6E8FEB86659FD93ull;
            g ^= g >> 32; g *= 0xFF51AFD7ED558CCDull; g ^= g >> 29;
            const u32 hb = pd_hi_bits(vh);
            const u32 mine = (hb << 27) | (u32)(i - a + 1);
            u32 slot = (u32)g & (kPdSlots - 1);
            bool done = false;
            for (int probes = 0; probes < kPdSlots && !done; probes++) {
                u32 e = s_idx[slot];
                if (e == 0u) {
                    const u32 old = atomicCAS(&s_idx[slot], 0u, mine);
                    if (old == 0u) {
                        if (vk > 1ull) __hip_atomic_store(&s_key[slot], (unsigned long long)vk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        done = true;
                    }
                    e = old == 0u ? mine : old;
                }
                if (!done && (e >> 27) == hb) {
                    // same hi bits: the key -- from LDS when it is there -- and the rest of hi where there is a rest
                    const u64 ri = a + (u64)((e & 0x7FFFFFFu) - 1u);
                    const unsigned long long k = __hip_atomic_load(&s_key[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    bool same = k != 0ull ? k == (unsigned long long)vk : skey[ri] == vk;
                    if (same && !(hb & 16u)) same = shi[ri] == vh;
                    done = same;
                }
                if (!done) slot = (slot + 1) & (kPdSlots - 1);
            }
            if (done) { atomicAdd(&s_cnt[slot], 1u); res = slot; }
            else s_fail = 1u;                       // table full
        }
        return res;
    };
    if (usable) {
        u64 vk[kPdMine], vh[kPdMine];             // (all of a thread's records in flight before the first probe)
#pragma unroll
        for (int k = 0; k < kPdMine; k++) {
            const u64 i = a + (u64)k * kPdThreads + threadIdx.x;
            vk[k] = i < b ? skey[i] : 0ull;
            vh[k] = i < b ? shi[i] : 0ull;
        }
#pragma unroll
        for (int k = 0; k < kPdMine; k++) {
            const u64 i = a + (u64)k * kPdThreads + threadIdx.x;
            if (i < b) myres[k] = insert(i, vk[k], vh[k]);
        }
        for (u64 i = a + (u64)kPdMine * kPdThreads + threadIdx.x; i < b; i += kPdThreads) lid[i] = insert(i, skey[i], shi[i]);
    }
    __syncthreads();
    // dense numbering of the occupied slots (slot order): thread t takes slots [8t, 8t + 8)
    constexpr int PER = kPdSlots / kPdThreads;
    u32 occ = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) occ += s_idx[threadIdx.x * PER + k] != 0u ? 1u : 0u;
    u32 incl = occ;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        u32 o = (u32)__shfl_up((int)incl, off);
        if (lane >= off) incl += o;
    }
    if (lane == 63) s_w[w] = incl;
    __syncthreads();
    u32 before = incl - occ, total = 0;
    for (int k = 0; k < kPdThreads / 64; k++) { if (k < w) before += s_w[k]; total += s_w[k]; }
    u32 j = before;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int sl = threadIdx.x * PER + k;
        const u32 e = s_idx[sl];
        if (e != 0u) {
            const u64 ri = a + (u64)((e & 0x7FFFFFFu) - 1u);
            s_idx[sl] = j;                          // (the slot's dense number from here on; only its own thread looks at it before the barrier)
            dkey[a + j] = skey[ri];
            dhi[a + j] = shi[ri];
            dcnt[a + j] = s_cnt[sl];
            j++;
        }
    }
    __syncthreads();
    if (s_fail) { if (threadIdx.x == 0) { *overflow = 1u; pcount[blockIdx.x] = 0; } return; }
#pragma unroll
    for (int k = 0; k < kPdMine; k++) {
        const u64 i = a + (u64)k * kPdThreads + threadIdx.x;
        if (i < b) lid[i] = myres[k] != kNoId ? s_idx[myres[k]] : kNoId;
    }
    for (u64 i = a + (u64)kPdMine * kPdThreads + threadIdx.x; i < b; i += kPdThreads) {
        const u32 sl = lid[i];
        if (sl != kNoId) lid[i] = s_idx[sl];
    }
    if (threadIdx.x == 0) pcount[blockIdx.x] = total;
}
template <class VALID>
inline void rec_dedupe(u64 nparts, const u64 *pstart, const u64 *skey, const u64 *shi, VALID valid, u32 *lid, u32 *pcount, u64 *dkey, u64 *dhi, u32 *dcnt,
                       u32 *overflow, const char *name = "rec_dedupe") {
    if (nparts == 0) return;
    prof_begin(name);
    hipLaunchKernelGGL((k_rec_dedupe<VALID>), dim3((unsigned)nparts), dim3(kPdThreads), 0, rt().stream, pstart, skey, shi, valid, lid, pcount, dkey, dhi, dcnt, overflow);
    prof_end();
    after_launch(name);
}

// ------------------------------------------------------------------ fixed-width record packing
// out[i * rec .. (i + 1) * rec) = the low `rec` bytes (1..8, little endian) of f(i).  A tile of 1024 records is assembled in LDS
// (the loads of f are coalesced: lane = record) and leaves as aligned 16-byte stores when `out` is 16-byte aligned (a tile is
// 1024 * rec bytes: always a multiple of 16).  (One lane per record with `rec` byte stores -- 64 lanes 5 bytes apart -- took 9 ms
// for the 1.66 G five-byte records of the 10 GB image.)
static constexpr int kPackTile = 1024;
template <class F>
__global__ void __launch_bounds__(kBlock) k_pack_records(u64 n, F f, u32 rec, u8 *out) {
    __shared__ __attribute__((aligned(16))) u8 s_b[kPackTile * 8];
    const u64 tiles = (n + kPackTile - 1) / kPackTile;
    for (u64 tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const u64 base = tile * kPackTile;
        const u32 cnt = n - base < (u64)kPackTile ? (u32)(n - base) : (u32)kPackTile;
#pragma unroll
        for (int j = 0; j < kPackTile / kBlock; j++) {
            const u32 k = (u32)j * kBlock + threadIdx.x;
            if (k < cnt) {
                const u64 v = f(base + k);
                for (u32 b = 0; b < rec; b++) s_b[k * rec + b] = (u8)(v >> (8 * b));
            }
        }
        __syncthreads();
        const u32 bytes = cnt * rec;
        u8 *dst = out + base * (u64)rec;
        if (((uintptr_t)dst & 15) == 0) {
            const u32 vecs = bytes / 16;
            for (u32 x = threadIdx.x; x < vecs; x += kBlock) reinterpret_cast<uint4 *>(dst)[x] = reinterpret_cast<const uint4 *>(s_b)[x];
            for (u32 x = vecs * 16 + threadIdx.x; x < bytes; x += kBlock) dst[x] = s_b[x];
        } else {
            for (u32 x = threadIdx.x; x < bytes; x += kBlock) dst[x] = s_b[x];
        }
        __syncthreads();
    }
}
template <class F>
inline void pack_records(u64 n, F f, u32 rec, u8 *out, const char *name = "pack_records") {
    if (n == 0) return;
    if (rec < 1 || rec > 8) throw Error(-22, "pack_records: record width out of range");
    prof_begin(name, n * rec);
    hipLaunchKernelGGL((k_pack_records<F>), dim3(grid_for((n + kPackTile - 1) / kPackTile, 1)), dim3(kBlock), 0, rt().stream, n, f, rec, out);
    prof_end();
    after_launch(name);
}

// ------------------------------------------------------------------ stream merge (induction pass C)
// A sequence of G segments in OUTPUT order.  Segment g is a literal run (sym, len) or a TAKE of `len` symbols from an axis T
// that the TAKE segments consume front to back, each where the one in front of it stopped.  T is given by its MAXIMAL runs
// (neighbours differ in symbol): erank(x) = run starts in [0, x), esym(k) / epos(k) = symbol / first position of run k.
// Result: the maximal runs of the concatenation, (symbol, first symbol position) per run -- written once, in order; nothing of
// segment or atom size is stored in between.
//     SEG:  REF locate(u64 g)  +  void fetch(REF, u32 &sym, IDX &len, bool &take)   (two dependent loads: the kernels issue all of a
//           lane's first loads, then all of its second ones);  u64 pre_before(u64 g) = segments in [0, g) that are no cells, REF plain(u64 t)
//           = cell t (tiles without such segments skip locate);  void eword(u64 w, u64 &bits, u64 &before) = word w of the run-start
//           vector over T and the starts in front of it;  u32 esym(u64 k);  u64 epos(u64 k)
// Three streaming passes over the segments, one workgroup per tile of kSmTile segments:
//     k_sm_sums    (sum of TAKE lengths, sum of lengths) per tile      -> scans: T position / symbol position at every tile start
//     k_sm_merge<false>  the tile's segments in LDS, a block scan gives every segment its T position; a TAKE segment
//                  [x, x + len) touches the runs erank(x + 1) - 1 .. erank(x + len) - 1 (two rank loads; neighbouring lanes hit
//                  the same words); a run head = an atom whose symbol differs from the atom in front of it -> heads per tile
//     k_sm_merge<true>   the same once more, now with the run index at every tile start: the heads are stored
// (Output order = segment order = T order: every access of the three passes runs forward through its array.  The form this
// replaces computed the T prefix of every cell with a scan of its own, stored it, gathered five arrays per cell to place
// packed atoms, wrote them, and merged them with another scan: 105 ms at level 0 of the 10 GB build against ~35.)
// segments per thread SPT = 4 or 8 (tiles of 1024 / 2048 segments), chosen per call: tiles of plain cells (level 0 of a read
// collection) run faster small -- 143 registers and three waves per SIMD at 8, seven at 4: emit 24.7 -> 19.9 ms, count 11.9 -> 9.9 --
// tiles that mix cells and pre-BWT runs faster large (the per-tile work -- two ranks in the kinds vector, the bases, four block
// barriers -- weighs more there: count 10.3 ms at 8, 13.7 at 4 on level 1 of the 10 GB build)
static constexpr u32 kSmNoSym = 0xFFFFFFFFu;
static constexpr u32 kSmInline = 8;                        // atoms behind the first a lane stores by itself; a TAKE that spans more runs of T is queued
                                                           // and copied by a kernel of its own, one lane per atom (a pre-BWT run of BWT markers can
                                                           // take tens of millions of runs: the deep levels of a read collection)
template <class IDX>
struct SmWide { IDX r, L, x, cnt; u64 k; IDX tile1; };     // run index / symbol position / T position of the segment, atoms behind the first, first of their runs;
                                                           // tile1 != 0 (one walk): r counts from the start of tile tile1 - 1, whose base the status words hold
template <class IDX>
struct SmPlan {
    u64 G = 0, tiles = 0;
    int spt = 8;
    IDX *xbase = nullptr, *lbase = nullptr, *hbase = nullptr;       // [tiles + 1] exclusive prefixes: TAKE symbols, symbols, run heads
    u32 *tlast = nullptr;                                           // [tiles] last symbol of every tile
    u64 take_total = 0, len_total = 0, heads = 0, atoms = 0;
    u64 wide_n = 0, wide_atoms = 0;                                 // queued TAKE segments and their atoms
    void release() {
        if (xbase) dev_free(xbase);
        if (lbase) dev_free(lbase);
        if (hbase) dev_free(hbase);
        if (tlast) dev_free(tlast);
        xbase = lbase = hbase = nullptr; tlast = nullptr;
    }
};
// (one WAVE per tile, eight loads in flight per lane, no barrier: with a workgroup per tile of 1024 segments this pass spent its
// time starting workgroups that lived for two dependent loads -- 6.4 ms for the 2.65 G cells of level 0 of the 10 GB build)
template <class SEG, class IDX, int SPT>
__global__ void __launch_bounds__(kBlock) k_sm_sums(u64 G, u64 tiles, SEG seg, IDX *tile_take, IDX *tile_len) {
    constexpr int kSmTile = kBlock * SPT;
    constexpr int TPW = 1;                          // tiles per wave (two small tiles per wave was measured slower: 7.2 vs 5.9 ms at level 0 of the 10 GB build)
    const int lane = threadIdx.x & 63;
    const u64 tile0 = ((u64)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * TPW;
    for (int tw = 0; tw < TPW; tw++) {
        const u64 tile = tile0 + tw;
        if (tile < tiles) {                              // (wave-uniform)
            const u64 base = tile * kSmTile;
            IDX a = 0, b = 0;
            for (int j0 = 0; j0 < kSmTile / 64; j0 += 8) {
                u32 sym[8]; IDX len[8]; bool take[8];
                typename SEG::Ref ref[8];
#pragma unroll
                for (int j = 0; j < 8; j++) { const u64 g = base + (u64)(j0 + j) * 64 + lane; ref[j] = seg.locate(g < G ? g : G - 1); }      // (clamped, not branched: all loads in flight)
#pragma unroll
                for (int j = 0; j < 8; j++) seg.fetch(ref[j], sym[j], len[j], take[j]);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const bool v = base + (u64)(j0 + j) * 64 + lane < G;
                    a += (v && take[j]) ? len[j] : (IDX)0;
                    b += v ? len[j] : (IDX)0;
                }
            }
            a = wave_reduce<IDX, Op::Sum>(a);
            b = wave_reduce<IDX, Op::Sum>(b);
            if (lane == 0) { tile_take[tile] = a; tile_len[tile] = b; }
        }
    }
}
// wide[]: [0] queued segments, [1] their atoms (count pass: totals; emit pass: wide[2] = queue fill).
// Count pass: the first segment of a tile counts as a head; tfirst / tlast (first and last symbol of the tile) let the caller
// take that back where the tile in front ends with the same symbol -- no look at the neighbouring tile from inside the kernel
// (three dependent loads on every tile's critical path when lane 0 worked it out by itself).
// The passes are bound by their instruction count (~100 vector instructions per segment in the first form, half of them 64-bit
// arithmetic in the 64-bit index build), so:
//   * positions inside a tile are offsets of type LT from the tile's bases -- 32 bits whenever the tile describes < 2^32 symbols
//     (a wave-uniform choice per tile);
//   * a tile without pre-BWT runs (level 0 of a read collection: all but 45 k of 2.6 G segments are cells) skips the per-segment
//     rank in the kinds vector;
//   * the two ranks of a TAKE segment come from ONE word of the T vector whenever its ends share it;
//   * the emit pass stages the tile's heads in LDS and writes them out linearly (a lane's heads are consecutive, so the direct
//     stores of a wave were 64 addresses 40 bytes apart: 24 ms for level 0 of the 10 GB build against 12 for the count pass).
// ---- decoupled look-back (the one-walk form of the stream merge).  A tile publishes 8-byte words that validate themselves
// (flag in the top two bits, written by ONE agent-scope store, read by agent-scope loads: a single naturally aligned granule
// needs no fence on either side, MI355X_MICROARCH.md "inter-workgroup visibility"):  0 = nothing yet, 1 = the tile's own sum,
// 2 = the sum of the tile and of every tile in front of it, 3 = the tile gave up (a wait ran out, or a tile in front gave up).
// Tiles are ordered by blockIdx: the hardware starts the workgroups of a grid in index order (per XCD queue), so the smallest
// unfinished tile is always running and waits only for finished ones.  HIP does not promise that order, hence the deadline:
// a tile that waits longer gives up, everybody behind it does the same at once, and the host takes the two-pass form.
// (Round 2 measured this shape for plain scans -- 2048 elements per tile -- and dropped it: the publication chain across eight
// L2s cost more than the second read of the inputs.  A tile of the stream merge is 10-30 us of work; there it pays.)
static constexpr u64 kLbAgg = 1ull << 62, kLbPre = 2ull << 62, kLbBad = 3ull << 62, kLbVal = (1ull << 62) - 1ull;
GRL_DEV void lb_store(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
GRL_DEV u64 lb_load(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
struct SmLb {
    u64 *st_head = nullptr;      // [tiles] flag | run heads (own, then inclusive)
    u64 *res = nullptr;          // [0] run heads in all, [1] != 0: some tile gave up
    u64 patience = 0;            // ticks of wall_clock64() (100 MHz) a tile waits for another one
};
// one wave: the sum of the values of tiles [0, t).  A round looks at the 64 x W tiles in front of `hi` (lane l at tiles
// hi - 1 - l - 64 j), nearest first; the first inclusive word ends the walk.  false: a tile in front gave up or the deadline passed.
// The window has to cover the tiles that have published their own sum but not yet their inclusive one -- the ones that are in
// THEIR look-back: throughput in tiles per microsecond x the latency of a look-back.  And the waiting has to be quiet: the status
// words of the tiles in flight are a few KB -- a handful of memory channels -- and every agent-scope load is a trip to the memory
// side, so a wave re-reads only the words it is still waiting for and sleeps longer every time.  (First form: all lanes re-read
// their words every 0.1 us.  Level 0 of the 10 GB build -- 2.6 M tiles of 1024 segments, 100+ tiles per microsecond -- then ran at
// the rate of the look-back: 39.5 ms with 64 tiles per round, 44.4 ms with 256, against 30.3 ms for count + emit.)
template <int W>
GRL_DEV bool lb_lookback(const u64 *status, u64 t, u64 patience, u64 &excl) {
    const int lane = threadIdx.x & 63;
    const u64 t0 = wall_clock64();
    u64 sum = 0, hi = t;
    bool ok = true, done = false;
    while (hi > 0 && ok && !done) {
        u64 v[W];
#pragma unroll
        for (int j = 0; j < W; j++) v[j] = 0;
        bool ready = false;
        int jp = W;                              // first row that holds an inclusive word (W: none)
        unsigned long long upto = ~0ull;         // ... and the lanes of that row up to and including it
        int nap = 1;
        while (!ready && ok) {
#pragma unroll
            for (int j = 0; j < W; j++) {
                const u64 back = (u64)lane + 64ull * (u64)j;                    // tile hi - 1 - back
                if (back >= hi) v[j] = kLbPre;                                  // (in front of tile 0: an inclusive zero)
                else if ((v[j] >> 62) == 0 && j <= jp) v[j] = lb_load(status + (hi - 1 - back));
            }
            ready = true; jp = W; upto = ~0ull;
#pragma unroll
            for (int j = 0; j < W; j++) {
                if (jp == W) {                                                  // (rows behind the first inclusive word do not matter)
                    const unsigned long long pm = __ballot((v[j] >> 62) >= 2), zm = __ballot((v[j] >> 62) == 0);
                    const unsigned long long up = pm ? (((pm & (~pm + 1ull)) << 1) - 1ull) : ~0ull;
                    if (zm & up) ready = false;
                    if (pm) { jp = j; upto = up; }
                }
            }
            if (!ready) {
                if (wall_clock64() - t0 > patience) ok = false;
                else {
                    for (int z = 0; z < nap; z++) __builtin_amdgcn_s_sleep(16);     // 16 x 64 clocks ~ 0.4 us per unit
                    if (nap < 8) nap++;
                }
            }
        }
        if (ok) {
            u64 part = 0;
            bool bad = false;
#pragma unroll
            for (int j = 0; j < W; j++) {
                const bool take = j < jp || (j == jp && ((upto >> lane) & 1ull));
                if (take) { part += v[j] & kLbVal; bad = bad || (v[j] >> 62) == 3; }
            }
            if (__ballot(bad)) ok = false;
            part = wave_reduce<u64, Op::Sum>(part);
            sum += __shfl(part, 0, 64);                                         // (the reduction lands in lane 0)
            done = jp != W;
            hi = hi > 64ull * W ? hi - 64ull * W : 0;
        }
    }
    excl = sum;
    return ok;
}
template <class LT, int SPT> struct SmShared {
    static constexpr int kSmTile = kBlock * SPT;
    u32 sym[kSmTile + kSmTile / 32];            // bit 31: TAKE (symbols are < 2^30); skewed by one slot per 32.  Emit: the staged heads' symbols
    LT len[kSmTile + kSmTile / 32];             // ... and their positions
    Pair<LT, LT> w2[4];
    u32 w1[4];
    u32 last[kBlock];
};
// MODE 0: count pass, 1: emit pass of the two-pass form, 2: ONE WALK (round 6) -- count and emit together; the run index at the tile's
// start (hb) is not an input but comes from a decoupled look-back over per-tile status words (SmLb), see k_sm_merge
template <class LT, class SEG, class IDX, int MODE, int SPT>
GRL_DEV void sm_tile(u64 G, const SEG &seg, const u64 base, const IDX xb, const IDX lb, IDX hb, const u32 prev_tile, const bool plain, const u64 ord0,
                     SmShared<LT, SPT> &S, IDX *tile_heads, IDX *tile_atoms, u32 *tfirst, u32 *tlast, unsigned long long *wide, SmWide<IDX> *queue,
                     u64 queue_cap, u32 *osym, IDX *ostart, const SmLb &lbk) {
    constexpr bool EMIT = MODE != 0;
    constexpr int TILE = kBlock * SPT;
    typedef Pair<LT, LT> P2;
    // striped loads (neighbouring lanes, neighbouring segments) -> LDS -> blocked (a lane's segments are consecutive)
    {
        u32 sym[SPT]; IDX len[SPT]; bool take[SPT];
        typename SEG::Ref ref[SPT];
        if (plain) {
#pragma unroll
            for (int j = 0; j < SPT; j++) { const u64 g = base + (u32)j * kBlock + threadIdx.x; ref[j] = seg.plain((g < G ? g : G - 1) - ord0); }
        } else {
#pragma unroll
            for (int j = 0; j < SPT; j++) { const u64 g = base + (u32)j * kBlock + threadIdx.x; ref[j] = seg.locate(g < G ? g : G - 1); }      // (clamped, not branched: all loads in flight)
        }
#pragma unroll
        for (int j = 0; j < SPT; j++) seg.fetch(ref[j], sym[j], len[j], take[j]);
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            const u32 k = (u32)j * kBlock + threadIdx.x;
            const bool v = base + k < G;
            S.sym[k + (k >> 5)] = v ? (sym[j] | (take[j] ? 0x80000000u : 0u)) : 0u;
            S.len[k + (k >> 5)] = v ? (LT)len[j] : (LT)0;
        }
    }
    __syncthreads();
    u32 sy[SPT]; LT ln[SPT];
    P2 acc(0);
#pragma unroll
    for (int i = 0; i < SPT; i++) {
        const u32 k = threadIdx.x * SPT + i;
        sy[i] = S.sym[k + (k >> 5)];
        ln[i] = S.len[k + (k >> 5)];
        acc.a += (sy[i] & 0x80000000u) ? ln[i] : (LT)0;
        acc.b += ln[i];
    }
    P2 tot;
    const P2 ex = block_excl_scan<P2>(acc, S.w2, &tot);
    const u64 g0 = base + (u64)threadIdx.x * SPT;
    LT xs[SPT], Ls[SPT];                     // offsets from (xb, lb)
    {
        LT x = ex.a, L = ex.b;
#pragma unroll
        for (int i = 0; i < SPT; i++) { xs[i] = x; Ls[i] = L; x += (sy[i] & 0x80000000u) ? ln[i] : (LT)0; L += ln[i]; }
    }
    // runs of T every TAKE segment touches: the rank word of its first symbol, and of its end where that is another word
    u64 k0[SPT]; u32 inner[SPT];             // first run touched; runs touched behind it (saturated: a segment with >= 2^32 - 1 inner runs is split by nobody -- see below)
    u64 k1m[SPT];                            // last run touched  (dropping this array for a recount in the saturated case made the compiler use MORE registers: 143 -> 160)
    {
        u64 aw[SPT], ab[SPT], bw[SPT], bb[SPT];
#pragma unroll
        for (int i = 0; i < SPT; i++) {
            const bool tk = (g0 + i < G) && (sy[i] & 0x80000000u);
            const u64 x1 = (u64)xb + (u64)xs[i] + 1;
            aw[i] = 0; ab[i] = 0;
            if (tk) seg.eword(x1 >> 6, aw[i], ab[i]);
        }
#pragma unroll
        for (int i = 0; i < SPT; i++) {
            const bool tk = (g0 + i < G) && (sy[i] & 0x80000000u);
            const u64 x1 = (u64)xb + (u64)xs[i] + 1, xe = x1 - 1 + (u64)ln[i];
            bw[i] = aw[i]; bb[i] = ab[i];
            if (tk && (xe >> 6) != (x1 >> 6)) seg.eword(xe >> 6, bw[i], bb[i]);
        }
#pragma unroll
        for (int i = 0; i < SPT; i++) {
            const bool tk = (g0 + i < G) && (sy[i] & 0x80000000u);
            const u64 x1 = (u64)xb + (u64)xs[i] + 1, xe = x1 - 1 + (u64)ln[i];
            const u64 ka = ab[i] + (u64)__builtin_popcountll(aw[i] & ((1ull << (x1 & 63)) - 1ull)) - 1;       // run holding the first symbol
            const u64 kb = bb[i] + (u64)__builtin_popcountll(bw[i] & ((1ull << (xe & 63)) - 1ull));           // run starts in [0, x + len)
            k0[i] = tk ? ka : 0;
            k1m[i] = tk ? kb - 1 : 0;
            const u64 in64 = tk ? kb - 1 - ka : 0;
            inner[i] = in64 > 0xFFFFFFFEull ? 0xFFFFFFFFu : (u32)in64;
        }
    }
    u32 fs[SPT], ls[SPT];
#pragma unroll
    for (int i = 0; i < SPT; i++) {
        const bool tk = (g0 + i < G) && (sy[i] & 0x80000000u);
        fs[i] = tk ? seg.esym(k0[i]) : (sy[i] & 0x7FFFFFFFu);
    }
#pragma unroll
    for (int i = 0; i < SPT; i++) ls[i] = inner[i] ? seg.esym(k1m[i]) : fs[i];
    // the symbol in front of my first segment: the last symbol of the lane in front of me; lane 0: of the tile in front
    {
        u32 mine = kSmNoSym;
#pragma unroll
        for (int i = 0; i < SPT; i++) if (g0 + i < G) mine = ls[i];
        S.last[threadIdx.x] = mine;
    }
    __syncthreads();
    u32 prev = threadIdx.x > 0 ? S.last[threadIdx.x - 1] : prev_tile;       // (count pass, lane 0: no symbol -> its first segment counts as a head)
    bool head[SPT];
    u64 nh = 0, na = 0;                      // (64 bits: a single segment can hold 2^32 runs in the 64-bit build)
    u32 nwide = 0;
    unsigned long long wide_atoms = 0;
#pragma unroll
    for (int i = 0; i < SPT; i++) {
        const bool v = g0 + i < G;
        const u64 in64 = inner[i] == 0xFFFFFFFFu ? k1m[i] - k0[i] : (u64)inner[i];
        head[i] = v && fs[i] != prev;
        nh += (head[i] ? 1u : 0u) + in64;
        na += (v ? 1u : 0u) + in64;
        if (in64 > (u64)kSmInline) { nwide++; wide_atoms += in64; }
        if (v) prev = ls[i];
    }
    if constexpr (!EMIT) {
        nh = wave_reduce<u64, Op::Sum>(nh);
        na = wave_reduce<u64, Op::Sum>(na);
        const unsigned long long anyw = __ballot(nwide != 0);
        if (anyw) {                            // (rare: one pair of atomics per wave that holds a wide segment)
            nwide = wave_reduce<u32, Op::Sum>(nwide);
            wide_atoms = wave_reduce<unsigned long long, Op::Sum>(wide_atoms);
            if ((threadIdx.x & 63) == 0) { atomicAdd(&wide[0], (unsigned long long)nwide); atomicAdd(&wide[1], wide_atoms); }
        }
        __shared__ u64 s_r[4][2];
        if ((threadIdx.x & 63) == 0) { s_r[threadIdx.x >> 6][0] = nh; s_r[threadIdx.x >> 6][1] = na; }
        __syncthreads();
        if (threadIdx.x == 0) {
            tile_heads[blockIdx.x] = (IDX)(s_r[0][0] + s_r[1][0] + s_r[2][0] + s_r[3][0]);
            tile_atoms[blockIdx.x] = (IDX)(s_r[0][1] + s_r[1][1] + s_r[2][1] + s_r[3][1]);
            tfirst[blockIdx.x] = fs[0];
            // the tile's last symbol: of the last lane that holds a segment (the lanes behind it hold none)
            const u64 nin = G - base < (u64)TILE ? G - base : (u64)TILE;
            tlast[blockIdx.x] = S.last[(nin - 1) / SPT];
        }
    } else {
        __shared__ Pair<u64, u64> s_h[4];
        __shared__ u64 s_lb[2];
        Pair<u64, u64> htot2;
        const u64 hl0 = block_excl_scan<Pair<u64, u64>>(Pair<u64, u64>(nh, na), s_h, &htot2).a;       // my first head, counted from the tile's
        const u64 htot = htot2.a;
        // ONE WALK: my heads are exact (the symbol in front of the tile was worked out by thread 0 at the start: k_sm_merge), so
        // the tile's own sum goes out at once.  The look-back itself comes as LATE as possible -- behind the loop that stages the
        // heads in LDS, whose inline atoms are the tile's last gathers: by then the tiles in front have published, and a second
        // trip to their words (each one a round trip to the memory side) is rare.  Only a tile whose heads do not fit LDS needs
        // its base before that loop; the queued segments carry their place relative to the tile (SmWide::tile1) and the launch
        // that copies them adds the base.
        const bool staged = htot <= (u64)TILE;                       // (uniform) the tile's heads fit the LDS arrays the segments came through
        u64 hbm = (u64)hb;
        bool live = true;
        auto look_back = [&] {                    // (all threads; a barrier inside)
            if (threadIdx.x < 64) {
                const u64 t = blockIdx.x;
                u64 excl = 0;
                const bool ok = lb_lookback<(SPT >= 8 ? 1 : 4)>(lbk.st_head, t, lbk.patience, excl);
                if (threadIdx.x == 0) {
                    lb_store(lbk.st_head + t, ok ? (kLbPre | ((excl + htot) & kLbVal)) : kLbBad);
                    if (!ok) lb_store(lbk.res + 1, 1ull);
                    else if (t + 1 == gridDim.x) lbk.res[0] = excl + htot;
                    s_lb[0] = excl; s_lb[1] = ok ? 1 : 0;
                }
            }
            __syncthreads();
            hbm = s_lb[0]; live = s_lb[1] != 0;
        };
        if constexpr (MODE == 2) {
            if (threadIdx.x == 0) { lb_store(lbk.st_head + blockIdx.x, kLbAgg | (htot & kLbVal)); tile_atoms[blockIdx.x] = (IDX)htot2.b; }
            const unsigned long long anyw = __ballot(nwide != 0);
            if (anyw) {                            // (rare: the atoms of the queued segments, for the launch that copies them)
                wide_atoms = wave_reduce<unsigned long long, Op::Sum>(wide_atoms);
                if ((threadIdx.x & 63) == 0) atomicAdd(&wide[1], wide_atoms);
            }
            if (!staged) look_back();
        }
        const IDX tile1 = (MODE == 2 && staged) ? (IDX)blockIdx.x + 1 : (IDX)0;      // queued segments: place relative to this tile
        const u64 qbase_r = (MODE == 2 && staged) ? 0 : hbm;
        if (live) {
        if (staged) {
            for (u32 k = threadIdx.x; k < (u32)htot; k += kBlock) S.sym[k] = kSmNoSym;      // (places of queued atoms stay marked: the wide kernel writes them)
            __syncthreads();
        }
        u64 r = hl0;
#pragma unroll
        for (int i = 0; i < SPT; i++) {
            if (g0 + i < G) {
                if (head[i]) {
                    if (staged) { S.sym[r] = fs[i]; S.len[r] = Ls[i]; }
                    else { osym[hbm + r] = fs[i]; ostart[hbm + r] = lb + (IDX)Ls[i]; }
                    r++;
                }
                if (sy[i] & 0x80000000u) {
                    const u64 in64 = inner[i] == 0xFFFFFFFFu ? k1m[i] - k0[i] : (u64)inner[i];
                    if (in64 > (u64)kSmInline) {
                        const u64 q = (u64)atomicAdd(&wide[2], 1ull);
                        if (q < queue_cap) queue[q] = SmWide<IDX>{(IDX)(qbase_r + r), lb + (IDX)Ls[i], xb + (IDX)xs[i], (IDX)in64, k0[i] + 1, tile1};
                    } else {
                        // (all of a segment's loads in flight first, clamped instead of branched, was tried: no faster -- 18.9 vs 19.4 ms on
                        // level 1 of the 10 GB build -- and 24 more registers)
                        const u64 x = (u64)xb + (u64)xs[i];
                        for (u32 a = 0; a < (u32)in64; a++) {
                            const u64 k = k0[i] + 1 + (u64)a;
                            const u32 sk = seg.esym(k);
                            const u64 off = seg.epos(k) - x;            // (< the segment's length)
                            if (staged) { S.sym[r + a] = sk; S.len[r + a] = Ls[i] + (LT)off; }
                            else { osym[hbm + r + a] = sk; ostart[hbm + r + a] = lb + (IDX)Ls[i] + (IDX)off; }
                        }
                    }
                    r += in64;
                }
            }
        }
        }
        if (staged) {
            if constexpr (MODE == 2) look_back();            // (its barrier is the one the write-out needs)
            else __syncthreads();
            if (live) {
                for (u32 k = threadIdx.x; k < (u32)htot; k += kBlock) {
                    const u32 sk = S.sym[k];
                    if (sk != kSmNoSym) { osym[hbm + k] = sk; ostart[hbm + k] = lb + (IDX)S.len[k]; }
                }
            }
        }
    }
}
template <class SEG, class IDX, int MODE, int SPT>
__global__ void __launch_bounds__(kBlock) k_sm_merge(u64 G, SEG seg, const IDX *xbase, const IDX *lbase, const IDX *hbase, IDX *tile_heads, IDX *tile_atoms,
                                                     u32 *tfirst, u32 *tlast, unsigned long long *wide, SmWide<IDX> *queue, u64 queue_cap, u32 *osym, IDX *ostart, SmLb lbk) {
    constexpr int kSmTile = kBlock * SPT;
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[sizeof(SmShared<IDX, SPT>)];
    const u64 base = (u64)blockIdx.x * kSmTile;
    // (loaded first: nothing below has to wait for them)
    const IDX xb = xbase[blockIdx.x], lb = lbase[blockIdx.x], lnext = lbase[blockIdx.x + 1];
    IDX hb = 0;
    u32 prev_tile = kSmNoSym;
    if constexpr (MODE == 1) { hb = hbase[blockIdx.x]; if (blockIdx.x > 0) prev_tile = tlast[blockIdx.x - 1]; }
    if constexpr (MODE == 2) {
        // the last symbol of the segment in front of the tile, by thread 0 itself: its record, and where it is a TAKE -- it ends where my
        // tile's T axis begins -- the run of T in front of xb.  Three dependent loads, issued before everything else: they are back when
        // the tile's own chain of four is.  (The count pass of the two-pass form leaves this to the scan over the tiles: there the
        // loads sat at the end of the tile.)
        if (threadIdx.x == 0 && blockIdx.x > 0) {
            u32 sym; IDX len; bool take;
            seg.fetch(seg.locate(base - 1), sym, len, take);
            if (take) {
                u64 bits, before;
                seg.eword((u64)xb >> 6, bits, before);
                sym = seg.esym(before + (u64)__builtin_popcountll(bits & ((1ull << ((u64)xb & 63)) - 1ull)) - 1);      // run starts in [0, xb) - 1
            }
            prev_tile = sym;
        }
    }
    const u64 gend = base + kSmTile < G ? base + kSmTile : G;
    const u64 ord0 = seg.pre_before(base), ord1 = seg.pre_before(gend);
    const bool plain = ord0 == ord1;                                   // (uniform) no pre-BWT run among the tile's segments
    if (sizeof(IDX) == 4 || (u64)(lnext - lb) >= 0xFFFFFFFFull)        // (uniform) offsets inside the tile in the index width ...
        sm_tile<IDX, SEG, IDX, MODE, SPT>(G, seg, base, xb, lb, hb, prev_tile, plain, ord0, *reinterpret_cast<SmShared<IDX, SPT> *>(s_raw), tile_heads, tile_atoms, tfirst, tlast,
                                     wide, queue, queue_cap, osym, ostart, lbk);
    else                                                               // ... or in 32 bits when the tile describes < 2^32 symbols
        sm_tile<u32, SEG, IDX, MODE, SPT>(G, seg, base, xb, lb, hb, prev_tile, plain, ord0, *reinterpret_cast<SmShared<u32, SPT> *>(s_raw), tile_heads, tile_atoms, tfirst, tlast,
                                     wide, queue, queue_cap, osym, ostart, lbk);
}
// heads of tile t without the provisional head of its first segment where the tile in front ends with the same symbol
template <class IDX>
struct SmHeadsIn {
    const IDX *heads; const u32 *tfirst; const u32 *tlast;
    GRL_DEV IDX operator()(u64 t) const { return heads[t] - ((t > 0 && tfirst[t] == tlast[t - 1]) ? (IDX)1 : (IDX)0); }
};
template <class IDX>
struct SmWideCountIn {
    const SmWide<IDX> *q;
    GRL_DEV u64 operator()(u64 i) const { return (u64)q[i].cnt; }
};
// the atoms of the queued TAKE segments, one lane per atom (entries in queue order, qbase = exclusive prefix of their counts)
template <class SEG, class IDX>
__global__ void __launch_bounds__(kBlock) k_sm_wide(u64 natoms, u64 nq, SEG seg, const SmWide<IDX> *queue, const u64 *qbase, u32 *osym, IDX *ostart,
                                                    const u64 *st_head /* one walk: inclusive run counts per tile */) {
    const u64 stride = (u64)gridDim.x * kBlock;
    for (u64 y = (u64)blockIdx.x * kBlock + threadIdx.x; y < natoms; y += stride) {
        u64 lo = 0, hi = nq;                   // last entry with qbase <= y
        while (lo + 1 < hi) { const u64 mid = (lo + hi) >> 1; if (qbase[mid] <= y) lo = mid; else hi = mid; }
        const SmWide<IDX> e = queue[lo];
        const u64 a = y - qbase[lo], k = e.k + a;
        u64 r = (u64)e.r + a;
        if (e.tile1 > (IDX)1) r += st_head[(u64)e.tile1 - 2] & kLbVal;       // (runs in front of the segment's tile = inclusive count of the tile in front of it)
        osym[r] = seg.esym(k);
        ostart[r] = e.L + (IDX)(seg.epos(k) - (u64)e.x);
    }
}
// passes 1 + 2: the plan's prefixes and totals (one host synchronisation); pass 3 follows through stream_merge_emit
template <class SEG, class IDX>
inline void stream_merge_count(u64 G, SEG seg, SmPlan<IDX> &plan, const char *name = "stream_merge", bool mostly_plain = false) {
    plan.release();
    plan = SmPlan<IDX>();
    plan.G = G;
    if (G == 0) return;
    static const int spt_env = dev_env("GRLBWT_SM_SPT") ? atoi(dev_env("GRLBWT_SM_SPT")) : 0;      // (experiments: 4 or 8 segments per thread at every level)
    plan.spt = spt_env == 4 || spt_env == 8 ? spt_env : (mostly_plain ? 4 : 8);
    const u64 kSmTile = (u64)kBlock * plan.spt;
    plan.tiles = (G + kSmTile - 1) / kSmTile;
    const u64 T = plan.tiles;
    plan.xbase = (IDX *)dev_alloc((T + 1) * sizeof(IDX));
    plan.lbase = (IDX *)dev_alloc((T + 1) * sizeof(IDX));
    plan.hbase = (IDX *)dev_alloc((T + 1) * sizeof(IDX));
    plan.tlast = (u32 *)dev_alloc(T * sizeof(u32));
    IDX *tatoms = (IDX *)dev_alloc((T + 1) * sizeof(IDX));
    u32 *tfirst = (u32 *)dev_alloc(T * sizeof(u32));
    prof_begin(std::string(name) + ".sums");
    if (plan.spt == 4) hipLaunchKernelGGL((k_sm_sums<SEG, IDX, 4>), dim3((unsigned)((T + kBlock / 64 - 1) / (kBlock / 64))), dim3(kBlock), 0, rt().stream, G, T, seg, plan.xbase, plan.lbase);
    else hipLaunchKernelGGL((k_sm_sums<SEG, IDX, 8>), dim3((unsigned)((T + kBlock / 64 - 1) / (kBlock / 64))), dim3(kBlock), 0, rt().stream, G, T, seg, plan.xbase, plan.lbase);
    prof_end();
    after_launch(name);
    u64 *dres = (u64 *)dev_alloc(6 * sizeof(u64));      // [0] TAKE symbols, [1] symbols, [2] heads, [3] atoms, [4] wide segments, [5] their atoms
    dev_memset(dres, 0, 6 * sizeof(u64));               // (the scans store IDX-wide totals into zeroed words)
    exclusive_scan_async<IDX, PtrIn<IDX>>(T, PtrIn<IDX>{plan.xbase}, plan.xbase, (IDX *)(dres + 0), plan.xbase + T, name);
    exclusive_scan_async<IDX, PtrIn<IDX>>(T, PtrIn<IDX>{plan.lbase}, plan.lbase, (IDX *)(dres + 1), plan.lbase + T, name);
    prof_begin(std::string(name) + ".count");
    if (plan.spt == 4)
        hipLaunchKernelGGL((k_sm_merge<SEG, IDX, 0, 4>), dim3((unsigned)T), dim3(kBlock), 0, rt().stream, G, seg, (const IDX *)plan.xbase, (const IDX *)plan.lbase,
                           (const IDX *)nullptr, plan.hbase, tatoms, tfirst, plan.tlast, (unsigned long long *)(dres + 4), (SmWide<IDX> *)nullptr, (u64)0, (u32 *)nullptr, (IDX *)nullptr, SmLb());
    else
        hipLaunchKernelGGL((k_sm_merge<SEG, IDX, 0, 8>), dim3((unsigned)T), dim3(kBlock), 0, rt().stream, G, seg, (const IDX *)plan.xbase, (const IDX *)plan.lbase,
                           (const IDX *)nullptr, plan.hbase, tatoms, tfirst, plan.tlast, (unsigned long long *)(dres + 4), (SmWide<IDX> *)nullptr, (u64)0, (u32 *)nullptr, (IDX *)nullptr, SmLb());
    prof_end();
    after_launch(name);
    exclusive_scan_async<IDX, SmHeadsIn<IDX>>(T, SmHeadsIn<IDX>{plan.hbase, tfirst, plan.tlast}, plan.hbase, (IDX *)(dres + 2), plan.hbase + T, name);
    exclusive_scan_async<IDX, PtrIn<IDX>>(T, PtrIn<IDX>{tatoms}, tatoms, (IDX *)(dres + 3), (IDX *)nullptr, name);
    u64 h[6];
    d2h(h, dres, 6 * sizeof(u64));
    dev_free(dres); dev_free(tatoms); dev_free(tfirst);
    plan.take_total = h[0]; plan.len_total = h[1]; plan.heads = h[2]; plan.atoms = h[3]; plan.wide_n = h[4]; plan.wide_atoms = h[5];
}
template <class SEG, class IDX>
inline void stream_merge_emit(SEG seg, SmPlan<IDX> &plan, u32 *osym, IDX *ostart, const char *name = "stream_merge") {
    if (plan.G == 0) return;
    const u64 nq = plan.wide_n;
    SmWide<IDX> *queue = (SmWide<IDX> *)dev_alloc((nq ? nq : 1) * sizeof(SmWide<IDX>));
    unsigned long long *wide = (unsigned long long *)dev_alloc(3 * sizeof(unsigned long long));
    dev_memset(wide, 0, 3 * sizeof(unsigned long long));
    prof_begin(std::string(name) + ".emit", plan.heads * (sizeof(u32) + sizeof(IDX)));
    if (plan.spt == 4)
        hipLaunchKernelGGL((k_sm_merge<SEG, IDX, 1, 4>), dim3((unsigned)plan.tiles), dim3(kBlock), 0, rt().stream, plan.G, seg, (const IDX *)plan.xbase,
                           (const IDX *)plan.lbase, (const IDX *)plan.hbase, (IDX *)nullptr, (IDX *)nullptr, (u32 *)nullptr, plan.tlast, wide, queue, nq, osym, ostart, SmLb());
    else
        hipLaunchKernelGGL((k_sm_merge<SEG, IDX, 1, 8>), dim3((unsigned)plan.tiles), dim3(kBlock), 0, rt().stream, plan.G, seg, (const IDX *)plan.xbase,
                           (const IDX *)plan.lbase, (const IDX *)plan.hbase, (IDX *)nullptr, (IDX *)nullptr, (u32 *)nullptr, plan.tlast, wide, queue, nq, osym, ostart, SmLb());
    prof_end();
    after_launch(name);
    if (nq) {
        u64 *qbase = (u64 *)dev_alloc((nq + 1) * sizeof(u64));
        exclusive_scan_async<u64, SmWideCountIn<IDX>>(nq, SmWideCountIn<IDX>{queue}, qbase, (u64 *)nullptr, qbase + nq, name);
        prof_begin(std::string(name) + ".wide", plan.wide_atoms * (sizeof(u32) + sizeof(IDX)));
        hipLaunchKernelGGL((k_sm_wide<SEG, IDX>), dim3(grid_for(plan.wide_atoms, kBlock)), dim3(kBlock), 0, rt().stream, plan.wide_atoms, nq, seg,
                           (const SmWide<IDX> *)queue, (const u64 *)qbase, osym, ostart, (const u64 *)nullptr);
        prof_end();
        after_launch(name);
        dev_free(qbase);
    }
    dev_free(queue); dev_free(wide);
}

// Status words of the look-back: memory of the runtime's own allocator (hipMalloc), kept between calls -- agent-scope polling
// across the XCDs' L2s is what the microarchitecture guide measured on such memory; the engine's arena is mapped memory.
inline u64 *lookback_scratch(size_t words) {
    static u64 *buf = nullptr;
    static size_t cap = 0;
    static int dev = -1;
    if (dev != rt().device) { buf = nullptr; cap = 0; dev = rt().device; }      // (another device: the old block stays with its device)
    if (words > cap) {
        if (buf) { GRL_HIP_CHECK(hipStreamSynchronize(rt().stream)); (void)hipFree(buf); buf = nullptr; cap = 0; }
        size_t want = words + words / 4 + 4096;
        GRL_HIP_CHECK(hipMalloc((void **)&buf, want * sizeof(u64)));
        cap = want;
    }
    return buf;
}
// The one-walk form: sums + scans as in stream_merge_count, then ONE merge kernel that finds the run heads and writes them; the
// run index at a tile's start comes from the look-back.  osym / ostart must hold `out_cap` entries (an upper bound of the runs:
// segments + runs of T).  Returns false when the walk gave up -- the queue of wide segments overflowed, or a tile waited too long
// for another one -- and the caller takes stream_merge_count + stream_merge_emit; the plan then holds nothing.
template <class SEG, class IDX>
inline bool stream_merge_onepass(u64 G, SEG seg, SmPlan<IDX> &plan, u32 *osym, IDX *ostart, u64 out_cap, u64 queue_cap, const char *name = "stream_merge",
                                 bool mostly_plain = false) {
    plan.release();
    plan = SmPlan<IDX>();
    plan.G = G;
    if (G == 0) return true;
    static const int spt_dev = dev_env("GRLBWT_DEV_SM1_SPT") ? atoi(dev_env("GRLBWT_DEV_SM1_SPT")) : 0;
    plan.spt = spt_dev == 4 || spt_dev == 8 ? spt_dev : (mostly_plain ? 4 : 8);
    const u64 kSmTile = (u64)kBlock * plan.spt;
    plan.tiles = (G + kSmTile - 1) / kSmTile;
    const u64 T = plan.tiles;
    plan.xbase = (IDX *)dev_alloc((T + 1) * sizeof(IDX));
    plan.lbase = (IDX *)dev_alloc((T + 1) * sizeof(IDX));
    IDX *tatoms = (IDX *)dev_alloc((T + 1) * sizeof(IDX));
    prof_begin(std::string(name) + ".sums");
    if (plan.spt == 4) hipLaunchKernelGGL((k_sm_sums<SEG, IDX, 4>), dim3((unsigned)((T + kBlock / 64 - 1) / (kBlock / 64))), dim3(kBlock), 0, rt().stream, G, T, seg, plan.xbase, plan.lbase);
    else hipLaunchKernelGGL((k_sm_sums<SEG, IDX, 8>), dim3((unsigned)((T + kBlock / 64 - 1) / (kBlock / 64))), dim3(kBlock), 0, rt().stream, G, T, seg, plan.xbase, plan.lbase);
    prof_end();
    after_launch(name);
    u64 *dres = (u64 *)dev_alloc(8 * sizeof(u64));      // [0] TAKE symbols, [1] symbols, [2] atoms, [3] -, [4] -, [5] atoms of the queued segments, [6] queued segments
    dev_memset(dres, 0, 8 * sizeof(u64));               // (the scans store IDX-wide totals into zeroed words)
    exclusive_scan_async<IDX, PtrIn<IDX>>(T, PtrIn<IDX>{plan.xbase}, plan.xbase, (IDX *)(dres + 0), plan.xbase + T, name);
    exclusive_scan_async<IDX, PtrIn<IDX>>(T, PtrIn<IDX>{plan.lbase}, plan.lbase, (IDX *)(dres + 1), plan.lbase + T, name);
    u64 *st = lookback_scratch(T + 2);
    dev_memset(st, 0, (T + 2) * sizeof(u64));
    SmLb lbk;
    lbk.st_head = st; lbk.res = st + T;
    lbk.patience = 200000000ull;                        // two seconds of the 100 MHz clock
    if (const char *pt = dev_env("GRLBWT_DEV_LB_PATIENCE")) lbk.patience = (u64)atoll(pt);      // (development builds: tools/gpu_lookback_giveup.py makes the tiles give up)
    SmWide<IDX> *queue = (SmWide<IDX> *)dev_alloc((queue_cap ? queue_cap : 1) * sizeof(SmWide<IDX>));
    unsigned long long *wide = (unsigned long long *)(dres + 4);      // [1] atoms of the queued segments, [2] queue fill
    (void)out_cap;
    prof_begin(std::string(name) + ".walk");
    if (plan.spt == 4)
        hipLaunchKernelGGL((k_sm_merge<SEG, IDX, 2, 4>), dim3((unsigned)T), dim3(kBlock), 0, rt().stream, G, seg, (const IDX *)plan.xbase, (const IDX *)plan.lbase,
                           (const IDX *)nullptr, (IDX *)nullptr, tatoms, (u32 *)nullptr, (u32 *)nullptr, wide, queue, queue_cap, osym, ostart, lbk);
    else
        hipLaunchKernelGGL((k_sm_merge<SEG, IDX, 2, 8>), dim3((unsigned)T), dim3(kBlock), 0, rt().stream, G, seg, (const IDX *)plan.xbase, (const IDX *)plan.lbase,
                           (const IDX *)nullptr, (IDX *)nullptr, tatoms, (u32 *)nullptr, (u32 *)nullptr, wide, queue, queue_cap, osym, ostart, lbk);
    prof_end();
    after_launch(name);
    exclusive_scan_async<IDX, PtrIn<IDX>>(T, PtrIn<IDX>{tatoms}, tatoms, (IDX *)(dres + 2), (IDX *)nullptr, name);
    d2d(dres + 7, lbk.res, sizeof(u64));                // run heads in all
    d2d(dres + 3, lbk.res + 1, sizeof(u64));            // a tile gave up
    u64 h[8];
    d2h(h, dres, 8 * sizeof(u64));
    dev_free(dres); dev_free(tatoms);
    plan.take_total = h[0]; plan.len_total = h[1]; plan.atoms = h[2]; plan.heads = h[7]; plan.wide_atoms = h[5]; plan.wide_n = h[6];
    const bool gave_up = h[3] != 0 || h[6] > queue_cap || plan.heads > out_cap;
    if (gave_up) { dev_free(queue); plan.release(); plan = SmPlan<IDX>(); plan.G = G; return false; }
    if (plan.wide_n) {
        const u64 nq = plan.wide_n;
        u64 *qbase = (u64 *)dev_alloc((nq + 1) * sizeof(u64));
        exclusive_scan_async<u64, SmWideCountIn<IDX>>(nq, SmWideCountIn<IDX>{queue}, qbase, (u64 *)nullptr, qbase + nq, name);
        prof_begin(std::string(name) + ".wide", plan.wide_atoms * (sizeof(u32) + sizeof(IDX)));
        hipLaunchKernelGGL((k_sm_wide<SEG, IDX>), dim3(grid_for(plan.wide_atoms, kBlock)), dim3(kBlock), 0, rt().stream, plan.wide_atoms, nq, seg,
                           (const SmWide<IDX> *)queue, (const u64 *)qbase, osym, ostart, (const u64 *)lbk.st_head);
        prof_end();
        after_launch(name);
        dev_free(qbase);
    }
    dev_free(queue);
    return true;
}

}   // namespace prim

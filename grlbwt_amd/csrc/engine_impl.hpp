// engine_impl.hpp -- the parse-then-induce engine (grlBWT exact path) over the
// device primitives of prim_hip.hpp.
//
// Included by engine_hip.hip once per index width:
//     #define GRL_NS grl32 / grl64,  #define GRL_IDX_T uint32_t / uint64_t
// (and, for the CPU test-suite only, by tests/hostsim over prim_sim.hpp).
//
// Reference map (file:line into ddiazdom/grlBWT):
//   a1  collection_stats            external/cdt/lib/utils.cpp:100-189      -> Engine::load_text
//   a2  lms_parsing::operator()     include/parsing_strategies.h:82-145     -> StartPred (phrase-start bit-vector)
//   a3  ext_hash_functor + hash_table::increment_value
//                                   exact_par_phase.hpp:14-42, hash_table.hpp:453-539 -> HashInsertFn
//   a5  dictionary ctor             exact_par_phase.hpp:106-183             -> CompactTableFn / DictBuildFn
//   a6  suffix_induction            exact_LMS_induction.h:94-158            -> Engine::dict_stage: Key0Fn + radix sort, then refinement by
//                                                                               symbol extension (ExtKeyFn, SegSortSmallFn, SegBig*Fn)
//   a7  produce_pre_bwt             exact_par_phase.cpp:136-242             -> GroupAccum*Fn/GroupDecideFn/GroupEmitFn
//   a8  produce_grammar             exact_par_phase.cpp:14-95               -> MetaPosFn + GrammarFn
//   a9  rank assignment             exact_par_phase.cpp:427-450             -> PhraseValFn / ScatterValFn
//   a10 ext_parse_functor/parse_text exact_par_phase.hpp:44-84, parsing_strategies.h:644-676 -> MapFn
//   a11 par_phase loop              exact_par_phase.cpp:338-366,496         -> Engine::parse_phase
//   a12 parse2bwt                   exact_ind_phase.cpp:603-672             -> Engine::first_bwt
//   a13 compute_hocc_size           exact_ind_phase.cpp:42-109              -> Engine::expand_split: ChainGen through prim::expand_count
//   a14 infer_lvl_bwt pass B        exact_ind_phase.cpp:143-258             -> ... and prim::expand_sort (expansion fused with the first split pass)
//   a15 infer_lvl_bwt pass C        exact_ind_phase.cpp:287-361             -> Engine::assemble_t: TermHeadEmitFn, PrePlaceFn, AsmSeg through
//                                                                               prim::stream_merge_count / stream_merge_emit
//   8e  collection-level mode       parsing_strategies.h:200-386 (thread ranges) -> Engine::dist_build (dist_round_t, dist_induce_level, dist_finish)
//   8f2 .rl_bwt consumers           scripts/*.cpp                            -> image_plain / image_rle / image_stats / image_split_runs / invert_image
//   8f3 FASTA/Q ingestion           external/bioparsers/lib/fastx_handler.cpp, kseq.h -> Engine::fastx_to_text
//   a16 bwt_buff_writer format      include/bwt_io.h:377-382,448-490        -> PackRunsFn
//   a17 final header widths         exact_ind_phase.cpp:274-276 @ level 0   -> Engine::finish
//
// Data layout in HBM: every level stays resident; text_0 in its native cell
// width, text_r (r>=1) as u32 cells  (rank<<2 | rep<<1 | is_terminator);
// run-length BWTs as struct-of-arrays (u32 sym[], idx_t len[]).

namespace GRL_NS {

using prim::u8;
using prim::u16;
using prim::u32;
using prim::u64;
typedef GRL_IDX_T idx_t;

// ------------------------------------------------------------------ buffers
template <class T>
struct DBuf {
    T *p = nullptr;
    u64 n = 0;
    DBuf() {}
    explicit DBuf(u64 n_) { alloc(n_); }
    DBuf(const DBuf &) = delete;
    DBuf &operator=(const DBuf &) = delete;
    DBuf(DBuf &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DBuf &operator=(DBuf &&o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
        return *this;
    }
    ~DBuf() { release(); }
    void alloc(u64 n_) { release(); p = (T *)prim::dev_alloc((n_ ? n_ : 1) * sizeof(T)); n = n_; }
    void release() { if (p) prim::dev_free(p); p = nullptr; n = 0; }
    void shrink(u64 n_) { if (p && n_ < n) { prim::dev_shrink(p, (n_ ? n_ : 1) * sizeof(T)); n = n_; } }      // the first n_ elements stay
    void zero() { prim::dev_memset(p, 0, n * sizeof(T)); }
    void fill_ff() { prim::dev_memset(p, 0xFF, n * sizeof(T)); }
    std::vector<T> to_host(u64 cnt) const { std::vector<T> h(cnt); prim::d2h(h.data(), p, cnt * sizeof(T)); return h; }
    T get(u64 i) const { T v; prim::d2h(&v, p + i, sizeof(T)); return v; }
};

static inline unsigned bitlen64(u64 v) { return v == 0 ? 0 : 64 - (unsigned)__builtin_clzll(v); }

// Fault injection for the multi-rank failure-agreement tests (tests/test_dist_gloo.py): compiled into the serial test stand-in
// only (prim::kIsDevice is false there).  The product library never reads these variables: no environment can make a
// production rank throw "injected by the test".
static inline bool test_fail_rank(const char *var, int me) {
    if constexpr (prim::kIsDevice) { (void)var; (void)me; return false; }
    else { const char *fr = getenv(var); return fr && atoi(fr) == me; }
}

// ----------------------------------------------------------------- searches
// first index i in [0,n) with a[i] > v  (n if none)
template <class T>
GRL_HD u64 upper_bound(const T *a, u64 n, T v) {
    u64 lo = 0, hi = n;
    while (lo < hi) { u64 mid = (lo + hi) >> 1; if (a[mid] <= v) lo = mid + 1; else hi = mid; }
    return lo;
}
// first index i in [0,n) with a[i] >= v  (n if none)
template <class T>
GRL_HD u64 lower_bound(const T *a, u64 n, T v) {
    u64 lo = 0, hi = n;
    while (lo < hi) { u64 mid = (lo + hi) >> 1; if (a[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}

// -------------------------------------------------------------- cell access
// level 0: raw symbols, rep == 1 everywhere (parsing_strategies.h:102-103), terminator == separator;
// level >= 1: u32 cell = rank<<2 | rep<<1 | is_terminator.
template <class cell_t, bool FIRST>
struct CellOps {
    cell_t sep;
    GRL_HD u32 sym(cell_t c) const { return FIRST ? (u32)c : (u32)(c >> 2); }
    GRL_HD bool rep(cell_t c) const { return FIRST ? true : (bool)((c >> 1) & 1); }
    GRL_HD bool isT(cell_t c) const { return FIRST ? (c == sep) : (bool)(c & 1); }
};

GRL_HD bool bit_at(const u64 *w, u64 i) { return (w[i >> 6] >> (i & 63)) & 1ull; }
// smallest i in [p, n) whose bit is set, n if there is none: 64 positions per step (the ends of very long phrases -- an N gap of
// 10^8 cells is ONE phrase -- are found through the bit-vectors, not cell by cell)
GRL_HD u64 next_set_bit(const u64 *w, u64 p, u64 n) {
    u64 r = n;
    if (p < n) {
        u64 wi = p >> 6;
        u64 x = w[wi] >> (p & 63);
        if (x) r = p + (u64)__builtin_ctzll(x);
        else {
            bool found = false;
            for (wi++; !found && (wi << 6) < n; wi++) { x = w[wi]; if (x) { r = (wi << 6) + (u64)__builtin_ctzll(x); found = true; } }
        }
        if (r > n) r = n;
    }
    return r;
}
// the same for the one-lane walkers of the cold kernels (table compaction, grammar walk, the bounds of a listed long phrase):
// eight words per step while they are empty -- a load and its latency per word made the 1.6 M empty words of a 10^8-cell phrase
// a 170 ms affair, five times per build.  (Not in the hashing kernels: inlined there, its sixteen registers of loads in flight
// pushed the level-0 kernel of the 10 GB build from 105 to 148 registers and 328 bytes of scratch per lane: 46 -> 122 ms.)
GRL_HD u64 next_set_bit_far(const u64 *w, u64 p, u64 n) {
    u64 r = n;
    if (p < n) {
        u64 wi = p >> 6;
        u64 x = w[wi] >> (p & 63);
        if (x) r = p + (u64)__builtin_ctzll(x);
        else {
            bool found = false;
            wi++;
            bool fast = true;
            while (fast && ((wi + 8) << 6) <= n) {
                u64 any = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) any |= w[wi + k];
                if (any) fast = false; else wi += 8;
            }
            for (; !found && (wi << 6) < n; wi++) { x = w[wi]; if (x) { r = (wi << 6) + (u64)__builtin_ctzll(x); found = true; } }
        }
        if (r > n) r = n;
    }
    return r;
}

// ------------------------------------------------------- a2: phrase starts
// A.1 of SURVEY.md: position p starts a phrase iff it starts a string or it is an
// LMS break: text[p-1] > text[p], type(p) = S, rep[p-1] = rep[p] = 1.
template <class cell_t, bool FIRST>
struct StartPred {
    const cell_t *t;
    CellOps<cell_t, FIRST> ops;
    GRL_DEV bool operator()(u64 p) const {
        if (p == 0) return true;
        cell_t cp = t[p - 1], c = t[p];
        if (ops.isT(cp)) return true;
        u32 sp = ops.sym(cp), s = ops.sym(c);
        if (!(sp > s) || !ops.rep(cp) || !ops.rep(c)) return false;
        u64 q = p;
        cell_t cq = c;
        for (;;) {                               // type(p): first unequal symbol to the right decides
            if (ops.isT(cq)) return false;       // equal run reaches the string end: L by definition
            cell_t nx = t[q + 1];
            u32 sn = ops.sym(nx);
            if (sn != s) return sn > s;
            q++;
            cq = nx;
        }
    }
};

struct PopcIn {
    const u64 *w;
    GRL_DEV u64 operator()(u64 i) const { return (u64)__builtin_popcountll(w[i]); }
};
struct PopcIn32 {
    const u64 *w;
    GRL_DEV u32 operator()(u64 i) const { return (u32)__builtin_popcountll(w[i]); }
};

// ------------------------------------------------- a3: phrase hashing/count
// phrase table: u64 keys and idx_t counts.  Two key forms (0 = empty):
//   generic   0 | tag:10 | ends-a-string:1 | len:12 | pos+1:40      the phrase is compared through its representative occurrence
//   exact     1 | len:3  | 0:4 | content:56                          byte cells, phrases of <= 7 cells: the key IS the phrase  Struct-of-arrays when few phrases
// are hot (level 0): the keys are written once and then read-mostly, so they stay cacheable, while the counts take
// the atomic traffic (16-byte slots were measured 4x slower there: the atomics on a hot phrase's count kept
// invalidating the line every probe of that phrase has to read).  16-byte (key, count) slots when most phrases are
// distinct: then every occurrence would fetch two random lines instead of one (hash_local).
static constexpr u64 kPosBits = 40;
static constexpr u64 kPosMask = (1ull << kPosBits) - 1;
static constexpr u64 kLenSat = 4095;               // lengths >= 4095 saturate; verified through the start bits
static constexpr u64 kExactKey = 1ull << 63;
GRL_HD u64 key_len(u64 k) { return (k >> kPosBits) & 0xFFFull; }
GRL_HD bool key_lastT(u64 k) { return (k >> (kPosBits + 12)) & 1ull; }     // the phrase ends with a terminator (known to the inserting lane)
GRL_HD u64 key_pos(u64 k) { return (k & kPosMask) - 1; }

// Phrase hash: two 32-bit multiplicative lanes per symbol (the walk is instruction-bound: a 64-bit
// multiply per byte cost ~4x the ALU work), folded into 64 bits and finalised once per phrase.
struct PhraseHash {
    u32 a, b;
    GRL_HD static PhraseHash init() { return PhraseHash{0x85A308D3u, 0x243F6A88u}; }
    GRL_HD void add(u32 v) {
        a = (a ^ v) * 0x9E3779B1u;
        b = (b + v) * 0x85EBCA6Bu;
        b ^= b >> 15;
    }
    GRL_HD u64 finish(u64 len) const {
        u64 h = ((u64)a << 32) | (u64)b;
        h ^= len * 0xC2B2AE3D27D4EB4Full;
        h ^= h >> 33; h *= 0xFF51AFD7ED558CCDull; h ^= h >> 33; h *= 0xC4CEB9FE1A85EC53ull; h ^= h >> 29;
        return h;
    }
};

// ---- phrase records of the partitioned naming (levels above 0, single-GPU rounds; prim::RecSort / prim::rec_dedupe) ----
// A phrase of at most cmax = min(7, 124 / b) cells (b = bits per symbol) IS a 128-bit record: symbol j at bits [j b, (j+1) b) of
// (lo, hi), hi bit 60 = the phrase ends a string, hi bits 61..63 = its length (0 = no record: the phrase is longer and goes
// through the hash table as before).  Equal records <=> equal phrases: no hashing, no look at the text to tell them apart.
static constexpr u64 kPhrLastT = 1ull << 60;
GRL_HD void rec_put(u64 &lo, u64 &hi, u32 sym, u32 j, int b) {
    const u32 o = j * (u32)b;
    if (o < 64) { lo |= (u64)sym << o; if (o + (u32)b > 64) hi |= (u64)sym >> (64 - o); }
    else hi |= (u64)sym << (o - 64);
}
GRL_HD u32 rec_sym(const prim::U128 &r, u32 j, int b) {
    const u32 o = j * (u32)b;
    u64 v;
    if (o < 64) { v = r.lo >> o; if (o + (u32)b > 64) v |= r.hi << (64 - o); }
    else v = r.hi >> (o - 64);
    return (u32)(v & ((1ull << b) - 1ull));
}
GRL_HD u32 rec_len(const prim::U128 &r) { return (u32)(r.hi >> 61); }
// A record travels through the partition sort (prim::RecSort) as two words: `hi` as it is (symbols above bit 64, flags, length) and
// key = lo ^ rec_g(hi).  The sort groups the records by a hash of the KEY alone (its passes read 8 bytes per record, no hash array);
// folding hi into the key keeps records that differ only above bit 64 -- long phrases with a common start -- out of one partition,
// and (key, hi) pairs are equal exactly when the records are.  rec_lo() undoes it.
GRL_HD u64 rec_g(u64 hi) { return hi * 0xD6E8FEB86659FD93ull; }
GRL_HD u64 rec_key(u64 lo, u64 hi) { return lo ^ rec_g(hi); }
GRL_HD u64 rec_lo(u64 key, u64 hi) { return key ^ rec_g(hi); }
struct RecValid {         // (by the hi word: the length field; the long phrases, which take the hash table, leave invalid records)
    GRL_DEV bool operator()(u64 hi) const { return (hi >> 61) != 0; }
};
static constexpr u32 kLongMark = 0x80000000u;      // in the per-occurrence slot array: the occurrence went through the hash table (slot in the low bits)

// One lane per text position; lanes on a phrase start hash the phrase, find/claim its
// slot and return the slot id (the caller counts it: prim::for_each_agg).
template <class cell_t, bool FIRST>
struct HashInsertFn {
    const cell_t *t;
    CellOps<cell_t, FIRST> ops;
    const u64 *startbits;
    const idx_t *wordbase;
    u64 *keys;
    u64 mask;
    u64 probe_limit;  // give up (overflow flag) after this many probes: the host retries with a larger table
    int ks = 0;       // key of slot s at keys[s << ks]: 0 = keys[] on its own, 1 = interleaved with the counts (16-byte slots)
    u32 *out_slot;    // [n_occ] slot of every phrase occurrence, text order
    u32 *scal;        // [1] error flag, [2..3] debug
    u64 n, n_occ;
    u64 *rep_pos = nullptr;   // [capacity] exact keys: position of the occurrence that claimed the slot (the dictionary reads the phrase there)
    // claim protocol of prim::for_each_agg: with claim_bits set, the slot id returned for the occurrence whose CAS created the
    // table entry carries prim::kClaimBit, and the kernel sets bit (position of that occurrence) -- one bit per distinct
    // phrase, from which the dictionary is compacted without a scan over the (sparse) table
    u64 *claim_bits = nullptr;
    static constexpr bool kClaims = true;
    GRL_DEV u64 claim_pos(u64 item) const { return item; }
    // HOT TABLE (level 0 of DNA-like texts: a few thousand phrases take 97 % of the occurrences).  hot_keys[0 .. hot_mask] is a
    // small dense table holding the phrases of a sample of the text, built before this pass and READ-ONLY during it: a
    // probe is a plain cached load into a few hundred KB that stay in every XCD's L2 (random loads: 255 G/s from a 4 MiB
    // window, 80 G/s from 16 MiB, 51 G/s from HBM -- tools/membench.hip; the one big sparse table kept every hot key on a
    // line of its own, 5 MB of them).  A phrase that is not there goes on to the big table (keys / mask), whose slot ids
    // follow the hot ones (slot_base).
    const u64 *hot_keys = nullptr;
    u64 hot_mask = 0;
    u32 slot_base = 0;
    // DIRECT INDEX (byte texts with at most 8 distinct cell values -- DNA reads: A C G T N and the separator): three bits of a cell
    // tell the values apart (dir_b0 < dir_b1 < dir_b2, found on the host from the byte histogram), so a phrase of <= 7 cells IS a
    // number below 2^22 -- its cells' 3-bit codes, a 1 above them for the length -- and that number is its slot: no hash, no
    // probe, no key compare, no claim.  Slots [0, kDirectSlots) in front of the table (slot_base), like the hot table this
    // replaces.  The representative position of a slot is whatever occurrence wrote dir_rep[] last (first_seen: once per
    // workgroup and slot, from the LDS count cache); DirectFixFn makes key, claim bit and so the dictionary entry afterwards.
    // (parsing_strategies.h:82-145 hashes every phrase.)
    static constexpr u32 kDirectSlots = 1u << 22;
    u32 dir_on = 0, dir_b0 = 0, dir_b1 = 0, dir_b2 = 0;
    u64 *dir_rep = nullptr;
    GRL_DEV u32 direct_index(u64 chunk, u32 len) const {      // chunk = the phrase's cells in its low bytes, len <= 7
        const u64 ones = 0x0101010101010101ull;
        const u64 c = ((chunk >> dir_b0) & ones) | (((chunk >> dir_b1) & ones) << 1) | (((chunk >> dir_b2) & ones) << 2);
        const u32 lo = (u32)c, hi = (u32)(c >> 32);
        const u32 plo = (lo & 7u) | ((lo >> 5) & 0x38u) | ((lo >> 10) & 0x1C0u) | ((lo >> 15) & 0xE00u);
        const u32 phi = (hi & 7u) | ((hi >> 5) & 0x38u) | ((hi >> 10) & 0x1C0u) | ((hi >> 15) & 0xE00u);
        const u32 top = 1u << (3u * len);
        return ((plo | (phi << 12)) & (top - 1u)) | top;
    }
    GRL_DEV void first_seen(u32 slot, u64 item) const { if (dir_on && slot < kDirectSlots) dir_rep[slot] = item; }
    // partitioned naming (see "phrase records"): with rec_k set, a phrase of <= rec_cmax cells leaves a record at its ordinal and
    // does NOT touch the table; longer ones take the table as before and mark their slot entry with kLongMark
    u64 *rec_k = nullptr; u64 *rec_hi = nullptr; int rec_b = 0; u32 rec_cmax = 0;      // (record of the occurrence with ordinal o: rec_k[o], rec_hi[o])
    u32 *long_count = nullptr;     // (sample pass: how many phrases are longer than rec_cmax)
    u64 walk_cap = ~0ull;          // (sample pass: a phrase longer than this is left out -- one lane walking a 10^8-cell phrase a second time: 6 s)
    // Byte cells: 4 phrases per lane at once through the exact-key path (process_batch).  A phrase of <= 7 cells is cut
    // out of ONE unaligned 8-byte load with the start bits and a zero-byte test for the terminator -- no loop -- its
    // bytes are the table key, so a probe that matches needs no look at a representative occurrence, and the text,
    // start-bit and table loads of the 4 phrases are in flight together (the kernel waits on dependent gathers).
    static constexpr bool kExact = FIRST && sizeof(cell_t) == 1;
#ifndef GRL_HASH_BATCH
#define GRL_HASH_BATCH 4
#endif
    // (Measured and dropped on the levels above 0, 10 GB build, level 1 = 964 M occurrences of 101 M phrases, 79 ms: exact keys
    // for phrases of <= 3 cells -- no look at the representative's text for two thirds of the occurrences -- 79 ms again;
    // the hashing alone, counts taken afterwards from the recorded slots: 53 + 49 ms.  The kernel runs at the rate of its
    // random count atomics and table probes, ~20 G/s each; a sort of the slot ids to count runs costs what the atomics do.)
    // (The same batching for the 4-byte cells of the levels above 0 -- phrases of <= 4 cells, generic keys, probe and
    // representative compare 2-4 phrases wide -- was built and measured at 80 vs 77 ms on level 1 of the 10 GB build:
    // there the two random HBM accesses per phrase run at the memory system's random-access rate, ~25 G/s, whatever
    // the number in flight per lane.  Dropped.)
    static constexpr int kBatch = kExact ? GRL_HASH_BATCH : 1;
    static constexpr u64 kCh = 8 / sizeof(cell_t);    // cells per 8-byte chunk
    GRL_DEV static u64 load8(const cell_t *a) { u64 v; __builtin_memcpy(&v, a, 8); return v; }   // unaligned 8-byte load
    // (single exit, loop flags instead of returns from inside the loops: these functions are inlined several times into
    // unrolled batch code, where hipcc 7.2 has mishandled divergent early exits -- see find_or_insert)
    GRL_DEV bool same_phrase(u64 q, u64 p, u64 len, bool check_bits) const {
        bool same = true;
        if (q + len > n) { scal[1] = 3; scal[2] = (u32)q; scal[3] = (u32)len; same = false; }
        if (same && check_bits) {
            // the length did not fit the key: the phrase at q must END where this one does -- from the start bits, 64 positions per
            // step (a bit test per cell made the comparison of two 5 M-cell phrases a 1 s affair)
            const u64 nq = next_set_bit(startbits, q + 1, n);
            const u64 eq = (nq >= n || ops.isT(t[nq - 1])) ? nq - 1 : nq;
            if (eq - q + 1 != len) same = false;
        }
        u64 j = 0;
        // 8 bytes per load, four loads in flight: long phrases are latency-bound per load
        for (; same && j + 4 * kCh <= len; j += 4 * kCh) {
            u64 d = 0;
#pragma unroll
            for (int x = 0; x < 4; x++) d |= load8(t + q + j + (u64)x * kCh) ^ load8(t + p + j + (u64)x * kCh);
            if (d) same = false;
        }
        for (; same && j + kCh <= len; j += kCh) if (load8(t + q + j) != load8(t + p + j)) same = false;
        for (; same && j < len; j++) if (t[q + j] != t[p + j]) same = false;
        return same;
    }
    GRL_DEV bool is_start(u64 p) const { return (startbits[p >> 6] >> (p & 63)) & 1ull; }
    GRL_DEV u32 operator()(u64 p) const { return is_start(p) ? process(p) : prim::kNoBucket; }
    // p is a phrase start: hash the phrase, find/claim its slot, record it for this occurrence
    GRL_DEV u32 process(u64 p) const {
        u64 w = startbits[p >> 6];
        u64 ord = (u64)wordbase[p >> 6] + (u64)__builtin_popcountll(w & ((1ull << (p & 63)) - 1ull));
        const bool pack = !kExact && rec_cmax != 0;      // (uniform) partitioned naming: short phrases become records
        bool fast = false;
        if (pack && p + (u64)rec_cmax + 1 <= n) {
            // Straight-line cut of a short phrase: the cells p .. p+cmax-1 and ONE window of start bits decide where it ends (at
            // the first terminator cell, or at the first phrase start behind p: that cell is shared with the next phrase) --
            // no loop of dependent loads, no hashing.  (The general walk below took 32 ms for the 964 M phrases of level 1.)
            const u64 b0 = p + 1;
            u64 bits = startbits[b0 >> 6] >> (b0 & 63);
            if ((b0 & 63) + (u64)rec_cmax > 64) bits |= startbits[(b0 >> 6) + 1] << (64 - (b0 & 63));
            cell_t cs[7];
#pragma unroll
            for (u32 j = 0; j < 7; j++) cs[j] = j < rec_cmax ? t[p + j] : cell_t(0);
            u32 m = ((u32)(bits << 1)) & ((1u << rec_cmax) - 1u) & ~1u;      // bit j: position p + j starts a phrase (j >= 1)
            u32 tm = 0;
#pragma unroll
            for (u32 j = 0; j < 7; j++) if (j < rec_cmax && ops.isT(cs[j])) tm |= 1u << j;
            m |= tm;
            if (m) {
                const u32 eo = (u32)__builtin_ctz(m);                          // offset of the phrase's last cell
                u64 klo = 0, khi = 0;
#pragma unroll
                for (u32 j = 0; j < 7; j++) if (j <= eo) rec_put(klo, khi, ops.sym(cs[j]), j, rec_b);
                const u64 len = (u64)eo + 1;
                if (ord >= n_occ) { scal[1] = 5; scal[2] = (u32)p; scal[3] = (u32)ord; }
                else if (rec_k) {
                    const u64 hi = khi | (((tm >> eo) & 1u) ? kPhrLastT : 0ull) | (len << 61);
                    rec_k[ord] = rec_key(klo, hi);
                    rec_hi[ord] = hi;
                    out_slot[ord] = 0;
                }
                fast = true;
            }
        }
        // (single exit, no return from the divergent branch above: see the compiler note in find_or_insert)
        u32 res = prim::kNoBucket;
        if (!fast) res = process_walk(p, ord, pack);
        return res;
    }
    // the general form: walk the phrase cell by cell (any length), hash it, find/claim its slot -- or leave its record
    // LONG PHRASES (a string of 10^8 equal symbols is ONE phrase; so is the whole of a level whose text has become one phrase):
    // one lane hashing 10^8 cells took 4 s.  A phrase the walk has followed for kLongWalk cells without reaching its end is LISTED
    // (giant_list) and left; prim::for_each_giant then takes one WAVE per listed phrase: lane 0 finds its bounds
    // (giant_bounds), every lane hashes one of 64 equal pieces (giant_piece), the piece hashes are mixed in lane order
    // (giant_mix) and lane 0 finishes with the result (process_giant: table lookup / insertion).  Which hash a phrase gets
    // depends on its length only, so every occurrence gets the same one -- and the hashing kernels hold no code for long phrases.
    u64 *giant_list = nullptr; u32 *giant_n = nullptr; u32 giant_cap = 0;
    static constexpr u64 kLongWalk = 4096;
    GRL_DEV void giant_bounds(u64 item, u64 &p, u64 &ee) const {
        p = item;
        const u64 ns = next_set_bit_far(startbits, p + 1, n);
        ee = (ns >= n || ops.isT(t[ns - 1])) ? ns - 1 : ns;
    }
    GRL_DEV u64 giant_piece(u64 p, u64 ee, int lane) const {
        const u64 len = ee - p + 1, piece = (len + 63) / 64;
        u64 x = p + (u64)lane * piece;
        const u64 xe = x + piece < ee + 1 ? x + piece : ee + 1;      // my cells: [x, xe)
        const u64 cnt = xe > x ? xe - x : 0;
        PhraseHash ph = PhraseHash::init();
        while (x + 4 * kCh <= xe) {
            u64 ck[4];
#pragma unroll
            for (int q4 = 0; q4 < 4; q4++) ck[q4] = load8(t + x + (u64)q4 * kCh);
#pragma unroll
            for (int q4 = 0; q4 < 4; q4++) {
#pragma unroll
                for (u64 j = 0; j < kCh; j++) ph.add(ops.sym((cell_t)(sizeof(cell_t) == 8 ? ck[q4] : (ck[q4] >> (8 * sizeof(cell_t) * j)))));
            }
            x += 4 * kCh;
        }
        while (x < xe) { ph.add(ops.sym(t[x])); x++; }
        return ph.finish(cnt);
    }
    GRL_HD static u64 giant_mix(u64 acc, u64 h) {
        acc = (acc ^ h) * 0xFF51AFD7ED558CCDull;
        return acc ^ (acc >> 32);
    }
    GRL_DEV u32 process_giant(u64 item, u64 acc, u64 ee) const {
        const u64 p = item;
        const u64 w = startbits[p >> 6];
        const u64 ord = (u64)wordbase[p >> 6] + (u64)__builtin_popcountll(w & ((1ull << (p & 63)) - 1ull));
        const bool pack = !kExact && rec_cmax != 0;
        return process_walk(p, ord, pack, &acc, ee);
    }
    GRL_DEV u32 process_walk(u64 p, u64 ord, bool pack, const u64 *giant_acc = nullptr, u64 known_end = 0) const {
        PhraseHash ph = PhraseHash::init();
        u64 e = p;                               // last cell taken so far
        cell_t c = t[p];
        ph.add(ops.sym(c));
        bool done = ops.isT(c);
        u64 klo = 0, khi = 0;
        if (pack) rec_put(klo, khi, ops.sym(c), 0, rec_b);
        // cells are taken 8 bytes at a time while that stays inside the text (one load covers a whole DNA phrase;
        // phrases of millions of cells -- e.g. N-runs -- would otherwise pay one memory latency per cell)
        bool capped = false, giant = false, use_giant = false;
        while (!done && e + 1 + kCh <= n) {
            if (e - p >= walk_cap) { capped = true; done = true; }
            if (!done && e - p >= kLongWalk) {
                if (giant_acc) { e = known_end; use_giant = true; }      // (the wave's visit: it knows the end and brings the hash)
                else {
                    giant = true;
                    const u32 gs = giant_n ? prim::atomic_add(giant_n, 1u) : giant_cap;
                    if (gs < giant_cap) giant_list[gs] = p; else scal[1] = 6;      // (the list takes every phrase that can be this long: see hash_local)
                }
                done = true;
            }
            if (!done) {
            u64 chunk = load8(t + e + 1);
            u64 b0 = e + 1;
            u64 bits = startbits[b0 >> 6] >> (b0 & 63);
            if ((b0 & 63) + kCh > 64) bits |= startbits[(b0 >> 6) + 1] << (64 - (b0 & 63));
#pragma unroll
            for (u64 j = 0; j < kCh; j++) {
                if (!done) {
                    cell_t cj = (cell_t)(sizeof(cell_t) == 8 ? chunk : (chunk >> (8 * sizeof(cell_t) * j)));
                    ph.add(ops.sym(cj));
                    e = b0 + j;
                    if (pack && e - p < (u64)rec_cmax) rec_put(klo, khi, ops.sym(cj), (u32)(e - p), rec_b);
                    done = ((bits >> j) & 1ull) || ops.isT(cj);
                }
            }
            }
        }
        bool ok = true;
        while (!done) {                           // tail of the text: cell by cell
            e++;
            if (e >= n) { scal[1] = 4; scal[2] = (u32)p; ok = false; done = true; }
            else {
                c = t[e];
                ph.add(ops.sym(c));
                if (pack && e - p < (u64)rec_cmax) rec_put(klo, khi, ops.sym(c), (u32)(e - p), rec_b);
                done = bit_at(startbits, e) || ops.isT(c);
            }
        }
        if (capped || giant) ok = false;
        if (ok && ord >= n_occ) { scal[1] = 5; scal[2] = (u32)p; scal[3] = (u32)ord; ok = false; }
        u32 found = prim::kNoBucket;
        if (ok && pack) {
            const u64 len = e - p + 1;
            if (len <= (u64)rec_cmax) {
                if (rec_k) {                       // the phrase is its record: nothing to look up
                    const u64 hi = khi | (ops.isT(t[e]) ? kPhrLastT : 0ull) | (len << 61);
                    rec_k[ord] = rec_key(klo, hi);
                    rec_hi[ord] = hi;
                    out_slot[ord] = 0;
                }
                ok = false;                        // (not an error: no table work for this phrase)
            } else {
                if (long_count) prim::atomic_add(long_count, 1u);
                if (rec_k) { rec_k[ord] = ord; rec_hi[ord] = 0ull; }      // (an invalid record; the sort's hash of the ordinal spreads them over the partitions)
            }
        }
        if (ok) {
            u64 len = e - p + 1;
            if (kExact && len <= 7) {              // same key form as process_batch: a phrase has ONE entry whichever path saw it
                u64 content = 0;
                for (u64 j = 0; j < len; j++) content |= (u64)t[p + j] << (8 * j);
                const u64 mine = kExactKey | (len << 60) | content;
                if (dir_on) found = direct_index(content, (u32)len);
                else
                found = insert_exact(mine, p, exact_hash(mine) & (hot_keys ? hot_mask : mask), 0, false);
            } else found = find_or_insert(p, len, use_giant ? PhraseHash{(u32)(*giant_acc >> 32), (u32)*giant_acc}.finish(len) : ph.finish(len), ops.isT(t[e]));
            if (found != prim::kNoBucket) out_slot[ord] = (found & ~prim::kClaimBit) | (pack ? kLongMark : 0u);
        }
        return found;
    }
    GRL_DEV void process_batch(const u64 *item, const bool *valid, u32 *slot) const { process_batch_exact<false>(item, valid, slot, nullptr, nullptr); }
    // the streaming form (prim::k_for_each_agg, STREAM): the kernel hands over the next phrase start and the ordinal of every
    // item -- no start-bit window, no rank lookup: ONE load per phrase (its cells) where the form above has five
    static constexpr bool kStream = kExact;
    GRL_DEV u64 ordinal_base(u64 p) const { return (u64)wordbase[p >> 6]; }               // (p a multiple of 64)
    // (the next phrase start behind p if it lies within 8 cells, any position further away otherwise: that is all the batch form
    // asks -- following the bits to the end of a phrase of 10^8 cells with one lane cost one_symbol_100M 0.17 s)
    GRL_DEV u64 next_item(u64 p) const {
        const u64 b0 = p + 1;
        u64 bits = startbits[b0 >> 6] >> (b0 & 63);
        if ((b0 & 63) > 56) bits |= startbits[(b0 >> 6) + 1] << (64 - (b0 & 63));
        bits &= 0xFFull;
        return bits ? b0 + (u64)__builtin_ctzll(bits) : p + 9;
    }
    // prim::name_stream's protocol (direct index only): a phrase named from its own cells, everything else deferred
    GRL_DEV u64 stream_load(u64 p) const { return load8(t + p); }
    GRL_DEV u32 stream_name(u64 p, u64 chunk, u64 next) const {
        const u64 nx = next - p;
        const u32 spos = nx < 8 ? (u32)nx : 8u;
        const u64 x = chunk ^ ((u64)ops.sep * 0x0101010101010101ull);
        const u64 z = (x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull;
        const u32 tpos = z ? (u32)(__builtin_ctzll(z) >> 3) : 8u;
        const u32 e = tpos == 0 ? 0u : (tpos < spos ? tpos : spos);
        const u32 len = e + 1;
        return len <= 7 ? direct_index(chunk, len) : prim::kDeferBucket;
    }
    GRL_DEV void stream_store(u64 ord, u32 slot) const {
        if (ord >= n_occ) { scal[1] = 5; scal[3] = (u32)ord; }
        else out_slot[ord] = slot;
    }
    GRL_DEV void process_batch_stream(const u64 *item, const bool *valid, u32 *slot, const u64 *next, const u64 *ord) const {
        process_batch_exact<true>(item, valid, slot, next, ord);
    }
    template <bool STREAM>
    GRL_DEV void process_batch_exact(const u64 *item, const bool *valid, u32 *slot, const u64 *next, const u64 *ordv) const {
        constexpr int B = kBatch;
        u64 chunk[B], wp[B], w0[B], w1[B], mine[B], idx[B], cur[B];
        idx_t wb[B];
        bool fast[B];
        const bool can = n >= 8;
#pragma unroll
        for (int j = 0; j < B; j++) {
            const u64 p = item[j];
            fast[j] = valid[j] && can && p + 8 <= n;
            const u64 pp = fast[j] ? p : 0;
            chunk[j] = can ? load8(t + pp) : 0ull;
            if constexpr (!STREAM) {
            wp[j] = startbits[pp >> 6];
            const u64 b1 = (pp + 1) >> 6;
            w0[j] = startbits[b1];
            w1[j] = startbits[b1 + 1];
            wb[j] = wordbase[pp >> 6];
            } else { wp[j] = w0[j] = w1[j] = 0; wb[j] = 0; }
        }
        const u64 sepx = (u64)ops.sep * 0x0101010101010101ull;
#pragma unroll
        for (int j = 0; j < B; j++) {
            const u64 p = item[j];
            u32 spos;
            if constexpr (STREAM) {
                const u64 nx = next[j] - p;                             // (>= 1 for a valid item: the next phrase start behind p)
                spos = nx < 8 ? (u32)nx : 8u;
            } else {
            const u32 sh = (u32)((p + 1) & 63);
            u64 bits = w0[j] >> sh;
            if (sh) bits |= w1[j] << (64 - sh);
            bits &= 0x7Full;                                            // start bits of cells p+1 .. p+7
            spos = bits ? (u32)__builtin_ctzll(bits) + 1u : 8u;
            }
            const u64 x = chunk[j] ^ sepx;
            const u64 z = (x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull;     // lowest set bit: first terminator byte
            const u32 tpos = z ? (u32)(__builtin_ctzll(z) >> 3) : 8u;  // cell index of the first terminator (8: none)
            const u32 e = tpos == 0 ? 0u : (tpos < spos ? tpos : spos);  // last cell of the phrase
            const u32 len = e + 1;
            fast[j] = fast[j] && len <= 7;
            const u64 content = chunk[j] & ((1ull << (8 * len)) - 1ull);
            if (dir_on) {                                                // (uniform) the phrase's number is its slot
                mine[j] = 0; cur[j] = 0;
                idx[j] = (u64)direct_index(chunk[j], len <= 7 ? len : 0u);
            } else {
            mine[j] = kExactKey | ((u64)len << 60) | content;
            idx[j] = exact_hash(mine[j]) & (hot_keys ? hot_mask : mask);
            }
        }
        if (!dir_on) {
#pragma unroll
        for (int j = 0; j < B; j++) cur[j] = hot_keys ? hot_keys[fast[j] ? idx[j] : 0] : prim::load_relaxed(&keys[(fast[j] ? idx[j] : 0) << ks]);
        }
#pragma unroll
        for (int j = 0; j < B; j++) {
            // (no continue/break/return inside divergent code here: see the note in find_or_insert)
            const u64 p = item[j];
            const bool ex = valid[j] && fast[j];
            u32 found = prim::kNoBucket;
            if (ex) found = dir_on ? (u32)idx[j] : insert_exact(mine[j], p, idx[j], cur[j], true);
            if (ex && found != prim::kNoBucket) {
                const u64 ord = STREAM ? ordv[j] : (u64)wb[j] + (u64)__builtin_popcountll(wp[j] & ((1ull << (p & 63)) - 1ull));
                if (ord >= n_occ) { scal[1] = 5; scal[2] = (u32)p; scal[3] = (u32)ord; found = prim::kNoBucket; }
                else out_slot[ord] = found & ~prim::kClaimBit;
            }
            if (valid[j] && !fast[j]) found = prim::kDeferBucket;      // the long cases go through process() later, 64 at a time
            slot[j] = found;
        }
    }
    GRL_DEV static u64 exact_hash(u64 mine) {
        u64 h = mine * 0x9E3779B97F4A7C15ull;
        h ^= h >> 32; h *= 0xD6E8FEB86659FD93ull; h ^= h >> 29;
        return h;
    }
    // find or claim the slot of an exact key; (sl, c) = first slot and, if `have_first`, the key already loaded from it
    GRL_DEV u32 insert_exact(u64 mine, u64 p, u64 sl, u64 c, bool have_first) const {
        u32 found = prim::kNoBucket;
        if (hot_keys) {                         // (sl, c) refer to the hot table: look the key up there first (read-only: plain loads)
            bool miss = false;
            bool first = have_first;
            while (found == prim::kNoBucket && !miss) {
                if (!first) c = hot_keys[sl];
                first = false;
                if (c == mine) found = (u32)sl;
                else if (c == 0) miss = true;
                else sl = (sl + 1) & hot_mask;
            }
            sl = exact_hash(mine) & mask;
            have_first = false;
        }
        bool stop = false;                      // the table overflowed elsewhere: give up (the host re-runs the pass)
        bool claimed = false;
        const bool was_hot = found != prim::kNoBucket;
        for (u64 probes = 0; probes < probe_limit && found == prim::kNoBucket && !stop; probes++) {
            if (probes || !have_first) {
                if (probes == 16 && prim::load_relaxed(&scal[1])) stop = true;
                c = prim::load_relaxed(&keys[sl << ks]);
            }
            if (c == 0) {
                u64 old = prim::atomic_cas(&keys[sl << ks], 0ull, mine);
                if (old == 0) { c = mine; rep_pos[sl] = p; claimed = true; } else c = old;
            }
            if (c == mine) found = (u32)sl + slot_base; else sl = (sl + 1) & mask;
        }
        if (found == prim::kNoBucket) scal[1] = 1;
        else if (!was_hot && claimed && claim_bits) found |= prim::kClaimBit;
        return found;
    }
    // claim or find the table slot of the phrase t[p .. p+len) whose (finalised) hash is h
    GRL_DEV u32 find_or_insert(u64 p, u64 len, u64 h, bool lastT) const {
        u64 lsat = len < kLenSat ? len : kLenSat;
        u64 hi = ((h >> 54) << 13) | ((u64)lastT << 12) | lsat;          // tag:10 | ends-a-string:1 | len:12 (bit 63 of the key stays clear)
        u64 mine = (hi << kPosBits) | (p + 1);
        u64 slot = h & mask;
        u32 hot_found = prim::kNoBucket;
        if (hot_keys) {                         // hot table first (read-only)
            u64 hs = h & hot_mask;
            bool miss = false;
            while (hot_found == prim::kNoBucket && !miss) {
                const u64 cur = hot_keys[hs];
                if (cur == 0) miss = true;
                else if ((cur >> kPosBits) == hi && same_phrase(key_pos(cur), p, len, lsat == kLenSat)) hot_found = (u32)hs;
                else hs = (hs + 1) & hot_mask;
            }
        }
        // NOTE: the result is carried in `found` and returned after the loop.  Returning from inside
        // the probe loop made hipcc 7.2 (gfx950) reuse the return register as a scratch under a
        // partial exec mask, so lanes that matched an existing key returned a stale value.
        u32 found = hot_found;
        bool stop = false;
        bool claimed = false;
        for (u64 probes = 0; probes < probe_limit && found == prim::kNoBucket && !stop; probes++) {
            // once probing gets long, look at the overflow flag (L1-bypassing load; done rarely: a coherent load of one
            // address by every lane was measured to serialise and cost 20 ms per 10 M phrases)
            if (probes == 16 && prim::load_relaxed(&scal[1])) stop = true;
            u64 cur = prim::load_relaxed(&keys[slot << ks]);   // L1-bypassing: a stale 0 from L1 would turn every later occurrence of a hot phrase into a CAS on one address
            if (cur == 0) {
                u64 old = prim::atomic_cas(&keys[slot << ks], 0ull, mine);
                if (old == 0) { cur = mine; claimed = true; }
                else cur = old;
            }
            bool hit = (cur == mine);
            if (!hit && (cur >> kPosBits) == hi) hit = same_phrase(key_pos(cur), p, len, lsat == kLenSat);
            if (hit) found = (u32)slot + slot_base;
            else slot = (slot + 1) & mask;
        }
        if (found == prim::kNoBucket) scal[1] = 1;   // table full / out of probes
        else if (claimed && claim_bits) found |= prim::kClaimBit;
        return found;
    }
};
struct SlotCountAdd {
    idx_t *counts; u64 cs;     // count of slot s at counts[s * cs]
    GRL_DEV void operator()(u32 slot, u32 c) const { prim::atomic_add(&counts[(u64)slot * cs], (idx_t)c); }
};
// Direct index, afterwards: one lane per slot of the direct region; a slot that counted occurrences becomes a table entry like
// the claimed ones -- the exact key rebuilt from the slot number (sym_of_code: the cell value of every 3-bit code), the claim bit
// at its representative's position -- so that the dictionary is compacted from the claim bits as for every other phrase.
struct DirectFixFn {
    const idx_t *counts; u64 *keys; const u64 *rep; u64 *claim_bits; u64 sym_of_code;      // (byte c of sym_of_code = cell value of code c)
    GRL_DEV void operator()(u64 s) const {
        if (s >= 8 && counts[s] != 0) {
            const u32 len = (u32)((63 - __builtin_clzll(s)) / 3);
            u64 content = 0;
            for (u32 j = 0; j < len; j++) content |= ((sym_of_code >> (8 * ((s >> (3 * j)) & 7ull))) & 0xFFull) << (8 * j);
            keys[s] = kExactKey | ((u64)len << 60) | content;
            const u64 p = rep[s];
            prim::atomic_or(&claim_bits[p >> 6], 1ull << (p & 63));
        }
    }
};
struct NoCountAdd {            // (experiments: the hashing pass without its count atomics)
    GRL_DEV void operator()(u32, u32) const {}
};

struct SampleCountIn {    // (slot occupied, its count): distinct phrases and occurrences of a sample's table
    const u64 *keys; const idx_t *counts;
    GRL_DEV prim::Pair<u64, u64> operator()(u64 i) const { return prim::Pair<u64, u64>(keys[i] != 0 ? 1ull : 0ull, (u64)counts[i]); }
};
struct OccIn {
    const u64 *keys; int ks;
    GRL_DEV u32 operator()(u64 i) const { return keys[i << ks] != 0 ? 1u : 0u; }
};
// The same functor over a SAMPLE of the positions: virtual index v -> block v / blk of the text (blocks `stride` apart),
// offset v % blk (prim::for_each_agg protocol)
template <class F>
struct SampledFn {
    F f; u64 blk, stride;
    static constexpr int kBatch = F::kBatch;
    static constexpr bool kClaims = F::kClaims;   // (claims are marked at the REAL position: claim_pos)
    u64 *claim_bits = nullptr;                    // = f.claim_bits
    GRL_DEV u64 map(u64 v) const { return (v / blk) * stride + (v % blk); }
    GRL_DEV u64 claim_pos(u64 v) const { return map(v); }
    GRL_DEV bool is_start(u64 v) const { return f.is_start(map(v)); }
    GRL_DEV u32 process(u64 v) const { return f.process(map(v)); }
    GRL_DEV u32 operator()(u64 v) const { return f(map(v)); }
    GRL_DEV void process_batch(const u64 *item, const bool *valid, u32 *slot) const {
        u64 m[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; j++) m[j] = map(item[j]);
        f.process_batch(m, valid, slot);
    }
};

// Partitioned naming, the pass over the text: ONE lane per position, nothing but the straight-line cut of HashInsertFn::process
// (a window of start bits + rec_cmax cell loads -> the phrase's 128-bit record at its ordinal).  Phrases that do not end within
// rec_cmax cells (and the few at the very end of the text) are only LISTED here; ListedFn sends them through the general walk and
// the hash table afterwards.  The record pass through prim::for_each_agg with the whole HashInsertFn inlined took 33 ms for the
// 964 M phrases of level 1 of the 10 GB build, 25 of them with neither the cell loads nor the record stores.
// (B = bits per symbol at compile time, hence cmax and every shift: the record is assembled in four 32-bit words by constant
// shifts, no loop or branch per cell.  The run-time form with 64-bit shifts ran ~250 vector + ~200 scalar instructions per 64
// positions and was bound by exactly that -- 27.7 ms at level 1; with run-time word indices the compiler kept the four words
// in scratch memory -- 66 ms.  B = 0: the run-time form, for symbol widths without an instance.)
template <int B>
GRL_HD void rec_put32(u32 &W0, u32 &W1, u32 &W2, u32 &W3, u32 s, int j) {      // symbol s at bits [j B, (j + 1) B) of W3:W2:W1:W0
    const int o = j * B, wi = o >> 5, sh = o & 31;
    const u32 a = s << sh;
    const u32 c = (sh + B > 32) ? s >> ((32 - sh) & 31) : 0u;
    if (wi == 0) { W0 |= a; W1 |= c; }
    else if (wi == 1) { W1 |= a; W2 |= c; }
    else if (wi == 2) { W2 |= a; W3 |= c; }
    else W3 |= a;
}
template <class cell_t, int B>
struct PhraseRecordFn {
    static constexpr int CMAX = B ? (124 / B < 7 ? 124 / B : 7) : 7;
    const cell_t *t; CellOps<cell_t, false> ops; const u64 *startbits; const idx_t *wordbase;
    u64 n, n_occ;
    u64 *rec_k; u64 *rec_hi; int rec_b; u32 rec_cmax;            // (B == 0 reads the last two)
    u64 *long_bits;       // bit p: the phrase starting at p is left to the walk (a counter for a list of them serialised the pass:
                          // 38 M same-address atomics at level 2 of the 10 GB build, 171 ms)
    u32 *scal;
    // (p is a phrase start, ord its ordinal: prim::for_each_set_bit over the start bits -- every lane of a wave on a phrase of its own,
    // consecutive lanes at consecutive ordinals.  Rounds 3-5 ran one lane per text POSITION: two lanes in three idle.)
    GRL_DEV void operator()(u64 p, u64 ord) const {
        bool listed = true;
        {
            const u32 cmax = B ? (u32)CMAX : rec_cmax;
            if (p + (u64)cmax + 1 <= n) {
                const u64 b0 = p + 1;
                u64 bits = startbits[b0 >> 6] >> (b0 & 63);
                if ((b0 & 63) + (u64)cmax > 64) bits |= startbits[(b0 >> 6) + 1] << (64 - (b0 & 63));
                cell_t cs[CMAX];
#pragma unroll
                for (int j = 0; j < CMAX; j++) cs[j] = (u32)j < cmax ? t[p + j] : cell_t(0);
                u32 m = ((u32)(bits << 1)) & ((1u << cmax) - 1u) & ~1u;      // bit j: position p + j starts a phrase (j >= 1)
                u32 tm = 0;
#pragma unroll
                for (int j = 0; j < CMAX; j++) tm |= ((u32)j < cmax && ops.isT(cs[j]) ? 1u : 0u) << j;
                m |= tm;
                if (m) {
                    const u32 eo = (u32)__builtin_ctz(m);                          // offset of the phrase's last cell
                    u64 klo = 0, khi = 0;
                    if constexpr (B != 0) {
                        u32 W0 = 0, W1 = 0, W2 = 0, W3 = 0;
#pragma unroll
                        for (int j = 0; j < CMAX; j++) rec_put32<B>(W0, W1, W2, W3, (u32)j <= eo ? ops.sym(cs[j]) : 0u, j);
                        klo = (u64)W0 | ((u64)W1 << 32); khi = (u64)W2 | ((u64)W3 << 32);
                    } else {
#pragma unroll
                        for (int j = 0; j < CMAX; j++) if ((u32)j <= eo) rec_put(klo, khi, ops.sym(cs[j]), (u32)j, rec_b);
                    }
                    const u64 len = (u64)eo + 1;
                    if (ord >= n_occ) { scal[1] = 5; scal[2] = (u32)p; scal[3] = (u32)ord; }
                    else {
                        const u64 hi = khi | (((tm >> eo) & 1u) ? kPhrLastT : 0ull) | (len << 61);
                        rec_k[ord] = rec_key(klo, hi);
                        rec_hi[ord] = hi;
                    }
                    listed = false;
                }
            }
        }
        if (listed) prim::atomic_or(long_bits + (p >> 6), 1ull << (p & 63));      // (rare, and every word another address; the bits start out zero)
    }
};
struct BitPositionsFn {      // the positions of the set bits of words[], in order: pos[base[w] ..] for word w
    const u64 *words; const u32 *base; u64 *pos;
    GRL_DEV void operator()(u64 w) const {
        u64 x = words[w];
        u64 k = base[w];
        while (x) { pos[k++] = w * 64 + (u64)__builtin_ctzll(x); x &= x - 1; }
    }
};
// HashInsertFn over a LIST of phrase starts (prim::for_each_agg protocol; claims are marked at the real position)
template <class F>
struct ListedFn {
    F f; const u64 *pos;
    static constexpr int kBatch = 1;
    static constexpr bool kClaims = F::kClaims;
    u64 *claim_bits = nullptr;                    // = f.claim_bits
    GRL_DEV u64 claim_pos(u64 v) const { return pos[v]; }
    GRL_DEV bool is_start(u64) const { return true; }
    GRL_DEV u32 process(u64 v) const { return f.process(pos[v]); }
    GRL_DEV u32 operator()(u64 v) const { return f.process(pos[v]); }
    GRL_DEV void process_batch(const u64 *, const bool *, u32 *) const {}
    GRL_DEV void first_seen(u32 slot, u64 v) const { f.first_seen(slot, pos[v]); }
};

// ------------------------------------------------------ a5: dictionary view
template <class cell_t, bool FIRST>
struct CompactTableFn {
    const cell_t *t;
    CellOps<cell_t, FIRST> ops;
    const u64 *startbits;
    const u64 *keys; const idx_t *counts;
    u64 *ph_pos; idx_t *ph_freq; u32 *ph_len; u32 *ph_slot; u8 *ph_lastT;
    int ks; u64 cs; const u64 *rep_pos;
    u64 n;                                         // cells of the text
    GRL_DEV void emit(u64 s, u64 k) const {        // table slot s is phrase k of the dictionary
        u64 k64 = keys[s << ks];
        // (one exit: an early return from the first branch cost the ends-a-string flag of a few phrases in the 64-bit build
        // -- the flag store of the second branch ran for lanes of the first; see the note in find_or_insert)
        u64 pos, len;
        bool lastT;
        if (k64 & kExactKey) {                     // the key is the phrase: length and last cell come from the key, the position from rep_pos
            len = (k64 >> 60) & 7ull;
            pos = rep_pos[s];
            lastT = ops.isT((cell_t)(k64 >> (8 * (len - 1))));
        } else {
            pos = key_pos(k64); len = key_len(k64);
            lastT = key_lastT(k64);                // carried in the key: no gather of the phrase's last cell
            if (len == kLenSat) {
                // saturated: the phrase ends at the first terminator or AT the next phrase start (LMS phrases share that cell) --
                // and the position behind a terminator is a start, so the next start bit behind pos decides (a walk cell by cell
                // took 18.7 s for ONE phrase of 10^8 cells)
                const u64 ns = next_set_bit_far(startbits, pos + 1, n);
                const u64 e = (ns >= n || ops.isT(t[ns - 1])) ? ns - 1 : ns;
                len = e - pos + 1;
            }
        }
        ph_pos[k] = pos; ph_freq[k] = counts[s * cs]; ph_len[k] = (u32)len; ph_slot[k] = (u32)s;
        ph_lastT[k] = lastT ? 1 : 0;
    }
};
// The distinct phrases from the claim bits of the hashing pass (bit p: the occurrence at text position p created its table
// entry): one lane per word of 64 positions; phrase numbers = rank of the claim, i.e. phrases in order of first claim.
// (Rounds 1-2 scanned the table -- three passes over 2^31 16-byte slots at level 1 of the 10 GB build, 22 ms, to find 101 M
// entries; the bit-vector has n / 8 bytes whatever the table size.)
struct ClaimSlotsFn {       // step 1, one lane per word: the table slot of every claim, in claim order (out_slot is read nearly in sequence)
    const u64 *claim_bits; const idx_t *cbase; const u64 *startbits; const idx_t *wordbase; const u32 *out_slot; u32 *ph_slot;
    GRL_DEV void operator()(u64 w) const {
        u64 m = claim_bits[w];
        if (m) {
            const u64 sb = startbits[w];
            const u64 ob = (u64)wordbase[w];
            u64 k = (u64)cbase[w];
            while (m) {
                const int b = __builtin_ctzll(m);
                m &= m - 1;
                ph_slot[k++] = out_slot[ob + (u64)__builtin_popcountll(sb & ((1ull << b) - 1ull))] & ~kLongMark;
            }
        }
    }
};
template <class cell_t, bool FIRST>
struct ClaimCompactFn {     // step 2, one lane per phrase: its table entry (a random gather per phrase, all of them in flight together --
                            // one lane walking the claims of a word took them one after the other: 13 ms for 159 M phrases, 2.2x the table scan it replaced)
    CompactTableFn<cell_t, FIRST> c;
    GRL_DEV void operator()(u64 k) const { c.emit((u64)c.ph_slot[k], k); }
};
// (Both run ONE WAVE PER PARTITION -- lane i handles partition i >> 6, elements (i & 63), + 64, ... of it: a partition's distinct records
// sit together in the staging area and its occurrences together in the sorted order, so every access is a coalesced run and nobody
// searches for its partition.  Rounds 3-5: one lane per phrase with a binary search over the partitions' bases -- 18 dependent
// loads per wave, 11 ms per 10 GB build -- and one lane per record that took its partition from a hash of the sorted key.)
struct PartPhraseFn {      // the distinct phrases of the partitioned naming: from the staging areas of their partitions
    const u32 *pbase; const u64 *pstart; const u64 *dkey; const u64 *dhi; const u32 *dcnt; u32 slot0;
    prim::U128 *ph_key; u64 *ph_pos; idx_t *ph_freq; u32 *ph_len; u32 *ph_slot; u8 *ph_lastT;
    u8 *ph_vflag;          // the two low bits of the phrase's value in the next text: (frequency > 1) << 1 | ends-a-string
    GRL_DEV void operator()(u64 i) const {
        const u64 p = i >> 6;
        const u64 k0 = pbase[p], cnt = (u64)pbase[p + 1] - k0, src0 = pstart[p];
        for (u64 j = i & 63; j < cnt; j += 64) {
            const u64 k = k0 + j, src = src0 + j;
            const u64 hi = dhi[src];
            const prim::U128 r(rec_lo(dkey[src], hi), hi);
            const u32 f = dcnt[src];
            ph_key[k] = r; ph_pos[k] = 0; ph_freq[k] = (idx_t)f; ph_len[k] = rec_len(r); ph_slot[k] = slot0 + (u32)k;
            ph_lastT[k] = (r.hi & kPhrLastT) ? 1 : 0;
            ph_vflag[k] = (u8)((f > 1 ? 2u : 0u) | ((r.hi & kPhrLastT) ? 1u : 0u));
        }
    }
};
struct PartValFn {         // the records of a partition, in sorted order -> the values of their phrases (read where GroupPhraseValFn put them)
    const u64 *pstart; const u32 *pbase; const u32 *lid; const u32 *slot_val; u32 slot0; u32 *out;
    const u8 *vflag;       // (single-GPU rounds: the slot holds the phrase's rank << 2, its two flag bits come from here -- see GroupPhraseValFn)
    GRL_DEV void operator()(u64 i) const {
        const u64 p = i >> 6;
        const u64 a = pstart[p], e = pstart[p + 1], k0 = (u64)pbase[p], v0 = (u64)slot0 + k0;
        for (u64 x = a + (i & 63); x < e; x += 256) {       // (four elements per step: their loads are in flight together)
            u32 l[4], v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) l[j] = x + 64u * j < e ? lid[x + 64u * j] : prim::kNoId;
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = l[j] == prim::kNoId ? 0u : (slot_val[v0 + (u64)l[j]] | (vflag ? (u32)vflag[k0 + (u64)l[j]] : 0u));
#pragma unroll
            for (int j = 0; j < 4; j++) if (x + 64u * j < e) out[x + 64u * j] = v[j];
        }
    }
};
struct PartCombineFn {     // the parse: occurrences that went through the table read their slot's value, the others take the value that came back
    const u32 *slot_val; const u32 *vals; u32 *text;
    GRL_DEV void operator()(u64 i) const { const u32 m = text[i]; text[i] = (m & kLongMark) ? slot_val[m & ~kLongMark] : vals[i]; }
};
struct LenIn {
    const u32 *l;
    GRL_DEV u32 operator()(u64 i) const { return l[i]; }
};
struct FreqLenIn {        // (frequency, length) of phrase i
    const idx_t *f; const u32 *l;
    GRL_DEV prim::Pair<u64, u64> operator()(u64 i) const { return prim::Pair<u64, u64>((u64)f[i], (u64)l[i]); }
};
GRL_HD u64 rank1(const u64 *words, const idx_t *base, u64 x);      // (defined with the rank bit-vectors below)
template <class cell_t, bool FIRST, int PER = 4>
struct DictBuildFn {      // one lane per PER consecutive dictionary positions: one phrase lookup, then a forward walk
    const cell_t *t;
    CellOps<cell_t, FIRST> ops;
    const u32 *ph_off; u64 D; u64 S; const u64 *ph_pos;
    u32 *dict_sym; u32 *dict_phr;
    const u64 *pw; const idx_t *pb;      // rank bit-vector of the phrase starts over the dictionary positions (nullptr: binary search)
    const prim::U128 *pkeys = nullptr; u64 pDs = 0; int pkb = 0;      // phrases [0, pDs) are given by their records (partitioned naming): no look at the text
    struct alignas(16) Quad { u32 v[4]; };
    GRL_DEV void operator()(u64 c) const {
        u64 q0 = c * PER, q1 = q0 + PER < S ? q0 + PER : S;
        // the phrase holding position q0: two loads through the bit-vector (a binary search over the phrase offsets was
        // 27 dependent probes per lane)
        u64 k = pw ? rank1(pw, pb, q0 + 1) - 1 : upper_bound<u32>(ph_off, D, (u32)q0) - 1;
        u64 nxt = ph_off[k + 1];
        u32 phr[PER], sym[PER];
#pragma unroll
        for (int j = 0; j < PER; j++) {
            u64 q = q0 + j;
            if (q < q1) {
                while (q >= nxt) { k++; nxt = ph_off[k + 1]; }
                phr[j] = (u32)k;
                sym[j] = (pkeys && k < pDs) ? rec_sym(pkeys[k], (u32)(q - ph_off[k]), pkb) : ops.sym(t[ph_pos[k] + (q - ph_off[k])]);
            }
        }
        if (q1 - q0 == PER) {     // 16-byte stores: a quarter of the store requests of scalar stores per array
#pragma unroll
            for (int j = 0; j < PER; j += 4) {
                *reinterpret_cast<Quad *>(dict_phr + q0 + j) = Quad{{phr[j], phr[j + 1], phr[j + 2], phr[j + 3]}};
                *reinterpret_cast<Quad *>(dict_sym + q0 + j) = Quad{{sym[j], sym[j + 1], sym[j + 2], sym[j + 3]}};
            }
        } else {
#pragma unroll
            for (int j = 0; j < PER; j++)
                if (q0 + j < q1) { dict_phr[q0 + j] = phr[j]; dict_sym[q0 + j] = sym[j]; }
        }
    }
};

// ------------------------------------------------ a6: dictionary suffix sort
// Order: lexicographic with the phrase end comparing as +infinity (the sentinel
// code is all-ones), equal suffixes of different phrases form one group.
// The suffix made of just the LAST cell of a phrase that does not end a string forms, with its equals, a group that is
// never valid (exact_par_phase.cpp:162: no pre-BWT entry, no rank) -- unless it is a whole one-cell phrase, whose value
// is read from its group.  A quarter of all dictionary suffixes are of that kind (one per phrase): they are left out of
// the sort, and so of the group stage, altogether.
// (flags above the left symbol of a suffix record, see SufRecT below: the suffix is the last cell of its phrase / that phrase ends a string)
static constexpr u32 kRecSym = 0x3FFFFFFFu, kRecFinal = 0x40000000u, kRecLastT = 0x80000000u;
struct SufKeep {
    const u32 *dict_phr; const u32 *ph_off; const u8 *ph_lastT;
    bool all = false;     // levels with long phrases keep every suffix: the doubling rounds (DoubleKeyFn) want a slot for every position
    GRL_DEV bool operator()(u64 q) const {
        const u32 k = dict_phr[q];
        return all || !(q + 1 == (u64)ph_off[k + 1] && !ph_lastT[k] && q != (u64)ph_off[k]);
    }
};
// RUN-AWARE KEYS (levels whose phrases hold long runs of one symbol: a 2 M-cell run of N makes one phrase of 2 M cells, and its
// 2 M suffixes a^rem X tie with those of every other copy of the run for rem / K rounds each -- quadratic: 311 s for a 20 MB
// input).  Below the window of K symbols a key carries, whenever the window lies INSIDE a run (rem >= K cells of the symbol a
// left in it), one class bit and the run's remaining length: the suffixes a^i x.. and a^j y.. (i < j) compare as x against a, so
// all runs followed by a smaller symbol come first, shortest first, then the runs followed by a larger symbol (or the phrase
// end, which compares as +infinity), longest first.  Equal keys then mean equal remaining runs, and the whole run is
// consumed at once: skip[q] grows by rem - K.  rb = 0: plain keys (every ordinary level).
struct RunKeys {
    const u32 *rem = nullptr;     // [S] cells from q to the end of its run of equal symbols, inside its phrase
    u32 *skip = nullptr;          // [S] cells the suffix at q has jumped beyond the rounds' common depth
    int rb = 0;                   // bits below the window: 1 + bitlen(longest phrase)
};
GRL_DEV u64 suffix_key0(const u32 *dict_sym, u64 q, u64 end, int K, int b, const RunKeys &rk = RunKeys(), u32 *jump = nullptr) {
    const u64 sent = (1ull << b) - 1;       // first K symbols packed b bits each
    u64 key = 0;
    for (int j = 0; j < K; j++) key = (key << b) | ((q + j < end) ? (u64)dict_sym[q + j] : sent);
    if (rk.rb) {
        u64 low = 0;
        if (q < end) {
            const u32 r = rk.rem[q];
            if (r >= (u32)K) {
                const u32 a = dict_sym[q];
                const u64 xe = q + (u64)r;                              // (<= end: runs stay inside their phrase)
                const bool up = xe >= end || dict_sym[xe] > a;
                const int R = rk.rb - 1;
                low = up ? ((1ull << R) | (((1ull << R) - 1ull) - (u64)r)) : (u64)r;
                if (jump) *jump = r - (u32)K;
            }
        }
        key = (key << rk.rb) | low;
    }
    return key;
}
struct RunChangeIn {      // 1 at the last cell of every run of equal symbols (runs end with their phrase)
    const u32 *dict_sym; const u64 *pw; u64 S;
    GRL_DEV u32 operator()(u64 q) const { return (q + 1 >= S || bit_at(pw, q + 1) || dict_sym[q + 1] != dict_sym[q]) ? 1u : 0u; }
};
struct RunEndsEmitFn {    // rex[q] = run ends in front of q; ends[k] = position of the k-th run end
    static constexpr bool kWaveEmit = false;
    u32 *rex; u32 *ends;
    GRL_DEV void operator()(u64 q, u32 ex, u32 v) const { rex[q] = ex; if (v) ends[ex] = (u32)q; }
};
struct RunRemFn {
    const u32 *rex; const u32 *ends; u32 *rem;
    GRL_DEV void operator()(u64 q) const { rem[q] = ends[rex[q]] - (u32)q + 1u; }
};
struct InitSkipFn {       // after the first sort: the suffixes whose first key lay inside a run have consumed that run
    const u32 *perm; const u32 *rem; u32 K; u32 *skip;
    GRL_DEV void operator()(u64 t) const { const u32 q = perm[t]; const u32 r = rem[q]; if (r >= K) skip[q] = r - K; }
};
struct SampleKey0Fn {     // keys at strided positions: the splitter sample of the sharded sort
    const u32 *dict_sym; const u32 *dict_phr; const u32 *ph_off; int K, b; u64 stride; u64 *out; RunKeys rk;
    GRL_DEV void operator()(u64 i) const { const u64 q = i * stride; out[i] = suffix_key0(dict_sym, q, ph_off[dict_phr[q] + 1], K, b, rk); }
};
struct SampleWeightFn {   // ... and the frequency of the phrase each sampled suffix belongs to (splitters by symbols described, not by suffixes)
    const u32 *dict_phr; const idx_t *ph_freq; u64 stride; u64 *out;
    GRL_DEV void operator()(u64 i) const { out[i] = (u64)ph_freq[dict_phr[i * stride]]; }
};
struct PhraseDropIn {     // 1 for a phrase whose last-cell suffix is left out (one scan over the PHRASES then places every kept suffix)
    const u32 *ph_off; const u8 *ph_lastT; bool all = false;
    GRL_DEV u32 operator()(u64 k) const { return (!all && !ph_lastT[k] && ph_off[k + 1] - ph_off[k] > 1) ? 1u : 0u; }
};
struct Key0KeepFn {       // (key, position) of the kept suffixes at position - (suffixes left out in front of the phrase)
    SufKeep keep; const u32 *dropcnt; const u32 *dict_sym; int K, b; u64 *ka; u32 *va; RunKeys rk;
    GRL_DEV void operator()(u64 q) const {
        if (keep(q)) {
            const u32 k = keep.dict_phr[q];
            const u64 i = q - (u64)dropcnt[k];
            ka[i] = suffix_key0(dict_sym, q, keep.ph_off[k + 1], K, b, rk);
            va[i] = (u32)q;
        }
    }
};
struct KeepRangeIn {      // 1 for the kept suffixes (SufKeep) among the positions q0 + i of my share of the dictionary
    SufKeep keep; u64 q0;
    GRL_DEV u32 operator()(u64 i) const { return keep(q0 + i) ? 1u : 0u; }
};
struct KeyRangeFn {       // (key, position, owner of the key's range) of those, compacted  (pos_base: the arrays describe MY part of the dictionary, positions travel as global ones)
    SufKeep keep; const u32 *ex; u64 q0; const u32 *dict_sym; int K, b; const u64 *spl; int N;
    u64 *lk; u32 *lp; u32 *own; u32 *idx; RunKeys rk; u32 pos_base = 0;
    u64 *lrec = nullptr; const idx_t *ph_freq = nullptr; u32 bwt_code = 0;      // (dictionary sharded by owner: what the group fold reads about the suffix travels with it)
    GRL_DEV void operator()(u64 i) const {
        const u64 q = q0 + i;
        if (keep(q)) {
            const u32 o = ex[i];
            const u64 key = suffix_key0(dict_sym, q, keep.ph_off[keep.dict_phr[q] + 1], K, b, rk);
            u32 d = 0;
            for (int r = 1; r < N; r++) if (key >= spl[r]) d = (u32)r;      // the LAST rank whose range starts at or below the key
            lk[o] = key; lp[o] = (u32)q + pos_base; own[o] = d; idx[o] = o;
            if (lrec) {
                const u32 k = keep.dict_phr[q];
                const u32 left = ((q == (u64)keep.ph_off[k]) ? bwt_code : dict_sym[q - 1]) | ((q + 1 == (u64)keep.ph_off[k + 1]) ? kRecFinal : 0u) | (keep.ph_lastT[k] ? kRecLastT : 0u);
                lrec[o] = ((u64)ph_freq[k] << 32) | (u64)left;
            }
        }
    }
};
struct ByteIn {
    const u8 *f;
    GRL_DEV u32 operator()(u64 i) const { return f[i]; }
};
// (the same without an exchange, for few ranks: every rank looks at ALL positions and keeps the records of its own key range)
struct OwnKeyFlagFn {     // flag[q] = 1 for the kept suffixes whose key lies in the range of rank `me`
    SufKeep keep; const u32 *dict_sym; int K, b; const u64 *spl; int N, me; u8 *flag; RunKeys rk;
    GRL_DEV void operator()(u64 q) const {
        u8 f = 0;
        if (keep(q)) {
            const u64 key = suffix_key0(dict_sym, q, keep.ph_off[keep.dict_phr[q] + 1], K, b, rk);
            int d = 0;
            for (int r = 1; r < N; r++) if (key >= spl[r]) d = r;
            f = d == me ? 1 : 0;
        }
        flag[q] = f;
    }
};
struct OwnKeyEmitFn {     // ... and their (key, position) records, compacted (positions ascending)
    static constexpr bool kWaveEmit = false;
    SufKeep keep; const u32 *dict_sym; int K, b; u64 *ka; u32 *perm; RunKeys rk;
    GRL_DEV void operator()(u64 q, u32 ex, u32 v) const {
        if (v) { ka[ex] = suffix_key0(dict_sym, q, keep.ph_off[keep.dict_phr[q] + 1], K, b, rk); perm[ex] = (u32)q; }
    }
};
struct GatherKeyPosFn {   // records in owner order
    const u32 *order; const u64 *lk; const u32 *lp; u64 *sk; u32 *sp;
    GRL_DEV void operator()(u64 j) const { const u32 i = order[j]; sk[j] = lk[i]; sp[j] = lp[i]; }
};
struct HeadFlagFn {       // hflag[t] = 1 where the sorted key changes
    const u64 *k; u8 *hflag;
    GRL_DEV void operator()(u64 t) const { hflag[t] = (t == 0 || k[t] != k[t - 1]) ? 1 : 0; }
};
struct FirstUnresolvedFn { // after the first sort: member of a group of > 1 suffixes whose key holds no sentinel (the suffix is at
                           // least K symbols long): what the refinement below has to look at again
    const u64 *k; const u8 *hflag; u64 S; u64 sent; u8 *uflag;
    GRL_DEV void operator()(u64 t) const {
        bool multi = !hflag[t] || (t + 1 < S && !hflag[t + 1]);
        uflag[t] = (multi && (k[t] & sent) != sent) ? 1 : 0;
    }
};
// Refinement by SYMBOL EXTENSION.  After the first pass a group of equal keys whose suffixes have not ended is re-sorted,
// inside the group, by the next K symbols of its members (read straight from the dictionary: one contiguous gather), and so
// on K symbols at a time until every group has ended or is a singleton.  Groups are contiguous slot ranges and homogeneous
// (equal keys end at the same place), so a round works on the list of still-unresolved slots only: no rank array over the
// dictionary, no inverse-permutation scatter, no rank gathers -- and nothing to exchange between ranks that own different
// key ranges.  (Rounds 1 and 2 used prefix doubling over positional ranks: every pass scattered the ranks of the re-sorted
// suffixes to their dictionary positions -- 85 ms of the 10 GB build -- and in the collection-level mode all-gathered them.)
// A group of at most kSegCap members is ordered by counting (every member counts the smaller keys of its group: the keys sit
// in neighbouring words); larger groups take two stable radix sorts together, by key and then by group.
static constexpr u32 kSegCap = 64;
struct GroupStartsFn {    // gstart[dense group id] = head slot ; gstart[G] = S
    const u8 *hflag; const u32 *ex; u64 S; u32 *gstart;
    GRL_DEV void operator()(u64 t) const {
        if (hflag[t]) gstart[ex[t]] = (u32)t;
        if (t == S - 1) gstart[ex[t] + hflag[t]] = (u32)S;
    }
};
struct ExtKeyFn {         // compact the unresolved slots; key = the next K symbols of each (sentinel behind the phrase end)
    // Where the phrase ends comes from the phrase-start bit-vector of the dictionary (S/8 bytes: it stays in the caches),
    // not from a per-position length array: an unresolved suffix has not ended within its first Lres symbols, so it ends
    // at the first phrase start in [q + Lres, q + Lres + K].
    const u32 *act; const u8 *uflag; const u32 *uex; const u32 *perm; const u8 *hflag; const u32 *dict_sym; const u64 *pw; u64 S;
    u64 Lres; int K, b;
    u32 *uslot; u32 *uq; u64 *ukey; u8 *uhead;
    RunKeys rk;
    GRL_DEV void operator()(u64 i) const {
        if (uflag[i]) {
            const u64 t = act ? (u64)act[i] : i;
            const u64 q = perm[t], x = q + Lres + (rk.rb ? (u64)rk.skip[q] : 0ull);
            // bits x .. x+K-1 of the start vector (K <= 16; the vector has a spare word behind position S)
            u64 w = 0;
            if (x < S) {
                w = pw[x >> 6] >> (x & 63);
                if ((x & 63) + (u64)K > 64) w |= pw[(x >> 6) + 1] << (64 - (x & 63));
            }
            u32 valid = x >= S ? 0u : (u32)K;                      // symbols of the key that lie inside the phrase
            const u32 starts = (u32)(w & ((1ull << K) - 1ull));
            if (x < S && starts) valid = (u32)__builtin_ctz(starts);
            if (x < S && x + valid > S) valid = (u32)(S - x);
            const u64 sent = (1ull << b) - 1;
            u64 key = 0;
            for (int j = 0; j < K; j++) key = (key << b) | (((u32)j < valid) ? (u64)dict_sym[x + (u64)j] : sent);
            if (rk.rb) {                           // (the run field of suffix_key0, with the phrase end taken from the start bits)
                u64 low = 0;
                if (valid == (u32)K) {
                    const u32 r = rk.rem[x];
                    if (r >= (u32)K) {
                        const u32 a = dict_sym[x];
                        const u64 xe = x + (u64)r;
                        const bool up = xe >= S || bit_at(pw, xe) || dict_sym[xe] > a;
                        const int R = rk.rb - 1;
                        low = up ? ((1ull << R) | (((1ull << R) - 1ull) - (u64)r)) : (u64)r;
                        rk.skip[q] += r - (u32)K;
                    }
                }
                key = (key << rk.rb) | low;
            }
            const u32 o = uex[i];
            uslot[o] = (u32)t; uq[o] = (u32)q; ukey[o] = key; uhead[o] = hflag[t];
        }
    }
};
struct SegStartFn {       // seg_start[k] = item of the k-th group head among the unresolved items ; seg_start[nseg] = U
    const u8 *uhead; const u32 *hex; u64 U; u32 *seg_start;
    GRL_DEV void operator()(u64 j) const {
        if (uhead[j]) seg_start[hex[j]] = (u32)j;
        if (j == U - 1) seg_start[hex[j] + uhead[j]] = (u32)U;
    }
};
struct SegSortSmallFn {   // one lane per item of a group of at most kSegCap items: stable rank among the group's keys
    const u8 *uhead; const u32 *hex; const u32 *seg_start; const u32 *uslot; const u32 *uq; const u64 *ukey; u64 sent; u32 cap;
    u32 *perm; u8 *hflag; u8 *unext;       // unext[position] = the item now at that position is still unresolved
    GRL_DEV void operator()(u64 j) const {
        const u32 s = hex[j] + uhead[j] - 1;
        const u32 a = seg_start[s], e = seg_start[s + 1];
        if (e - a <= cap) {
            const u64 k = ukey[j];
            u32 less = 0, eq_before = 0, eq_total = 0;
            for (u32 x = a; x < e; x++) {
                const u64 kx = ukey[x];
                less += kx < k ? 1u : 0u;
                const u32 same = kx == k ? 1u : 0u;
                eq_total += same;
                eq_before += (same && x < (u32)j) ? 1u : 0u;
            }
            const u32 pos = a + less + eq_before;
            const u32 dst = uslot[pos];
            perm[dst] = uq[j];
            if (eq_before == 0 && pos != a) hflag[dst] = 1;           // first of its key (the group's first slot is a head already)
            unext[pos] = (eq_total >= 2 && (k & sent) != sent) ? 1 : 0;
        }
    }
};
struct SegBigIn {         // 1 for the items of groups above kSegCap
    const u8 *uhead; const u32 *hex; const u32 *seg_start; u32 cap;
    GRL_DEV u32 operator()(u64 j) const { const u32 s = hex[j] + uhead[j] - 1; return (seg_start[s + 1] - seg_start[s] > cap) ? 1u : 0u; }
};
struct SegBigGatherFn {
    const u8 *uhead; const u32 *hex; const u32 *seg_start; const u32 *bex; const u64 *ukey; u32 cap; u32 *bitem; u64 *bkey; u32 *bidx;
    GRL_DEV void operator()(u64 j) const {
        const u32 s = hex[j] + uhead[j] - 1;
        if (seg_start[s + 1] - seg_start[s] > cap) { const u32 p = bex[j]; bitem[p] = (u32)j; bkey[p] = ukey[j]; bidx[p] = p; }
    }
};
struct SegBigSegKeyFn {   // second sort key of the key-sorted large items: their group
    const u32 *i1; const u32 *bitem; const u8 *uhead; const u32 *hex; u32 *key2;
    GRL_DEV void operator()(u64 p) const { const u32 j = bitem[i1[p]]; key2[p] = hex[j] + uhead[j] - 1; }
};
struct SegBigWriteFn {    // large items now ordered by (group, key): back into the groups' slots
    const u32 *s2; const u32 *i2; const u32 *bitem; const u64 *ukey; const u32 *uq; const u32 *uslot; u64 n; u64 sent;
    u32 *perm; u8 *hflag; u8 *unext;
    GRL_DEV void operator()(u64 p) const {
        const u32 src = bitem[i2[p]], seg = s2[p];
        const u64 k = ukey[src];
        const bool first = p == 0 || s2[p - 1] != seg, last = p + 1 == n || s2[p + 1] != seg;
        const bool eq_prev = !first && ukey[bitem[i2[p - 1]]] == k, eq_next = !last && ukey[bitem[i2[p + 1]]] == k;
        const u32 jt = bitem[p], dst = uslot[jt];                     // the p-th large item's place: same groups, same sizes, same order
        perm[dst] = uq[src];
        if (!first && !eq_prev) hflag[dst] = 1;
        unext[jt] = ((eq_prev || eq_next) && (k & sent) != sent) ? 1 : 0;
    }
};
// DOUBLING ROUNDS (single-GPU rounds, levels with long phrases, once the symbol extension has run a few rounds without
// finishing: long phrases that are NOT runs -- strictly monotone ramps over a large alphabet -- shared by several strings
// keep their suffix groups tied for their whole length, K symbols per round).  The suffix at position q, sorted to depth
// d(q) = Ld + skip[q], takes as its key the current group of the suffix at q + d(q) (Larsson-Sadakane): groups are totally
// ordered consistently with the final order, so members whose targets lie in different groups are ordered for good, and
// members whose targets share a group T agree on depth(T) more symbols.  Key = group number << 1 | (T resolved): equal keys
// with a resolved T are equal suffixes (the segment sorts' "still unresolved" test reads the low bit: sent = 1); a suffix
// that has ended takes the largest key.
struct SlotOfFn {
    const u32 *perm; u32 *slot_of;
    GRL_DEV void operator()(u64 t) const { slot_of[perm[t]] = (u32)t; }
};
struct UresInitFn {       // the unresolved flags of the active list, by slot
    const u32 *act; const u8 *uflag; u8 *ures;
    GRL_DEV void operator()(u64 i) const { if (uflag[i]) ures[act ? (u64)act[i] : i] = 1; }
};
struct DoubleKeyFn {
    const u32 *act; const u8 *uflag; const u32 *uex; const u32 *perm; const u8 *hflag;
    const u32 *dict_phr; const u32 *ph_off; const u32 *slot_of; const u32 *rank_ex; const u8 *ures; const u32 *skip;
    u64 Ld, Sg;
    u32 *uslot; u32 *uq; u64 *ukey; u8 *uhead; u32 *nskip;
    GRL_DEV void operator()(u64 i) const {
        if (uflag[i]) {
            const u64 t = act ? (u64)act[i] : i;
            const u64 q = perm[t];
            const u64 sk = skip[q], x = q + Ld + sk, end = ph_off[dict_phr[q] + 1];
            u64 key = (Sg << 1) | 1ull;                     // the suffix has ended: behind every real group, resolved
            u64 ns = sk;
            if (x < end) {
                const u32 st = slot_of[x];
                const bool tu = ures[st] != 0;
                key = ((u64)(rank_ex[st] + hflag[st] - 1) << 1) | (tu ? 0ull : 1ull);
                if (tu) ns = sk + Ld + (u64)skip[x];
            }
            const u32 o = uex[i];
            uslot[o] = (u32)t; uq[o] = (u32)q; ukey[o] = key; uhead[o] = hflag[t]; nskip[o] = (u32)ns;
        }
    }
};
struct AfterDoubleFn {    // item j: its position's new depth; place j: which position sits there now and whether it is still unresolved
    const u32 *uq; const u32 *nskip; const u32 *uslot; const u32 *perm; const u8 *unext; u32 *skip; u32 *slot_of; u8 *ures;
    GRL_DEV void operator()(u64 j) const {
        skip[uq[j]] = nskip[j];
        const u32 sl = uslot[j];
        slot_of[perm[sl]] = sl;
        ures[sl] = unext[j];
    }
};
struct DenseGidFn {       // final dense group id of every slot
    const u8 *hflag; const u32 *ex; u32 *gid;
    GRL_DEV void operator()(u64 t) const { gid[t] = ex[t] + hflag[t] - 1; }
};

// -------------------------------------------- a7: groups -> pre-BWT + ranks
// Per dictionary position q (coalesced pass): left symbol (or the BWT marker for a whole phrase)
// and the frequency of its phrase, so that the pass over the sorted suffixes needs ONE gather.
// (the 64-bit build's record has four spare bytes: they carry the phrase, which the whole-phrase suffixes need)
template <int IB> struct SufRecT;
template <> struct alignas(8) SufRecT<4> { u32 freq; u32 left; GRL_HD void set_phr(u32) {} GRL_HD u32 phr(const u32 *dict_phr, u32 q) const { return dict_phr[q]; } };
template <> struct alignas(16) SufRecT<8> { u64 freq; u32 left; u32 k; GRL_HD void set_phr(u32 v) { k = v; } GRL_HD u32 phr(const u32 *, u32) const { return k; } };
typedef SufRecT<sizeof(idx_t)> SufRec;      // one 8/16-byte gather per sorted suffix
// left carries two flags above the symbol (symbols are < 2^30): the suffix is the last cell of its phrase, and that
// phrase ends a string -- what the group decision needs from the group's first member.
// What the group fold reads per sorted suffix: from the array SuffixRecFn streamed out (RecArray), or computed where it is
// needed (RecCompute: five gathers per member instead of one, no array over the whole dictionary -- the collection-level mode
// from GRLBWT_DIST_REC_FLY_MIN ranks on (default 8), where a rank folds 1/N of the suffixes and the array would still be S
// records of 8-16 bytes: 9.9 GB at level 2 of the 10 GB collection)
struct RecArray {
    const SufRec *rec;
    GRL_DEV SufRec operator()(u32, u32 q) const { return rec[q]; }
};
struct RecWire {           // by arrival index: the 8-byte record that came with the suffix's (key, position) from the owner of the position (frequency << 32 | left symbol and flags)
    const u64 *rec;
    GRL_DEV SufRecT<8> operator()(u32, u32 a) const { const u64 x = rec[a]; SufRecT<8> r; r.freq = x >> 32; r.left = (u32)x; r.k = 0; return r; }
};
struct RecSlot {           // by sorted slot: what the owners of the positions answered (dictionary sharded by owner; the phrase number travels in the record in both index widths)
    const SufRecT<8> *rec;
    GRL_DEV SufRecT<8> operator()(u32 j, u32) const { return rec[j]; }
};
struct RecCompute {
    const u32 *dict_sym; const u32 *dict_phr; const u32 *ph_off; const idx_t *ph_freq; const u8 *ph_lastT; u32 bwt_code;
    u32 phr_base = 0;             // (my part of a dictionary sharded by owner: phrase numbers travel as global ones)
    GRL_DEV SufRec operator()(u32, u32 q) const {
        const u32 k = dict_phr[q];
        SufRec r;
        r.freq = ph_freq[k];
        r.left = (((u64)q == (u64)ph_off[k]) ? bwt_code : dict_sym[q - 1]) | (((u64)q + 1 == (u64)ph_off[k + 1]) ? kRecFinal : 0u) | (ph_lastT[k] ? kRecLastT : 0u);
        r.set_phr(k + phr_base);
        return r;
    }
};
struct SuffixRecFn {
    const u32 *dict_sym; const u32 *dict_phr; const u32 *ph_off; const idx_t *ph_freq; const u8 *ph_lastT; u32 bwt_code;
    SufRec *rec;
    GRL_DEV void operator()(u64 q) const {
        u32 k = dict_phr[q];
        SufRec r;
        r.freq = ph_freq[k];
        r.left = ((q == ph_off[k]) ? bwt_code : dict_sym[q - 1]) | ((q + 1 == ph_off[k + 1]) ? kRecFinal : 0u) | (ph_lastT[k] ? kRecLastT : 0u);
        r.set_phr(k);
        rec[q] = r;
    }
};
// Per equal-suffix group: min/max of the left symbol, sum of frequencies, "contains a whole
// phrase".  Groups of <= kGroupChunk members (the overwhelming majority) are folded by ONE LANE
// PER GROUP with plain stores (GroupAccumSmallFn, every lane busy); larger groups are cut into chunks
// of kGroupChunk consecutive members, each folded by its first lane and combined with one set of
// atomics per chunk (GroupAccumLargeFn; 32x fewer same-address atomics than one per member).
static constexpr u32 kGroupChunk = 32;
template <class REC>
struct GroupAccumSmallFn {
    const u32 *perm; const u32 *gstart; REC rec; const u32 *dict_phr;
    u32 bwt_code;
    u32 *gmin; u32 *gmax; idx_t *gacc; u8 *gfull; u8 *gflag; u32 *pslot;
    u32 *gphr;            // non-null: the whole phrase of a group is recorded BY GROUP (sequential store) instead of pslot[phrase] = group
    bool slot_mode = false;           // ... as the SLOT of that member, not its phrase number (dictionary sharded by owner: the slot gives its position and owner)
    GRL_DEV void operator()(u64 g) const {
        const u32 t0 = gstart[g], t1 = gstart[g + 1];
        const bool large = t1 - t0 > kGroupChunk;   // folded by GroupAccumLargeFn with atomics: start from the identities
        u32 mn = 0xFFFFFFFFu, mx = 0, first = 0; idx_t acc = 0; u8 fl = 0;
        // (one exit, no early return: see the note in find_or_insert.  Four members per iteration with their gathers issued
        // together was measured slower: 59 vs 56 ms at 10 GB -- the average group has 2.2 members, the padding gathers cost
        // more than the overlap gives)
        const u32 te = large ? t0 : t1;
        for (u32 j = t0; j < te; j++) {
            const u32 q = perm[j];
            const auto r = rec(j, q);
            if (j == t0) first = r.left;
            u32 left = r.left & kRecSym;
            mn = left < mn ? left : mn; mx = left > mx ? left : mx;
            acc += (idx_t)r.freq;
            if (left == bwt_code) {                 // a whole phrase: the group its metasymbol will be read from
                fl = 1;
                const u32 k = slot_mode ? (u32)j : r.phr(dict_phr, q);
                if (gphr) gphr[g] = k; else pslot[k] = (u32)g;
            }
        }
        gmin[g] = mn; gmax[g] = mx; gacc[g] = acc; gfull[g] = fl;
        if (!large) {
            // the group decision (GroupDecideFn) from what is already in registers
            bool valid = !(first & kRecFinal) || (first & kRecLastT);     // exact_par_phase.cpp:162
            bool ranked = valid && (mn != mx || fl);                       // :187
            gflag[g] = (valid ? 1 : 0) | (ranked ? 2 : 0) | (t1 - t0 > 1 ? 4 : 0);
        }
    }
};
struct GroupChunksIn {    // chunks of a group above kGroupChunk members (0 for the others)
    const u32 *gstart;
    GRL_DEV u32 operator()(u64 g) const { const u32 sz = gstart[g + 1] - gstart[g]; return sz > kGroupChunk ? (sz + kGroupChunk - 1) / kGroupChunk : 0u; }
};
template <class REC>
struct GroupAccumLargeFn {   // one lane per chunk of a large group (a lane per slot spent 27 ms at 10 GB finding out it had nothing to do)
    const u32 *perm; const u32 *coff; u64 G; const u32 *gstart; REC rec; const u32 *dict_phr;
    u32 bwt_code;
    u32 *gmin; u32 *gmax; idx_t *gacc; u8 *gfull; u32 *pslot; u32 *gphr;
    bool slot_mode = false;
    GRL_DEV void operator()(u64 c) const {
        const u32 g = (u32)upper_bound<u32>(coff, G, (u32)c) - 1;     // the group with coff[g] <= c < coff[g + 1]
        const u32 t1 = gstart[g + 1], t = gstart[g] + ((u32)c - coff[g]) * kGroupChunk;
        const u32 te = t + kGroupChunk < t1 ? t + kGroupChunk : t1;
        u32 mn = 0xFFFFFFFFu, mx = 0; idx_t acc = 0; u8 fl = 0;
        // (eight members per step, their gathers in flight together: a lane that took its 32 members one dependent pair of loads
        // after the other spent 7.4 ms per 10 GB build here; padding is rare -- only a group's last chunk is short)
        constexpr u32 kB = 8;
        for (u32 j0 = t; j0 < te; j0 += kB) {
            u32 q[kB];
#pragma unroll
            for (u32 x = 0; x < kB; x++) q[x] = perm[j0 + x < te ? j0 + x : j0];
            decltype(rec(j0, q[0])) r[kB];
#pragma unroll
            for (u32 x = 0; x < kB; x++) r[x] = rec(j0 + x < te ? j0 + x : j0, q[x]);
#pragma unroll
            for (u32 x = 0; x < kB; x++) {
                if (j0 + x < te) {
                    const u32 left = r[x].left & kRecSym;
                    mn = left < mn ? left : mn; mx = left > mx ? left : mx;
                    acc += (idx_t)r[x].freq;
                    if (left == bwt_code) {
                        fl = 1;
                        const u32 k = slot_mode ? (u32)(j0 + x) : r[x].phr(dict_phr, q[x]);
                        if (gphr) gphr[g] = k; else pslot[k] = g;
                    }
                }
            }
        }
        prim::atomic_min(&gmin[g], mn);
        prim::atomic_max(&gmax[g], mx);
        prim::atomic_add(&gacc[g], acc);
        if (fl) gfull[g] = 1;
    }
};
enum : u8 { GF_VALID = 1, GF_RANKED = 2, GF_MULTI = 4 };
template <class REC>
struct GroupDecideFn {
    const u32 *perm; const u32 *gstart; REC rec;
    const u32 *gmin; const u32 *gmax; const u8 *gfull;
    u8 *gflag;
    GRL_DEV void operator()(u64 g) const {
        u32 t0 = gstart[g], size = gstart[g + 1] - t0;
        if (size <= kGroupChunk) return;                            // decided by GroupAccumSmallFn
        const u32 first = rec(t0, perm[t0]).left;                   // (the flags of the group's first member, as the small fold reads them)
        bool pfinal = (first & kRecFinal) != 0;
        bool valid = !pfinal || (first & kRecLastT);                // exact_par_phase.cpp:162
        bool ranked = valid && (gmin[g] != gmax[g] || gfull[g]);    // :187
        gflag[g] = (valid ? GF_VALID : 0) | (ranked ? GF_RANKED : 0) | (size > 1 ? GF_MULTI : 0);
    }
};
struct FlagIn {
    const u8 *f; u8 m;
    GRL_DEV u32 operator()(u64 i) const { return (f[i] & m) ? 1u : 0u; }
};
struct RankedValidIn {    // (group is ranked, group is valid)
    const u8 *f;
    GRL_DEV prim::Pair<u32, u32> operator()(u64 i) const { const u8 x = f[i]; return prim::Pair<u32, u32>((x & GF_RANKED) ? 1u : 0u, (x & GF_VALID) ? 1u : 0u); }
};
struct SplitPairEmitFn {  // the two prefixes into arrays of their own
    static constexpr bool kWaveEmit = false;
    u32 *a; u32 *b;
    GRL_DEV void operator()(u64 i, prim::Pair<u32, u32> ex, prim::Pair<u32, u32>) const { a[i] = ex.a; b[i] = ex.b; }
};
struct GroupEmitFn {      // (m_off, p_off: metasymbols / pre-BWT entries of the ranks in front of me when the groups are sharded)
    const u8 *gflag; const u32 *grank; const u32 *pidx; const u32 *gmin; const idx_t *gacc; const u32 *gstart; const u32 *perm;
    u32 bwt_code, hocc_code, m_off, p_off;
    u32 *psym; idx_t *plen; u8 *has_hocc; u32 *repq; u32 *u_to_p0; u32 *pu0;
    GRL_DEV void operator()(u64 g) const {
        u8 f = gflag[g];
        if (f & GF_VALID) {
            u32 j = pidx[g];
            pu0[j] = m_off + grank[g];   // metasymbols in front of this pre-BWT run (grank is the exclusive count of ranked groups)
            u32 s = gmin[g];
            if (f & GF_RANKED) {
                s = (f & GF_MULTI) ? hocc_code : bwt_code;
                u32 u = grank[g];
                has_hocc[u] = (f & GF_MULTI) ? 1 : 0;
                repq[u] = perm ? perm[gstart[g]] : gstart[g];      // (perm null: the representative's SLOT, for positions kept as (owner, offset))
                u_to_p0[u] = p_off + j;
            }
            psym[j] = s;
            plen[j] = gacc[g];
        }
    }
};

struct PreToMetaFn {     // merged pre-BWT run -> number of metasymbols in front of it (the value of its first member)
    const u32 *pu0; const u32 *merged; u32 *p_to_u;
    GRL_DEV void operator()(u64 j) const { if (j == 0 || merged[j] != merged[j - 1]) p_to_u[merged[j]] = pu0[j]; }
};
struct ComposeMapFn {    // metasymbol -> merged pre-BWT run
    const u32 *u_to_p0; const u32 *merged; u32 *u_to_p;
    GRL_DEV void operator()(u64 u) const { u_to_p[u] = merged[u_to_p0[u]]; }
};

// ------------------------------------------------------------- a8: grammar
// meta[q] = metasymbol of the suffix at dictionary position q if its group is ranked and has > 1 member (the marked
// positions, phr_marks, exact_par_phase.cpp:203-205), else 0 (metasymbols are >= sigma3 > 0), so that the grammar walk
// below reads meta[] sequentially and needs no further lookups.  Filled from the SORTED side: slot t knows its dense
// group gid[t] (non-decreasing: ginfo[] is read in order) and only the members of marked groups scatter through perm[]
// (the earlier form, by dictionary position through rank[q] -> gid -> ginfo, was two dependent random gathers for
// every position: 57 ms of the 10 GB build).
struct PackGroupInfoFn {
    const u32 *grank; const u8 *gflag; u32 m_off; u32 *ginfo;
    GRL_DEV void operator()(u64 g) const {       // rank < 2^30 (alphabet bound of the next level)
        ginfo[g] = ((m_off + grank[g]) << 1) | (((gflag[g] & (GF_RANKED | GF_MULTI)) == (GF_RANKED | GF_MULTI)) ? 1u : 0u);
    }
};
struct MarkedIn {          // 1 for the slots MetaPosFn would write (members of marked groups)
    const u32 *gid; const u32 *ginfo;
    GRL_DEV u32 operator()(u64 t) const { return ginfo[gid[t]] & 1u; }
};
struct MetaPairFn {        // (position << 32 | metasymbol) of the marked slots, compacted: what a rank tells the others
    const u32 *perm; const u32 *gid; const u32 *ginfo; const u32 *ex; u32 sigma3; u64 *pairs;
    const u32 *pown = nullptr; u32 *own = nullptr;      // (positions as (owner, offset): the owner of slot t, copied beside the pair)
    GRL_DEV void operator()(u64 t) const {
        u32 gi = ginfo[gid[t]];
        if (gi & 1u) { pairs[ex[t]] = ((u64)perm[t] << 32) | (u64)((gi >> 1) + sigma3); if (own) own[ex[t]] = pown[t]; }
    }
};
// The grammar walk reads ONE array: dm[q] = meta << 32 | last-cell-of-its-phrase << 31 | that-phrase-ends-a-string << 30 | symbol
// (symbols and metasymbols are < 2^30).  A walk then touches one or two neighbouring 64-byte lines per metasymbol.  (Rounds 1-2 kept
// meta[], dict_sym[], the phrase-start bit-vector, dict_phr[] and ph_lastT[] apart: five random lines for 8 bytes of
// output, 161 GB fetched to write 4 GB at 10 GB -- 46 ms.)
static constexpr u32 kDmEnd = 0x80000000u, kDmLastT = 0x40000000u, kDmSym = 0x3FFFFFFFu;
struct DictMetaInitFn {    // streaming: low half of dm[] from the dictionary, meta = 0  (q0: dm[] describes the positions from q0 on)
    const u32 *dict_sym; const u32 *dict_phr; const u32 *ph_off; const u8 *ph_lastT; u64 *dm; u64 q0 = 0;
    GRL_DEV void operator()(u64 i) const {
        const u64 q = q0 + i;
        const u32 k = dict_phr[q];
        const bool end = q + 1 == (u64)ph_off[k + 1];
        dm[i] = (u64)(dict_sym[q] | (end ? kDmEnd : 0u) | ((end && ph_lastT[k]) ? kDmLastT : 0u));
    }
};
struct MetaPosFn {         // by sorted slot t: gid[] and ginfo[] are read in order, only the marked suffixes scatter (the high halves of dm[])
    const u32 *perm; const u32 *gid; const u32 *ginfo; u32 sigma3; u64 *dm;
    GRL_DEV void operator()(u64 t) const {
        u32 gi = ginfo[gid[t]];
        if (gi & 1u) reinterpret_cast<u32 *>(dm)[2 * (u64)perm[t] + 1] = (gi >> 1) + sigma3;
    }
};
struct ApplyMetaPairsFn {  // the same from (position << 32 | metasymbol) pairs (sharded dictionary stage; q0: first position of dm[])
    const u64 *pairs; u64 *dm; u64 q0 = 0;
    GRL_DEV void operator()(u64 i) const { const u64 p = pairs[i]; reinterpret_cast<u32 *>(dm)[2 * ((p >> 32) - q0) + 1] = (u32)p; }
};
// Collection-level mode, grammar passes sharded by the OWNER of a dictionary position (round 5): rank g merged -- and holds the
// walk array dm[] of -- the positions [sbase[g], sbase[g + 1]).  Marks and walk requests travel as 64-bit records with the
// position in the high half; a record goes to the rank whose part holds its position.
struct PosOwnerFn {        // own[i] = that rank, for record i
    const u64 *rec; const u64 *sbase; int N; u32 *own;
    GRL_DEV void operator()(u64 i) const {
        const u64 q = rec[i] >> 32;
        u32 d = 0;
        for (int r = 1; r < N; r++) if (q >= sbase[r]) d = (u32)r;      // the LAST rank whose part starts at or below q (empty parts in front of it share the start)
        own[i] = d;
    }
};
struct ExtCompactFn {      // ExtKeyFn without the key: the unresolved slots compacted, and the request for each one's next K symbols (position << 32 | item)
    const u32 *act; const u8 *uflag; const u32 *uex; const u32 *perm; const u8 *hflag;
    u32 *uslot; u32 *uq; u8 *uhead; u64 *req;
    GRL_DEV void operator()(u64 i) const {
        if (uflag[i]) {
            const u64 t = act ? (u64)act[i] : i;
            const u32 q = perm[t], o = uex[i];
            uslot[o] = (u32)t; uq[o] = q; uhead[o] = hflag[t]; req[o] = ((u64)q << 32) | o;
        }
    }
};
struct ExtCompactAiFn {    // the same when the sorted values are ARRIVAL INDICES (records travelled with the suffixes): the position for the request comes from pos_of[]
    const u32 *act; const u8 *uflag; const u32 *uex; const u32 *perm; const u8 *hflag; const u32 *pos_of;
    u32 *uslot; u32 *uq; u8 *uhead; u64 *req;
    const u64 *bounds = nullptr; int N = 0; u32 *own = nullptr;      // (positions as (owner, offset): the owner of an arrival is the rank it came from)
    GRL_DEV void operator()(u64 i) const {
        if (uflag[i]) {
            const u64 t = act ? (u64)act[i] : i;
            const u32 a = perm[t], o = uex[i];
            uslot[o] = (u32)t; uq[o] = a; uhead[o] = hflag[t]; req[o] = ((u64)pos_of[a] << 32) | o;
            if (own) { u32 d = 0; for (int r = 1; r < N; r++) if ((u64)a >= bounds[r]) d = (u32)r; own[o] = d; }
        }
    }
};
struct ArrivalOwnerFn {    // own[t] = the rank the arrival index[t] came from (arrivals sit in sender order: bounds[g] <= a < bounds[g + 1])
    const u32 *index; const u64 *bounds; int N; u32 *own;
    GRL_DEV void operator()(u64 t) const {
        const u64 a = index[t];
        u32 d = 0;
        for (int r = 1; r < N; r++) if (a >= bounds[r]) d = (u32)r;
        own[t] = d;
    }
};
struct GatherU32Fn {       // out[t] = table[index[t]]
    const u32 *index; const u32 *table; u32 *out;
    GRL_DEV void operator()(u64 t) const { out[t] = table[index[t]]; }
};
struct GatherU64Fn {       // out[t] = table[index[t]]
    const u32 *index; const u64 *table; u64 *out;
    GRL_DEV void operator()(u64 t) const { out[t] = table[index[t]]; }
};
struct IotaU32Fn {
    u32 *out;
    GRL_DEV void operator()(u64 i) const { out[i] = (u32)i; }
};
struct GroupPosPairFn {    // (position of the group's whole-phrase member << 32 | metasymbol rank) of the groups that hold one, compacted; gslot[g] = that member's slot
    const u8 *gfull; const u32 *gslot; const u32 *grank; u32 m_off; const u32 *ex; u64 *pairs;
    const u32 *perm; const u32 *pown; u32 *own;
    GRL_DEV void operator()(u64 g) const {
        if (gfull[g]) { const u32 t = gslot[g]; pairs[ex[g]] = ((u64)perm[t] << 32) | (u64)(m_off + grank[g]); own[ex[g]] = pown[t]; }
    }
};
struct ApplyPosPairsFn {   // on the owner: rank of the phrase whose first cell sits at my position (pair >> 32) - s0
    const u64 *pairs; const u32 *dict_phr; u64 s0; u32 *phrase_rank;
    GRL_DEV void operator()(u64 i) const { const u64 p = pairs[i]; phrase_rank[dict_phr[(p >> 32) - s0]] = (u32)p; }
};
struct ExtKeyOwnerFn {     // on the owner: symbols [x, x + K) of the suffix at MY position (req >> 32) - s0, x = that + Lres (ExtKeyFn's key, no run field)
    const u64 *req; const u32 *dict_sym; const u64 *pw; u64 S, s0, Lres; int K, b; u64 *ans;
    GRL_DEV void operator()(u64 i) const {
        const u64 x = (req[i] >> 32) - s0 + Lres;
        u64 w = 0;
        if (x < S) {
            w = pw[x >> 6] >> (x & 63);
            if ((x & 63) + (u64)K > 64) w |= pw[(x >> 6) + 1] << (64 - (x & 63));
        }
        u32 valid = x >= S ? 0u : (u32)K;
        const u32 starts = (u32)(w & ((1ull << K) - 1ull));
        if (x < S && starts) valid = (u32)__builtin_ctz(starts);
        if (x < S && x + valid > S) valid = (u32)(S - x);
        const u64 sent = (1ull << b) - 1;
        u64 key = 0;
        for (int j = 0; j < K; j++) key = (key << b) | (((u32)j < valid) ? (u64)dict_sym[x + (u64)j] : sent);
        ans[i] = key;
    }
};
struct ExtAnswerFn {       // ukey[item] = the owner's answer
    const u64 *req; const u64 *ans; u64 *ukey;
    GRL_DEV void operator()(u64 i) const { ukey[(u32)req[i]] = ans[i]; }
};
struct RecRequestFn {      // (position << 32 | slot) of every sorted suffix of my key range
    const u32 *perm; u64 *req;
    GRL_DEV void operator()(u64 t) const { req[t] = ((u64)perm[t] << 32) | t; }
};
struct RecLocalFn {        // on the owner, streaming: what the group fold wants to know about the suffix at each of MY positions
    RecCompute rc; SufRecT<8> *out;
    GRL_DEV void operator()(u64 q) const {
        const u32 k = rc.dict_phr[q];
        SufRecT<8> r;
        r.freq = (u64)rc.ph_freq[k];
        r.left = ((q == (u64)rc.ph_off[k]) ? rc.bwt_code : rc.dict_sym[q - 1]) | ((q + 1 == (u64)rc.ph_off[k + 1]) ? kRecFinal : 0u) | (rc.ph_lastT[k] ? kRecLastT : 0u);
        r.k = k + rc.phr_base;
        out[q] = r;
    }
};
struct RecOwnerFn {        // ... and the answer to a request: ONE gather (computed per request it was five dependent ones: 83 ms per rank at N = 2 of the 10 GB collection)
    const u64 *req; const SufRecT<8> *local; u64 s0; SufRecT<8> *ans;
    GRL_DEV void operator()(u64 i) const { ans[i] = local[(req[i] >> 32) - s0]; }
};
struct RecAnswerFn {       // recs[slot] = the owner's answer
    const u64 *req; const SufRecT<8> *ans; SufRecT<8> *recs;
    GRL_DEV void operator()(u64 i) const { recs[(u32)req[i]] = ans[i]; }
};
struct WalkRequestFn {     // (position of the representative << 32 | my metasymbol) for every metasymbol of my key range
    const u32 *repq; u64 *req;
    const u32 *perm = nullptr; const u32 *pown = nullptr; u32 *own = nullptr;      // (positions as (owner, offset): repq[] holds the representative's SLOT)
    GRL_DEV void operator()(u64 u) const {
        if (own) { const u32 t = repq[u]; req[u] = ((u64)perm[t] << 32) | u; own[u] = pown[t]; }
        else req[u] = ((u64)repq[u] << 32) | u;
    }
};
struct WalkAnswerFn {      // answers come back in the order the requests left: g1 << 32 | g0 of the metasymbol in the request's low half
    const u64 *req; const u64 *ans; u32 *g0; u32 *g1;
    GRL_DEV void operator()(u64 i) const { const u32 u = (u32)req[i]; const u64 a = ans[i]; g0[u] = (u32)a; g1[u] = (u32)(a >> 32); }
};
struct DmStopBitsFn {      // one lane per word: bit q = dm[q] is marked or the last cell of its phrase (where a grammar walk stops)
    const u64 *dm; u64 S; u64 *bits;
    GRL_DEV void operator()(u64 w) const {
        u64 m = 0;
        for (u64 j = 0; j < 64 && w * 64 + j < S; j++) { const u64 e = dm[w * 64 + j]; if ((e >> 32) || ((u32)e & kDmEnd)) m |= 1ull << j; }
        bits[w] = m;
    }
};
struct GrammarFn {
    const u32 *repq; const u64 *dm;
    u32 MD;
    u32 *g0; u32 *g1;
    const u64 *stops = nullptr; u64 S = 0;      // levels with very long phrases: where the walks stop, as a bit-vector (DmStopBitsFn)
    const u64 *req = nullptr; u64 q0 = 0; u64 *ans = nullptr;      // walks asked for by other ranks: start at (req[u] >> 32) - q0, answer g1 << 32 | g0
    GRL_DEV void operator()(u64 u) const {
        u64 x = req ? (req[u] >> 32) - q0 : (u64)repq[u];
        u64 prev = dm[x];
        u32 a = MD, b = 0;
        bool done = false;
        if ((u32)prev & kDmEnd) { b = (u32)prev & kDmSym; done = true; }               // the representative is the last cell of its phrase (:38-41)
        if (!done && stops) {                      // jump to the cell in front of the stop (a walk over 10^8 cells took 14 s)
            const u64 y = next_set_bit_far(stops, x + 1, S);
            if (y > x + 1) { x = y - 1; prev = dm[x]; }
        }
        while (!done) {
            const u64 e = dm[++x];
            const u32 m = (u32)(e >> 32);
            if (m) { a = (u32)prev & kDmSym; b = m; done = true; }                      // :49-80
            else if ((u32)e & kDmEnd) {                                                 // x is the last cell of the phrase (:81-85)
                b = ((u32)e & kDmLastT) ? ((u32)e & kDmSym) : ((u32)prev & kDmSym);
                done = true;
            } else prev = e;
        }
        if (ans) ans[u] = ((u64)b << 32) | (u64)a;
        else { g0[u] = a; g1[u] = b; }
    }
};

// ------------------------------------------- a9 + a10: ranks -> next text
struct PhraseValFn {      // pslot[k] = equal-suffix group of phrase k's whole-phrase suffix (recorded by the group accumulation)
    const u32 *pslot; const idx_t *ph_freq; const u8 *ph_lastT; const u32 *grank;
    u32 *phrase_val;
    GRL_DEV void operator()(u64 k) const {
        u32 r = grank[pslot[k]];
        phrase_val[k] = (r << 2) | ((ph_freq[k] > 1) ? 2u : 0u) | (ph_lastT[k] ? 1u : 0u);
    }
};
// Single-GPU rounds: the value of a phrase goes from the GROUP of its whole-phrase suffix straight to the table slot the parse
// reads it from -- one gather (the phrase's slot and flags, packed) and one scatter per phrase.  (Rounds 1-2: pslot[phrase] =
// group scattered from the sorted side, the rank gathered back per phrase, the value scattered to the slot: two random
// scatters of D stores each, ~23 G/s in arrays of this size -- tools/membench.hip.)
struct PackPhraseInfoFn {  // pinfo[k] = slot | (freq > 1) << 32 | ends-a-string << 33
    const u32 *ph_slot; const idx_t *ph_freq; const u8 *ph_lastT; u64 *pinfo;
    GRL_DEV void operator()(u64 k) const { pinfo[k] = (u64)ph_slot[k] | ((ph_freq[k] > 1) ? (1ull << 32) : 0ull) | (ph_lastT[k] ? (1ull << 33) : 0ull); }
};
struct GroupPhraseValFn {
    const u8 *gfull; const u32 *gphr; const u32 *grank; const u64 *pinfo; u32 *slot_val;
    // Phrases [0, Ds) are the record phrases of the partitioned naming: their slots are slot0 + k, so the rank goes there with ONE
    // random store and the two flag bits of the value are OR-ed in where the value is read (PartValFn: a byte per phrase, read next to
    // the value).  Only the table phrases behind them -- 1-12 % -- take the gather of (slot, flags) this pass used to pay for everyone
    // (two random accesses per phrase: 27 ms per 10 GB build).
    u64 Ds; u32 slot0;
    GRL_DEV void operator()(u64 g) const {
        if (gfull[g]) {
            const u32 k = gphr[g];
            if ((u64)k < Ds) slot_val[slot0 + k] = grank[g] << 2;
            else {
                const u64 pi = pinfo[(u64)k - Ds];
                slot_val[(u32)pi] = (grank[g] << 2) | ((pi >> 32) & 1ull ? 2u : 0u) | ((pi >> 33) & 1ull ? 1u : 0u);
            }
        }
    }
};
struct ScatterValFn {
    const u32 *ph_slot; const u32 *val; u32 *slot_val;
    GRL_DEV void operator()(u64 k) const { slot_val[ph_slot[k]] = val[k]; }
};
struct MapFn {            // one lane per FOUR consecutive cells: a 16-byte load, four independent gathers, a 16-byte store
    const u32 *slot_val; u32 *text; u64 n;      // (one cell per lane ran the 2.9 G cells of level 0 of the 10 GB build at 1.9 TB/s: 12 ms)
    struct alignas(16) Quad { u32 v[4]; };
    GRL_DEV void operator()(u64 j) const {
        const u64 i = 4 * j;
        if (i + 4 <= n) {
            const Quad q = *reinterpret_cast<const Quad *>(text + i);
            *reinterpret_cast<Quad *>(text + i) = Quad{{slot_val[q.v[0]], slot_val[q.v[1]], slot_val[q.v[2]], slot_val[q.v[3]]}};
        } else {
            for (u64 x = i; x < n; x++) text[x] = slot_val[text[x]];
        }
    }
};

// ------------------------------------------------------------- rank bitmaps
// A sorted list of boundary positions on an axis of `nbits` positions becomes a bit-vector
// with a popcount prefix per 64-bit word, so "how many boundaries are < x" (the merge rank
// of x against the list) costs two loads instead of a binary search.
struct RankBits {
    DBuf<u64> words;
    DBuf<idx_t> base;
};
struct BuildBitsFn {      // one lane per 16 consecutive (sorted, distinct) positions: OR them into their words
    const idx_t *pos; u64 count; u64 *words;
    u64 ws = 1;           // word i of the bit-vector sits at words[i * ws] (interleaved layouts: RankCell keeps a word and its rank side by side)
    struct alignas(16) Chunk { idx_t v[16 / sizeof(idx_t)]; };     // 16 bytes of positions per load
    GRL_DEV void operator()(u64 j) const {
        u64 i0 = j * 16, i1 = i0 + 16 < count ? i0 + 16 : count;
        idx_t x16[16];
        if (i1 - i0 == 16 && ((uintptr_t)pos & 15) == 0) {
            constexpr int per = 16 / (int)sizeof(idx_t);
#pragma unroll
            for (int c = 0; c < 16 / per; c++) {
                Chunk ch = *reinterpret_cast<const Chunk *>(pos + i0 + c * per);
#pragma unroll
                for (int k = 0; k < per; k++) x16[c * per + k] = ch.v[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) x16[k] = (i0 + k < i1) ? pos[i0 + k] : (idx_t)0;
        }
        u64 cur = (u64)x16[0] >> 6, m = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (i0 + k < i1) {
                u64 x = x16[k], w = x >> 6;
                if (w != cur) { prim::atomic_or(&words[cur * ws], m); m = 0; cur = w; }
                m |= 1ull << (x & 63);
            }
        }
        prim::atomic_or(&words[cur * ws], m);
    }
};
// # boundaries in [0, x)
GRL_HD u64 rank1(const u64 *words, const idx_t *base, u64 x) {
    return (u64)base[x >> 6] + (u64)__builtin_popcountll(words[x >> 6] & ((1ull << (x & 63)) - 1ull));
}
static inline void build_rankbits(RankBits &rb, const idx_t *pos, u64 count, u64 nbits, const char *name) {
    u64 nw = nbits / 64 + 2;
    rb.words.alloc(nw);
    rb.base.alloc(nw + 1);
    rb.words.zero();
    prim::for_each((count + 15) / 16, BuildBitsFn{pos, count, rb.words.p}, name);
    prim::exclusive_scan_nosync<idx_t>(nw, PopcIn{rb.words.p}, rb.base.p, true, name);
}

struct BuildBits32Fn {    // the same for 32-bit positions (phrase offsets of the dictionary)
    const u32 *pos; u64 count; u64 *words;
    GRL_DEV void operator()(u64 j) const {
        u64 i0 = j * 16, i1 = i0 + 16 < count ? i0 + 16 : count;
        u64 cur = (u64)pos[i0] >> 6, m = 0;
        for (u64 i = i0; i < i1; i++) {
            u64 x = pos[i], w = x >> 6;
            if (w != cur) { prim::atomic_or(&words[cur], m); m = 0; cur = w; }
            m |= 1ull << (x & 63);
        }
        prim::atomic_or(&words[cur], m);
    }
};
static inline void build_rankbits32(RankBits &rb, const u32 *pos, u64 count, u64 nbits, const char *name) {
    u64 nw = nbits / 64 + 2;
    rb.words.alloc(nw);
    rb.base.alloc(nw + 1);
    rb.words.zero();
    prim::for_each((count + 15) / 16, BuildBits32Fn{pos, count, rb.words.p}, name);
    prim::exclusive_scan_nosync<idx_t>(nw, PopcIn{rb.words.p}, rb.base.p, true, name);
}

// ------------------------------------------------------------ run utilities
template <class L>
struct IdxIn {
    const L *p;
    GRL_DEV idx_t operator()(u64 i) const { return (idx_t)p[i]; }
};
struct SymHeadIn64 {
    const u32 *s;
    GRL_DEV u64 operator()(u64 t) const { return (t == 0 || s[t] != s[t - 1]) ? 1ull : 0ull; }
};
struct IdxIn64 {
    const idx_t *p;
    GRL_DEV u64 operator()(u64 i) const { return (u64)p[i]; }
};
typedef prim::Pair<idx_t, idx_t> HeadLen;      // (#run heads, #symbols) scanned together
struct HeadLenIn {
    const u32 *s; const idx_t *len;
    GRL_DEV HeadLen operator()(u64 t) const { return HeadLen((t == 0 || s[t] != s[t - 1]) ? (idx_t)1 : (idx_t)0, len[t]); }
};
struct DiffFn {
    const idx_t *start; idx_t *len;
    GRL_DEV void operator()(u64 r) const { len[r] = start[r + 1] - start[r]; }
};

struct Runs {
    DBuf<u32> sym;
    DBuf<idx_t> len;      // (pass C leaves only `pos` behind: what reads the lengths there goes through RunLen / Engine::need_len)
    DBuf<idx_t> pos;      // optional: first symbol position of every run + total (R + 1 entries), kept from the merge that made the runs
    u64 R = 0;
    u64 n = 0;            // symbols described (set by merge_runs)
};
// length of run i, from the lengths or from the runs' position prefix (a difference array of R entries, written and read once
// per level, was 6 ms at level 0 of the 10 GB build)
struct RunLen {
    const idx_t *len; const idx_t *pos;
    GRL_DEV idx_t operator()(u64 i) const { return len ? len[i] : (idx_t)(pos[i + 1] - pos[i]); }
};
struct RunLenIn64 {
    RunLen l;
    GRL_DEV u64 operator()(u64 i) const { return (u64)l(i); }
};

// merge adjacent equal symbols (bwt_io.h push_back/inc_freq_last idiom): -> maximal runs.
// One fused scan gives every input run its output run index and its symbol offset; the scan hands both straight to
// the run heads (no prefix array of n pairs in between).
struct MergeEmitFn {
    static constexpr bool kWaveEmit = false;
    const u32 *sym; u64 n;
    u32 *osym; idx_t *ostart; u32 *map;
    GRL_DEV void operator()(u64 t, HeadLen ex, HeadLen v) const {
        if (v.a) { osym[ex.a] = sym[t]; ostart[ex.a] = ex.b; }
        if (t == n - 1) ostart[ex.a + v.a] = ex.b + v.b;
        if (map) map[t] = (u32)(ex.a + v.a - 1);
    }
};
static inline Runs merge_runs(const u32 *sym, const idx_t *len, u64 n, u32 *merged_index = nullptr) {
    Runs out;
    if (n == 0) { out.sym.alloc(0); out.len.alloc(0); return out; }
    // the number of output runs is not known before the scan: heads land in scratch sized for the worst case
    DBuf<u32> hsym(n);
    DBuf<idx_t> ostart(n + 1);
    HeadLen tot = prim::exclusive_scan_emit<HeadLen>(n, HeadLenIn{sym, len}, MergeEmitFn{sym, n, hsym.p, ostart.p, merged_index}, "merge_runs.scan");
    u64 R = (u64)tot.a;
    out.n = (u64)tot.b;
    out.len.alloc(R);
    prim::for_each(R, DiffFn{ostart.p, out.len.p}, "merge_runs.len");
    // the heads' scratch (sized for the worst case) becomes the symbol array as it is, unless it is much larger than what it
    // holds (a copy of 4 R bytes -- 6.6 GB at level 0 of the 10 GB build -- for nothing but a tighter allocation otherwise)
    if (R * 2 >= n) out.sym = std::move(hsym);
    else { out.sym.alloc(R); prim::d2d(out.sym.p, hsym.p, R * sizeof(u32)); }
    out.pos = std::move(ostart);      // (pass C of the next level wants exactly this prefix: no scan of the lengths there)
    out.R = R;
    return out;
}

// --------------------------------------------------------- a12: parse2bwt
struct CellSymFn {
    const u32 *t; u32 *sym; idx_t *len;
    GRL_DEV void operator()(u64 i) const { sym[i] = t[i] >> 2; len[i] = 1; }
};

template <class F>
struct StoreFn {          // out[i] = f(i): materialise an expensive scan input once
    F f; idx_t *out;
    GRL_DEV void operator()(u64 i) const { out[i] = f(i); }
};

// ----------------------------------------------------- a13/a14: induction
// Grammar cell of a metasymbol for the chain walks: g0 | has_hocc<<31 in the low word, g1 in the high word, so that a
// chain step is ONE random 8-byte load (symbols are < 2^30).
struct PackGrammarFn {
    const u32 *g0; const u32 *g1; const u8 *has_hocc; u64 *gp;
    GRL_DEV void operator()(u64 u) const { gp[u] = ((u64)g1[u] << 32) | (u64)(g0[u] | (has_hocc[u] ? 0x80000000u : 0u)); }
};
struct ChainCountFn {
    const u32 *nsym; const u64 *gp; u32 sigma3;
    GRL_DEV idx_t operator()(u64 i) const {
        u64 c0 = gp[nsym[i]];
        idx_t c = (c0 & 0x80000000ull) ? 1 : 0;
        u32 nx = (u32)(c0 >> 32);
        while (nx >= sigma3) { c++; nx = (u32)(gp[nx - sigma3] >> 32); }
        return c;
    }
};
struct TakeCountIn {      // 1 if the run's symbol has hidden occurrences (it drops a TAKE cell)
    const u32 *nsym; const u64 *gp;
    GRL_DEV u64 operator()(u64 i) const { return (gp[nsym[i]] & 0x80000000ull) ? 1ull : 0ull; }
};
enum { CELLS_SEPARATE = 0, CELLS_PACKED = 1, CELLS_FUSED = 2 };
template <int MODE>
struct ChainExpandFn {
    const u32 *nsym; RunLen nlen; const u64 *gp; const idx_t *eoff;
    u32 sigma3, take_code;
    u32 *ekey; idx_t *eidx; u32 *esym; idx_t *elen; u64 *epack; u32 *term;
    int kb, lb;                                   // CELLS_FUSED: epack[e] = sym << (kb+lb) | len << kb | bucket
    GRL_DEV void put(u64 e, u32 key, u32 sym, idx_t f) const {
        if (MODE == CELLS_FUSED) { epack[e] = ((u64)sym << (kb + lb)) | ((u64)f << kb) | (u64)key; return; }
        ekey[e] = key;
        if (MODE == CELLS_PACKED) epack[e] = ((u64)sym << 32) | (u64)f;
        else { eidx[e] = (idx_t)e; esym[e] = sym; elen[e] = f; }
    }
    GRL_DEV void operator()(u64 i) const {
        u32 cur = nsym[i];
        idx_t f = nlen(i);
        u64 e = eoff[i];
        u64 c = gp[cur];
        if (c & 0x80000000ull) put(e++, cur, take_code, f);
        u32 nx = (u32)(c >> 32);
        while (nx >= sigma3) {
            u32 b = nx - sigma3;
            put(e++, b, (u32)c & 0x7FFFFFFFu, f);
            c = gp[b];
            nx = (u32)(c >> 32);
        }
        term[i] = nx;                                       // exact_ind_phase.cpp:257 write_sym
    }
};
// The same walk as a generator for prim::expand_count / expand_sort (chain expansion fused with the first pass of the
// bucket split; protocol in prim_hip.hpp): nodes are metasymbols, a node's record is its packed grammar cell, keys are
// the fused cells sym << (kb+lb) | len << kb | bucket, finish() rewrites the run's symbol (exact_ind_phase.cpp:257).
struct ChainGen {
    const u32 *nsym; RunLen nlen; const u64 *gp;
    u32 sigma3, take_code;
    u32 *term;
    int kb, lb;
    GRL_DEV u32 start(u64 i) const { return nsym[i]; }
    GRL_DEV u64 node(u32 u) const { return gp[u]; }
    GRL_DEV bool owns(u64 rec) const { return (rec & 0x80000000ull) != 0; }
    GRL_DEV bool more(u64 rec) const { return (u32)(rec >> 32) >= sigma3; }
    GRL_DEV u32 next(u64 rec) const { return (u32)(rec >> 32) - sigma3; }
    GRL_DEV u64 item_bits(u64 i) const { return (u64)nlen(i) << kb; }
    GRL_DEV u64 key_own(u32 u, u64 ib) const { return ((u64)take_code << (kb + lb)) | ib | (u64)u; }
    GRL_DEV u64 key_step(u64 rec, u32 b, u64 ib) const { return ((u64)((u32)rec & 0x7FFFFFFFu) << (kb + lb)) | ib | (u64)b; }
    GRL_DEV void finish(u64 i, u64 rec) const { term[i] = (u32)(rec >> 32); }
};
struct NarrowCellFn {
    const u64 *in; u32 *out;
    GRL_DEV void operator()(u64 i) const { out[i] = (u32)in[i]; }
};
struct GatherCellFn {
    const idx_t *perm; const u32 *esym; const idx_t *elen; u32 *ssym; idx_t *slen;
    GRL_DEV void operator()(u64 t) const { u64 e = perm[t]; ssym[t] = esym[e]; slen[t] = elen[e]; }
};

// --------------------------------------------------------- a15: assemble
struct NotCodeIn {
    const u32 *sym; u32 code;
    GRL_DEV idx_t operator()(u64 i) const { return sym[i] != code ? (idx_t)1 : (idx_t)0; }
};
// induced cells in bucket-major order, in one of three forms: ONE 64-bit word per cell, sym | len | bucket (the word
// the split sorted, when the three fields fit); bucket array + packed payload (sym<<32 | len); bucket array + two
// separate arrays (64-bit build with a run of >= 2^32 symbols).
struct CellView {
    const u64 *fused; int kb, lb;
    const u32 *skey; const u64 *packed; const u32 *ssym; const idx_t *slen;
    u32 u0;                 // first bucket of the piece being assembled (collection-level mode: buckets are owned by ranges)
    const u32 *fused32;     // the one-word form in 32 bits (bucket + length + symbol bits <= 32: level 0 of DNA collections)
    GRL_DEV u32 key(u64 t) const {
        return (fused32 ? (fused32[t] & ((1u << kb) - 1u)) : fused ? (u32)(fused[t] & ((1ull << kb) - 1ull)) : skey[t]) - u0;
    }
    GRL_DEV u32 sym(u64 t) const {
        return fused32 ? (fused32[t] >> (kb + lb)) : fused ? (u32)(fused[t] >> (kb + lb)) : packed ? (u32)(packed[t] >> 32) : ssym[t];
    }
    GRL_DEV idx_t len(u64 t) const {
        return fused32 ? (idx_t)((fused32[t] >> kb) & ((1u << lb) - 1u))
                       : fused ? (idx_t)((fused[t] >> kb) & ((1ull << lb) - 1ull)) : packed ? (idx_t)(packed[t] & 0xFFFFFFFFull) : slen[t];
    }
    GRL_DEV void load(u64 t, u32 &s, idx_t &l) const {      // symbol and length from ONE load of the cell
        if (fused32) { const u32 w = fused32[t]; s = w >> (kb + lb); l = (idx_t)((w >> kb) & ((1u << lb) - 1u)); }
        else if (fused) { const u64 w = fused[t]; s = (u32)(w >> (kb + lb)); l = (idx_t)((w >> kb) & ((1ull << lb) - 1ull)); }
        else if (packed) { const u64 w = packed[t]; s = (u32)(w >> 32); l = (idx_t)(w & 0xFFFFFFFFull); }
        else { s = ssym[t]; l = slen[t]; }
    }
};
struct CellHeadIn {     // 1 where a cell does not merge with its predecessor (other bucket or other symbol): the reference's n_runs
    CellView c;
    GRL_DEV u64 operator()(u64 t) const { return (t == 0 || c.key(t) != c.key(t - 1) || c.sym(t) != c.sym(t - 1)) ? 1ull : 0ull; }
};
// ---- pass C as a stream merge (prim::stream_merge_*) -------------------------------------------------------------
// Output order = pre-BWT order with every HOCC run replaced by the cells of its buckets.  A "segment" is a non-HOCC
// pre-BWT run or a cell; segment index of cell t: nhb[j] + t (j = pre-BWT run of its bucket), of pre-BWT run j:
// nhb[j] + #cells of buckets in front of it.  TAKE segments (TAKE cells, BWT-marker runs) tile the T axis (the symbols
// of the rewritten BWT_{r+1}) in output order, so the T position of a segment is the sum of the TAKE lengths in front of
// it, and BWT_r is the stream of the literal segments and of the pieces of T the TAKE segments cut out, with equal
// neighbours merged.  prim::stream_merge_* does exactly that in three forward passes over the segments; what it needs:
//   * which segments are pre-BWT runs: a bit-vector over the segment axis with ranks (RankCell), the non-HOCC runs compacted
//     in order (nh_sym / nh_len) -- the cells are the other segments, in their own order;
//   * the MAXIMAL runs of the rewritten BWT_{r+1} (the chain walks rewrite symbols, neighbours may have become equal): esym /
//     epos, and their starts as a bit-vector over T with ranks.
// (Rounds 1-3: a scan gave every cell its T prefix (stored), two bit-vectors over T their first output atom through a rank
// each, a kernel gathered five arrays per cell to place packed atoms, another scan merged the atoms: 346 GB of traffic for
// 42 GB priced, 206 ms of the 10 GB build.)
typedef prim::Pair<idx_t, idx_t> HoccBwt;      // (HOCC-marker symbols, BWT-marker symbols) scanned together over the pre-BWT
struct PreScanIn {
    const u32 *sym; const idx_t *len; u32 hocc_code, bwt_code;
    GRL_DEV HoccBwt operator()(u64 j) const {
        u32 s = sym[j];
        return HoccBwt(s == hocc_code ? len[j] : (idx_t)0, s == bwt_code ? len[j] : (idx_t)0);
    }
};
// first cell whose bucket is >= u (cells are sorted by bucket)
GRL_DEV u64 cell_lower_bound(const CellView &c, u64 E, u32 u) {
    u64 lo = 0, hi = E;
    while (lo < hi) { u64 mid = (lo + hi) >> 1; if (c.key(mid) < u) lo = mid + 1; else hi = mid; }
    return lo;
}
struct alignas(16) RankCell { u64 w; u64 b; };        // a word of a bit-vector and the number of set bits in front of it: ONE 16-byte load per rank
GRL_DEV u64 rank_in(const RankCell *rc, u64 x) {      // set bits in [0, x)
    const RankCell c = rc[x >> 6];
    return c.b + (u64)__builtin_popcountll(c.w & ((1ull << (x & 63)) - 1ull));
}
struct RankCellPopcIn {
    const RankCell *rc;
    GRL_DEV u64 operator()(u64 i) const { return (u64)__builtin_popcountll(rc[i].w); }
};
struct RankCellBaseEmitFn {
    static constexpr bool kWaveEmit = false;
    RankCell *rc;
    GRL_DEV void operator()(u64 i, u64 ex, u64) const { rc[i].b = ex; }
};
struct PrePlaceFn {       // non-HOCC pre-BWT runs, compacted in order: segment index, symbol, length
    const u32 *psym; const idx_t *plen; const idx_t *nhb; const u32 *u_to_p; u64 M; CellView c; u64 E;
    const u32 *p_to_u; const idx_t *first_cell;      // optional O(1) forms of the two searches (nullptr: search)
    u32 hocc_code;
    idx_t *seg_pos; u32 *nh_sym; idx_t *nh_len;
    GRL_DEV void operator()(u64 j) const {
        u32 s = psym[j];
        if (s != hocc_code) {
            u64 ustar = p_to_u ? (u64)p_to_u[j] : lower_bound<u32>(u_to_p, M, (u32)j);      // metasymbols whose pre-BWT run lies in front of j
            u64 cs = first_cell ? (u64)first_cell[ustar] : cell_lower_bound(c, E, (u32)ustar);   // ... and their cells
            const u64 i = (u64)nhb[j];
            seg_pos[i] = (idx_t)(i + cs);
            nh_sym[i] = s;
            nh_len[i] = plen[j];
        }
    }
};
// first_cell[m] = first cell whose bucket is >= m (m in [0, M]) = exclusive prefix of the bucket sizes.  The bucket heads
// record where their bucket starts (+1: 0 = empty) and where the bucket in front of them ends; sizes -> one scan.
// (Filling the gaps between heads by loops was quadratic at the deep levels: 24 cells, 65 M metasymbols.)
struct BucketEdgesFn {
    CellView c; u64 E; idx_t *bstart1; idx_t *bend;
    GRL_DEV void operator()(u64 t) const {
        const u32 k = c.key(t);
        if (t == 0) bstart1[k] = 1;
        else { const u32 pk = c.key(t - 1); if (pk != k) { bstart1[k] = (idx_t)(t + 1); bend[pk] = (idx_t)t; } }
        if (t == E - 1) bend[k] = (idx_t)E;
    }
};
struct BucketSizeIn {
    const idx_t *bstart1; const idx_t *bend;
    GRL_DEV idx_t operator()(u64 m) const { return bstart1[m] ? bend[m] - (bstart1[m] - 1) : (idx_t)0; }
};
// Stable merge of N bucket-sorted blocks of one-word cells (the cell exchange of the collection-level induction: block g =
// the cells of my buckets that rank g made, rank order inside a bucket = order of the slices).  Per block the bucket edges
// (BucketEdgesFn over the block), per bucket the blocks' sizes in rank order -> one scan over the buckets gives every
// (block, bucket) run its place; a cell then moves ONCE.  (A stable radix sort over the bucket bits did the same with two to
// four passes over all cells: 31 ms per rank at N = 2 on the 10 GB collection.)
struct BlockTotalIn {     // cells of bucket k over all blocks
    const idx_t *bstart1; const idx_t *bend; int N; u64 M;
    GRL_DEV idx_t operator()(u64 k) const {
        idx_t c = 0;
        for (int g = 0; g < N; g++) { const u64 j = (u64)g * M + k; if (bstart1[j]) c += bend[j] - (bstart1[j] - 1); }
        return c;
    }
};
struct BlockOffsetsFn {   // off[g][k]: what to add to a cell's index inside block g to get its place (wraps: unsigned arithmetic)
    const idx_t *bstart1; const idx_t *bend; const idx_t *base; int N; u64 M; idx_t *off;
    GRL_DEV void operator()(u64 k) const {
        idx_t run = base[k];
        for (int g = 0; g < N; g++) {
            const u64 j = (u64)g * M + k;
            if (bstart1[j]) { off[j] = run - (bstart1[j] - 1); run += bend[j] - (bstart1[j] - 1); }
        }
    }
};
template <class K>
struct BlockPlaceFn {
    const K *in; K kmask; u32 u0; const idx_t *off; K *out;
    GRL_DEV void operator()(u64 t) const { const K c = in[t]; out[(idx_t)(off[(u32)(c & kmask) - u0] + (idx_t)t)] = c; }
};
// the maximal runs of the rewritten BWT_{r+1}: run k starts one (exact_ind_phase.cpp:257 rewrites the symbols, the lengths stay)
struct TermHeadIn {
    const u32 *term;
    GRL_DEV idx_t operator()(u64 k) const { return (k == 0 || term[k] != term[k - 1]) ? (idx_t)1 : (idx_t)0; }
};
struct TermHeadEmitFn {
    static constexpr bool kWaveEmit = false;
    const u32 *term; const idx_t *Tpos; u32 *esym; idx_t *epos;
    GRL_DEV void operator()(u64 k, idx_t ex, idx_t v) const { if (v) { esym[ex] = term[k]; epos[ex] = Tpos[k]; } }
};
// the segment stream of one level (prim::stream_merge protocol)
struct AsmSeg {
    const RankCell *kinds;                      // over the segment axis: bit g = segment g is a (non-HOCC) pre-BWT run
    const u32 *nh_sym; const idx_t *nh_len;     // those runs in order
    CellView c; u32 take_code;
    const RankCell *tstarts; const u32 *esym_; const idx_t *epos_;
    struct Ref { u64 idx; bool pre; };          // where the record of a segment sits: pre-BWT run idx of the compacted ones, or cell idx
    GRL_DEV Ref locate(u64 g) const {
        const RankCell kc = kinds[g >> 6];
        const u64 ord = kc.b + (u64)__builtin_popcountll(kc.w & ((1ull << (g & 63)) - 1ull));
        const bool pre = (kc.w >> (g & 63)) & 1ull;
        return Ref{pre ? ord : g - ord, pre};
    }
    GRL_DEV void fetch(const Ref &r, u32 &sym, idx_t &len, bool &take) const {
        if (r.pre) { sym = nh_sym[r.idx]; len = nh_len[r.idx]; }
        else c.load(r.idx, sym, len);
        take = sym == take_code;
    }
    GRL_DEV u64 pre_before(u64 g) const { return rank_in(kinds, g); }
    GRL_DEV Ref plain(u64 t) const { return Ref{t, false}; }
    GRL_DEV u64 erank(u64 x) const { return rank_in(tstarts, x); }
    GRL_DEV void eword(u64 w, u64 &bits, u64 &before) const { const RankCell c = tstarts[w]; bits = c.w; before = c.b; }
    GRL_DEV u32 esym(u64 k) const { return esym_[k]; }
    GRL_DEV u64 epos(u64 k) const { return (u64)epos_[k]; }
};

// ------------------------------------------------------- a16: .rl_bwt image
struct RunRecordFn {      // sym | len << (8 sb): the record of run i as one word (sb + fb <= 8)
    const u32 *sym; RunLen len; u32 sb;
    GRL_DEV u64 operator()(u64 i) const { return (u64)sym[i] | ((u64)len(i) << (8 * sb)); }
};
struct PackRunsFn {
    const u32 *sym; RunLen len; u32 sb, fb; u8 *out; u32 hdr;      // hdr: bytes in front of the first record (16, or 0 for a part)
    GRL_DEV void operator()(u64 i) const {
        const u32 rec = sb + fb;
        u8 *p = out + hdr + i * (u64)rec;
        u64 s = sym[i], l = len(i);
        // records of 4 or 8 bytes (DNA: 1+3; tokens: 2+2 ... ) are one aligned store, not `rec` byte stores
        if (rec == 4 && ((uintptr_t)out & 3) == 0) { *reinterpret_cast<u32 *>(p) = (u32)(s | (l << (8 * sb))); return; }
        if (rec == 8 && ((uintptr_t)out & 7) == 0) { *reinterpret_cast<u64 *>(p) = s | (l << (8 * sb)); return; }
        for (u32 b = 0; b < sb; b++) p[b] = (u8)(s >> (8 * b));
        for (u32 b = 0; b < fb; b++) p[sb + b] = (u8)(l >> (8 * b));
    }
};

// -------------------------------------------------------------- stats (a1)
struct ForeignByteIn {     // 1 for a sampled cell whose value is not in the set the statistics found (a borrowed text that changed since)
    const u8 *t; u64 stride; u64 p0, p1, p2, p3;
    GRL_DEV u64 operator()(u64 i) const {
        const u32 c = t[i * stride];
        const u64 w = (c >> 6) == 0 ? p0 : ((c >> 6) == 1 ? p1 : ((c >> 6) == 2 ? p2 : p3));
        return ((w >> (c & 63)) & 1ull) ? 0ull : 1ull;
    }
};
template <class cell_t>
struct EqIn {
    const cell_t *t; cell_t v;
    GRL_DEV u64 operator()(u64 i) const { return t[i] == v ? 1ull : 0ull; }
};
template <class cell_t>
struct CellIn {
    const cell_t *t;
    GRL_DEV u64 operator()(u64 i) const { return (u64)t[i]; }
};

// ------------------------------------------- collection-level multi-GPU pieces
// The per-round dictionary merge (join_thread_phrases, parsing_strategies.h:277-386) is partitioned by content: a hash of
// a phrase names the rank that merges it, so the same phrase from every shard meets on ONE rank and every rank inserts
// about its own share of the phrases (an all-gather of the lists made every rank insert all of them).
template <class cell_t, bool FIRST>
struct PhraseOwnerFn {
    const cell_t *t; CellOps<cell_t, FIRST> ops; const u64 *pos; const u32 *len; u32 size; u32 *owner; u32 *idx;
    const prim::U128 *pkeys; u64 pDs; int pkb;      // partitioned naming: phrases [0, pDs) are records, not text positions
    GRL_DEV void operator()(u64 k) const {
        const u64 o = pos[k], l = len[k];
        PhraseHash ph = PhraseHash::init();
        if (k < pDs) {
            const prim::U128 r = pkeys[k];
            for (u64 j = 0; j < l; j++) ph.add(rec_sym(r, (u32)j, pkb));
        } else {
            // (eight independent loads per step: one lane walks the whole phrase, and a load and its latency per cell made a 5 M-cell
            // phrase -- an N gap -- a 0.36 s affair)
            u64 j = 0;
            for (; j + 8 <= l; j += 8) {
                cell_t c[8];
#pragma unroll
                for (int x = 0; x < 8; x++) c[x] = t[o + j + (u64)x];
#pragma unroll
                for (int x = 0; x < 8; x++) ph.add((u32)ops.sym(c[x]));
            }
            for (; j < l; j++) ph.add((u32)ops.sym(t[o + j]));
        }
        const u64 h = ph.finish(l) * 0x9E3779B97F4A7C15ull;        // remixed: the table below takes its slot and tag from the plain hash
        owner[k] = (u32)(((h >> 32) * (u64)size) >> 32);
        idx[k] = (u32)k;
    }
};
struct KeyBoundFn {       // first element of every key value in a sorted key array; with `pre`, the prefix value there too
    const u32 *k; u64 n; const u32 *pre; u64 *out;
    GRL_DEV void operator()(u64 d) const {
        const u64 i = lower_bound<u32>(k, n, (u32)d);
        out[2 * d] = i;
        out[2 * d + 1] = pre ? (u64)pre[i] : 0ull;
    }
};
struct SendPhraseFn {     // my phrases in owner order: length and frequency
    const u32 *order; const u32 *len; const idx_t *freq; u32 *slen; u32 *sfreq;      // (a shard's parse has < 2^32 phrases: its frequencies fit u32)
    GRL_DEV void operator()(u64 i) const { const u32 k = order[i]; slen[i] = len[k]; sfreq[i] = (u32)freq[k]; }
};
// ... and their cells in the exchange format (u32  sym<<2 | rep<<1 | is_terminator on every level), straight from the text.
// One lane per 16 consecutive output cells: one search for the phrase of the first cell, then a forward walk.
template <class cell_t, bool FIRST>
struct SendCellsFn {
    const cell_t *t; CellOps<cell_t, FIRST> ops; const u32 *order; const u64 *pos; const u32 *soff; u64 D, S; u32 *out;
    const u64 *pw; const idx_t *pb;      // rank bit-vector of the phrase starts over the output positions
    const prim::U128 *pkeys; u64 pDs; int pkb;      // partitioned naming: phrases [0, pDs) are records, not text positions
    struct alignas(16) Quad { u32 v[4]; };
    // (the cell that travels is symbol << 2 | ends-a-string: the "repeated" bit of a parse cell is a function of its symbol and
    // is not part of a record, so no sender includes it -- equal phrases must be equal words on every rank)
    GRL_DEV void operator()(u64 j) const {
        const u64 q0 = j * 16, q1 = q0 + 16 < S ? q0 + 16 : S;
        u64 i = rank1(pw, pb, q0 + 1) - 1;
        u64 nxt = soff[i + 1], cur = order[i], off = q0 - soff[i];
        // a record phrase is read ONCE and shifted down by one symbol per cell (rec_sym's general 128-bit extraction per cell
        // made this kernel instruction-bound: 23 ms per rank at N = 2 on the 10 GB collection)
        u64 lo = 0, hi = 0, left = 0;          // the record's remaining symbols (next one at bit 0), cells left in it
        bool lastT = false;
        const u64 smask = (1ull << pkb) - 1ull;
        auto open_record = [&](u64 skip) {
            const prim::U128 r = pkeys[cur];
            left = (u64)rec_len(r) - skip;
            lastT = (r.hi & kPhrLastT) != 0;
            lo = r.lo; hi = r.hi & (kPhrLastT - 1ull);
            const u32 sh = (u32)skip * (u32)pkb;                       // < 128
            if (sh >= 64) { lo = hi >> (sh - 64); hi = 0; }
            else if (sh) { lo = (lo >> sh) | (hi << (64 - sh)); hi >>= sh; }
        };
        if (cur < pDs) open_record(off);
        u32 v[16];
#pragma unroll
        for (int x = 0; x < 16; x++) {
            const u64 q = q0 + x;
            if (q < q1) {
                while (q >= nxt) { i++; nxt = soff[i + 1]; cur = order[i]; off = 0; if (cur < pDs) open_record(0); }
                if (cur < pDs) {
                    left--;
                    v[x] = ((u32)(lo & smask) << 2) | ((left == 0 && lastT) ? 1u : 0u);
                    lo = (lo >> pkb) | (hi << (64 - pkb)); hi >>= pkb;      // (1 <= pkb <= 30)
                } else {
                    const cell_t c = t[pos[cur] + off];
                    v[x] = ((u32)ops.sym(c) << 2) | (ops.isT(c) ? 1u : 0u);
                }
                off++;
            }
        }
        if (q1 - q0 == 16) {
#pragma unroll
            for (int x = 0; x < 16; x += 4) *reinterpret_cast<Quad *>(out + q0 + x) = Quad{{v[x], v[x + 1], v[x + 2], v[x + 3]}};
        } else {
#pragma unroll
            for (int x = 0; x < 16; x++) if (q0 + x < q1) out[q0 + x] = v[x];
        }
    }
};
struct ListCellsFn {      // cells of the phrases a list names, packed in list order (16 consecutive cells per lane)
    const u64 *pos; const u32 *off; u64 D, S; const u32 *cells; u32 *out;
    const u64 *pw; const idx_t *pb;
    struct alignas(16) Quad { u32 v[4]; };
    GRL_DEV void operator()(u64 j) const {
        const u64 q0 = j * 16, q1 = q0 + 16 < S ? q0 + 16 : S;
        u64 k = rank1(pw, pb, q0 + 1) - 1;
        u64 nxt = off[k + 1], src = pos[k] + (q0 - off[k]);
        u32 v[16];
#pragma unroll
        for (int x = 0; x < 16; x++) {
            const u64 q = q0 + x;
            if (q < q1) {
                while (q >= nxt) { k++; nxt = off[k + 1]; src = pos[k]; }
                v[x] = cells[src++];
            }
        }
        if (q1 - q0 == 16) {
#pragma unroll
            for (int x = 0; x < 16; x += 4) *reinterpret_cast<Quad *>(out + q0 + x) = Quad{{v[x], v[x + 1], v[x + 2], v[x + 3]}};
        } else {
#pragma unroll
            for (int x = 0; x < 16; x++) if (q0 + x < q1) out[q0 + x] = v[x];
        }
    }
};
struct AddLenFn {
    idx_t *len; u64 add;
    GRL_DEV void operator()(u64) const { *len = (idx_t)((u64)*len + add); }
};
struct OffToPosFn {
    const u32 *off; u64 *pos;
    GRL_DEV void operator()(u64 k) const { pos[k] = off[k]; }
};
struct ScatterU32Fn {     // out[order[i]] = v[i]
    const u32 *order; const u32 *v; u32 *out;
    GRL_DEV void operator()(u64 i) const { out[order[i]] = v[i]; }
};
// phrases given as a list (offset, length, weight) over a cell buffer
struct ListInsertFn {
    const u32 *cells; const u64 *off; const u32 *len; const u32 *weight;
    u64 *keys; idx_t *counts; u64 mask;      // key = tag:24 | (list index + 1):40
    u32 *list_slot; u32 *scal;
    u8 *is_rep;                              // [list entries], zeroed: set for the entry that creates its slot (the phrase's representative)
    GRL_DEV void operator()(u64 i) const {
        const u64 o = off[i], l = len[i];
        PhraseHash ph = PhraseHash::init();
        {
            u64 j = 0;                           // (eight independent loads per step: see PhraseOwnerFn)
            for (; j + 8 <= l; j += 8) {
                u32 c[8];
#pragma unroll
                for (int x = 0; x < 8; x++) c[x] = cells[o + j + (u64)x];
#pragma unroll
                for (int x = 0; x < 8; x++) ph.add(c[x] >> 2);
            }
            for (; j < l; j++) ph.add(cells[o + j] >> 2);
        }
        const u64 h = ph.finish(l);
        const u64 tag = h >> kPosBits;
        const u64 mine = (tag << kPosBits) | (i + 1);
        u64 slot = h & mask;
        u32 found = prim::kNoBucket;
        for (u64 probes = 0; probes <= mask && found == prim::kNoBucket; probes++) {
            u64 cur = prim::load_relaxed(&keys[slot]);
            if (cur == 0) {
                u64 old = prim::atomic_cas(&keys[slot], 0ull, mine);
                cur = (old == 0) ? mine : old;
                if (old == 0) is_rep[i] = 1;
            }
            bool hit = (cur == mine);
            if (!hit && (cur >> kPosBits) == tag) {
                const u64 i2 = (cur & kPosMask) - 1;
                if (len[i2] == l) {
                    const u64 o2 = off[i2];
                    u32 diff = 0;
                    u64 j = 0;
                    for (; j + 8 <= l && diff == 0; j += 8) {
#pragma unroll
                        for (int x = 0; x < 8; x++) diff |= cells[o2 + j + (u64)x] ^ cells[o + j + (u64)x];
                    }
                    for (; j < l; j++) diff |= cells[o2 + j] ^ cells[o + j];
                    hit = diff == 0;
                }
            }
            if (hit) found = (u32)slot; else slot = (slot + 1) & mask;
        }
        if (found == prim::kNoBucket) scal[1] = 1;
        else {
            prim::atomic_add(&counts[found], (idx_t)weight[i]);
            list_slot[i] = found;
        }
    }
};
// The representative of a merged phrase is the list entry whose CAS created the table slot (its index sits in the key); an
// owner numbers its phrases in the order of their representatives.  Which duplicate wins is a race, so the numbering can
// differ between runs -- it is the owner's alone (every rank receives the owner's part as it is) and nothing downstream
// depends on the order of a dictionary.  (Until round 3 the representative was the SMALLEST list index per slot, found by
// one atomicMin per list entry: 13 ms per rank at N = 2 on the 10 GB collection for a determinism nobody needs.)
struct SlotWinnerFn {     // slot_min[slot] = list index of the entry that created the slot (all-ones for an empty slot)
    const u64 *keys; u32 *slot_min;
    GRL_DEV void operator()(u64 s) const { const u64 k = keys[s]; slot_min[s] = k ? (u32)((k & kPosMask) - 1) : 0xFFFFFFFFu; }
};

struct ListPhraseFn {     // phrase k = rank of its representative among the representatives (list order)
    const u32 *cells; const u64 *off; const u32 *len; const u32 *list_slot; const u8 *is_rep; const u32 *rep_ex;
    const idx_t *counts;
    u64 *ph_pos; idx_t *ph_freq; u32 *ph_len; u8 *ph_lastT;
    GRL_DEV void operator()(u64 i) const {
        if (is_rep[i]) {
            const u32 s = list_slot[i];
            const u32 k = rep_ex[i];
            ph_pos[k] = off[i]; ph_freq[k] = counts[s]; ph_len[k] = len[i];
            ph_lastT[k] = (u8)(cells[off[i] + len[i] - 1] & 1u);
        }
    }
};
struct ListValFn {        // value of the i-th phrase of the lists I merged: through its representative's number in the merged dictionary
    const u32 *list_slot; const u32 *slot_min; const u32 *rep_ex; const u32 *gval; u64 my_first; u32 *val;
    GRL_DEV void operator()(u64 i) const { val[i] = gval[my_first + rep_ex[slot_min[list_slot[i]]]]; }
};

// ---- distributed dictionary stage functors ---------------------------------------------
// Suffixes are partitioned over the ranks by ranges of their packed first-pass key (equal suffixes have
// equal keys, so a group never spans two ranks and rank order = sorted order).  Positional ranks are
// global slots (base of the rank + local slot); after every pass the ranks of the (re)sorted suffixes
// are exchanged as (position, rank) pairs.
struct ApplyPairsFn {     // value[(pair >> 32) - base] = low 32 bits
    const u64 *pairs; u32 *value; u64 base = 0;
    GRL_DEV void operator()(u64 i) const { u64 p = pairs[i]; value[(p >> 32) - base] = (u32)p; }
};
struct OwnPhraseIn {      // 1 if the whole-phrase suffix of phrase k sorted into my slots
    const u32 *pslot;
    GRL_DEV u32 operator()(u64 k) const { return pslot[k] != 0xFFFFFFFFu ? 1u : 0u; }
};
struct OwnPhrasePairFn {  // (phrase << 32 | metasymbol rank) for those
    const u32 *pslot; const u32 *ex; const u32 *grank; u32 m_off; u64 *pairs;
    GRL_DEV void operator()(u64 k) const {
        const u32 g = pslot[k];
        if (g != 0xFFFFFFFFu) pairs[ex[k]] = ((u64)k << 32) | (u64)(m_off + grank[g]);
    }
};
struct PhraseValDistFn {
    const u32 *phrase_rank; const idx_t *ph_freq; const u8 *ph_lastT; u32 *phrase_val;
    GRL_DEV void operator()(u64 k) const {
        phrase_val[k] = (phrase_rank[k] << 2) | ((ph_freq[k] > 1) ? 2u : 0u) | (ph_lastT[k] ? 1u : 0u);
    }
};

// ---- collection-level induction functors ----------------------------------------------
// The output of a level (BWT_r) is cut at pre-BWT run boundaries into one contiguous piece per rank ("owner").  An owner
// holds the buckets (metasymbols) whose HOCC runs lie in its piece, so it receives exactly those cells from every rank,
// and the stretch of the rewritten BWT_{r+1} its piece consumes (consumption is monotone in output order).  With cells
// and that stretch in hand a piece is an ordinary single-GPU pass C.
struct PreBwtLenIn {      // length of pre-BWT run j if it is a BWT-marker run (what a piece takes straight from the rewritten BWT_{r+1})
    const u32 *sym; const idx_t *len; u32 bwt_code;
    GRL_DEV u64 operator()(u64 j) const { return sym[j] == bwt_code ? (u64)len[j] : 0ull; }
};
struct OwnerSplitFn {     // lane d in [0, size]: first pre-BWT run of owner d, its first bucket, symbols and BWT-marker symbols in front
    const idx_t *Ppos; const HoccBwt *PHB; const u32 *u_to_p; u64 P, M, n_r; int size;
    u64 *out /*[size+1][4]*/;
    GRL_DEV void operator()(u64 d) const {
        const u64 chunk = (n_r + (u64)size - 1) / (u64)size;
        const u64 target = d * chunk < n_r ? d * chunk : n_r;
        const u64 p = (d == (u64)size) ? P : lower_bound<idx_t>(Ppos, P, (idx_t)target);
        out[4 * d] = p;
        out[4 * d + 1] = lower_bound<u32>(u_to_p, M, (u32)p);
        out[4 * d + 2] = (u64)Ppos[p];
        out[4 * d + 3] = (u64)PHB[p].b;
    }
};
struct CellBoundsFn {     // lane d: first of my cells whose bucket belongs to owner d or a later one
    CellView c; u64 E; const u64 *split; u64 *out;
    GRL_DEV void operator()(u64 d) const { out[d] = cell_lower_bound(c, E, (u32)split[4 * d + 1]); }
};
struct CellTakeOffIn {    // TAKE length of cell off + t
    CellView c; u32 take_code; u64 off;
    GRL_DEV u64 operator()(u64 t) const { return c.sym(off + t) == take_code ? (u64)c.len(off + t) : 0ull; }
};
struct WindowRunsFn {     // lane d: my runs of BWT_{r+1} that overlap [a, b) (local symbol coordinates): first run, count
    const idx_t *Tpos; u64 R; const u64 *ab; u64 *out /*[size][2]*/;
    GRL_DEV void operator()(u64 d) const {
        const u64 a = ab[2 * d], b = ab[2 * d + 1];
        u64 k0 = 0, cnt = 0;
        if (a < b) {
            k0 = upper_bound<idx_t>(Tpos, R, (idx_t)a) - 1;               // the run holding symbol a
            cnt = lower_bound<idx_t>(Tpos, R, (idx_t)b) - k0;             // ... up to the run holding symbol b - 1
        }
        out[2 * d] = k0; out[2 * d + 1] = cnt;
    }
};
template <class LT>       // LT: the type the lengths travel in (u32 whenever the longest run of the level fits: 8 instead of 12 bytes per run)
struct WindowSendFn {     // the runs of every owner's window, clipped to the window, in owner order
    const idx_t *Tpos; const u32 *term; const u64 *ab; const u64 *k0cnt; const u64 *soff /*[size+1]*/; int size;
    u32 *ssym; LT *slen;
    GRL_DEV void operator()(u64 y) const {
        int d = 0;
        while (d + 1 < size && y >= soff[d + 1]) d++;
        const u64 k = k0cnt[2 * d] + (y - soff[d]);
        const u64 a = ab[2 * d], b = ab[2 * d + 1];
        const u64 s = (u64)Tpos[k] > a ? (u64)Tpos[k] : a, e = (u64)Tpos[k + 1] < b ? (u64)Tpos[k + 1] : b;
        ssym[y] = term[k];
        slen[y] = (LT)(e - s);
    }
};
struct NarrowIdxFn {
    const idx_t *in; u32 *out;
    GRL_DEV void operator()(u64 i) const { out[i] = (u32)in[i]; }
};
struct WidenLenFn {
    const u32 *in; idx_t *out;
    GRL_DEV void operator()(u64 i) const { out[i] = (idx_t)in[i]; }
};
struct RebaseFn {         // out[i] = in[i] - sub
    const u32 *in; u32 sub; u32 *out;
    GRL_DEV void operator()(u64 i) const { out[i] = in[i] - sub; }
};
struct PieceMetaFn {      // metasymbols in front of every pre-BWT run of my piece, relative to the piece (no p_to_u table at hand)
    const u32 *u_to_p; u64 M; u32 p0, u0; u32 *out;
    GRL_DEV void operator()(u64 j) const { out[j] = (u32)lower_bound<u32>(u_to_p, M, p0 + (u32)j) - u0; }
};
struct IotaIdxFn {
    idx_t *v;
    GRL_DEV void operator()(u64 i) const { v[i] = (idx_t)i; }
};

// ---- f3 ingestion: FASTA/FASTQ -> one string per line (external/bioparsers/lib/fastx_handler.cpp:7-58 over kseq.h:179-220) ----
// The reference reads records one by one through kseq; here the file is a table of lines.  Every line is a HEADER (starts a
// record), a SEQUENCE line (its bytes belong to the record above) or SKIPPED ('+' and quality lines of FASTQ); a record's
// string is the concatenation of its sequence lines, each without the "\r" kseq drops (kseq.h:141), followed by the
// separator, and with FX_REVCOMP by its reverse complement and another separator.  Two layouts are classified in parallel:
// no '+' line outside headers (FASTA, any wrapping, blank lines), and strict four-line FASTQ records (verified on the
// device); anything else (FASTQ over several lines, damaged records) is walked record by record by one lane like kseq does
// (FxSeqClassFn) -- the byte work stays parallel in every case.
enum : u8 { FX_SKIP = 0, FX_HDR = 1, FX_SEQ = 2 };
struct FxNlWordsFn {      // one lane per 64 input bytes: bit b of words[w] = (in[64w + b] == '\n')
    const u8 *in; u64 n; u64 *words;
    GRL_DEV void operator()(u64 w) const {
        const u64 x0 = w * 64;
        u64 m = 0;
        if (x0 + 64 <= n && (((uintptr_t)in) & 15) == 0) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                struct alignas(16) Quad { u32 v[4]; };
                const Quad qd = *reinterpret_cast<const Quad *>(in + x0 + 16 * q);
                const u32 *ws = qd.v;
#pragma unroll
                for (int k = 0; k < 4; k++)
#pragma unroll
                    for (int b = 0; b < 4; b++)
                        if (((ws[k] >> (8 * b)) & 0xFFu) == 10u) m |= 1ull << (16 * q + 4 * k + b);
            }
        } else {
            for (u64 b = 0; b < 64 && x0 + b < n; b++) if (in[x0 + b] == 10) m |= 1ull << b;
        }
        words[w] = m;
    }
};
struct FxNlPosFn {        // end of every line: position of its '\n'
    const u64 *words; const idx_t *base; u64 *nlpos;
    GRL_DEV void operator()(u64 w) const {
        u64 m = words[w], k = (u64)base[w];
        while (m) { const int b = __builtin_ctzll(m); nlpos[k++] = w * 64 + (u64)b; m &= m - 1; }
    }
};
struct FxLine { u64 start; u64 len; u8 first, last; };
GRL_DEV FxLine fx_line(const u8 *in, const u64 *nlpos, u64 i) {
    FxLine L;
    L.start = i ? nlpos[i - 1] + 1 : 0;
    L.len = nlpos[i] - L.start;
    L.first = L.len ? in[L.start] : 0;
    L.last = L.len ? in[L.start + L.len - 1] : 0;
    return L;
}
GRL_DEV bool fx_is_hdr(u8 c) { return c == '>' || c == '@'; }
struct FxPlusIn {         // 1 for a line that starts with '+' (headers start with '>' or '@')
    const u8 *in; const u64 *nlpos;
    GRL_DEV u64 operator()(u64 i) const { return fx_line(in, nlpos, i).first == '+' ? 1ull : 0ull; }
};
struct FxLastNonEmptyIn { // (index + 1) of a non-empty line
    const u64 *nlpos;
    GRL_DEV u64 operator()(u64 i) const { const u64 s = i ? nlpos[i - 1] + 1 : 0; return nlpos[i] > s ? i + 1 : 0ull; }
};
GRL_DEV u64 fx_kept(const FxLine &L) { return L.len - ((L.len > 1 && L.last == '\r') ? 1 : 0); }     // kseq.h:141
struct FxFastqBadIn {     // 1 if record k is not a strict four-line FASTQ record
    const u8 *in; const u64 *nlpos;
    GRL_DEV u64 operator()(u64 k) const {
        const FxLine h = fx_line(in, nlpos, 4 * k), s = fx_line(in, nlpos, 4 * k + 1), p = fx_line(in, nlpos, 4 * k + 2),
                     q = fx_line(in, nlpos, 4 * k + 3);
        bool ok = fx_is_hdr(h.first) && h.len >= 1 && p.first == '+';
        ok = ok && !(s.len && (fx_is_hdr(s.first) || s.first == '+'));
        ok = ok && fx_kept(s) == fx_kept(q);
        return ok ? 0ull : 1ull;
    }
};
struct FxClassFn {        // class and provisional kept length of every line
    const u8 *in; const u64 *nlpos; u64 fastq_lines;     // > 0: strict FASTQ over the first fastq_lines lines
    bool fastq;
    u8 *cls; u64 *raw;
    GRL_DEV void operator()(u64 i) const {
        const FxLine L = fx_line(in, nlpos, i);
        u8 c;
        if (fastq) c = (i < fastq_lines) ? ((i & 3) == 0 ? FX_HDR : (i & 3) == 1 ? FX_SEQ : FX_SKIP) : FX_SKIP;
        else c = fx_is_hdr(L.first) ? FX_HDR : FX_SEQ;
        cls[i] = c;
        raw[i] = c == FX_SEQ ? fx_kept(L) : 0;
    }
};
// kseq's record walk (kseq.h:179-220) line by line, ONE lane: for the layouts the parallel rules above do not cover
// (FASTQ records over several lines, '+' lines in odd places, a quality string that stops early).  Sequential like the
// reference's parser -- such files are rare and small; the four-line and FASTA layouts never come here.  Outputs the same
// (class, kept length) per line as the parallel path; res[0] = first line that no longer counts (kseq gives up at a
// record whose quality string has another length than its sequence: that record and everything behind it is not written).
struct FxSeqClassFn {
    const u8 *in; u64 n; const u64 *nlpos; u64 m; bool tail;
    u8 *cls; u64 *kept; u64 *res;
    GRL_DEV void operator()(u64) const {
        int st = 0;                                   // 0 looking for a header, 1 sequence lines, 2 quality lines
        u64 seq_l = 0, qual_l = 0, cur_hdr = 0, stop = m;
        u8 qlast = 0, qprev = 0;
        bool halted = false;
        for (u64 i = 0; i < m && !halted; i++) {
            const FxLine L = fx_line(in, nlpos, i);
            const bool last_line = i == m - 1;
            u8 c = FX_SKIP;
            u64 k = 0;
            if (st == 0) {                            // kseq.h:183-186: characters are skipped up to the first '>' or '@'
                u64 pos = L.len;
                for (u64 x = 0; x < L.len && pos == L.len; x++) if (fx_is_hdr(in[L.start + x])) pos = x;
                if (pos < L.len) {
                    if (last_line && tail && pos == L.len - 1) { halted = true; stop = i; }      // the header character ends the input: no record
                    else { c = FX_HDR; st = 1; seq_l = 0; cur_hdr = i; }
                }
            } else if (st == 1) {                     // kseq.h:194-199
                if (L.len == 0) {
                } else if (fx_is_hdr(L.first)) {
                    if (last_line && tail && L.len == 1) { halted = true; stop = i; }
                    else { c = FX_HDR; seq_l = 0; cur_hdr = i; }
                } else if (L.first == '+') {
                    if (last_line && tail) { halted = true; stop = cur_hdr; }                     // :211 no quality string: -2
                    else { st = 2; qual_l = 0; qlast = 0; qprev = 0; }
                } else {
                    c = FX_SEQ;
                    seq_l += L.len; k = L.len;
                    if (seq_l > 1 && L.last == '\r') { seq_l--; k--; }                            // kseq.h:141
                }
            } else {                                  // kseq.h:214: quality lines until the string is as long as the sequence
                if (L.len >= 2) { qprev = in[L.start + L.len - 2]; qlast = L.last; }
                else if (L.len == 1) { qprev = qlast; qlast = L.last; }
                qual_l += L.len;
                if (qual_l > 1 && qlast == '\r') { qual_l--; qlast = qprev; qprev = 0; }
                if (qual_l >= seq_l || last_line) {
                    st = 0;
                    if (qual_l != seq_l) { halted = true; stop = cur_hdr; }                        // :217 -2: the conversion ends here
                }
            }
            cls[i] = c; kept[i] = k;
        }
        if (!halted && st == 2 && seq_l != 0) stop = cur_hdr;                                      // '+' line, then the input ends
        res[0] = stop;
    }
};
struct FxStopFn {         // lines from `stop` on do not count
    u64 stop; u8 *cls; u64 *kept;
    GRL_DEV void operator()(u64 i) const { if (i >= stop) { cls[i] = FX_SKIP; kept[i] = 0; } }
};
struct FxIsHdrIn {
    const u8 *cls;
    GRL_DEV u64 operator()(u64 i) const { return cls[i] == FX_HDR ? 1ull : 0ull; }
};
struct FxHdrLinesFn {     // hline[r] = line of the r-th header
    const u8 *cls; const u64 *hrank; u64 m; u64 R; u64 *hline;
    GRL_DEV void operator()(u64 i) const {
        if (cls[i] == FX_HDR) hline[hrank[i]] = i;
        if (i == 0) hline[R] = m;
    }
};
// A line that is just "\r": kseq keeps the '\r' only when it is the first byte of the record's sequence (the string must be
// longer than one byte for the strip, kseq.h:141)
struct FxKeptFn {
    const u8 *in; const u64 *nlpos; const u8 *cls; const u64 *hrank; const u64 *hline; const u64 *praw; u64 *kept;
    GRL_DEV void operator()(u64 i) const {
        u64 k = 0;
        if (cls[i] == FX_SEQ) {
            const FxLine L = fx_line(in, nlpos, i);
            k = fx_kept(L);
            if (L.len == 1 && L.last == '\r') {
                const u64 r = hrank[i] - 1;                       // (a sequence line always has a header in front)
                k = (praw[i] - praw[hline[r]] == 0) ? 1 : 0;
            }
        }
        kept[i] = k;
    }
};
struct FxLineRec { u64 start, out_base, rc_base, kept_cls; };      // kept << 2 | class
struct FxLineRecFn {
    const u64 *nlpos; const u8 *cls; const u64 *hrank; const u64 *hline; const u64 *kept; const u64 *pk; bool rc;
    FxLineRec *rec;
    GRL_DEV void operator()(u64 i) const {
        FxLineRec L;
        L.start = i ? nlpos[i - 1] + 1 : 0;
        L.kept_cls = (kept[i] << 2) | cls[i];
        L.out_base = 0; L.rc_base = 0;
        if (cls[i] == FX_SEQ) {
            const u64 r = hrank[i] - 1, b = pk[hline[r]], lr = pk[hline[r + 1]] - b, j = pk[i] - b;
            if (!rc) L.out_base = pk[i] + r;
            else { const u64 B = 2 * b + 2 * r; L.out_base = B + j; L.rc_base = B + 2 * lr - j; }
        }
        rec[i] = L;
    }
};
GRL_DEV u8 fx_comp(u8 c) {   // dna_string::comp (external/bioparsers/lib/dna_string.cpp:6-14); 0 = no complement
    return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c == 10 ? 10 : c == 149 ? 168 : c == 151 ? 155 :
           c == 155 ? 151 : c == 168 ? 149 : 0;
}
struct FxCopyFn {         // one lane per input byte
    const u8 *in; u64 n; const u64 *words; const idx_t *base; const FxLineRec *rec; const u64 *hrank; bool rc;
    u8 *out; u64 *bad_rec;
    GRL_DEV void operator()(u64 x) const {
        const u8 c = in[x];
        if (c != 10) {
            const u64 i = rank1(words, base, x);
            const FxLineRec L = rec[i];
            const u64 t = x - L.start;
            if ((L.kept_cls & 3) == FX_SEQ && t < (L.kept_cls >> 2)) {
                out[L.out_base + t] = c;
                if (rc) {
                    const u8 cc = fx_comp(c);
                    if (cc) out[L.rc_base - t] = cc;
                    else prim::atomic_min(bad_rec, hrank[i] - 1);
                }
            }
        }
    }
};
struct FxBadPosFn {       // the last byte without complement of record `r` (the first the reference meets: it walks a record backwards)
    const u8 *in; const u64 *words; const idx_t *base; const FxLineRec *rec; const u64 *hrank; u64 r; u64 x0; u64 *bad_pos;
    GRL_DEV void operator()(u64 d) const {
        const u64 x = x0 + d;
        const u8 c = in[x];
        if (c != 10) {
            const u64 i = rank1(words, base, x);
            const FxLineRec L = rec[i];
            if ((L.kept_cls & 3) == FX_SEQ && x - L.start < (L.kept_cls >> 2) && hrank[i] - 1 == r && fx_comp(c) == 0)
                prim::atomic_max(bad_pos, x + 1);
        }
    }
};
struct FxSepFn {          // the separator(s) of every record
    const u64 *hline; const u64 *pk; bool rc; u8 sep; u8 *out;
    GRL_DEV void operator()(u64 r) const {
        const u64 b = pk[hline[r]], lr = pk[hline[r + 1]] - b;
        if (!rc) out[b + lr + r] = sep;
        else { const u64 B = 2 * b + 2 * r; out[B + lr] = sep; out[B + 2 * lr + 1] = sep; }
    }
};

// ---- .rl_bwt consumers (scripts/grl2plain.cpp, scripts/reverse_bwt.cpp + fm_index.h:79-83) ------
struct PtrU32In {
    const u32 *p;
    GRL_DEV u32 operator()(u64 i) const { return p[i]; }
};
struct SepLenIn {         // total length of the runs of one symbol
    const u32 *sym; const idx_t *len; u32 code;
    GRL_DEV u64 operator()(u64 i) const { return sym[i] == code ? (u64)len[i] : 0ull; }
};
struct UnpackRunsFn {     // (sym: sb bytes LE, len: fb bytes LE) records -> arrays
    const u8 *img; u32 sb, fb; u32 *sym; idx_t *len;
    GRL_DEV void operator()(u64 i) const {
        const u8 *p = img + 16 + i * (u64)(sb + fb);
        u64 s = 0, l = 0;
        for (u32 b = 0; b < sb; b++) s |= (u64)p[b] << (8 * b);
        for (u32 b = 0; b < fb; b++) l |= (u64)p[sb + b] << (8 * b);
        sym[i] = (u32)s; len[i] = (idx_t)l;
    }
};
struct ImageLenIn {        // run length straight from the packed records (64-bit sum: decides the index width of a consumer)
    const u8 *img; u32 sb, fb;
    GRL_DEV u64 operator()(u64 i) const {
        const u8 *p = img + 16 + i * (u64)(sb + fb) + sb;
        u64 l = 0;
        for (u32 b = 0; b < fb; b++) l |= (u64)p[b] << (8 * b);
        return l;
    }
};
struct PlainRunsFn {      // scripts/grl2plain.cpp:30-45: one output byte per BWT position, (char)sym, optional null replacement
    const u32 *rsym; const u64 *rw; const idx_t *rb; int null_char; u8 *out;
    GRL_DEV void operator()(u64 i) const {
        u32 sy = rsym[rank1(rw, rb, i + 1) - 1];
        if (sy == 0 && null_char >= 0) sy = (u32)null_char;
        out[i] = (u8)sy;
    }
};
struct RleExportFn {      // scripts/grlbwt2rle.cpp:22-30: .syms as uint8, .len as uint32 (the reference's casts)
    const u32 *rsym; const idx_t *rlen; u8 *syms; u32 *lens;
    GRL_DEV void operator()(u64 i) const { syms[i] = (u8)rsym[i]; lens[i] = (u32)rlen[i]; }
};
// split_runs (scripts/split_runs.cpp:44-109).  A run is first cut into pieces of at most L = 2^bits - 1 symbols
// (`while(len>max_length)`), every piece then at the multiples of the block size B that fall strictly inside it; a piece
// that STARTS on a block boundary (other than position 0) is preceded by a zero-length record of its own symbol (the
// reference arrives there with acc_block == block_size and emits `push_back(sym, 0)`, :87-90).  Each boundary in (0, n)
// therefore costs exactly one extra record, so the first record of piece j of run i sits at
//   (#pieces before it) + (#multiples of B in (0, start of the piece)).
struct PieceCountFn {     // number of L-pieces of every run
    const idx_t *len; u64 L;
    GRL_DEV idx_t operator()(u64 i) const { u64 l = len[i]; return (idx_t)(l == 0 ? 1 : (l + L - 1) / L); }
};
struct SplitRunsFn {      // one lane per L-piece
    const u32 *rsym; const idx_t *rlen; const idx_t *rpos; const idx_t *jbase; const u64 *jw; const idx_t *jb;
    u64 L, B;
    u32 *osym; idx_t *olen;
    GRL_DEV void operator()(u64 x) const {
        const u64 i = rank1(jw, jb, x + 1) - 1;          // run owning piece x
        const u64 j = x - (u64)jbase[i];
        const u64 l = rlen[i];
        const u64 k = l == 0 ? 0 : (l + L - 1) / L - 1;  // full pieces peeled off in front of the remainder
        u64 cur = (u64)rpos[i] + j * L;
        const u64 end = cur + (j < k ? L : l - k * L);
        const u32 sy = rsym[i];
        u64 o = x + ((B && cur > 0) ? (cur - 1) / B : 0);
        if (B) {
            if (cur > 0 && cur % B == 0) { osym[o] = sy; olen[o] = 0; o++; }
            for (u64 nb = (cur / B + 1) * B; nb < end; nb += B) { osym[o] = sy; olen[o] = (idx_t)(nb - cur); o++; cur = nb; }
        }
        osym[o] = sy; olen[o] = (idx_t)(end - cur);
    }
};
struct RunSymFn {         // for_each_agg protocol: every run is a work item, its bucket is its symbol
    static constexpr int kBatch = 1;
    static constexpr bool kClaims = false;
    GRL_DEV u64 claim_pos(u64 i) const { return i; }
    const u32 *rsym;
    GRL_DEV bool is_start(u64) const { return true; }
    GRL_DEV u32 process(u64 i) const { return rsym[i]; }
    GRL_DEV u32 operator()(u64 i) const { return rsym[i]; }
};
struct BinAddFn {
    u64 *bins;
    GRL_DEV void operator()(u32 b, u32 c) const { prim::atomic_add(&bins[b & 255u], (u64)c); }
};
struct LenFitIn {         // 1 if the run length needs exactly `cls` bytes class (1: <=255, 2: <=65535, 3: more)
    const idx_t *len; int cls;
    GRL_DEV u64 operator()(u64 i) const {
        u64 l = len[i];
        int c = l <= 255 ? 1 : (l <= 65535 ? 2 : 3);
        return c == cls ? 1ull : 0ull;
    }
};
struct LenKeyFn {
    const idx_t *len; u32 *key; u32 *val;
    GRL_DEV void operator()(u64 i) const { key[i] = (u32)len[i]; val[i] = (u32)i; }
};
struct ExpandRunsFn {     // grl2plain: symbol of every BWT position through the run-start bitmap
    const u32 *rsym; const u64 *rw; const idx_t *rb; u32 *out; idx_t *idx;
    GRL_DEV void operator()(u64 i) const { out[i] = rsym[rank1(rw, rb, i + 1) - 1]; idx[i] = (idx_t)i; }
};
struct LfScatterFn {      // LF[order[j]] = j  (order = stable sort of positions by symbol)
    const idx_t *order; idx_t *lf;
    GRL_DEV void operator()(u64 j) const { lf[order[j]] = (idx_t)j; }
};
struct InvertLenFn {      // string i: backward LF walk from row i until its own terminator comes back
    const u32 *bwt; const idx_t *lf; u32 sep; idx_t *slen;
    GRL_DEV void operator()(u64 i) const {
        u64 row = i, l = 1;
        for (u32 c = bwt[row]; c != sep; c = bwt[row]) { l++; row = lf[row]; }
        slen[i] = (idx_t)l;
    }
};
template <class cell_t>
struct InvertWriteFn {
    const u32 *bwt; const idx_t *lf; const idx_t *off; u32 sep; cell_t *text;
    GRL_DEV void operator()(u64 i) const {
        u64 end = off[i + 1] - 1, row = i;
        text[end] = (cell_t)sep;
        for (u32 c = bwt[row]; c != sep; c = bwt[row]) { text[--end] = (cell_t)c; row = lf[row]; }
    }
};

// ---- the same inversion indexed by RUNS (scripts/fm_index.h:79-83 computes LF from the symbol's rank; over the run-length
// BWT the rank of a run's first symbol is a prefix sum over the RUNS of that symbol).  A stable sort of the runs by symbol and
// a scan of their lengths in that order give LF of every run start; LF inside a run is linear, so one record per run --
// (LF(start) - start, symbol) -- and the run-start bit-vector replace the per-position arrays: 28-36 bytes per RUN instead
// of 20-36 per symbol, which is what lets the 10 GB headline image (1.66 G runs, 10^10 symbols) round-trip on one device.
template <int IB> struct RunRecT;
template <> struct alignas(8) RunRecT<4> { u32 delta; u32 sym; };
template <> struct alignas(16) RunRecT<8> { u64 delta; u32 sym; u32 pad; };
typedef RunRecT<sizeof(idx_t)> RunRec;
struct RankCellFn {
    const u64 *words; const idx_t *base; RankCell *rc;
    GRL_DEV void operator()(u64 i) const { rc[i] = RankCell{words[i], (u64)base[i]}; }
};
struct IotaFn {
    idx_t *v;
    GRL_DEV void operator()(u64 i) const { v[i] = (idx_t)i; }
};
struct RunOrderLenIn {    // length of the j-th run in (symbol, position) order
    const idx_t *rlen; const idx_t *order;
    GRL_DEV idx_t operator()(u64 j) const { return rlen[order[j]]; }
};
struct RunLfEmitFn {      // emit side of that scan: ex = LF of the run's first position
    static constexpr bool kWaveEmit = false;
    const idx_t *order; const idx_t *rpos; const u32 *rsym; RunRec *rec;
    GRL_DEV void operator()(u64 j, idx_t ex, idx_t) const {
        const u64 k = order[j];
        RunRec r;
        r.delta = (decltype(r.delta))(ex - rpos[k]);      // (wraps below zero: row + delta is taken modulo 2^bits)
        r.sym = rsym[k];
        rec[k] = r;
    }
};
GRL_DEV u64 run_of_row(const RankCell *rc, u64 row) {      // index of the run holding BWT position row
    const RankCell c = rc[(row + 1) >> 6];
    return c.b + (u64)__builtin_popcountll(c.w & ((1ull << ((row + 1) & 63)) - 1ull)) - 1;
}
struct RunInvertLenFn {
    const RankCell *rc; const RunRec *rec; u32 sep; idx_t *slen;
    GRL_DEV void operator()(u64 i) const {
        idx_t row = (idx_t)i;
        u64 l = 1;
        RunRec r = rec[run_of_row(rc, (u64)row)];
        while (r.sym != sep) { l++; row = (idx_t)(row + (idx_t)r.delta); r = rec[run_of_row(rc, (u64)row)]; }
        slen[i] = (idx_t)l;
    }
};
template <class cell_t>
struct RunInvertWriteFn {
    const RankCell *rc; const RunRec *rec; const idx_t *off; u32 sep; cell_t *text;
    GRL_DEV void operator()(u64 i) const {
        u64 end = off[i + 1] - 1;
        idx_t row = (idx_t)i;
        text[end] = (cell_t)sep;
        RunRec r = rec[run_of_row(rc, (u64)row)];
        while (r.sym != sep) { text[--end] = (cell_t)r.sym; row = (idx_t)(row + (idx_t)r.delta); r = rec[run_of_row(rc, (u64)row)]; }
    }
};

template <class cell_t>
struct RunInvertTailFn {    // the last `tail` cells of string i (separator included), right-aligned in slot i of `tail` cells; the rest of the slot stays as it was
    const RankCell *rc; const RunRec *rec; u32 sep; u64 tail; cell_t *out; idx_t *got;
    GRL_DEV void operator()(u64 i) const {
        u64 end = (i + 1) * tail - 1, l = 1;
        idx_t row = (idx_t)i;
        out[end] = (cell_t)sep;
        RunRec r = rec[run_of_row(rc, (u64)row)];
        while (r.sym != sep && l < tail) { out[--end] = (cell_t)r.sym; l++; row = (idx_t)(row + (idx_t)r.delta); r = rec[run_of_row(rc, (u64)row)]; }
        got[i] = (idx_t)l;
    }
};

// =========================================================================
struct RoundInfo {
    u64 n_in = 0, D = 0, S = 0, M = 0, parse_size = 0, sigma = 0, max_phrase_len = 0, sort_iters = 0, table_retries = 0;
};
struct LevelInfo {
    u64 R_next = 0, E = 0, P = 0, G = 0, A = 0, R = 0, n = 0;
    u64 Esteps = 0, Emerged = 0;      // measured only while profiling (SURVEY 8d's E'_r and E_r): chain steps, cells after the in-bucket merge
};
struct Stats {
    u64 n_strings = 0, n_syms = 0, min_sym = 0, max_sym = 0, max_sym_freq = 0, sb = 0, fb = 0;
};
struct Timers {
    double classify = 0, hash = 0, dict_sort = 0, dict_groups = 0, emit = 0, ind_expand = 0, ind_sort = 0, ind_assemble = 0,
           stats = 0, finish = 0;
};

struct LevelData {
    u32 sigma = 0, M = 0;
    DBuf<u32> g0, g1;
    DBuf<u8> has_hocc;
    DBuf<u32> u_to_p;        // metasymbol -> index of the (merged) pre-BWT run emitted by its group
    DBuf<u32> p_to_u;        // (merged) pre-BWT run -> number of metasymbols whose run lies in front of it (single-GPU dictionary stage)
    Runs prebwt;
    // collection-level mode: the pre-BWT stays with the rank whose key range produced it (round 5) -- prebwt, u_to_p and p_to_u
    // then describe MY part only (run indices and metasymbol counts relative to it), metasymbols [u0, u0 + Ml) are mine
    bool pre_local = false;
    u32 u0 = 0, Ml = 0;
    RoundInfo info;
};

static inline double now_s() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
struct StageTimer {
    double *acc;
    u64 peak0;
    const char *name;
    explicit StageTimer(double *a, const char *nm = "?") : acc(a), name(nm) { prim::stage_begin(); peak0 = prim::pool_stage_begin(); }
    ~StageTimer() {
        prim::stage_end(acc);
        prim::pool_stage_end(peak0, name);
    }
};

class Engine {
  public:
    ~Engine() { prim::stages_drop(); }   // open stage clocks point into tm
    int cell_bytes = 1;
    const void *text0 = nullptr;      // device pointer (owned by own0 or borrowed)
    DBuf<u8> own0;
    u64 n0 = 0;
    Stats stats;
    Timers tm;
    std::vector<LevelData> levels;    // one per parsing round
    std::vector<LevelInfo> linfo;
    DBuf<u32> cur_text;               // text of the current level (level >= 1)
    u64 cur_n = 0;
    u32 cur_sigma = 0;
    bool parse_done = false;
    Runs bwt;                         // BWT of level `bwt_level`
    int bwt_level = -1;
    DBuf<u8> image;                   // .rl_bwt bytes (device)
    u64 sym_present[4] = {0, 0, 0, 0};  // byte texts: the cell values that occur in MY text (stats_t)
    bool sym_present_known = false;
    u64 image_bytes = 0;              // of the WHOLE image
    u64 image_runs = 0;
    // what `image` holds: all of it, or (collection-level mode with Comm::keep_parts) the bytes [image_part_off, + image_part_bytes)
    u64 image_part_off = 0, image_part_bytes = 0;
    bool keep_texts = false;          // debug/parity: keep every level's text
    std::vector<DBuf<u32>> kept_texts;
    std::vector<Runs> kept_bwts;      // debug/parity: BWT of every level (index = level)

    // ---- a1 ------------------------------------------------------------
    template <class cell_t>
    void stats_t(const cell_t *t, u64 n, const u64 *hist256 = nullptr) {
        StageTimer st(&tm.stats, "stats");
        cell_t sep;
        prim::d2h(&sep, t + (n - 1), sizeof(cell_t));
        u64 mn, mx, F = n;
        if (sizeof(cell_t) == 1) {
            u64 h[256];
            if (hist256) std::memcpy(h, hist256, sizeof h);      // counted while the text was uploaded (file loader)
            else prim::byte_histogram((const u8 *)t, n, h);
            mn = 0; while (h[mn] == 0) mn++;
            mx = 255; while (h[mx] == 0) mx--;
            F = 0; for (int i = 0; i < 256; i++) if (h[i] > F) F = h[i];      // utils.cpp:161-175
            stats.n_strings = h[(u64)sep];
            for (int i = 0; i < 4; i++) sym_present[i] = 0;
            for (int i = 0; i < 256; i++) if (h[i]) sym_present[i >> 6] |= 1ull << (i & 63);
            sym_present_known = true;
        } else {
            mn = prim::reduce_min<u64>(n, CellIn<cell_t>{t}, "stats.min");
            mx = prim::reduce_max<u64>(n, CellIn<cell_t>{t}, "stats.max");
            stats.n_strings = prim::reduce_sum<u64>(n, EqIn<cell_t>{t, sep}, "stats.nstr");
        }
        if ((u64)sep != mn) throw prim::Error(-84, "Error: the file is ill formed");   // utils.cpp:177-180
        stats.n_syms = n; stats.min_sym = mn; stats.max_sym = mx; stats.max_sym_freq = F;
        if (mx + 8 >= (1ull << 30)) throw prim::Error(-75, "symbols >= 2^30 are not supported by this build");
        stats.sb = (bitlen64(mx + 4) + 7) / 8;                                          // a17
        stats.fb = (bitlen64(F) + 7) / 8;
    }

    void load_text(const void *dev_cells, u64 n, int w, const u64 *hist256 = nullptr) {
        if (n == 0 || !(w == 1 || w == 2 || w == 4 || w == 8)) throw prim::Error(-22, "bad input size or cell width");
        if (sizeof(idx_t) == 4 && n >= 0xFFFFFF00ull) throw prim::Error(-75, "input too large for the 32-bit index build");
        if (n >= kPosMask) throw prim::Error(-75, "input too large");
        cell_bytes = w; text0 = dev_cells; n0 = n;
        prim::rt().tag = -1; prim::rt().phase = 0;
        // scratch for the whole build in one slab: ~22x the input for the 32-bit index build, ~30x for the 64-bit one
        prim::pool_reserve((size_t)(n * (u64)w) * (sizeof(idx_t) == 4 ? 22 : 30));
        levels.clear(); linfo.clear(); kept_texts.clear(); kept_bwts.clear();
        parse_done = false; bwt_level = -1; image_bytes = 0;
        tm = Timers();
        switch (w) {
            case 1: stats_t<u8>((const u8 *)dev_cells, n, hist256); break;
            case 2: stats_t<u16>((const u16 *)dev_cells, n); break;
            case 4: stats_t<u32>((const u32 *)dev_cells, n); break;
            default: stats_t<u64>((const u64 *)dev_cells, n); break;
        }
        cur_n = n;
        cur_sigma = (u32)(stats.max_sym + 1);
    }
    void upload_text(const void *host_cells, u64 n, int w) {
        own0.alloc(n * (u64)w + 16);
        prim::h2d(own0.p, host_cells, n * (u64)w);
        load_text(own0.p, n, w);
    }

    // ---- one parsing round (par_round, exact_par_phase.cpp:374-497) ------
    // The round is cut in three pieces so that the collection-level multi-GPU driver can put its
    // exchanges between them:  hash_local (a2-a5: breaks, phrase table, distinct phrases of THIS
    // text) -> dict_stage (a5-a8 on a set of distinct phrases, local or merged over all ranks)
    // -> emit_local (a9/a10: metasymbols back into the local parse).
    struct LocalParse {
        u64 n_occ = 0, D = 0, S = 0, cap = 0;
        u32 maxlen = 0;
        DBuf<u32> next_text;       // slot id of every phrase occurrence, then the parse itself
        DBuf<u64> ph_pos; DBuf<idx_t> ph_freq; DBuf<u32> ph_len, ph_slot, ph_off; DBuf<u8> ph_lastT;
        // partitioned naming (levels above 0): phrases [0, Ds) are given by their records; what the emission needs to send the
        // phrases' values back to text order
        u64 Ds = 0; int part_bits = 0, rec_b = 0; u32 slot0 = 0;
        DBuf<prim::U128> ph_key;
        DBuf<u32> lid, pbase;
        DBuf<u8> ph_vflag;         // record phrases: the two flag bits of their values (PartPhraseFn)
        DBuf<u64> pstart;          // first record of every partition in the sorted order (the emission walks the partitions)
        prim::RecSort psort;
        void clear() {
            n_occ = D = S = cap = Ds = 0; maxlen = 0; part_bits = rec_b = 0; slot0 = 0;
            next_text.release(); ph_pos.release(); ph_freq.release(); ph_len.release(); ph_slot.release(); ph_off.release(); ph_lastT.release();
            ph_key.release(); ph_vflag.release(); pstart.release(); lid.release(); pbase.release(); psort.release();
        }
    };

    // phrase hashing launch: one lane per text position, counts pre-aggregated in LDS when few phrases dominate
    // (a tiled variant -- 64 positions per lane, text staged in LDS, per-tile de-duplication of <= 7-byte phrases --
    // was built and measured at 1.9-3.2 ms vs 2.2 ms for this form on 101 MB of reads, and dropped)
    template <class cell_t, bool FIRST>
    void launch_hash(HashInsertFn<cell_t, FIRST> f, idx_t *counts, u64 cs, u64 n, bool aggregate) {
        prim::for_each_agg(n, f, SlotCountAdd{counts, cs}, aggregate, "hash_phrases");
    }

    template <class cell_t, bool FIRST>
    void hash_local(const cell_t *t, u64 n, CellOps<cell_t, FIRST> ops, LocalParse &P, LevelData &L, bool allow_part = false) {
        const u64 nwords = (n + 63) / 64;
        DBuf<u64> startbits(nwords + 1);
        DBuf<idx_t> wordbase(nwords + 1);
        u64 n_occ;
        {
            StageTimer st(&tm.classify, "classify");
            prim::start_bitvector(n, t, ops, StartPred<cell_t, FIRST>{t, ops}, startbits.p, "lms_breaks");
            n_occ = (u64)prim::exclusive_scan<idx_t>(nwords, PopcIn{startbits.p}, wordbase.p, true, "phrase_ordinals");
        }
        P.n_occ = n_occ;
        L.info.parse_size = n_occ;
        // (phrase ordinals are idx_t: the 64-bit build takes parses of 2^32 phrases and more -- level 0 of a 24.9 GB collection has
        // 7.3 G; what stays 32 bits wide are slot ids, dictionary positions and a shard's frequencies on the wire: checked where they are made)
        if (sizeof(idx_t) == 4 && n_occ >= 0xFFFFFFF0ull) throw prim::Error(-75, "parse too large for u32 phrase ordinals");

        // ---- partitioned naming (levels above 0, single-GPU rounds): a phrase of <= cmax cells is a 128-bit record; the records
        // are grouped by a hash prefix into partitions that fit an LDS table (prim::RecSort), de-duplicated and counted there
        // (prim::rec_dedupe), and the phrases' values travel back through the sort's passes in reverse (emit_local).  Only the
        // longer phrases (1 % at level 1, ~12 % at levels 2-3 of the 10 GB build) still go through the hash table below.
        // Round 2's table took 3 random HBM accesses per occurrence (probe, verification, count atomic: 51 / 51 / 18 G/s) and the
        // emission a fourth; this path streams.  GRLBWT_NO_PART=1 switches it off; GRLBWT_PART_MIN_OCC sets the smallest level.
        int rec_b = 0;
        u32 rec_cmax = 0;
        bool part = false;
        if (allow_part && !FIRST && sizeof(cell_t) == 4 && !prim::dev_env("GRLBWT_NO_PART")) {
            rec_b = (int)bitlen64(L.sigma > 1 ? (u64)L.sigma - 1 : 1);
            rec_cmax = (u32)std::min<int>(7, 124 / rec_b);
            const u64 min_occ = getenv("GRLBWT_PART_MIN_OCC") ? (u64)atoll(getenv("GRLBWT_PART_MIN_OCC")) : ((u64)1 << 20);
            part = rec_cmax >= 2 && n_occ >= min_occ;
            if (!part) rec_cmax = 0;
        }
        int part_bits = 0;
        if (part) {
            // partitions of at most ~5000 records: even a level whose phrases are ALL distinct fits the LDS tables (8192 entries)
            part_bits = 4;
            while (part_bits < 20 && (n_occ >> part_bits) > 5000) part_bits++;
            // (GRLBWT_PART_BITS: the tests make partitions too large for their table, to take the fallback on the device)
            if (const char *pb = getenv("GRLBWT_PART_BITS")) part_bits = std::max(1, std::min(20, atoi(pb)));
        }

        // ---- a3: hash every phrase occurrence ------------------------------
        // table capacity: the number of distinct phrases is unknown and usually << the number of
        // occurrences (20 k vs 30 M at level 0 of DNA reads): estimate the distinct fraction on a prefix,
        // size the table for load <= ~0.6, and grow x4 (re-running the pass) if a lane runs out of
        // probes because the prefix was not representative; cap_max = 2*n_occ always fits.
        u64 cap_max = 1024;
        while (cap_max < 2 * n_occ && cap_max < (1ull << 31)) cap_max <<= 1;      // slot ids are u32 with bit 31 spare (prim::kClaimBit)
        P.next_text.alloc(n_occ);
        DBuf<u32> scal(8);                        // [1] error flag, [2..3] debug, [4] giant phrases listed
        const u64 giant_cap = n / HashInsertFn<cell_t, FIRST>::kLongWalk + 2;      // (a listed phrase has more than kLongWalk cells of its own)
        DBuf<u64> giant_list(giant_cap);
        u64 cap = cap_max;
        double frac = 1.0;
        u64 s_blk = 0, s_stride = 0, s_n = 0, s_distinct = 0;      // the sample (blocks of s_blk cells, s_stride apart) and its distinct phrases
        u64 s_long = 0;                                             // partitioned naming: phrase occurrences of the sample that are too long for a record
        {
            // Table capacity from a sample of 2^20 cells (256 blocks spread evenly over the text) hashed into a table of its
            // own: its distinct fraction `frac`, extrapolated to the whole text.  That over-sizes the table whenever repetition
            // is global rather than local (level 1 of the 10 GB build: 964 M occurrences of 101 M phrases, 90 % distinct within
            // any 2^20 cells -> 2^31 slots), and measured that is the better side to err on: the pass runs FASTER on the sparse
            // table (75 ms at 2^31 slots, 83 ms at 2^28: fewer probes), zeroing it costs what that gains, and the compaction
            // afterwards does not scan the table (claim bits, below).  Estimating the number of distinct phrases from the
            // sample's abundance classes (Chao1) was tried and is 10-20x too low on this data (heterogeneous abundances):
            // two overflow re-runs per level.  An exact count (one more hashing pass over a 1/64 slice of the hash space)
            // costs about what a right-sized table saves.
            StageTimer st(&tm.hash, "hash");
            // (first level of a large text: 2^22 cells, so that the sample's phrases -- the hot table below -- cover the occurrences well)
            const u64 want_s = (FIRST && n >= (1ull << 26)) ? (1ull << 22) : (1ull << 20);
            const u64 n_s = n < want_s ? n : want_s;
            if (n_s < n) {
                const u64 blk = 4096, nblk = n_s / blk, stride = n / nblk;
                s_blk = blk; s_stride = stride; s_n = n_s;
                u64 cap_s = 1024;
                while (cap_s < 2 * n_s) cap_s <<= 1;
                DBuf<u64> tk(cap_s), trep;
                DBuf<idx_t> tc(cap_s);
                if (HashInsertFn<cell_t, FIRST>::kExact) trep.alloc(cap_s);
                tk.zero(); tc.zero(); scal.zero();
                typedef HashInsertFn<cell_t, FIRST> HF;
                HF fs{t, ops, startbits.p, wordbase.p, tk.p, cap_s - 1, cap_s, 0, P.next_text.p, scal.p, n, n_occ, trep.p};
                fs.rec_b = rec_b; fs.rec_cmax = rec_cmax;       // (partitioned naming: only the long phrases reach the sample's table)
                fs.walk_cap = HF::kLongWalk;        // (the sample leaves the long phrases out: they are listed and hashed by waves in the real pass)
                prim::for_each_agg(n_s, SampledFn<HF>{fs, blk, stride}, SlotCountAdd{tc.p, 1}, true, "hash_sample");
                const prim::Pair<u64, u64> so = prim::reduce_sum<prim::Pair<u64, u64>>(cap_s, SampleCountIn{tk.p, tc.p}, "hash_sample_count");      // (one reduction: one synchronisation)
                const u64 d_s = so.a;
                const u64 occ_s = std::max<u64>(so.b, 1);
                if (part) s_long = occ_s;
                frac = (double)d_s / (double)occ_s;
                if (frac > 1.0) frac = 1.0;
                s_distinct = d_s;
                u64 want = (u64)(1.7 * frac * (double)n_occ) + 4096;            // target load <= ~0.6 if the sample is representative
                if (part) want = (u64)(3.0 * (double)s_long * ((double)n / (double)n_s)) + 4096;   // the table only sees the long phrases: room for all of them being distinct
                cap = 4096;
                while (cap < want) cap <<= 1;
                if (cap > cap_max) cap = cap_max;
                if (const char *ov = prim::dev_env("GRLBWT_TABLE_LOG2")) {      // experiments: "l0,l1,..." log2 of the slots per level (0 = keep)
                    int lvl = prim::rt().tag, k = 0;
                    const char *q = ov;
                    while (k < lvl && *q) { if (*q == ',') k++; q++; }
                    const int lg = (k == lvl) ? atoi(q) : 0;
                    if (lg >= 10 && lg <= 40) cap = std::min<u64>((u64)1 << lg, cap_max);
                }
                if (getenv("GRLBWT_TABLE_TRACE")) fprintf(stderr, "[grlbwt] level %d: sample %llu occurrences, %llu distinct -> table %llu slots for %llu occurrences\n",
                                                          prim::rt().tag, (unsigned long long)occ_s, (unsigned long long)d_s, (unsigned long long)cap, (unsigned long long)n_occ);
            }
        }
        DBuf<u64> keys, rep_pos;
        DBuf<u64> claim(nwords + 1);             // bit p: the phrase occurrence starting at p created its table entry
        DBuf<idx_t> counts;
        // LDS pre-aggregation of the counts pays when few distinct phrases take most occurrences (level 0 of
        // DNA: 20 k phrases, 30 M occurrences); with mostly-distinct phrases the cache only thrashes.
        // (GRLBWT_FORCE_DIRECT_INDEX=1: the tests take the direct index -- below -- on texts too small for a sample)
        const bool force_direct = HashInsertFn<cell_t, FIRST>::kExact && !part && getenv("GRLBWT_FORCE_DIRECT_INDEX") != nullptr;
        const bool aggregate = frac < 0.25 || force_direct;
        // Table layout.  With few hot phrases the keys stay on lines of their own (the atomics on a hot count would keep
        // invalidating the key every probe reads: measured 4x slower interleaved).  With mostly-distinct phrases every
        // occurrence fetches a random key AND a random count: 16-byte (key, count) slots make that one line.
        const bool interleaved = !aggregate;      // (10 GB build: 0.6 % faster than separate arrays on the same box)
        const int ks = interleaved ? 1 : 0;
        const u64 cs = interleaved ? 16 / sizeof(idx_t) : 1;
        const idx_t *counts_p = nullptr;
        // Hot table (HashInsertFn::hot_keys): when few phrases dominate, the phrases of the sample get a small dense table of
        // their own in front of the big one -- slots [0, cap_hot) -- filled by hashing the sample once more (claims on: they
        // are dictionary phrases like the others) and read-only in the pass over the text.  GRLBWT_NO_HOT_TABLE=1 switches it off.
        u64 cap_hot = 0;
        DBuf<u64> rec_k, rec_hi;
        if (part) { rec_k.alloc(n_occ); rec_hi.alloc(n_occ); }
        // Direct index instead (HashInsertFn::direct_index): byte texts with at most 8 distinct cell values, told apart by three of
        // their bits.  GRLBWT_NO_DIRECT_INDEX=1 keeps the hot table.
        bool direct = false;
        u32 dir_b[3] = {0, 0, 0};
        u64 sym_of_code = 0;
        if (HashInsertFn<cell_t, FIRST>::kExact && aggregate && (s_n || force_direct) && !part && sym_present_known && !prim::test_env("GRLBWT_NO_DIRECT_INDEX")) {
            int syms[256], ns = 0;
            for (int i = 0; i < 256; i++) if ((sym_present[i >> 6] >> (i & 63)) & 1ull) syms[ns++] = i;
            for (u32 a = 0; a < 8 && !direct && ns <= 8; a++) for (u32 b = a + 1; b < 8 && !direct; b++) for (u32 c = b + 1; c < 8 && !direct; c++) {
                u32 seen = 0; bool ok = true; u64 soc = 0;
                for (int k = 0; k < ns && ok; k++) {
                    const u32 code = ((syms[k] >> a) & 1u) | (((syms[k] >> b) & 1u) << 1) | (((syms[k] >> c) & 1u) << 2);
                    if (seen & (1u << code)) ok = false;
                    seen |= 1u << code;
                    soc |= (u64)syms[k] << (8 * code);
                }
                if (ok) { direct = true; dir_b[0] = a; dir_b[1] = b; dir_b[2] = c; sym_of_code = soc; }
            }
        }
        if (direct && (const void *)t == text0 && text0 != (const void *)own0.p) {      // (level 0 of a text the caller lent)
            // The direct index trusts the set of cell values the statistics found: a value outside it would take another symbol's
            // code and two phrases one slot.  The engine's own copies cannot change; a buffer the caller lent (grlbwt_text_attach_device)
            // must not -- a strided sample of it is looked at again here (the whole text would cost another 4 ms per 10 GB).
            if constexpr (sizeof(cell_t) == 1) {
                const u64 stride = n >= (1ull << 22) ? n >> 20 : 1, m = n / stride;
                const u64 foreign = prim::reduce_sum<u64>(m, ForeignByteIn{(const u8 *)t, stride, sym_present[0], sym_present[1], sym_present[2], sym_present[3]}, "hash_prepare");
                if (foreign) throw prim::Error(-22, "the attached text has changed since its statistics were taken (a cell value that was not there)");
            }
        }
        if (direct) {
            cap_hot = HashInsertFn<cell_t, FIRST>::kDirectSlots;
            if (cap_max > (1ull << 30)) cap_max = 1ull << 30;             // slot ids of both regions stay below 2^31
            if (cap > cap_max) cap = cap_max;
            if (getenv("GRLBWT_TABLE_TRACE")) fprintf(stderr, "[grlbwt] level %d: direct index on bits %u, %u, %u of a cell (%llu slots in front of the table)\n", prim::rt().tag,
                                                      dir_b[0], dir_b[1], dir_b[2], (unsigned long long)cap_hot);
        } else
        if (aggregate && s_n && !part && !prim::test_env("GRLBWT_NO_HOT_TABLE")) {
            cap_hot = 1024;
            while (cap_hot < 4 * s_distinct) cap_hot <<= 1;               // load <= 0.25: short probe chains
            if (cap_max > (1ull << 30)) cap_max = 1ull << 30;             // slot ids of both tables stay below 2^31
            if (cap > cap_max) cap = cap_max;
            if (getenv("GRLBWT_TABLE_TRACE")) fprintf(stderr, "[grlbwt] level %d: hot table of %llu slots for the %llu phrases of the sample\n", prim::rt().tag,
                                                      (unsigned long long)cap_hot, (unsigned long long)s_distinct);
        }
        {
            StageTimer st(&tm.hash, "hash");
            typedef HashInsertFn<cell_t, FIRST> HF;
            for (;;) {
                idx_t *cnt;
                if (interleaved) {
                    keys.alloc(2 * cap); keys.zero();
                    cnt = (idx_t *)(keys.p + 1);          // the count lives in the second half of the slot
                } else {
                    keys.alloc(cap_hot + cap); counts.alloc(cap_hot + cap);
                    keys.zero(); counts.zero();
                    cnt = counts.p;
                }
                counts_p = cnt;
                if (HF::kExact) rep_pos.alloc(cap_hot + cap);                          // (written by the lanes that claim a slot)
                scal.zero();
                claim.zero();
                u64 probe_limit = (cap == cap_max) ? cap : 96;
                HF f{t, ops, startbits.p, wordbase.p, keys.p + cap_hot, cap - 1, probe_limit, ks,
                     P.next_text.p, scal.p, n, n_occ, rep_pos.p ? rep_pos.p + cap_hot : nullptr, claim.p};
                if (part) { f.rec_k = rec_k.p; f.rec_hi = rec_hi.p; f.rec_b = rec_b; f.rec_cmax = rec_cmax; }
                f.giant_list = giant_list.p; f.giant_n = scal.p + 4; f.giant_cap = (u32)std::min<u64>(giant_cap, 0xFFFFFFFFull);
                if (direct) { f.dir_on = 1; f.dir_b0 = dir_b[0]; f.dir_b1 = dir_b[1]; f.dir_b2 = dir_b[2]; f.dir_rep = rep_pos.p; f.slot_base = (u32)cap_hot; }
                else if (cap_hot) {
                    HF fh{t, ops, startbits.p, wordbase.p, keys.p, cap_hot - 1, cap_hot, 0, P.next_text.p, scal.p, n, n_occ, rep_pos.p, claim.p};
                    fh.walk_cap = HF::kLongWalk;            // (a phrase of thousands of cells has no business in the hot table: the pass over the text
                                                            // would find it there and compare it cell by cell with itself)
                    prim::for_each_agg(s_n, SampledFn<HF>{fh, s_blk, s_stride, claim.p}, NoCountAdd{}, false, "hash_hot");
                    f.hot_keys = keys.p; f.hot_mask = cap_hot - 1; f.slot_base = (u32)cap_hot;
                }
                if constexpr (!FIRST) {
                    if (part && !getenv("GRLBWT_PART_ONE_PASS")) {
                        // the records by a lean streaming pass of their own, the long phrases from a list through the table
                        prim::dev_memset(P.next_text.p, 0, n_occ * sizeof(u32));
                        DBuf<u64> lbits(nwords + 1);
                        lbits.zero();
                        auto records = [&](auto bw) {
                            prim::for_each_set_bit<idx_t>(n, startbits.p, wordbase.p, PhraseRecordFn<cell_t, decltype(bw)::value>{t, ops, startbits.p, wordbase.p, n, n_occ,
                                                                                                                           rec_k.p, rec_hi.p, rec_b, rec_cmax, lbits.p, scal.p}, "hash_phrases");
                        };
                        switch (rec_b) {
#define GRL_REC_CASE(BW) case BW: records(std::integral_constant<int, BW>()); break;
                            GRL_REC_CASE(10) GRL_REC_CASE(11) GRL_REC_CASE(12) GRL_REC_CASE(13) GRL_REC_CASE(14) GRL_REC_CASE(15) GRL_REC_CASE(16)
                            GRL_REC_CASE(17) GRL_REC_CASE(18) GRL_REC_CASE(19) GRL_REC_CASE(20) GRL_REC_CASE(21) GRL_REC_CASE(22) GRL_REC_CASE(23)
                            GRL_REC_CASE(24) GRL_REC_CASE(25) GRL_REC_CASE(26) GRL_REC_CASE(27) GRL_REC_CASE(28) GRL_REC_CASE(29) GRL_REC_CASE(30)
#undef GRL_REC_CASE
                            default: records(std::integral_constant<int, 0>()); break;      // (narrow symbols: small levels)
                        }
                        DBuf<u32> lbase(nwords + 1);
                        const u64 nl = (u64)prim::exclusive_scan<u32>(nwords, PopcIn32{lbits.p}, lbase.p, false, "hash_long_list");
                        DBuf<u64> lpos(nl ? nl : 1);
                        if (nl) prim::for_each(nwords, BitPositionsFn{lbits.p, lbase.p, lpos.p}, "hash_long_list");
                        lbits.release(); lbase.release();
                        if (nl) prim::for_each_agg(nl, ListedFn<HF>{f, lpos.p, claim.p}, SlotCountAdd{cnt, cs}, false, "hash_long_phrases");
                    } else launch_hash<cell_t, FIRST>(f, cnt, cs, n, aggregate);
                } else if (direct && n_occ < 0xFFFFFFF0ull && !prim::dev_env("GRLBWT_NO_NAME_STREAM")) {
                    // the direct index in a kernel of its own (prim::name_stream): what it cannot name -- phrases of more than 7 cells,
                    // the last cells of the text -- is marked and takes the general code from a list afterwards
                    DBuf<u64> lbits(nwords + 1);
                    lbits.zero();
                    prim::name_stream(n, f, SlotCountAdd{cnt, cs}, lbits.p, "hash_phrases");
                    DBuf<u32> lbase(nwords + 1);
                    const u64 nl = (u64)prim::exclusive_scan<u32>(nwords, PopcIn32{lbits.p}, lbase.p, false, "hash_long_list");
                    DBuf<u64> lpos(nl ? nl : 1);
                    if (nl) prim::for_each(nwords, BitPositionsFn{lbits.p, lbase.p, lpos.p}, "hash_long_list");
                    lbits.release(); lbase.release();
                    if (nl) prim::for_each_agg(nl, ListedFn<HF>{f, lpos.p, claim.p}, SlotCountAdd{cnt, cs}, aggregate, "hash_long_phrases");
                } else launch_hash<cell_t, FIRST>(f, cnt, cs, n, aggregate);
                std::vector<u32> sc = scal.to_host(8);
                if (sc[4] && !sc[1]) {            // the phrases too long for one lane: one wave each
                    prim::for_each_giant((u64)sc[4], giant_list.p, f, SlotCountAdd{cnt, cs}, "hash_giant_phrases");
                    sc = scal.to_host(8);
                }
                if (sc[1] == 1) {
                    if (cap == cap_max) throw prim::Error(-28, "phrase hash table overflow");
                    cap = cap * 4 > cap_max ? cap_max : cap * 4;
                    L.info.table_retries++;
                    continue;
                }
                if (sc[1]) throw prim::Error(-71, "phrase hashing: consistency check " + std::to_string(sc[1]) + " failed (" +
                                                       std::to_string(sc[2]) + ", " + std::to_string(sc[3]) + ")");
                if (direct) prim::for_each(cap_hot, DirectFixFn{counts_p, keys.p, rep_pos.p, claim.p, sym_of_code}, "hash_direct_fix");
                break;
            }
        }
        P.cap = cap_hot + cap;

        // ---- partitioned naming: group the records, de-duplicate and count per partition -------------------------------
        u64 Ds = 0;
        DBuf<u32> dcnt;
        DBuf<u64> dkey, dhi;                      // the partitions' distinct records (staging: partition p's at pstart[p] ..)
        if (part) {
            StageTimer st(&tm.hash, "hash");
            DBuf<u64> rec_k2(n_occ), rec_hi2(n_occ);
            const int res = P.psort.forward(rec_k.p, rec_hi.p, rec_k2.p, rec_hi2.p, n_occ, part_bits, "phrase_part");
            DBuf<u64> skey, shi;
            if (res) { skey = std::move(rec_k2); shi = std::move(rec_hi2); dkey = std::move(rec_k); dhi = std::move(rec_hi); }
            else { skey = std::move(rec_k); shi = std::move(rec_hi); dkey = std::move(rec_k2); dhi = std::move(rec_hi2); }
            const u64 nparts = (u64)1 << part_bits;
            DBuf<u64> &pstart = P.pstart;
            pstart.alloc(nparts + 1);
            prim::for_each(nparts + 1, prim::RecBoundsFn{skey.p, n_occ, part_bits, nparts, pstart.p}, "phrase_part.bounds");
            P.lid.alloc(n_occ); P.pbase.alloc(nparts + 1); dcnt.alloc(n_occ);
            DBuf<u32> pcount(nparts), ovf(1);
            ovf.zero();
            prim::rec_dedupe(nparts, pstart.p, skey.p, shi.p, RecValid{}, P.lid.p, pcount.p, dkey.p, dhi.p, dcnt.p, ovf.p, "phrase_dedupe");
            if (ovf.get(0)) {
                // a partition with more distinct phrases than its LDS table takes (or an injected limit in the tests): this level
                // goes through the hash table after all
                if (getenv("GRLBWT_TABLE_TRACE")) fprintf(stderr, "[grlbwt] level %d: a phrase partition overflowed, falling back to the hash table\n", prim::rt().tag);
                L.info.table_retries++;
                skey.release(); shi.release(); dkey.release(); dhi.release(); keys.release(); counts.release(); rep_pos.release(); claim.release();
                startbits.release(); wordbase.release(); dcnt.release();
                P.clear();
                hash_local<cell_t, FIRST>(t, n, ops, P, L, false);
                return;
            }
            Ds = (u64)prim::exclusive_scan<u32>(nparts, PtrU32In{pcount.p}, P.pbase.p, true, "phrase_dedupe.scan");
            skey.release(); shi.release();                // the sorted records are no longer needed (the emission walks the partitions: pstart, lid)
            P.Ds = Ds; P.part_bits = part_bits; P.rec_b = rec_b; P.slot0 = (u32)P.cap;
            if (P.cap + Ds >= (1ull << 32)) throw prim::Error(-75, "phrase tables beyond 2^32 entries");
            P.cap += Ds;                                  // values of the record phrases live behind the table's slots
        }

        // ---- a5: distinct phrases of this text (from the claim bits: the table itself is not scanned) -------------------
        {
            StageTimer st(&tm.dict_sort, "dict_sort");
            DBuf<idx_t> cbase(nwords + 1);
            const u64 Dl = (u64)prim::exclusive_scan<idx_t>(nwords, PopcIn{claim.p}, cbase.p, false, "table_compact");
            const u64 D = Ds + Dl;
            if (D >= 0xFFFFFFF0ull) throw prim::Error(-75, "dictionary too large (>= 2^32 phrases)");
            P.D = D;
            P.ph_pos.alloc(D); P.ph_freq.alloc(D); P.ph_len.alloc(D); P.ph_slot.alloc(D); P.ph_lastT.alloc(D); P.ph_off.alloc(D + 1);
            if (part) {                                   // phrases [0, Ds): from the partitions' staging areas
                P.ph_key.alloc(Ds); P.ph_vflag.alloc(Ds);
                prim::for_each(((u64)1 << part_bits) * 64, PartPhraseFn{P.pbase.p, P.pstart.p, dkey.p, dhi.p, dcnt.p, P.slot0, P.ph_key.p, P.ph_pos.p, P.ph_freq.p,
                                                                     P.ph_len.p, P.ph_slot.p, P.ph_lastT.p, P.ph_vflag.p}, "phrase_dedupe.phrases");
                dkey.release(); dhi.release(); dcnt.release();
            }
            // ... and the phrases of the table behind them
            prim::for_each(nwords, ClaimSlotsFn{claim.p, cbase.p, startbits.p, wordbase.p, P.next_text.p, P.ph_slot.p + Ds}, "table_compact");
            prim::for_each(Dl, ClaimCompactFn<cell_t, FIRST>{CompactTableFn<cell_t, FIRST>{t, ops, startbits.p, keys.p, counts_p, P.ph_pos.p + Ds,
                                                             P.ph_freq.p + Ds, P.ph_len.p + Ds, P.ph_slot.p + Ds, P.ph_lastT.p + Ds, ks, cs, rep_pos.p, n}}, "table_compact");
            wordbase.release(); claim.release();
            // (frequencies and lengths summed by ONE reduction, the offsets by a scan whose total is known already: two host
            // synchronisations instead of four)
            const prim::Pair<u64, u64> fl = prim::reduce_sum<prim::Pair<u64, u64>>(D, FreqLenIn{P.ph_freq.p, P.ph_len.p}, "dict_totals");
            if (fl.a != n_occ) throw prim::Error(-71, "phrase frequencies (" + std::to_string(fl.a) + ") do not add up to the parse size (" +
                                                           std::to_string(n_occ) + ")");
            P.maxlen = prim::reduce_max<u32>(D, LenIn{P.ph_len.p}, "dict_maxlen");   // (an atomicMax per insert serialised on one address)
            const u64 S64 = fl.b;
            if (S64 >= 0xFFFFFFF0ull) throw prim::Error(-75, "dictionary too large (>= 2^32 symbols)");
            prim::exclusive_scan_nosync<u32>(D, LenIn{P.ph_len.p}, P.ph_off.p, true, "dict_offsets");
            P.S = S64;
        }
    }

    // the exchanges of the collection-level mode; nullptr = this engine holds the whole collection
    struct Comm {
        int rank = 0, size = 1;
        void *user = nullptr;
        int (*ag)(void *, const void *, void *, u64) = nullptr;
        int (*a2a)(void *, const void *, const u64 *, const u64 *, void *, const u64 *, const u64 *) = nullptr;
        bool stream_ordered = false;      // callbacks enqueue on the engine's stream: no host synchronisation around them
        bool keep_parts = false;          // the image stays where its runs were induced: no all-gather at the end (dist_finish)
        // bytes this rank sends to OTHER ranks, by exchange site (profiling builds: "@xfer:<site>" entries of the launch profile,
        // what tools/gpu_scale_projection.py prices the fabric with); named("...") tags the next exchange
        mutable const char *what = "unnamed";
        const Comm &named(const char *w) const { what = w; return *this; }
        void account(u64 bytes_to_others) const {
            if (prim::rt().profile) {
                auto &a = prim::rt().prof[std::string("@xfer:") + what + (prim::rt().tag >= 0 ? "#" + (prim::rt().phase ? std::string(1, prim::rt().phase) : std::string()) + std::to_string(prim::rt().tag) : std::string())];
                a.launches += 1; a.bytes += bytes_to_others;
            }
            what = "unnamed";
        }
        void allgather(const void *send, void *recv, u64 bytes) const {
            if (!stream_ordered) prim::sync();
            if (ag(user, send, recv, bytes) != 0) throw prim::Error(-5, "allgather callback failed");
        }
        // A failure that only this rank can have (its phrase table overflows, its device runs out of memory, an internal
        // check fails) must not leave the others waiting in the next exchange: the local code records it here (fail()) and
        // carries on to the next counter exchange with harmless values; every counter exchange carries the flag, and every
        // rank raises there.
        mutable int pending = 0;
        mutable std::string pending_msg;
        void fail(const prim::Error &e) const { if (!pending) { pending = e.code ? e.code : -71; pending_msg = e.what(); } }
        std::vector<u64> allgather_u64(const std::vector<u64> &mine) const {
            const u64 c = mine.size(), w = c + 1;
            std::vector<u64> m2(mine);
            m2.push_back((u64)(u32)(-pending));
            DBuf<u64> s(w), r(w * size);
            prim::h2d(s.p, m2.data(), w * 8);
            allgather(s.p, r.p, w * 8);
            std::vector<u64> all = r.to_host(w * size), out(c * size);
            for (int g = 0; g < size; g++) {
                if (all[g * w + c])
                    throw prim::Error(-(int)all[g * w + c], g == rank ? pending_msg : "rank " + std::to_string(g) + " failed (error " +
                                                                                       std::to_string(-(long long)all[g * w + c]) + ")");
                for (u64 i = 0; i < c; i++) out[g * c + i] = all[g * w + i];
            }
            return out;
        }
        // counts in elements of `elem` bytes; send blocks packed in destination order, receive blocks in source order.
        // `max_block` = the largest block of the whole exchange in elements (all ranks pass the same value: they hold the
        // count matrix): blocks above the limit go in several rounds, the same number on every rank.  (torch 2.10 + RCCL
        // 2.26 delivers HALF of an all-to-all block of 2 GB, silently, and is fine at 1 GiB: tools/gpu_rccl_sizes.py.)
        void alltoall(const void *send, const std::vector<u64> &scnt, void *recv, const std::vector<u64> &rcnt, u64 elem, u64 max_block) const {
            static const u64 limit = prim::test_env("GRLBWT_A2A_BLOCK") ? (u64)atoll(prim::test_env("GRLBWT_A2A_BLOCK")) : ((u64)256 << 20);
            const u64 per = std::max<u64>(limit / elem, 1);
            const u64 rounds = std::max<u64>((max_block + per - 1) / per, 1);
            std::vector<u64> sb(size), so(size), rb(size), ro(size), sbase(size + 1, 0), rbase(size + 1, 0);
            for (int g = 0; g < size; g++) { sbase[g + 1] = sbase[g] + scnt[g]; rbase[g + 1] = rbase[g] + rcnt[g]; }
            account((sbase[size] - scnt[rank]) * elem);
            // my own block never leaves the device: a plain copy (GRLBWT_A2A_SELF_VIA_COMM=1: through the callback like the
            // others -- the tests do that so that a one-rank RCCL run still moves data through RCCL)
            static const bool self_via_comm = getenv("GRLBWT_A2A_SELF_VIA_COMM") != nullptr;
            if (!self_via_comm) {
                if (scnt[rank] != rcnt[rank]) throw prim::Error(-71, "alltoallv: my own block has two sizes");
                prim::d2d((char *)recv + rbase[rank] * elem, (const char *)send + sbase[rank] * elem, scnt[rank] * elem);
                if (size == 1) return;
            }
            if (!stream_ordered) prim::sync();
            for (u64 k = 0; k < rounds; k++) {
                for (int g = 0; g < size; g++) {
                    const bool skip = !self_via_comm && g == rank;
                    const u64 s0 = std::min(k * per, scnt[g]), s1 = skip ? s0 : std::min((k + 1) * per, scnt[g]);
                    const u64 r0 = std::min(k * per, rcnt[g]), r1 = skip ? r0 : std::min((k + 1) * per, rcnt[g]);
                    sb[g] = (s1 - s0) * elem; so[g] = (sbase[g] + s0) * elem;
                    rb[g] = (r1 - r0) * elem; ro[g] = (rbase[g] + r0) * elem;
                }
                if (a2a(user, send, sb.data(), so.data(), recv, rb.data(), ro.data()) != 0) throw prim::Error(-5, "alltoallv callback failed");
            }
        }
        // variable-length all-gather of a typed device array -> dense concatenation in rank order.
        // same_counts: `base` already holds the prefix of every rank's count (an earlier call with the same counts).
        // Every rank sends its block straight to its place in every peer's result through the all-to-all callback (xGMI is
        // point to point: N - 1 direct writes per block); no padding to the longest block, no staging copy, no unpacking
        // pass.  dest: the caller's own array of base[size] elements (then the returned buffer is empty).
        template <class T>
        DBuf<T> allgather_v(const T *send, u64 count, std::vector<u64> &base, bool same_counts = false, T *dest = nullptr) const {
            if (!same_counts || base.size() != (size_t)size + 1) {
                std::vector<u64> cnt = allgather_u64({count});
                base.assign(size + 1, 0);
                for (int g = 0; g < size; g++) base[g + 1] = base[g] + cnt[g];
            }
            if (base[rank + 1] - base[rank] != count) throw prim::Error(-71, "allgather_v: counts changed between calls");
            account(count * sizeof(T) * (u64)(size - 1));
            DBuf<T> dense;
            if (!dest) { dense.alloc(base[size]); dest = dense.p; }
            static const bool self_via_comm = getenv("GRLBWT_A2A_SELF_VIA_COMM") != nullptr;
            if (!self_via_comm || !a2a) prim::d2d(dest + base[rank], send, count * sizeof(T));
            if ((size == 1 && !self_via_comm) || !a2a) return dense;
            static const u64 limit = prim::test_env("GRLBWT_A2A_BLOCK") ? (u64)atoll(prim::test_env("GRLBWT_A2A_BLOCK")) : ((u64)256 << 20);
            const u64 per = std::max<u64>(limit / sizeof(T), 1);
            u64 mx = 0;
            for (int g = 0; g < size; g++) mx = std::max(mx, base[g + 1] - base[g]);
            const u64 rounds = std::max<u64>((mx + per - 1) / per, 1);
            std::vector<u64> sb(size), so(size), rb(size), ro(size);
            if (!stream_ordered) prim::sync();
            for (u64 k = 0; k < rounds; k++) {
                const u64 s0 = std::min(k * per, count), s1 = std::min((k + 1) * per, count);
                for (int g = 0; g < size; g++) {
                    const bool skip = !self_via_comm && g == rank;
                    const u64 cg = base[g + 1] - base[g], r0 = std::min(k * per, cg), r1 = skip ? r0 : std::min((k + 1) * per, cg);
                    sb[g] = skip ? 0 : (s1 - s0) * sizeof(T); so[g] = s0 * sizeof(T);
                    rb[g] = (r1 - r0) * sizeof(T); ro[g] = (base[g] + r0) * sizeof(T);
                }
                if (a2a(user, send, sb.data(), so.data(), dest, rb.data(), ro.data()) != 0) throw prim::Error(-5, "alltoallv callback failed");
            }
            return dense;
        }
    };

    // ---- collection-level mode: records addressed to the OWNER of a dictionary position -------------------------------------
    // rec[i] = position << 32 | payload.  Sorted (stable) by the rank whose part of the merged dictionary holds the position
    // (dsb = sbase[0..N] on the device); cnt[d] = records for rank d.
    // (own_of given: positions travel as (owner, offset in the owner's part) -- own_of[i] is record i's owner, the position field its
    // offset; that is how a dictionary of 2^32 symbols and more is addressed with 32-bit fields)
    static u64 test_dict_part_pad() {
        static const u64 pad = prim::test_env("GRLBWT_TEST_DICT_PART_PAD") ? (u64)atoll(prim::test_env("GRLBWT_TEST_DICT_PART_PAD")) : 0;
        return pad;
    }
    void bucket_by_owner(const Comm &C, const u64 *dsb, DBuf<u64> &rec, u64 n, std::vector<u64> &cnt, const char *name, DBuf<u32> *own_of = nullptr) {
        const int N = C.size;
        int obits = (int)bitlen64((u64)N - 1);
        if (obits < 1) obits = 1;
        DBuf<u32> own, own2(n);
        DBuf<u64> rec2(n), bound(2 * ((u64)N + 1));
        if (own_of) own = std::move(*own_of);
        else { own.alloc(n); prim::for_each(n, PosOwnerFn{rec.p, dsb, N, own.p}, name); }
        const int res = prim::sort_pairs<u32, u64>(own.p, rec.p, own2.p, rec2.p, n, 0, obits, name);
        prim::for_each((u64)N + 1, KeyBoundFn{res ? own2.p : own.p, n, nullptr, bound.p}, name);
        std::vector<u64> bh = bound.to_host(2 * ((u64)N + 1));
        cnt.assign(N, 0);
        for (int d = 0; d < N; d++) cnt[d] = bh[2 * (d + 1)] - bh[2 * d];
        if (res) rec = std::move(rec2);
    }
    // Ask the owners: the requests req[0..n) travel to the owners of their positions, `answer(requests, count, out)` fills one A
    // per request there, the answers come back.  Afterwards req[] is in owner order and back[i] answers req[i].  A failure only
    // this rank can have (memory, an internal check) is agreed on at the counter exchanges: every rank raises.
    template <class A, class F>
    void owner_round_trip(const Comm &C, const u64 *dsb, DBuf<u64> &req, u64 n, DBuf<A> &back, F answer, const char *name, const char *xname, DBuf<u32> *own_of = nullptr) {
        const int N = C.size, me = C.rank;
        std::vector<u64> scnt(N, 0), rcnt(N, 0);
        try { bucket_by_owner(C, dsb, req, n, scnt, name, own_of); }
        catch (const prim::Error &e) { C.fail(e); std::fill(scnt.begin(), scnt.end(), 0); }
        std::vector<u64> mat = C.allgather_u64(scnt);            // (raises on every rank if one of them failed above)
        u64 nr = 0, maxb = 0;
        for (int g = 0; g < N; g++) {
            rcnt[g] = mat[(u64)g * N + me]; nr += rcnt[g];
            for (int d = 0; d < N; d++) maxb = std::max(maxb, mat[(u64)g * N + d]);
        }
        DBuf<u64> mine;
        DBuf<A> ans;
        try { mine.alloc(nr); ans.alloc(nr); back.alloc(n); } catch (const prim::Error &e) { C.fail(e); }
        C.allgather_u64({});                                     // (the bulk exchanges below have no way back)
        C.named(xname).alltoall(req.p, scnt, mine.p, rcnt, 8, maxb);
        try { answer(mine.p, nr, ans.p); } catch (const prim::Error &e) { C.fail(e); }
        C.allgather_u64({});
        C.named(xname).alltoall(ans.p, rcnt, back.p, scnt, sizeof(A), maxb);
    }

    // a5-a8 on D distinct phrases given as (position in t, length, frequency, ends-with-terminator);
    // fills L (grammar, has_hocc, pre-BWT, M) and phrase_val[k] = rank<<2 | (freq>1)<<1 | lastT.
    // With a communicator (collection-level mode: t, ph_* are the MERGED dictionary, identical on every rank) the suffix
    // sort and the group stage are sharded over the ranks by ranges of the packed first-pass key: equal suffixes have equal
    // keys, so a group never spans two ranks and rank order = sorted order; the refinement reads only the dictionary, so every
    // rank finishes its own key range without talking to the others; the group stage's outputs are all-gathered.  The O(S)
    // streaming passes (dictionary, grammar walk) stay replicated.  Same kernels in both modes.
    template <class cell_t, bool FIRST>
    void dict_stage(const Comm *C, const cell_t *t, CellOps<cell_t, FIRST> ops, u64 D, u64 S, u32 maxlen, const u64 *ph_pos, const idx_t *ph_freq,
                    const u32 *ph_off, const u8 *ph_lastT, u32 sigma, LevelData &L, DBuf<u32> &phrase_val,
                    const u32 *fused_ph_slot = nullptr, u32 *fused_slot_val = nullptr,        // (both set: the values go straight to the slots)
                    const prim::U128 *pkeys = nullptr, u64 pDs = 0, int pkb = 0,              // (phrases [0, pDs) given by their records)
                    u32 pslot0 = 0,                                   // (... and their slots are pslot0 + k: GroupPhraseValFn)
                    const std::vector<u64> *dbase = nullptr,          // (collection-level mode: rank g merged the phrases [dbase[g], dbase[g + 1])
                    const std::vector<u64> *sbase = nullptr,          //  = the dictionary positions [sbase[g], sbase[g + 1]))
                    bool sharded_dict = false,                        // (t, ph_* describe MY part of the dictionary only: see below)
                    u64 maxfreq = ~0ull) {                            // (the largest phrase frequency of the round, where the caller knows it)
        L.info.D = D; L.info.S = S; L.info.max_phrase_len = maxlen;
        // DICTIONARY SHARDED BY OWNER (collection-level mode, round 5): every rank holds the phrases it merged and nothing of the
        // others' -- t, ph_pos, ph_freq, ph_off, ph_lastT are LOCAL arrays of Dl phrases / Sl symbols, a dictionary position
        // travels as the global number s0 + local offset, a phrase as d0 + local number.  What a rank needs to know about a position
        // it does not own it ASKS of the owner (owner_round_trip): the next K symbols of an unresolved suffix in every refinement
        // round, and what the group fold reads per member (frequency, left symbol, flags, phrase).  No replicated dictionary, no
        // all-gather of the merged phrases, no O(S) or O(D) pass that every rank repeats.
        const bool sharded = C && sharded_dict && dbase && sbase;
        const u64 d0 = sharded ? (*dbase)[C->rank] : 0, s0 = sharded ? (*sbase)[C->rank] : 0;
        const u64 Dl = sharded ? (*dbase)[C->rank + 1] - d0 : D, Sl = sharded ? (*sbase)[C->rank + 1] - s0 - test_dict_part_pad() : S;
        // With every frequency below 2^32 the 8 bytes the group fold reads about a suffix (frequency, left symbol, flags) travel WITH
        // its (key, position) record in the sample-sort exchange: no round trip for them afterwards (24 bytes per suffix over the
        // fabric and a gather pass on the owner -- 3.3 GB and ~14 ms per rank at N = 8 of the 10 GB collection).  The values a rank
        // sorts are then ARRIVAL INDICES; pos_arr[] / rec_arr[] give the position and the record of an arrival.
        static const bool rec_round_trip = prim::test_env("GRLBWT_DIST_REC_ROUND_TRIP") != nullptr;
        const bool carry = sharded && maxfreq < 0xFFFFFFFFull && !rec_round_trip;
        // In this form a position is (owner, OFFSET in the owner's part): the owner of an arrival is the rank it came from (arrivals
        // sit in sender order), so 32-bit fields address a dictionary whose parts are each below 2^32 symbols, whatever their sum.
        if (sharded && !carry && S >= 0xFFFFFFF0ull) throw prim::Error(-75, "dictionary too large (>= 2^32 symbols and a phrase frequency >= 2^32, or GRLBWT_DIST_REC_ROUND_TRIP)");
        const u64 sq = carry ? 0 : s0;           // what the owner subtracts from a position it is asked about
        DBuf<u32> pos_arr, perm_ai, pown;        // (carry) offset of every arrival; the sorted arrival indices once perm holds offsets again; owner by slot
        DBuf<u64> abounds;                       // (carry) arrivals [abounds[g], abounds[g + 1]) came from rank g
        DBuf<u64> rec_arr;                       // (carry) record of every arrival
        DBuf<u64> dsb;                           // sbase[] on the device (owner of a position)
        if (C && sbase) { dsb.alloc((u64)C->size + 1); prim::h2d(dsb.p, sbase->data(), ((u64)C->size + 1) * 8); }
        DBuf<u32> dict_sym(Sl), dict_phr(Sl);
        RankBits pbits;                          // phrase starts over the dictionary positions (dictionary build, suffix refinement)
        {
            StageTimer st(&tm.dict_sort, "dict_sort");
            build_rankbits32(pbits, ph_off, Dl, Sl + 1, "dict_build");
            // (4 positions per lane: 16 per lane, four phrases walked one after the other, was latency-bound -- 42 ms at 10 GB)
            prim::for_each((Sl + 3) / 4, DictBuildFn<cell_t, FIRST, 4>{t, ops, ph_off, Dl, Sl, ph_pos, dict_sym.p, dict_phr.p, pbits.words.p, pbits.base.p,
                                                                       pkeys, pDs, pkb}, "dict_build");
        }
        // ---- a6: sort all phrase suffixes (radix on the first K symbols + refinement by symbol extension) ----------
        u64 Sg = S;                              // my slots of the sorted order (all of them without a communicator)
        DBuf<u32> perm, gid, gstart;
        u64 G = 0;
        // (with a communicator the sort and the group stage below work on THIS rank's key range -- sizes, memory and
        // termination differ from rank to rank: a failure is recorded (Comm::fail) and raised by every rank at the counter
        // exchange behind the group stage)
        auto sort_local = [&] {
            StageTimer st(&tm.dict_sort, "dict_sort");
            int b = (int)bitlen64(sigma);
            if (b < 1) b = 1;
            // as many symbols as fit 64 key bits per pass (up to 8 radix passes over all suffixes in the first one)
            // run-aware keys (RunKeys) for levels with long phrases, i.e. long runs of one symbol (GRLBWT_RUN_KEYS_MIN: from which
            // phrase length on; the tests set 0)
            static const u64 run_min = getenv("GRLBWT_RUN_KEYS_MIN") ? (u64)atoll(getenv("GRLBWT_RUN_KEYS_MIN")) : 512;
            RunKeys rk;
            DBuf<u32> run_rem, run_skip;
            if ((u64)maxlen >= run_min && S > 0) rk.rb = 1 + (int)bitlen64((u64)maxlen);
            // long phrases on one GPU: every suffix keeps its slot (no "last cell" suffixes left out), so that the refinement can
            // switch to doubling rounds (DoubleKeyFn) when the symbol extension does not finish in GRLBWT_DOUBLING_AFTER rounds
            const bool longmode = !C && (u64)maxlen >= run_min && S > 0;
            static const u64 dbl_after = getenv("GRLBWT_DOUBLING_AFTER") ? (u64)atoll(getenv("GRLBWT_DOUBLING_AFTER")) : 24;
            int K = (64 - rk.rb) / b;
            if (K < 1) K = 1;
            if (K > 16) K = 16;
            // (GRLBWT_SORT_KMAX: fewer symbols in the first sort's key -- fewer radix passes, more left to the refinement)
            static const int kmax = prim::dev_env("GRLBWT_SORT_KMAX") ? atoi(prim::dev_env("GRLBWT_SORT_KMAX")) : 16;
            if (kmax >= 1 && K > kmax) K = kmax;
            if ((u64)K > (u64)maxlen + 1) K = (int)maxlen + 1;
            if (rk.rb && K * b + rk.rb > 64) rk.rb = 0;                     // (symbols too wide to share a key with a run field: plain keys)
            if (rk.rb) {
                DBuf<u32> rex(S), ends(S);
                run_rem.alloc(S); run_skip.alloc(S);
                run_skip.zero();
                prim::exclusive_scan_emit<u32>(S, RunChangeIn{dict_sym.p, pbits.words.p, S}, RunEndsEmitFn{rex.p, ends.p}, "suffix_runs");
                prim::for_each(S, RunRemFn{rex.p, ends.p, run_rem.p}, "suffix_runs");
                rk.rem = run_rem.p; rk.skip = run_skip.p;
            }
            const int kbits = K * b + rk.rb;                                // sort bits of a key
            const u64 sent = ((1ull << b) - 1ull) << rk.rb;                 // (the last symbol of a key's window: all ones = the phrase has ended)
            // (GRLBWT_SEG_CAP: the tests lower the limit so that ordinary inputs take the large-group path too)
            static const u32 cap = getenv("GRLBWT_SEG_CAP") ? (u32)atoi(getenv("GRLBWT_SEG_CAP")) : kSegCap;
            DBuf<u64> ka;
            {
                const SufKeep keep{dict_phr.p, ph_off, ph_lastT, longmode};
                if (!C) {
                    DBuf<u32> dropcnt(D + 1);
                    const u64 dropped = prim::exclusive_scan<u32>(D, PhraseDropIn{ph_off, ph_lastT, longmode}, dropcnt.p, true, "suffix_keep");
                    Sg = S - dropped;
                    ka.alloc(Sg); perm.alloc(Sg);
                    prim::for_each(S, Key0KeepFn{keep, dropcnt.p, dict_sym.p, K, b, ka.p, perm.p, rk}, "suffix_keys0");
                } else {
                    // Sample-sort exchange.  Every rank makes the (key, position) records of ITS 1/N of the dictionary positions,
                    // groups them by the rank that owns their key range (splitters from a strided sample: replicated data, same
                    // on every rank) and sends them there: a rank computes S/N keys and receives the records of its own range,
                    // positions ascending (rank order = position order, the owner grouping is stable).  (Round 2: every rank
                    // computed the keys of ALL S positions and scanned them all to find its own -- two replicated O(S) passes,
                    // 36 ms per rank at any N on the 10 GB collection: profiles/r03/scale_projection_10GB.json.)
                    const int N = C->size, me = C->rank;
                    // With few ranks the exchange costs more than it saves: at N = 2 every rank would send half of its records --
                    // 4.6 GB over ONE xGMI link at level 2 of the 10 GB build, 40-75 ms -- to save 10 ms of key computation.  Below
                    // GRLBWT_SORT_EXCHANGE_MIN ranks (default 4) every rank looks at all S positions and keeps its own key range:
                    // two replicated streaming passes, nothing on the wire.
                    static const int xmin = prim::test_env("GRLBWT_SORT_EXCHANGE_MIN") ? atoi(prim::test_env("GRLBWT_SORT_EXCHANGE_MIN")) : 4;
                    const bool exchange = N >= xmin;
                    std::vector<u64> scnt(N, 0), rcnt(N, 0);
                    DBuf<u64> sk;
                    DBuf<u32> sp;
                    u64 Sown = 0;
                    try {
                        // (GRLBWT_TEST_FAIL_RANK_SORT=<rank>: the tests make one rank fail here)
                        if (test_fail_rank("GRLBWT_TEST_FAIL_RANK_SORT", me)) throw prim::Error(-71, "suffix refinement does not terminate (injected by the test)");
                        const u64 ns = S < 8192 ? S : 8192, stride = S / ns;
                        DBuf<u64> samp(ns), dspl(N);
                        prim::for_each(ns, SampleKey0Fn{dict_sym.p, dict_phr.p, ph_off, K, b, stride, samp.p, rk}, "dist.sample_keys");
                        std::vector<u64> hs = samp.to_host(ns), spl(N, 0);
                        // A key range is also the rank's piece of the level's OUTPUT (the pre-BWT stays where it was sorted, round 5).
                        // Where the dictionary is small against the text -- level 0 of read collections: 10^6 suffixes for 10^10 symbols --
                        // the dictionary stage costs nothing and the induction everything: the splitters then cut the sample by the
                        // symbols its suffixes stand for (phrase frequencies), not by their number.  Measured on the 10 GB collection at
                        // N = 8 (profiles/r05): pass C of a piece costs by its cells and runs, not by its symbols -- slowest / fastest rank of
                        // the induction 55 / 35 ms by count, 55 / 32 ms by mass, largest image part 1.9 -> 2.5 GB: left off.
                        if (S * 16 < L.info.n_in && prim::dev_env("GRLBWT_DIST_SPLIT_BY_MASS")) {      // (opt-in: measured no better -- see below)
                            DBuf<u64> sw(ns);
                            prim::for_each(ns, SampleWeightFn{dict_phr.p, ph_freq, stride, sw.p}, "dist.sample_keys");
                            std::vector<u64> hw = sw.to_host(ns);
                            std::vector<std::pair<u64, u64>> kw(ns);
                            u64 tot = 0;
                            for (u64 i = 0; i < ns; i++) { kw[i] = {hs[i], hw[i]}; tot += hw[i]; }
                            std::sort(kw.begin(), kw.end());
                            u64 acc = 0;
                            int d = 1;
                            for (u64 i = 0; i < ns && d < N; i++) {
                                while (d < N && acc >= tot / (u64)N * (u64)d) { spl[d] = kw[i].first; d++; }
                                acc += kw[i].second;
                            }
                            for (; d < N; d++) spl[d] = ~0ull;                             // (nothing left for the last ranks)
                        } else {
                        std::sort(hs.begin(), hs.end());
                        for (int d = 1; d < N; d++) spl[d] = hs[(u64)d * ns / N];          // rank d owns keys in [spl[d], spl[d + 1])
                        }
                        prim::h2d(dspl.p, spl.data(), (u64)N * 8);
                        if (!exchange) {
                            DBuf<u8> mine(S);
                            prim::for_each(S, OwnKeyFlagFn{keep, dict_sym.p, K, b, dspl.p, N, me, mine.p, rk}, "suffix_keys0");
                            Sown = prim::reduce_sum<u64>(S, ByteIn{mine.p}, "suffix_keep");
                            ka.alloc(Sown); perm.alloc(Sown);
                            prim::exclusive_scan_emit<u32>(S, ByteIn{mine.p}, OwnKeyEmitFn{keep, dict_sym.p, K, b, ka.p, perm.p, rk}, "suffix_keys0");
                        }
                        const u64 q0 = exchange ? S * (u64)me / (u64)N : 0, q1 = exchange ? S * (u64)(me + 1) / (u64)N : 0, nq = q1 - q0;
                        DBuf<u32> kex(nq + 1);
                        const u64 nk = prim::exclusive_scan<u32>(nq, KeepRangeIn{keep, q0}, kex.p, false, "suffix_keep");
                        DBuf<u64> lk(nk), bound(2 * ((u64)N + 1));
                        DBuf<u32> lp(nk), own(nk), own2(nk), idx(nk), idx2(nk);
                        prim::for_each(nq, KeyRangeFn{keep, kex.p, q0, dict_sym.p, K, b, dspl.p, N, lk.p, lp.p, own.p, idx.p, rk}, "suffix_keys0");
                        int obits = (int)bitlen64((u64)N - 1);
                        if (obits < 1) obits = 1;
                        const int res = prim::sort_pairs<u32, u32>(own.p, idx.p, own2.p, idx2.p, nk, 0, obits, "dist.key_owner_sort");
                        sk.alloc(nk); sp.alloc(nk);
                        prim::for_each(nk, GatherKeyPosFn{res ? idx2.p : idx.p, lk.p, lp.p, sk.p, sp.p}, "dist.key_owner_sort");
                        prim::for_each((u64)N + 1, KeyBoundFn{res ? own2.p : own.p, nk, nullptr, bound.p}, "dist.owner_bounds");
                        std::vector<u64> bh = bound.to_host(2 * ((u64)N + 1));
                        for (int d = 0; d < N; d++) scnt[d] = bh[2 * (d + 1)] - bh[2 * d];
                    } catch (const prim::Error &e) { C->fail(e); std::fill(scnt.begin(), scnt.end(), 0); }
                    std::vector<u64> mat = C->allgather_u64(scnt);           // (raises on every rank if one of them failed above)
                    if (exchange) {
                        u64 maxb = 0;
                        Sg = 0;
                        for (int g = 0; g < N; g++) {
                            rcnt[g] = mat[(u64)g * N + me];
                            Sg += rcnt[g];
                            for (int d = 0; d < N; d++) maxb = std::max(maxb, mat[(u64)g * N + d]);
                        }
                        try { ka.alloc(Sg); perm.alloc(Sg); } catch (const prim::Error &e) { C->fail(e); }
                        C->allgather_u64({});                                // (the bulk exchanges below have no way back)
                        C->named("sort.sample_keys").alltoall(sk.p, scnt, ka.p, rcnt, 8, maxb);
                        C->named("sort.sample_pos").alltoall(sp.p, scnt, perm.p, rcnt, 4, maxb);
                    } else Sg = Sown;
                }
            }
            gid.alloc(Sg); gstart.alloc(Sg + 1);
            DBuf<u8> hflag(Sg), uflag(Sg);
            DBuf<u32> ex(Sg + 1);
            {
                DBuf<u64> kb(Sg);
                DBuf<u32> vb(Sg);
                const u64 *ks = ka.p;
                if (prim::sort_pairs<u64, u32>(ka.p, perm.p, kb.p, vb.p, Sg, 0, kbits, "suffix_sort0")) {
                    std::swap(perm, vb);         // the result sits in the second buffer: take it, no copy
                    ks = kb.p;
                }
                if (rk.rb) prim::for_each(Sg, InitSkipFn{perm.p, rk.rem, (u32)K, rk.skip}, "suffix_runs");
                prim::for_each(Sg, HeadFlagFn{ks, hflag.p}, "suffix_heads");
                prim::for_each(Sg, FirstUnresolvedFn{ks, hflag.p, Sg, sent, uflag.p}, "suffix_unresolved");
            }                                   // (no host synchronisation: the buffers that go out of scope here are reused in stream order)
            ka.release();
            u64 Lres = (u64)K, iters = 1;        // Lres symbols (incl. a possible sentinel) resolved so far
            DBuf<u32> act;                       // slots still unresolved (empty = all slots), ascending
            u64 A = Sg;
            bool refined = false;
            bool doubling = false;               // (see DoubleKeyFn)
            u64 Ld = 0, dbl_rounds = 0;
            DBuf<u32> slot_of, rank_ex, nskip;
            DBuf<u8> ures;
            for (;;) {
                DBuf<u32> uex(A + 1);
                const u64 U = prim::exclusive_scan<u32>(A, ByteIn{uflag.p}, uex.p, false, "suffix_unresolved_scan");
                if (U == 0) break;
                if (Lres > (u64)maxlen + (u64)K || dbl_rounds > 80) throw prim::Error(-71, "suffix refinement does not terminate");
                if (!doubling && longmode && iters > dbl_after) {
                    doubling = true;
                    Ld = Lres;                   // depth of the suffix at q from here on: Ld + skip[q]
                    if (!rk.rb) { run_skip.alloc(S); run_skip.zero(); rk.skip = run_skip.p; }
                    slot_of.alloc(S); slot_of.fill_ff();
                    prim::for_each(Sg, SlotOfFn{perm.p, slot_of.p}, "suffix_doubling");
                    ures.alloc(Sg); ures.zero();
                    prim::for_each(A, UresInitFn{refined ? act.p : nullptr, uflag.p, ures.p}, "suffix_doubling");
                    rank_ex.alloc(Sg + 1);
                }
                DBuf<u32> uslot(U), uq(U), hex(U + 1);
                DBuf<u64> ukey(U);
                DBuf<u8> uhead(U), unext(U);
                if (doubling) {
                    prim::exclusive_scan_nosync<u32>(Sg, ByteIn{hflag.p}, rank_ex.p, false, "suffix_doubling");      // group numbers of all slots
                    nskip.alloc(U);
                    prim::for_each(A, DoubleKeyFn{refined ? act.p : nullptr, uflag.p, uex.p, perm.p, hflag.p, dict_phr.p, ph_off, slot_of.p, rank_ex.p,
                                                  ures.p, rk.skip, Ld, Sg, uslot.p, uq.p, ukey.p, uhead.p, nskip.p}, "suffix_doubling");
                    dbl_rounds++;
                } else
                prim::for_each(A, ExtKeyFn{refined ? act.p : nullptr, uflag.p, uex.p, perm.p, hflag.p, dict_sym.p, pbits.words.p, S, Lres, K, b,
                                           uslot.p, uq.p, ukey.p, uhead.p, rk}, "suffix_keys");
                const u64 sent_r = doubling ? 1ull : sent;                    // "still unresolved" bit(s) of a key
                const int kbits_r = doubling ? (int)bitlen64((Sg << 1) | 1ull) : kbits;
                const u64 nseg = prim::exclusive_scan<u32>(U, ByteIn{uhead.p}, hex.p, false, "suffix_heads");
                DBuf<u32> seg_start(nseg + 1), bex(U + 1);
                prim::for_each(U, SegStartFn{uhead.p, hex.p, U, seg_start.p}, "suffix_gstart");
                prim::for_each(U, SegSortSmallFn{uhead.p, hex.p, seg_start.p, uslot.p, uq.p, ukey.p, sent_r, cap, perm.p, hflag.p, unext.p}, "suffix_sort.small");
                const u64 NB = prim::exclusive_scan<u32>(U, SegBigIn{uhead.p, hex.p, seg_start.p, cap}, bex.p, false, "suffix_sort.big_scan");
                if (NB) {                        // groups above kSegCap: by key, then (stable) by group
                    DBuf<u32> bitem(NB), bidx(NB), bidx2(NB), key2(NB), key2b(NB);
                    DBuf<u64> bkey(NB), bkey2(NB);
                    prim::for_each(U, SegBigGatherFn{uhead.p, hex.p, seg_start.p, bex.p, ukey.p, cap, bitem.p, bkey.p, bidx.p}, "suffix_sort.big_gather");
                    const u32 *i1 = prim::sort_pairs<u64, u32>(bkey.p, bidx.p, bkey2.p, bidx2.p, NB, 0, kbits_r, "suffix_sort") ? bidx2.p : bidx.p;
                    u32 *i1o = (i1 == bidx.p) ? bidx2.p : bidx.p;
                    prim::for_each(NB, SegBigSegKeyFn{i1, bitem.p, uhead.p, hex.p, key2.p}, "suffix_sort.big_groups");
                    int sbits = (int)bitlen64(nseg);
                    if (sbits < 1) sbits = 1;
                    const int res = prim::sort_pairs<u32, u32>(key2.p, (u32 *)i1, key2b.p, i1o, NB, 0, sbits, "suffix_sort");
                    prim::for_each(NB, SegBigWriteFn{res ? key2b.p : key2.p, res ? i1o : i1, bitem.p, ukey.p, uq.p, uslot.p, NB, sent_r,
                                                     perm.p, hflag.p, unext.p}, "suffix_refine");
                }
                if (doubling) prim::for_each(U, AfterDoubleFn{uq.p, nskip.p, uslot.p, perm.p, unext.p, rk.skip, slot_of.p, ures.p}, "suffix_doubling");
                act = std::move(uslot);
                uflag = std::move(unext);
                A = U;
                refined = true;
                if (!doubling) Lres += (u64)K;
                iters++;
            }
            pbits.base.release();
            G = prim::exclusive_scan<u32>(Sg, ByteIn{hflag.p}, ex.p, false, "suffix_heads");
            prim::for_each(Sg, GroupStartsFn{hflag.p, ex.p, Sg, gstart.p}, "suffix_gstart");
            prim::for_each(Sg, DenseGidFn{hflag.p, ex.p, gid.p}, "suffix_gid");
            L.info.sort_iters = iters;
        };
        // The same with the dictionary sharded by owner.  Every rank runs the same sequence of collectives (the refinement goes on
        // until NO rank has an unresolved suffix left); a failure only this rank can have is recorded and raised by every rank at
        // the next counter exchange, and the rank-local sections in between are skipped once one is pending.
        auto sort_sharded = [&] {
            StageTimer st(&tm.dict_sort, "dict_sort");
            const int N = C->size, me = C->rank;
            auto local = [&](auto &&fn) { if (!C->pending) { try { fn(); } catch (const prim::Error &e) { C->fail(e); } } };
            int b = (int)bitlen64(sigma);
            if (b < 1) b = 1;
            int K = 64 / b;
            if (K < 1) K = 1;
            if (K > 16) K = 16;
            static const int kmax = prim::dev_env("GRLBWT_SORT_KMAX") ? atoi(prim::dev_env("GRLBWT_SORT_KMAX")) : 16;
            if (kmax >= 1 && K > kmax) K = kmax;
            if ((u64)K > (u64)maxlen + 1) K = (int)maxlen + 1;
            const int kbits = K * b;
            const u64 sent = (1ull << b) - 1ull;
            static const u32 cap = getenv("GRLBWT_SEG_CAP") ? (u32)atoi(getenv("GRLBWT_SEG_CAP")) : kSegCap;
            const SufKeep keep{dict_phr.p, ph_off, ph_lastT, false};
            // splitters from a sample every rank takes of its own part
            u64 nsl = 0;
            DBuf<u64> samp;
            local([&] {
                // (GRLBWT_TEST_FAIL_RANK_SORT=<rank>: the tests make one rank fail here)
                if (test_fail_rank("GRLBWT_TEST_FAIL_RANK_SORT", me)) throw prim::Error(-71, "suffix refinement does not terminate (injected by the test)");
                nsl = Sl < 64 ? Sl : std::max<u64>(64, 8192 / (u64)N);
                if (nsl > Sl) nsl = Sl;
                samp.alloc(nsl);
                if (nsl) prim::for_each(nsl, SampleKey0Fn{dict_sym.p, dict_phr.p, ph_off, K, b, Sl / nsl, samp.p, RunKeys()}, "dist.sample_keys");
            });
            if (C->pending) nsl = 0;
            std::vector<u64> sbb;
            DBuf<u64> alls = C->named("sort.splitter_sample").template allgather_v<u64>(samp.p, nsl, sbb);
            std::vector<u64> hs = alls.to_host(sbb[N]), spl(N, 0);
            std::sort(hs.begin(), hs.end());
            if (!hs.empty()) for (int d = 1; d < N; d++) spl[d] = hs[(u64)d * hs.size() / N];
            std::vector<u64> scnt(N, 0), rcnt(N, 0);
            DBuf<u64> sk, ka, sr;
            DBuf<u32> sp;
            local([&] {
                DBuf<u64> dspl(N);
                prim::h2d(dspl.p, spl.data(), (u64)N * 8);
                DBuf<u32> kex(Sl + 1);
                const u64 nk = prim::exclusive_scan<u32>(Sl, KeepRangeIn{keep, 0}, kex.p, false, "suffix_keep");
                DBuf<u64> lk(nk), bound(2 * ((u64)N + 1));
                DBuf<u32> lp(nk), own(nk), own2(nk), idx(nk), idx2(nk);
                DBuf<u64> lr(carry ? nk : 0);
                prim::for_each(Sl, KeyRangeFn{keep, kex.p, 0, dict_sym.p, K, b, dspl.p, N, lk.p, lp.p, own.p, idx.p, RunKeys(), (u32)sq,
                                              carry ? lr.p : nullptr, ph_freq, sigma + 1}, "suffix_keys0");
                int obits = (int)bitlen64((u64)N - 1);
                if (obits < 1) obits = 1;
                const int res = prim::sort_pairs<u32, u32>(own.p, idx.p, own2.p, idx2.p, nk, 0, obits, "dist.key_owner_sort");
                sk.alloc(nk); sp.alloc(nk);
                prim::for_each(nk, GatherKeyPosFn{res ? idx2.p : idx.p, lk.p, lp.p, sk.p, sp.p}, "dist.key_owner_sort");
                if (carry) { sr.alloc(nk); prim::for_each(nk, GatherU64Fn{res ? idx2.p : idx.p, lr.p, sr.p}, "dist.key_owner_sort"); }
                prim::for_each((u64)N + 1, KeyBoundFn{res ? own2.p : own.p, nk, nullptr, bound.p}, "dist.owner_bounds");
                std::vector<u64> bh = bound.to_host(2 * ((u64)N + 1));
                for (int d = 0; d < N; d++) scnt[d] = bh[2 * (d + 1)] - bh[2 * d];
            });
            if (C->pending) std::fill(scnt.begin(), scnt.end(), 0);
            std::vector<u64> mat = C->allgather_u64(scnt);           // (raises on every rank if one of them failed above)
            u64 maxb = 0;
            Sg = 0;
            for (int g = 0; g < N; g++) {
                rcnt[g] = mat[(u64)g * N + me];
                Sg += rcnt[g];
                for (int d = 0; d < N; d++) maxb = std::max(maxb, mat[(u64)g * N + d]);
            }
            // (known to every rank alike: slots are 32-bit)
            for (int d = 0; d < N; d++) { u64 sd = 0; for (int g = 0; g < N; g++) sd += mat[(u64)g * N + d]; if (sd >= 0xFFFFFFF0ull) throw prim::Error(-75, "a key range of the dictionary's suffixes has >= 2^32 of them: use more ranks"); }
            local([&] {
                ka.alloc(Sg); perm.alloc(Sg);
                if (carry) {
                    pos_arr.alloc(Sg); rec_arr.alloc(Sg); abounds.alloc((u64)N + 1);
                    std::vector<u64> ab((u64)N + 1, 0);
                    for (int g = 0; g < N; g++) ab[g + 1] = ab[g] + rcnt[g];
                    prim::h2d(abounds.p, ab.data(), ((u64)N + 1) * 8);
                }
            });
            C->allgather_u64({});                                // (the bulk exchanges below have no way back)
            C->named("sort.sample_keys").alltoall(sk.p, scnt, ka.p, rcnt, 8, maxb);
            C->named("sort.sample_pos").alltoall(sp.p, scnt, carry ? pos_arr.p : perm.p, rcnt, 4, maxb);
            if (carry) C->named("sort.sample_rec").alltoall(sr.p, scnt, rec_arr.p, rcnt, 8, maxb);
            sk.release(); sp.release(); sr.release();
            if (carry) local([&] { prim::for_each(Sg, IotaU32Fn{perm.p}, "suffix_keys0"); });      // the values of the sort: arrival indices
            DBuf<u8> hflag, uflag;
            DBuf<u32> ex;
            local([&] {
                gid.alloc(Sg); gstart.alloc(Sg + 1); hflag.alloc(Sg); uflag.alloc(Sg); ex.alloc(Sg + 1);
                DBuf<u64> kb(Sg);
                DBuf<u32> vb(Sg);
                const u64 *ks = ka.p;
                if (prim::sort_pairs<u64, u32>(ka.p, perm.p, kb.p, vb.p, Sg, 0, kbits, "suffix_sort0")) { std::swap(perm, vb); ks = kb.p; }
                prim::for_each(Sg, HeadFlagFn{ks, hflag.p}, "suffix_heads");
                prim::for_each(Sg, FirstUnresolvedFn{ks, hflag.p, Sg, sent, uflag.p}, "suffix_unresolved");
            });
            ka.release();
            u64 Lres = (u64)K, iters = 1, A = Sg;
            DBuf<u32> act;
            bool refined = false;
            for (;;) {
                u64 U = 0;
                DBuf<u32> uex;
                local([&] { uex.alloc(A + 1); U = A ? prim::exclusive_scan<u32>(A, ByteIn{uflag.p}, uex.p, false, "suffix_unresolved_scan") : 0; });
                if (C->pending) U = 0;
                std::vector<u64> us = C->allgather_u64({U});     // (every rank goes on while ANY rank has unresolved suffixes: the owners answer)
                u64 Uany = 0;
                for (u64 v : us) Uany = std::max(Uany, v);
                if (Uany == 0) break;
                if (Lres > (u64)maxlen + (u64)K) throw prim::Error(-71, "suffix refinement does not terminate");      // (the same on every rank)
                DBuf<u32> uslot, uq, hex, uown;
                DBuf<u64> ukey, req, back;
                DBuf<u8> uhead, unext;
                local([&] {
                    uslot.alloc(U); uq.alloc(U); hex.alloc(U + 1); ukey.alloc(U); req.alloc(U); uhead.alloc(U); unext.alloc(U);
                    if (carry) uown.alloc(U);
                    if (carry) prim::for_each(A, ExtCompactAiFn{refined ? act.p : nullptr, uflag.p, uex.p, perm.p, hflag.p, pos_arr.p, uslot.p, uq.p, uhead.p, req.p,
                                                                abounds.p, N, uown.p}, "suffix_keys");
                    else prim::for_each(A, ExtCompactFn{refined ? act.p : nullptr, uflag.p, uex.p, perm.p, hflag.p, uslot.p, uq.p, uhead.p, req.p}, "suffix_keys");
                });
                if (C->pending) U = 0;
                owner_round_trip<u64>(*C, dsb.p, req, U, back, [&](const u64 *rq, u64 nrq, u64 *out) {
                    prim::for_each(nrq, ExtKeyOwnerFn{rq, dict_sym.p, pbits.words.p, Sl, sq, Lres, K, b, out}, "suffix_keys");
                }, "dist.ext_owner_sort", "sort.ext_keys", carry && !C->pending && uown.p ? &uown : nullptr);
                local([&] {
                    if (U == 0) { act = std::move(uslot); uflag = std::move(unext); A = 0; refined = true; return; }      // (nothing of mine left: I only answer)
                    prim::for_each(U, ExtAnswerFn{req.p, back.p, ukey.p}, "suffix_keys");
                    const u64 nseg = prim::exclusive_scan<u32>(U, ByteIn{uhead.p}, hex.p, false, "suffix_heads");
                    DBuf<u32> seg_start(nseg + 1), bex(U + 1);
                    prim::for_each(U, SegStartFn{uhead.p, hex.p, U, seg_start.p}, "suffix_gstart");
                    prim::for_each(U, SegSortSmallFn{uhead.p, hex.p, seg_start.p, uslot.p, uq.p, ukey.p, sent, cap, perm.p, hflag.p, unext.p}, "suffix_sort.small");
                    const u64 NB = prim::exclusive_scan<u32>(U, SegBigIn{uhead.p, hex.p, seg_start.p, cap}, bex.p, false, "suffix_sort.big_scan");
                    if (NB) {                        // groups above kSegCap: by key, then (stable) by group
                        DBuf<u32> bitem(NB), bidx(NB), bidx2(NB), key2(NB), key2b(NB);
                        DBuf<u64> bkey(NB), bkey2(NB);
                        prim::for_each(U, SegBigGatherFn{uhead.p, hex.p, seg_start.p, bex.p, ukey.p, cap, bitem.p, bkey.p, bidx.p}, "suffix_sort.big_gather");
                        const u32 *i1 = prim::sort_pairs<u64, u32>(bkey.p, bidx.p, bkey2.p, bidx2.p, NB, 0, kbits, "suffix_sort") ? bidx2.p : bidx.p;
                        u32 *i1o = (i1 == bidx.p) ? bidx2.p : bidx.p;
                        prim::for_each(NB, SegBigSegKeyFn{i1, bitem.p, uhead.p, hex.p, key2.p}, "suffix_sort.big_groups");
                        int sbits = (int)bitlen64(nseg);
                        if (sbits < 1) sbits = 1;
                        const int res = prim::sort_pairs<u32, u32>(key2.p, (u32 *)i1, key2b.p, i1o, NB, 0, sbits, "suffix_sort");
                        prim::for_each(NB, SegBigWriteFn{res ? key2b.p : key2.p, res ? i1o : i1, bitem.p, ukey.p, uq.p, uslot.p, NB, sent,
                                                         perm.p, hflag.p, unext.p}, "suffix_refine");
                    }
                    act = std::move(uslot);
                    uflag = std::move(unext);
                    A = U;
                    refined = true;
                });
                Lres += (u64)K;
                iters++;
            }
            pbits.base.release();
            local([&] {
                G = prim::exclusive_scan<u32>(Sg, ByteIn{hflag.p}, ex.p, false, "suffix_heads");
                prim::for_each(Sg, GroupStartsFn{hflag.p, ex.p, Sg, gstart.p}, "suffix_gstart");
                prim::for_each(Sg, DenseGidFn{hflag.p, ex.p, gid.p}, "suffix_gid");
                if (carry) {                     // from here on perm[] holds positions (offsets, owner in pown[]) again; the arrival indices stay for the fold's records
                    DBuf<u32> pp(Sg);
                    pown.alloc(Sg);
                    prim::for_each(Sg, GatherU32Fn{perm.p, pos_arr.p, pp.p}, "suffix_gid");
                    prim::for_each(Sg, ArrivalOwnerFn{perm.p, abounds.p, N, pown.p}, "suffix_gid");
                    perm_ai = std::move(perm);
                    perm = std::move(pp);
                    pos_arr.release();
                }
            });
            L.info.sort_iters = iters;
        };
        if (!C) sort_local();
        else if (sharded) sort_sharded();
        else { try { sort_local(); } catch (const prim::Error &e) { C->fail(e); } }

        // ---- a7: equal-suffix groups -> pre-BWT, ranks -----------------------
        const u32 bwt_code = sigma + 1, hocc_code = sigma + 2, sigma3 = sigma + 3;
        DBuf<u32> grank, pidx, gmin, gmax;
        DBuf<idx_t> gacc;
        DBuf<u8> gfull, gflag;
        DBuf<u32> repq, pslot;                   // pslot[k] = my group holding phrase k's whole-phrase suffix (all ones: not mine)
        DBuf<u32> gphr;                          // ... or, single-GPU rounds, gphr[g] = the whole phrase of group g (see GroupPhraseValFn)
        const bool fused_vals = !C && fused_ph_slot && fused_slot_val;
        u64 M, P0;
        {
            StageTimer st(&tm.dict_groups, "dict_groups");
            u64 Ml = 0, P0l = 0;
            // (dictionary sharded by owner: what the fold reads per member comes from the owners of the members' positions, by slot)
            DBuf<SufRecT<8>> recs;
            if (sharded && !carry) {
                DBuf<u64> rq;
                DBuf<SufRecT<8>> back;
                u64 nrq = 0;
                if (!C->pending) { try { rq.alloc(Sg); prim::for_each(Sg, RecRequestFn{perm.p, rq.p}, "suffix_records"); nrq = Sg; } catch (const prim::Error &e) { C->fail(e); nrq = 0; } }
                const RecCompute rcl{dict_sym.p, dict_phr.p, ph_off, ph_freq, ph_lastT, sigma + 1, (u32)d0};
                owner_round_trip<SufRecT<8>>(*C, dsb.p, rq, nrq, back, [&](const u64 *r, u64 nr, SufRecT<8> *out) {
                    DBuf<SufRecT<8>> mine(Sl);
                    prim::for_each(Sl, RecLocalFn{rcl, mine.p}, "suffix_records");
                    prim::for_each(nr, RecOwnerFn{r, mine.p, s0, out}, "suffix_records.answers");
                }, "dist.rec_owner_sort", "group.records");
                if (!C->pending) { try { recs.alloc(Sg); prim::for_each(nrq, RecAnswerFn{rq.p, back.p, recs.p}, "suffix_records"); } catch (const prim::Error &e) { C->fail(e); } }
            }
            auto groups_local = [&] {
            grank.alloc(G + 1); pidx.alloc(G + 1); gmin.alloc(G); gmax.alloc(G); gacc.alloc(G); gfull.alloc(G); gflag.alloc(G);
            if (fused_vals || carry) gphr.alloc(G); else pslot.alloc(D);      // (carry: gphr[g] = slot of the group's whole-phrase member)
            if (C && !carry) pslot.fill_ff();    // (sharded: phrases whose whole-phrase suffix sorted elsewhere keep the mark)
            {
                static const int fly_min = prim::test_env("GRLBWT_DIST_REC_FLY_MIN") ? atoi(prim::test_env("GRLBWT_DIST_REC_FLY_MIN")) : 8;
                const bool fly = C && C->size >= fly_min;
                DBuf<SufRec> rec;
                DBuf<u32> coff(G + 1);
                if (carry) {
                    prim::for_each(G, GroupAccumSmallFn<RecWire>{perm_ai.p, gstart.p, RecWire{rec_arr.p}, nullptr, bwt_code,
                                                                 gmin.p, gmax.p, gacc.p, gfull.p, gflag.p, nullptr, gphr.p, true}, "group_accum");
                } else if (sharded) {
                    prim::for_each(G, GroupAccumSmallFn<RecSlot>{perm.p, gstart.p, RecSlot{recs.p}, nullptr, bwt_code,
                                                                 gmin.p, gmax.p, gacc.p, gfull.p, gflag.p, pslot.p, gphr.p}, "group_accum");
                } else if (!fly) {
                    rec.alloc(S);
                    prim::for_each(S, SuffixRecFn{dict_sym.p, dict_phr.p, ph_off, ph_freq, ph_lastT, bwt_code, rec.p}, "suffix_records");
                    prim::for_each(G, GroupAccumSmallFn<RecArray>{perm.p, gstart.p, RecArray{rec.p}, dict_phr.p, bwt_code,
                                                                  gmin.p, gmax.p, gacc.p, gfull.p, gflag.p, pslot.p, gphr.p}, "group_accum");
                } else {
                    const RecCompute rc{dict_sym.p, dict_phr.p, ph_off, ph_freq, ph_lastT, bwt_code};
                    prim::for_each(G, GroupAccumSmallFn<RecCompute>{perm.p, gstart.p, rc, dict_phr.p, bwt_code,
                                                                    gmin.p, gmax.p, gacc.p, gfull.p, gflag.p, pslot.p, gphr.p}, "group_accum");
                }
                const u64 NC = prim::exclusive_scan<u32>(G, GroupChunksIn{gstart.p}, coff.p, true, "group_accum_large");
                if (carry) {
                    prim::for_each(NC, GroupAccumLargeFn<RecWire>{perm_ai.p, coff.p, G, gstart.p, RecWire{rec_arr.p}, nullptr, bwt_code,
                                                                  gmin.p, gmax.p, gacc.p, gfull.p, nullptr, gphr.p, true}, "group_accum_large");
                    prim::for_each(G, GroupDecideFn<RecWire>{perm_ai.p, gstart.p, RecWire{rec_arr.p}, gmin.p, gmax.p, gfull.p, gflag.p}, "group_decide");
                } else if (sharded) {
                    prim::for_each(NC, GroupAccumLargeFn<RecSlot>{perm.p, coff.p, G, gstart.p, RecSlot{recs.p}, nullptr, bwt_code,
                                                                  gmin.p, gmax.p, gacc.p, gfull.p, pslot.p, gphr.p}, "group_accum_large");
                    prim::for_each(G, GroupDecideFn<RecSlot>{perm.p, gstart.p, RecSlot{recs.p}, gmin.p, gmax.p, gfull.p, gflag.p}, "group_decide");
                } else
                if (!fly) prim::for_each(NC, GroupAccumLargeFn<RecArray>{perm.p, coff.p, G, gstart.p, RecArray{rec.p}, dict_phr.p, bwt_code,
                                                                         gmin.p, gmax.p, gacc.p, gfull.p, pslot.p, gphr.p}, "group_accum_large");
                else prim::for_each(NC, GroupAccumLargeFn<RecCompute>{perm.p, coff.p, G, gstart.p, RecCompute{dict_sym.p, dict_phr.p, ph_off, ph_freq, ph_lastT, bwt_code},
                                                                      dict_phr.p, bwt_code, gmin.p, gmax.p, gacc.p, gfull.p, pslot.p, gphr.p}, "group_accum_large");
            }
            if (!sharded) prim::for_each(G, GroupDecideFn<RecCompute>{perm.p, gstart.p, RecCompute{dict_sym.p, dict_phr.p, ph_off, ph_freq, ph_lastT, bwt_code},
                                                                       gmin.p, gmax.p, gfull.p, gflag.p}, "group_decide");
            {   // ranks of the ranked groups and pre-BWT index of the valid ones: one scan of pairs (one host synchronisation)
                const prim::Pair<u32, u32> tot = prim::exclusive_scan_emit<prim::Pair<u32, u32>>(G, RankedValidIn{gflag.p}, SplitPairEmitFn{grank.p, pidx.p}, "group_ranks");
                Ml = tot.a; P0l = tot.b;
            }
            };
            if (!C) groups_local();
            else if (!C->pending) { try { groups_local(); } catch (const prim::Error &e) { C->fail(e); Ml = 0; P0l = 0; } }
            u64 Moff = 0, P0off = 0;
            M = Ml; P0 = P0l;
            std::vector<u64> bbM, bbP;
            if (C) {
                std::vector<u64> cnts = C->allgather_u64({Ml, P0l});
                bbM.assign(C->size + 1, 0); bbP.assign(C->size + 1, 0);
                for (int g = 0; g < C->size; g++) { bbM[g + 1] = bbM[g] + cnts[2 * g]; bbP[g + 1] = bbP[g] + cnts[2 * g + 1]; }
                Moff = bbM[C->rank]; P0off = bbP[C->rank];
                M = bbM[C->size]; P0 = bbP[C->size];
            }
            if ((u64)sigma3 + M + 8 >= (1ull << 30)) throw prim::Error(-75, "alphabet of the next level >= 2^30");
            L.M = (u32)M;
            L.has_hocc.alloc(Ml); repq.alloc(Ml);
            DBuf<u32> psym0(P0l);
            DBuf<idx_t> plen0(P0l);
            DBuf<u32> u_to_p0(Ml), pu0(P0l);
            // Collection-level mode: the pre-BWT stays where its groups were sorted (round 5).  My key range's groups are a
            // contiguous piece of the sorted order, so what they emit is a contiguous piece of the level's pre-BWT: the induction
            // takes exactly these pieces as the ranks' output pieces (dist_induce_level) and nobody needs anybody else's runs.
            // (Rounds 1-4 all-gathered the emitted runs, their metasymbol counts and the metasymbol -> run map to every rank --
            // 8.9 GB sent per rank of the 10 GB collection at N = 2, 17.8 GB received at any N -- and every rank merged and
            // scanned the WHOLE pre-BWT; GRLBWT_DIST_REPLICATED_PREBWT=1 keeps that form.)  Runs are merged inside a piece
            // only: a run cut by a piece boundary stays two runs, which describe the same symbols.
            static const bool replicated_pre = prim::test_env("GRLBWT_DIST_REPLICATED_PREBWT") != nullptr || prim::test_env("GRLBWT_DIST_REPLICATED_INDUCTION") != nullptr;      // (the replicated induction wants the whole pre-BWT)
            const bool pre_local = C && !replicated_pre;
            prim::for_each(G, GroupEmitFn{gflag.p, grank.p, pidx.p, gmin.p, gacc.p, gstart.p, carry ? nullptr : perm.p, bwt_code, hocc_code, pre_local ? 0u : (u32)Moff,
                                          pre_local ? 0u : (u32)P0off, psym0.p, plen0.p, L.has_hocc.p, repq.p, u_to_p0.p, pu0.p}, "prebwt_emit");
            if (C) {                             // every rank's (Ml, P0l) is known: one exchange per array
                if (!pre_local) {
                    psym0 = C->named("prebwt.sym").template allgather_v<u32>(psym0.p, P0l, bbP, true);
                    plen0 = C->named("prebwt.len").template allgather_v<idx_t>(plen0.p, P0l, bbP, true);
                    pu0 = C->named("prebwt.pu").template allgather_v<u32>(pu0.p, P0l, bbP, true);
                }
                L.has_hocc = C->named("grammar.has_hocc").template allgather_v<u8>(L.has_hocc.p, Ml, bbM, true);
                if (!pre_local) u_to_p0 = C->named("prebwt.u_to_p").template allgather_v<u32>(u_to_p0.p, Ml, bbM, true);
            }
            const u64 P0e = pre_local ? P0l : P0, Me = pre_local ? Ml : M;      // what this rank's pre-BWT arrays describe
            L.pre_local = pre_local; L.u0 = pre_local ? (u32)Moff : 0u; L.Ml = (u32)Me;
            DBuf<u32> merged(P0e);
            L.prebwt = merge_runs(psym0.p, plen0.p, P0e, merged.p);
            L.prebwt.pos.release();
            L.u_to_p.alloc(Me);
            prim::for_each(Me, ComposeMapFn{u_to_p0.p, merged.p, L.u_to_p.p}, "prebwt_map");
            L.p_to_u.alloc(L.prebwt.R);
            prim::for_each(P0e, PreToMetaFn{pu0.p, merged.p, L.p_to_u.p}, "prebwt_map");
            // ---- a8: grammar ------------------------------------------------
            L.g0.alloc(M); L.g1.alloc(M);
            u32 MD = sigma3 + (u32)M + 1;
            {
                DBuf<u32> ginfo(G);
                prim::for_each(G, PackGroupInfoFn{grank.p, gflag.p, (u32)Moff, ginfo.p}, "grammar_ginfo");
                static const bool replicated_grammar = prim::test_env("GRLBWT_DIST_REPLICATED_GRAMMAR") != nullptr;
                if (C && sbase && (!replicated_grammar || sharded)) {      // (a dictionary sharded by owner has no other form)
                    // Sharded by the owner of the dictionary position (round 5): I hold dm[] of MY part of the dictionary only.  The marks
                    // of my groups go to the owners of their positions, the walks of my metasymbols are done by the owners of their
                    // representatives and the answers come back.  (Rounds 1-4: dm[] of the WHOLE dictionary on every rank, every rank's
                    // marks all-gathered to everybody and applied by everybody -- at every N: 24 + 8 ms and 10 GB per rank at N = 8 of the
                    // 10 GB collection, 4.8 GB of pairs sent per rank.)
                    const int N = C->size, me = C->rank;
                    const u64 s0 = carry ? 0 : (*sbase)[me], Sme = (*sbase)[me + 1] - (*sbase)[me] - test_dict_part_pad();      // (carry: positions are offsets in the owner's part)
                    DBuf<u64> dm;
                    // records (position << 32 | payload) -> sorted by the owner of the position, counts per owner
                    auto by_owner = [&](DBuf<u64> &rec, u64 n, std::vector<u64> &cnt, const char *name, DBuf<u32> *own_of) {
                        bucket_by_owner(*C, dsb.p, rec, n, cnt, name, own_of);
                    };
                    std::vector<u64> mcnt(N, 0), wcnt(N, 0), mrc(N, 0), wrc(N, 0);
                    DBuf<u64> mp, wq;
                    try {
                        DBuf<u32> mown, wown;
                        dm.alloc(Sme);
                        prim::for_each(Sme, DictMetaInitFn{dict_sym.p, dict_phr.p, ph_off, ph_lastT, dm.p, sharded ? 0 : s0}, "grammar_init");      // (sharded dictionary: the arrays are my part already)
                        DBuf<u32> mex(Sg + 1);
                        const u64 nm = prim::exclusive_scan<u32>(Sg, MarkedIn{gid.p, ginfo.p}, mex.p, false, "dist.mark_scan");
                        mp.alloc(nm);
                        if (carry) { mown.alloc(nm); wown.alloc(Ml); }
                        prim::for_each(Sg, MetaPairFn{perm.p, gid.p, ginfo.p, mex.p, sigma3, mp.p, pown.p, mown.p}, "dist.mark_pairs");
                        by_owner(mp, nm, mcnt, "dist.mark_owner_sort", carry ? &mown : nullptr);
                        wq.alloc(Ml);
                        prim::for_each(Ml, WalkRequestFn{repq.p, wq.p, perm.p, pown.p, wown.p}, "dist.walk_requests");
                        by_owner(wq, Ml, wcnt, "dist.walk_owner_sort", carry ? &wown : nullptr);
                    } catch (const prim::Error &e) { C->fail(e); std::fill(mcnt.begin(), mcnt.end(), 0); std::fill(wcnt.begin(), wcnt.end(), 0); }
                    std::vector<u64> both(mcnt);
                    both.insert(both.end(), wcnt.begin(), wcnt.end());
                    std::vector<u64> mat = C->allgather_u64(both);           // (raises on every rank if one of them failed above)
                    u64 nmr = 0, nwr = 0, maxm = 0, maxw = 0;
                    for (int g = 0; g < N; g++) {
                        mrc[g] = mat[(u64)g * 2 * N + me]; wrc[g] = mat[(u64)g * 2 * N + N + me];
                        nmr += mrc[g]; nwr += wrc[g];
                        for (int d = 0; d < N; d++) { maxm = std::max(maxm, mat[(u64)g * 2 * N + d]); maxw = std::max(maxw, mat[(u64)g * 2 * N + N + d]); }
                    }
                    DBuf<u64> mine, wmine, wans, wback;
                    try { mine.alloc(nmr); wmine.alloc(nwr); wans.alloc(nwr); wback.alloc(Ml); } catch (const prim::Error &e) { C->fail(e); }
                    C->allgather_u64({});                                    // (the bulk exchanges below have no way back)
                    C->named("grammar.mark_pairs").alltoall(mp.p, mcnt, mine.p, mrc, 8, maxm);
                    C->named("grammar.walk_requests").alltoall(wq.p, wcnt, wmine.p, wrc, 8, maxw);
                    mp.release();
                    try {
                        prim::for_each(nmr, ApplyMetaPairsFn{mine.p, dm.p, s0}, "grammar_marks");
                        mine.release();
                        DBuf<u64> stops;             // (very long phrases only: the walks jump to their stops)
                        if (maxlen >= 4096 || prim::test_env("GRLBWT_GRAMMAR_JUMP")) {
                            stops.alloc((Sme + 63) / 64 + 1);
                            prim::for_each((Sme + 63) / 64, DmStopBitsFn{dm.p, Sme, stops.p}, "grammar_marks");
                        }
                        prim::for_each(nwr, GrammarFn{nullptr, dm.p, MD, nullptr, nullptr, stops.p, Sme, wmine.p, s0, wans.p}, "grammar");
                    } catch (const prim::Error &e) { C->fail(e); }
                    C->allgather_u64({});
                    C->named("grammar.walk_answers").alltoall(wans.p, wrc, wback.p, wcnt, 8, maxw);
                    DBuf<u32> g0l(Ml), g1l(Ml);
                    prim::for_each(Ml, WalkAnswerFn{wq.p, wback.p, g0l.p, g1l.p}, "grammar");
                    C->named("grammar.g0").template allgather_v<u32>(g0l.p, Ml, bbM, true, L.g0.p);
                    C->named("grammar.g1").template allgather_v<u32>(g1l.p, Ml, bbM, true, L.g1.p);
                } else {
                DBuf<u64> dm(S);
                prim::for_each(S, DictMetaInitFn{dict_sym.p, dict_phr.p, ph_off, ph_lastT, dm.p}, "grammar_init");
                if (!C) prim::for_each(Sg, MetaPosFn{perm.p, gid.p, ginfo.p, sigma3, dm.p}, "grammar_marks");
                else {                           // the marked positions of every rank's groups, as (position, metasymbol) pairs
                    DBuf<u32> mex(Sg + 1);
                    const u64 nm = prim::exclusive_scan<u32>(Sg, MarkedIn{gid.p, ginfo.p}, mex.p, false, "dist.mark_scan");
                    DBuf<u64> mp(nm);
                    prim::for_each(Sg, MetaPairFn{perm.p, gid.p, ginfo.p, mex.p, sigma3, mp.p}, "dist.mark_pairs");
                    std::vector<u64> bb;
                    DBuf<u64> all = C->named("grammar.mark_pairs").template allgather_v<u64>(mp.p, nm, bb);
                    prim::for_each(bb[C->size], ApplyMetaPairsFn{all.p, dm.p}, "grammar_marks");
                }
                DBuf<u64> stops;                 // (very long phrases only: the walks jump to their stops)
                if (maxlen >= 4096 || prim::test_env("GRLBWT_GRAMMAR_JUMP")) {
                    stops.alloc((S + 63) / 64 + 1);
                    prim::for_each((S + 63) / 64, DmStopBitsFn{dm.p, S, stops.p}, "grammar_marks");
                }
                if (!C) prim::for_each(M, GrammarFn{repq.p, dm.p, MD, L.g0.p, L.g1.p, stops.p, S}, "grammar");
                else {                           // every rank walks for its own metasymbols; the cells are all-gathered
                    DBuf<u32> g0l(Ml), g1l(Ml);
                    prim::for_each(Ml, GrammarFn{repq.p, dm.p, MD, g0l.p, g1l.p, stops.p, S}, "grammar");
                    C->named("grammar.g0").template allgather_v<u32>(g0l.p, Ml, bbM, true, L.g0.p);
                    C->named("grammar.g1").template allgather_v<u32>(g1l.p, Ml, bbM, true, L.g1.p);
                }
                }
            }
            // ---- a9: metasymbol of every phrase --------------------------------
            if (fused_vals) {
                const u64 Dt = D - pDs;                  // the table phrases: (slot, flags) packed for one gather
                DBuf<u64> pinfo(Dt);
                prim::for_each(Dt, PackPhraseInfoFn{fused_ph_slot + pDs, ph_freq + pDs, ph_lastT + pDs, pinfo.p}, "phrase_values");
                prim::for_each(G, GroupPhraseValFn{gfull.p, gphr.p, grank.p, pinfo.p, fused_slot_val, pDs, pslot0}, "slot_values");
            } else phrase_val.alloc(sharded ? Dl : D);
            if (fused_vals) {}
            else if (!C) prim::for_each(D, PhraseValFn{pslot.p, ph_freq, ph_lastT, grank.p, phrase_val.p}, "phrase_values");
            else if (carry) {
                // the groups that hold a whole phrase know the POSITION of that member: (position, metasymbol) pairs go to the owner of the
                // position, which knows the phrase that starts there -- no array over all D phrases on any rank
                const int N = C->size, me = C->rank;
                std::vector<u64> scnt(N, 0), rcnt(N, 0);
                DBuf<u64> fp;
                if (!C->pending) { try {
                    DBuf<u32> gex(G + 1);
                    const u64 nf = prim::exclusive_scan<u32>(G, ByteIn{gfull.p}, gex.p, false, "dist.full_scan");
                    fp.alloc(nf);
                    DBuf<u32> fown(nf);
                    prim::for_each(G, GroupPosPairFn{gfull.p, gphr.p, grank.p, (u32)Moff, gex.p, fp.p, perm.p, pown.p, fown.p}, "dist.full_pairs");
                    bucket_by_owner(*C, dsb.p, fp, nf, scnt, "dist.full_pairs", &fown);
                } catch (const prim::Error &e) { C->fail(e); std::fill(scnt.begin(), scnt.end(), 0); } }
                std::vector<u64> mat = C->allgather_u64(scnt);
                u64 got = 0, maxb = 0;
                for (int g = 0; g < N; g++) { rcnt[g] = mat[(u64)g * N + me]; got += rcnt[g]; for (int d = 0; d < N; d++) maxb = std::max(maxb, mat[(u64)g * N + d]); }
                // (checked for EVERY rank's column on every rank -- the matrix and the owners' phrase ranges are known to all: a rank that
                // threw alone here would leave the others waiting in the exchange below)
                for (int d = 0; d < N; d++) {
                    u64 col = 0;
                    for (int g = 0; g < N; g++) col += mat[(u64)g * N + d];
                    if (col != (*dbase)[d + 1] - (*dbase)[d]) throw prim::Error(-71, "dist dictionary: whole-phrase suffix count does not match the phrase count");
                }
                DBuf<u64> mine(got);
                DBuf<u32> phrase_rank(Dl);
                C->named("phrase.rank_pairs").alltoall(fp.p, scnt, mine.p, rcnt, 8, maxb);
                prim::for_each(got, ApplyPosPairsFn{mine.p, dict_phr.p, 0, phrase_rank.p}, "dist.apply_phrase_ranks");      // (offsets in my part)
                prim::for_each(Dl, PhraseValDistFn{phrase_rank.p, ph_freq, ph_lastT, phrase_val.p}, "phrase_values");
            } else {                             // a whole-phrase suffix sits on the rank that owns its key: (phrase, metasymbol) pairs
                DBuf<u32> phrase_rank(sharded ? Dl : D), fex(D + 1);
                const u64 nf = prim::exclusive_scan<u32>(D, OwnPhraseIn{pslot.p}, fex.p, true, "dist.full_scan");
                DBuf<u64> fp(nf);
                prim::for_each(D, OwnPhrasePairFn{pslot.p, fex.p, grank.p, (u32)Moff, fp.p}, "dist.full_pairs");
                if (dbase) {
                    // The value of a phrase is wanted by ONE rank: the one that merged it and answers its senders (dist_round_t).  The
                    // pairs are in phrase order and the owners' phrases are contiguous ranges, so every owner's pairs are one block:
                    // an all-to-all of 8 bytes per phrase in all, where rounds 1-4 all-gathered all D pairs to every rank.
                    const int N = C->size, me = C->rank;
                    std::vector<u64> scnt(N, 0), rcnt(N, 0);
                    u64 prev = 0;
                    for (int g = 0; g < N; g++) {
                        const u64 hi = (*dbase)[g + 1] ? (u64)fex.get((*dbase)[g + 1]) : 0;
                        scnt[g] = hi - prev; prev = hi;
                    }
                    std::vector<u64> mat = C->allgather_u64(scnt);
                    u64 got = 0, maxb = 0;
                    for (int g = 0; g < N; g++) { rcnt[g] = mat[(u64)g * N + me]; got += rcnt[g]; for (int d = 0; d < N; d++) maxb = std::max(maxb, mat[(u64)g * N + d]); }
                    for (int d = 0; d < N; d++) {      // (every rank's column on every rank: see above)
                        u64 col = 0;
                        for (int g = 0; g < N; g++) col += mat[(u64)g * N + d];
                        if (col != (*dbase)[d + 1] - (*dbase)[d]) throw prim::Error(-71, "dist dictionary: whole-phrase suffix count does not match the phrase count");
                    }
                    DBuf<u64> mine(got);
                    C->named("phrase.rank_pairs").alltoall(fp.p, scnt, mine.p, rcnt, 8, maxb);
                    if (sharded) {               // (my phrases' arrays are local: ranks, frequencies and values by local phrase number)
                        prim::for_each(got, ApplyPairsFn{mine.p, phrase_rank.p, d0}, "dist.apply_phrase_ranks");
                        prim::for_each(got, PhraseValDistFn{phrase_rank.p, ph_freq, ph_lastT, phrase_val.p}, "phrase_values");
                    } else {
                    prim::for_each(got, ApplyPairsFn{mine.p, phrase_rank.p}, "dist.apply_phrase_ranks");
                    // (only my own range of phrase_val is filled and read)
                    prim::for_each(got, PhraseValDistFn{phrase_rank.p + (*dbase)[me], ph_freq + (*dbase)[me], ph_lastT + (*dbase)[me], phrase_val.p + (*dbase)[me]}, "phrase_values");
                    }
                } else {
                std::vector<u64> bb;
                DBuf<u64> allf = C->named("phrase.rank_pairs").template allgather_v<u64>(fp.p, nf, bb);
                if (bb[C->size] != D) throw prim::Error(-71, "dist dictionary: whole-phrase suffix count does not match the phrase count");
                prim::for_each(D, ApplyPairsFn{allf.p, phrase_rank.p}, "dist.apply_phrase_ranks");
                prim::for_each(D, PhraseValDistFn{phrase_rank.p, ph_freq, ph_lastT, phrase_val.p}, "phrase_values");
                }
            }
        }
        L.info.M = M;
    }

    // a10: the local parse: slot id of every occurrence -> (rank<<2 | rep<<1 | T) of its phrase
    // (slot_val given: the dictionary stage has already put every phrase's value at its slot)
    // (rank_only: the slots of the record phrases hold rank << 2, the flag bits are P.ph_vflag's -- dict_stage with fused values)
    void emit_local(LocalParse &P, const u32 *val_of_local_phrase, DBuf<u32> *slot_val_filled = nullptr, bool rank_only = false) {
        StageTimer st(&tm.emit, "emit");
        DBuf<u32> own;
        if (!slot_val_filled) {
            own.alloc(P.cap);
            prim::for_each(P.D, ScatterValFn{P.ph_slot.p, val_of_local_phrase, own.p}, "slot_values");
        }
        const u32 *sv = slot_val_filled ? slot_val_filled->p : own.p;
        if (P.Ds || P.part_bits) {
            // partitioned naming: the record at sorted position i takes the value of its phrase (its partition's phrases are
            // neighbours in the value array), the values go back to text order through the sort's passes in reverse, and the
            // occurrences that went through the table read theirs from their slot
            DBuf<u32> va(P.n_occ), vb(P.n_occ), vc(P.n_occ);
            prim::for_each(((u64)1 << P.part_bits) * 64, PartValFn{P.pstart.p, P.pbase.p, P.lid.p, sv, P.slot0, va.p, rank_only ? P.ph_vflag.p : nullptr}, "emit_part.values");
            P.pstart.release();
            P.psort.backward(va.p, vb.p, vc.p, "emit_part.back");
            prim::for_each(P.n_occ, PartCombineFn{sv, vc.p, P.next_text.p}, "emit_parse");
            P.psort.release(); P.lid.release(); P.pbase.release(); P.ph_key.release(); P.ph_vflag.release();
        } else prim::for_each((P.n_occ + 3) / 4, MapFn{sv, P.next_text.p, P.n_occ}, "emit_parse");
    }

    void finish_round(LocalParse &P, LevelData &L, u64 n_strings_for_termination, u64 n_occ_for_termination) {
        if (keep_texts) {
            DBuf<u32> cp(P.n_occ);
            prim::d2d(cp.p, P.next_text.p, P.n_occ * sizeof(u32));
            prim::sync();
            kept_texts.push_back(std::move(cp));
        }
        cur_text = std::move(P.next_text);
        cur_n = P.n_occ;
        cur_sigma = L.M;
        levels.push_back(std::move(L));
        if (n_occ_for_termination == n_strings_for_termination) parse_done = true;      // exact_par_phase.cpp:496
    }

    template <class cell_t, bool FIRST>
    void par_round_t(const cell_t *t, u64 n, u32 sigma, cell_t sep) {
        CellOps<cell_t, FIRST> ops{sep};
        prim::rt().tag = (int)levels.size();
        prim::rt().phase = 'p';
        LevelData L;
        L.sigma = sigma;
        L.info.n_in = n;
        L.info.sigma = sigma;
        LocalParse P;
        hash_local<cell_t, FIRST>(t, n, ops, P, L, true);
        DBuf<u32> phrase_val, slot_val(P.cap);
        dict_stage<cell_t, FIRST>(nullptr, t, ops, P.D, P.S, P.maxlen, P.ph_pos.p, P.ph_freq.p, P.ph_off.p, P.ph_lastT.p, sigma, L, phrase_val,
                                  P.ph_slot.p, slot_val.p, P.ph_key.p, P.Ds, P.rec_b, P.slot0);
        emit_local(P, nullptr, &slot_val, true);
        finish_round(P, L, stats.n_strings, P.n_occ);
    }

    // returns true when the parsing phase is complete
    bool parse_round() {
        if (parse_done) return true;
        if (levels.empty()) {
            switch (cell_bytes) {
                case 1: par_round_t<u8, true>((const u8 *)text0, n0, cur_sigma, (u8)stats.min_sym); break;
                case 2: par_round_t<u16, true>((const u16 *)text0, n0, cur_sigma, (u16)stats.min_sym); break;
                case 4: par_round_t<u32, true>((const u32 *)text0, n0, cur_sigma, (u32)stats.min_sym); break;
                default: par_round_t<u64, true>((const u64 *)text0, n0, cur_sigma, (u64)stats.min_sym); break;
            }
        } else {
            DBuf<u32> t = std::move(cur_text);
            par_round_t<u32, false>(t.p, cur_n, cur_sigma, 0u);
        }
        if (levels.size() > 64) throw prim::Error(-75, "too many parsing rounds");
        return parse_done;
    }
    int parse_phase() {                                          // exact_par_phase.cpp:338-366
        while (!parse_round()) {}
        return (int)levels.size();
    }

    // ---- a12 ---------------------------------------------------------------
    void first_bwt() {
        if (!parse_done) throw prim::Error(-22, "parse phase not finished");
        prim::rt().tag = (int)levels.size();
        prim::rt().phase = 'i';
        StageTimer st(&tm.ind_assemble, "ind_assemble");
        DBuf<u32> s(cur_n);
        DBuf<idx_t> l(cur_n);
        prim::for_each(cur_n, CellSymFn{cur_text.p, s.p, l.p}, "parse2bwt");
        bwt = merge_runs(s.p, l.p, cur_n);
        bwt_level = (int)levels.size();
        cur_text.release();
        linfo.assign(levels.size() + 1, LevelInfo());
        linfo[bwt_level].R = bwt.R;
        linfo[bwt_level].n = cur_n;
        if (keep_texts) { kept_bwts.clear(); kept_bwts.resize(levels.size() + 1); keep_bwt(bwt_level); }
    }
    RunLen run_len() const { return RunLen{bwt.len.p, bwt.pos.p}; }      // (one of the two is there)
    void need_len() {                          // the lengths as an array (inspection, the collection-level mode's edges)
        if (!bwt.len.p && bwt.pos.p) { bwt.len.alloc(bwt.R); prim::for_each(bwt.R, DiffFn{bwt.pos.p, bwt.len.p}, "merge_runs.len"); }
    }
    void keep_bwt(int lvl) {
        need_len();
        Runs c;
        c.sym.alloc(bwt.R); c.len.alloc(bwt.R); c.R = bwt.R;
        prim::d2d(c.sym.p, bwt.sym.p, bwt.R * sizeof(u32));
        prim::d2d(c.len.p, bwt.len.p, bwt.R * sizeof(idx_t));
        prim::sync();
        kept_bwts[lvl] = std::move(c);
    }

    // ---- a13-a15: BWT_r from BWT_{r+1} -------------------------------------
    // passes A+B (exact_ind_phase.cpp:42-109,143-258) over the runs this engine holds of BWT_{r+1}: chain walks through
    // the level's grammar, cells split by bucket (stable).  The cells stay in c_* (bucket-major), term[i] = rewritten symbol
    // of run i.  `maxrun` = longest run of BWT_{r+1} (of ALL shards in the collection-level mode: it fixes the cell layout).
    u64 expand_split(LevelData &L, DBuf<u32> &term, u64 maxrun, int &kb, int &lb) {
        const u32 sigma3 = L.sigma + 3, bwt_code = L.sigma + 1, take_code = bwt_code;
        const u64 R = bwt.R, M = L.M;
        DBuf<idx_t> eoff;
        DBuf<u32> ssym; DBuf<idx_t> slen;
        DBuf<u64> gp;                           // packed grammar cells (chain walks)
        u64 E = 0;
        DBuf<u32> skey;                         // bucket of every induced cell, bucket-major order
        DBuf<u64> spack;                        // (sym<<32 | len) of every induced cell, same order (packed path)
        DBuf<u64> sfused;                       // sym | len | bucket in one word per cell (fused path)
        DBuf<u32> sfused32;                     // the same word in 32 bits when the three fields fit (kb + lb + sbits <= 32 < 2^32: kb < 32)
        kb = (int)bitlen64(L.M > 0 ? L.M - 1 : 0);
        if (kb < 1) kb = 1;
        {
            const int bits = kb;
            {
                StageTimer st(&tm.ind_expand, "ind_expand");
                gp.alloc(M);
                prim::for_each(M, PackGrammarFn{L.g0.p, L.g1.p, L.has_hocc.p, gp.p}, "induce_pack_grammar");
            }
            const int sbits = (int)bitlen64((u64)sigma3);
            prim::XsPlan plan;
            lb = (int)bitlen64(maxrun);
            if (lb < 1) lb = 1;
            // Whenever bucket, run length and symbol fit 64 bits together (always at DNA scales), the cell IS the sort
            // key: the split moves 8 bytes per cell and pass instead of 12, and holds 16 instead of 24 bytes per cell.
            // (GRLBWT_CELL_LAYOUT=packed|separate: the tests take the wider layouts on inputs that would never need them)
            const char *force = getenv("GRLBWT_CELL_LAYOUT");
            const bool fused = kb + lb + sbits <= 64 && !force;
            const bool cell32 = fused && kb + lb + sbits <= 32 && kb < 32 && !getenv("GRLBWT_NO_CELL32");
            if (getenv("GRLBWT_TABLE_TRACE")) fprintf(stderr, "[grlbwt] induction level %d: %llu runs, longest %llu, cell bits: bucket %d + length %d + symbol %d\n",
                                                      prim::rt().tag, (unsigned long long)R, (unsigned long long)maxrun, kb, lb, sbits);
            // otherwise the payload (sym, len) rides through the split as one u64 whenever every run length fits 32 bits
            const bool packed = !fused && maxrun < 0xFFFFFFFFull && !(force && force[0] == 's');
            bool done = false;
            if (fused) {
                // chain expansion fused with the first pass of the bucket split (prim::expand_*): the cells are never
                // written in run order, there is no offset array and no scan over the runs
                const ChainGen gen{bwt.sym.p, run_len(), gp.p, sigma3, take_code, term.p, kb, lb};
                {
                    StageTimer st(&tm.ind_expand, "ind_expand");
                    E = prim::expand_count(R, gen, bits, plan, "induce");
                }
                // 4-byte cells when everything fits (level 0 of the 10 GB DNA build: 17 + 8 + 7 bits): every pass of the split, pass C
                // and -- in the collection-level mode -- the cell exchange move half the bytes.  (A function of the layout alone:
                // the ranks of a collection-level build agree on it.)
                if (plan.ok && cell32) {
                    DBuf<u32> ef(E), ef2(E);
                    StageTimer st(&tm.ind_sort, "ind_sort");
                    int res = prim::expand_sort<ChainGen, u32>(gen, plan, ef.p, ef2.p, "induce");
                    sfused32 = std::move(res ? ef2 : ef);
                    done = true;
                } else if (plan.ok) {
                    DBuf<u64> ef(E), ef2(E);
                    StageTimer st(&tm.ind_sort, "ind_sort");
                    int res = prim::expand_sort(gen, plan, ef.p, ef2.p, "induce");
                    sfused = std::move(res ? ef2 : ef);
                    done = true;
                }
            }
            plan.release();
            if (!done) {     // offsets of every run's cells (an item with more than 32 cells, or cells that do not fit one word)
                StageTimer st(&tm.ind_expand, "ind_expand");
                eoff.alloc(R + 1);
                prim::for_each(R, StoreFn<ChainCountFn>{ChainCountFn{bwt.sym.p, gp.p, sigma3}, eoff.p}, "induce_count");
                E = (u64)prim::exclusive_scan<idx_t>(R, IdxIn<idx_t>{eoff.p}, eoff.p, true, "induce_count_scan");
            }
            if (done) {
            } else if (fused) {
                DBuf<u64> ef(E), ef2(E);
                {
                    StageTimer st(&tm.ind_expand, "ind_expand");
                    prim::for_each(R, ChainExpandFn<CELLS_FUSED>{bwt.sym.p, run_len(), gp.p, eoff.p, sigma3, take_code,
                                                                 nullptr, nullptr, nullptr, nullptr, ef.p, term.p, kb, lb}, "induce_expand");
                }
                StageTimer st(&tm.ind_sort, "ind_sort");
                int res = prim::sort_keys<u64, 1>(ef.p, ef2.p, E, 0, bits, "induce_split");
                if (cell32) {                    // (the same form as the fused kernel's: the layout decides, not the path taken)
                    sfused32.alloc(E);
                    prim::for_each(E, NarrowCellFn{res ? ef2.p : ef.p, sfused32.p}, "induce_split");
                } else sfused = std::move(res ? ef2 : ef);
                prim::sync();
            } else if (packed) {
                DBuf<u32> ekey(E), ekey2(E);
                DBuf<u64> ep(E), ep2(E);
                {
                    StageTimer st(&tm.ind_expand, "ind_expand");
                    prim::for_each(R, ChainExpandFn<CELLS_PACKED>{bwt.sym.p, run_len(), gp.p, eoff.p, sigma3, take_code,
                                                                  ekey.p, nullptr, nullptr, nullptr, ep.p, term.p, 0, 0}, "induce_expand");
                }
                StageTimer st(&tm.ind_sort, "ind_sort");
                int res = prim::sort_pairs<u32, u64>(ekey.p, ep.p, ekey2.p, ep2.p, E, 0, bits, "induce_split");
                skey = std::move(res ? ekey2 : ekey);
                spack = std::move(res ? ep2 : ep);
                prim::sync();
            } else {
                DBuf<u32> ekey(E), ekey2(E);
                DBuf<u32> esym(E);
                DBuf<idx_t> eidx(E), eidx2(E), elen(E);
                {
                    StageTimer st(&tm.ind_expand, "ind_expand");
                    prim::for_each(R, ChainExpandFn<CELLS_SEPARATE>{bwt.sym.p, run_len(), gp.p, eoff.p, sigma3, take_code,
                                                                    ekey.p, eidx.p, esym.p, elen.p, nullptr, term.p, 0, 0}, "induce_expand");
                }
                StageTimer st(&tm.ind_sort, "ind_sort");
                ssym.alloc(E); slen.alloc(E);
                int res = prim::sort_pairs<u32, idx_t>(ekey.p, eidx.p, ekey2.p, eidx2.p, E, 0, bits, "induce_split");
                prim::for_each(E, GatherCellFn{res ? eidx2.p : eidx.p, esym.p, elen.p, ssym.p, slen.p}, "induce_gather");
                skey = std::move(res ? ekey2 : ekey);
                prim::sync();
            }
        }
        // the cells move to the engine: pass C drops them before its run merge (peak memory)
        c_skey = std::move(skey); c_ssym = std::move(ssym); c_slen = std::move(slen); c_spack = std::move(spack);
        c_sfused = std::move(sfused); c_sfused32 = std::move(sfused32); c_gp = std::move(gp);
        return E;
    }
    // the received blocks (rcnt[g] cells from rank g, back to back in `in`) merged by bucket; buckets [u0, u0 + M)
    template <class K>
    void merge_cell_blocks(DBuf<K> &in, const std::vector<u64> &rcnt, int kb, int lb, u32 u0, u64 M, u64 Er) {
        const int N = (int)rcnt.size();
        DBuf<idx_t> bstart1((u64)N * M), bend((u64)N * M), base(M + 1), off((u64)N * M);
        bstart1.zero(); bend.zero();
        u64 ro = 0;
        for (int g = 0; g < N; g++) {
            if (rcnt[g]) {
                const K *blk = in.p + ro;
                CellView cv{sizeof(K) == 8 ? (const u64 *)blk : nullptr, kb, lb, nullptr, nullptr, nullptr, nullptr, u0, sizeof(K) == 4 ? (const u32 *)blk : nullptr};
                prim::for_each(rcnt[g], BucketEdgesFn{cv, rcnt[g], bstart1.p + (u64)g * M, bend.p + (u64)g * M}, "dist.merge_cells");
            }
            ro += rcnt[g];
        }
        prim::exclusive_scan_nosync<idx_t>(M, BlockTotalIn{bstart1.p, bend.p, N, M}, base.p, true, "dist.merge_cells");
        prim::for_each(M, BlockOffsetsFn{bstart1.p, bend.p, base.p, N, M, off.p}, "dist.merge_cells");
        DBuf<K> out(Er);
        const K kmask = (K)((kb >= (int)(8 * sizeof(K))) ? ~K(0) : ((K(1) << kb) - 1));
        ro = 0;
        for (int g = 0; g < N; g++) {
            if (rcnt[g]) prim::for_each(rcnt[g], BlockPlaceFn<K>{in.p + ro, kmask, u0, off.p + (u64)g * M, out.p}, "dist.merge_cells.scatter");
            ro += rcnt[g];
        }
        in = std::move(out);
    }
    CellView cell_view(int kb, int lb, u32 u0 = 0) const { return CellView{c_sfused.p, kb, lb, c_skey.p, c_spack.p, c_ssym.p, c_slen.p, u0, c_sfused32.p}; }
    u64 level_maxrun() {
        StageTimer st(&tm.ind_expand, "ind_expand");
        return prim::reduce_max<u64>(bwt.R, RunLenIn64{run_len()}, "induce_maxrun");
    }
    // what pass C assembles: a contiguous piece of the level's pre-BWT, the metasymbols (buckets) whose HOCC runs lie in it
    // (indices relative to the piece), and the number of symbols the piece describes.  The whole level on one GPU.
    struct AsmIn { const u32 *psym; const idx_t *plen; u64 P; const u32 *u_to_p; const u32 *p_to_u; u64 M; u32 sigma; u64 n_out; };
    void assemble(const AsmIn &in, LevelInfo &I, const CellView &cells, u64 E, DBuf<u32> &term, int r) {
        StageTimer st(&tm.ind_assemble, "ind_assemble");
        DBuf<idx_t> Tpos;
        u64 Tsum;
        if (bwt.pos.p) { Tpos = std::move(bwt.pos); Tsum = bwt.n; }       // the run merge that produced BWT_{r+1} left its prefix behind
        else {
            Tpos.alloc(bwt.R + 1);
            Tsum = (u64)prim::exclusive_scan<idx_t>(bwt.R, IdxIn<idx_t>{bwt.len.p}, Tpos.p, true, "asm.Tpos");
        }
        assemble_t(in, I, cells, E, Tpos, Tsum, term, r);
    }
    void release_level(LevelData &L) {           // the level's grammar is no longer needed
        L.g0.release(); L.g1.release(); L.has_hocc.release(); L.u_to_p.release(); L.p_to_u.release(); L.prebwt.sym.release(); L.prebwt.len.release();
    }

    void induce_level() {
        if (bwt_level <= 0) throw prim::Error(-22, "no level left to induce");
        const int r = bwt_level - 1;
        prim::rt().tag = r;
        prim::rt().phase = 'i';
        LevelData &L = levels[r];
        const u64 R = bwt.R;
        LevelInfo &I = linfo[r];
        I.R_next = R; I.P = L.prebwt.R;
        DBuf<u32> term(R);
        int kb, lb;
        const u64 E = expand_split(L, term, level_maxrun(), kb, lb);
        I.E = E;
        const CellView cells = cell_view(kb, lb);
        if (prim::rt().profile) {              // SURVEY 8d's E'_r and E_r for the roofline accounting (bench.py)
            I.Esteps = E - prim::reduce_sum<u64>(R, TakeCountIn{bwt.sym.p, c_gp.p}, "stat.take_cells");
            I.Emerged = prim::reduce_sum<u64>(E, CellHeadIn{cells}, "stat.merged_cells");
        }
        assemble(AsmIn{L.prebwt.sym.p, L.prebwt.len.p, L.prebwt.R, L.u_to_p.p, L.p_to_u.p, L.M, L.sigma, L.info.n_in}, I, cells, E, term, r);
        bwt_level = r;
        if (r == 0 && bwt.len.p) bwt.pos.release();      // (no level below wants the prefix -- unless it stands for the lengths)
        I.R = bwt.R;
        I.n = L.info.n_in;
        if (keep_texts) keep_bwt(r);
        release_level(L);
    }
    // pass C (exact_ind_phase.cpp:287-361): BWT_r from the pre-BWT, the induced cells and the rewritten BWT_{r+1}
    void assemble_t(const AsmIn &L, LevelInfo &I, const CellView &cells, u64 E, DBuf<idx_t> &Tpos, u64 Tsum, DBuf<u32> &term, int r) {
        const u32 bwt_code = L.sigma + 1, hocc_code = L.sigma + 2, take_code = bwt_code;
        const u64 R = bwt.R, P = L.P, M = L.M;
        // ---- the maximal runs of the rewritten BWT_{r+1} and their starts over the T axis
        DBuf<u32> esym(R ? R : 1);
        DBuf<idx_t> epos(R + 1);
        const u64 Re = R ? (u64)prim::exclusive_scan_emit<idx_t>(R, TermHeadIn{term.p}, TermHeadEmitFn{term.p, Tpos.p, esym.p, epos.p}, "asm.truns") : 0;
        Tpos.release(); term.release();
        bwt.sym.release(); bwt.len.release();
        const u64 nwT = Tsum / 64 + 2;
        DBuf<RankCell> tstarts(nwT);
        tstarts.zero();
        if (Re) prim::for_each((Re + 15) / 16, BuildBitsFn{epos.p, Re, reinterpret_cast<u64 *>(tstarts.p), 2}, "asm.tbits");
        prim::exclusive_scan_emit_nosync<u64>(nwT, RankCellPopcIn{tstarts.p}, RankCellBaseEmitFn{tstarts.p}, "asm.tbits");
        // ---- the non-HOCC pre-BWT runs, compacted, and where they sit among the segments
        DBuf<idx_t> nhb(P + 1);
        const u64 NH = (u64)prim::exclusive_scan<idx_t>(P, NotCodeIn{L.psym, hocc_code}, nhb.p, true, "asm.nhb");
        const u64 G = NH + E;
        I.G = G;
        DBuf<idx_t> seg_pos(NH ? NH : 1), nh_len(NH ? NH : 1);
        DBuf<u32> nh_sym(NH ? NH : 1);
        {
            // where the cells of the buckets in front of a pre-BWT run end: a table over the metasymbols when the runs are
            // many (one pass over the cells), a binary search per run when they are few (level 0: 45 k runs, 2.6 G cells)
            DBuf<idx_t> first_cell;
            if (P * 32 > E && L.p_to_u) {
                first_cell.alloc(M + 1);
                DBuf<idx_t> bstart1(M), bend(M);
                bstart1.zero(); bend.zero();
                prim::for_each(E, BucketEdgesFn{cells, E, bstart1.p, bend.p}, "asm.first_cell");
                prim::exclusive_scan_nosync<idx_t>(M, BucketSizeIn{bstart1.p, bend.p}, first_cell.p, true, "asm.first_cell");
            }
            prim::for_each(P, PrePlaceFn{L.psym, L.plen, nhb.p, L.u_to_p, M, cells, E, L.p_to_u, first_cell.p, hocc_code,
                                         seg_pos.p, nh_sym.p, nh_len.p}, "asm.pre_place");
        }
        nhb.release();
        const u64 nwG = G / 64 + 2;
        DBuf<RankCell> kinds(nwG);
        kinds.zero();
        if (NH) prim::for_each((NH + 15) / 16, BuildBitsFn{seg_pos.p, NH, reinterpret_cast<u64 *>(kinds.p), 2}, "asm.kinds");
        prim::exclusive_scan_emit_nosync<u64>(nwG, RankCellPopcIn{kinds.p}, RankCellBaseEmitFn{kinds.p}, "asm.kinds");
        seg_pos.release();
        // ---- the stream merge: count (T positions, run heads), then emit
        const AsmSeg seg{kinds.p, nh_sym.p, nh_len.p, cells, take_code, tstarts.p, esym.p, epos.p};
        prim::SmPlan<idx_t> plan;
        auto check_totals = [&] {
            if (plan.take_total != Tsum) { plan.release(); throw prim::Error(-71, "induction: BWT_{r+1} consumption mismatch (level " + std::to_string(r) + ": " +
                                                                                           std::to_string(plan.take_total) + " vs " + std::to_string(Tsum) + ")"); }
            if (plan.len_total != L.n_out) { plan.release(); throw prim::Error(-71, "induction: BWT of level " + std::to_string(r) + " describes " + std::to_string(plan.len_total) +
                                                                                           " symbols, the level has " + std::to_string(L.n_out)); }
        };
        Runs out;
        u64 Ro = 0;
        // ONE WALK (round 6): the run heads are found and written by one pass over the segments, the run index at every tile's start
        // comes from a look-back across the tiles (prim::stream_merge_onepass) -- the count pass (28 of 125 ms of pass C at 10 GB)
        // re-derived exactly what the emit pass derives.  The runs are written into arrays sized for an upper bound (every segment
        // a run + every run of T a run) and the arrays give their tails back.  GRLBWT_ASM_TWO_PASS=1 (the tests) and a walk that
        // gave up take the two-pass form.
        // Where the segments are almost all plain cells (level 0 of a read collection: 2.6 M tiles of 1024 segments, 100+ tiles per
        // microsecond) the look-back costs what the count pass it saves costs -- 41 ms against 36.6 with tiles of 1024 segments, 35.1
        // with tiles of 2048 (GRLBWT_DEV_SM1_SPT=8) -- and needs 12 bytes per SEGMENT of upper-bound arrays (32 GB at level 0 of the
        // 10 GB build, where the runs take 20): the two-pass form stays.
        // (GRLBWT_ASM_ONE_WALK=1: the tests take the one-walk form at every level, plain or not)
        const bool two_pass = getenv("GRLBWT_ASM_TWO_PASS") != nullptr;
        const bool plain_too = getenv("GRLBWT_ASM_ONE_WALK") != nullptr;
        const bool mostly_plain = NH * 64 < G;
        bool done = false;
        if (!two_pass && (!mostly_plain || plain_too)) {
            const u64 cap = G + Re + 1;
            // queued segments: a TAKE that spans more than kSmInline further runs of T -- at most Re / kSmInline of them
            const u64 qcap = Re / prim::kSmInline + 1;
            out.sym.alloc(cap); out.pos.alloc(cap + 1);
            try { done = prim::stream_merge_onepass<AsmSeg, idx_t>(G, seg, plan, out.sym.p, out.pos.p, cap, qcap, "asm", mostly_plain); } catch (...) { plan.release(); throw; }
            if (done) {
                check_totals();
                Ro = plan.heads;
                out.sym.shrink(Ro); out.pos.shrink(Ro + 1);
            } else {
                out.sym.release(); out.pos.release();
                if (getenv("GRLBWT_TABLE_TRACE")) fprintf(stderr, "[grlbwt] level %d: the one-walk form of pass C gave up, taking count + emit\n", r);
            }
        }
        if (!done) {
            prim::stream_merge_count<AsmSeg, idx_t>(G, seg, plan, "asm", mostly_plain);      // (few pre-BWT runs among the segments: small tiles, see prim_hip.hpp)
            check_totals();
            Ro = plan.heads;
            out.sym.alloc(Ro); out.pos.alloc(Ro + 1);
            try { prim::stream_merge_emit<AsmSeg, idx_t>(seg, plan, out.sym.p, out.pos.p, "asm"); } catch (...) { plan.release(); throw; }
        }
        I.A = plan.atoms;
        plan.release();
        const idx_t total = (idx_t)L.n_out;
        prim::h2d(out.pos.p + Ro, &total, sizeof(idx_t));
        // everything but the runs can go before their lengths are taken (peak memory)
        kinds.release(); nh_sym.release(); nh_len.release(); tstarts.release(); esym.release(); epos.release();
        release_cells();
        out.R = Ro; out.n = L.n_out;            // (no length array: RunLen reads the prefix)
        bwt = std::move(out);
    }
    // the induced cells of the level being assembled (owned here so that pass C can drop them before the run merge)
    DBuf<u32> c_skey, c_ssym, c_sfused32; DBuf<idx_t> c_slen; DBuf<u64> c_spack, c_sfused, c_gp;
    void release_cells() { c_skey.release(); c_ssym.release(); c_slen.release(); c_spack.release(); c_sfused.release(); c_sfused32.release(); c_gp.release(); }

    void induce_phase() {                                        // exact_ind_phase.cpp:674-697
        first_bwt();
        while (bwt_level > 0) induce_level();
    }

    // ---- a16/a17: .rl_bwt image in HBM --------------------------------------
    void finish() {
        if (bwt_level != 0) throw prim::Error(-22, "induction not finished");
        prim::rt().tag = -1; prim::rt().phase = 0;
        StageTimer st(&tm.finish, "finish");
        u32 sb = (u32)stats.sb, fb = (u32)stats.fb;
        image_bytes = 16 + bwt.R * (u64)(sb + fb);
        image.alloc(image_bytes);
        u8 hdr[16] = {0};
        for (int i = 0; i < 8; i++) { hdr[i] = (u8)((u64)sb >> (8 * i)); hdr[8 + i] = (u8)((u64)fb >> (8 * i)); }
        prim::h2d(image.p, hdr, 16);
        // records of up to 8 bytes go through prim::pack_records (tiles assembled in LDS, 16-byte stores); wider ones one lane per run.
        // (Four runs per lane, their 4 x 5 bytes put together in registers, was measured at 15.5 ms against 9 for one lane per run on
        // the 1.66 G runs of the 10 GB image: the loads of a lane's four runs are what is 20 bytes apart then.)
        if (sb + fb <= 8 && fb < 8) prim::pack_records(bwt.R, RunRecordFn{bwt.sym.p, run_len(), sb}, sb + fb, image.p + 16, "pack_rl_bwt");
        else prim::for_each(bwt.R, PackRunsFn{bwt.sym.p, run_len(), sb, fb, image.p, 16u}, "pack_rl_bwt");
        image_runs = bwt.R;
        image_part_off = 0; image_part_bytes = image_bytes;
        // "results are complete when a call returns" (include/grlbwt_hip.h): the image pointer may be handed to another
        // stream (torch, a copy engine) right after the build, so the engine's stream is drained here
        prim::sync();
    }
    // =====================================================================
    // collection-level multi-GPU: this engine holds one record shard
    // =====================================================================
    u64 g_n_strings = 0, g_n_syms = 0;      // collection-wide totals (distributed mode)

    // collection_stats over all shards: separator, alphabet, header widths
    void dist_stats(const Comm &C) {
        std::vector<u64> mine(8 + 256, 0);
        mine[0] = stats.n_syms; mine[1] = stats.n_strings; mine[2] = stats.min_sym; mine[3] = stats.max_sym;
        mine[4] = sizeof(idx_t);                // the ranks exchange idx_t arrays: one width for the whole collection
        if (cell_bytes == 1) {
            u64 h[256];
            prim::byte_histogram((const u8 *)text0, n0, h);
            for (int i = 0; i < 256; i++) mine[8 + i] = h[i];
        }
        std::vector<u64> all = C.allgather_u64(mine);
        const u64 w = mine.size();
        u64 n = 0, ns = 0, mn = ~0ull, mx = 0;
        std::vector<u64> hist(256, 0);
        for (int g = 0; g < C.size; g++) {
            n += all[g * w]; ns += all[g * w + 1];
            if (all[g * w + 2] < mn) mn = all[g * w + 2];
            if (all[g * w + 3] > mx) mx = all[g * w + 3];
            for (int i = 0; i < 256; i++) hist[i] += all[g * w + 8 + i];
        }
        // (a context picks its index width from its OWN shard unless GRLBWT_FLAG_FORCE_IDX64 is set: shards on either side of
        // 2^32 - 256 cells, or shards below it of a collection above it, must be refused by every rank alike)
        for (int g = 0; g < C.size; g++)
            if (all[g * w + 4] != sizeof(idx_t))
                throw prim::Error(-22, "the ranks disagree on the index width: create every context with GRLBWT_FLAG_FORCE_IDX64 when the collection has >= 2^32 - 256 cells");
        // every shard ends with the separator == its own minimum; it must be the global minimum too
        // (decided from the gathered values so that every rank takes the same branch)
        for (int g = 0; g < C.size; g++)
            if (all[g * w + 2] != mn) throw prim::Error(-84, "Error: the file is ill formed");
        u64 F = n;
        if (cell_bytes == 1) { F = 0; for (int i = 0; i < 256; i++) if (hist[i] > F) F = hist[i]; }
        if (sizeof(idx_t) == 4 && n >= 0xFFFFFF00ull)
            throw prim::Error(-75, "collection too large for the 32-bit index build: create every context with GRLBWT_FLAG_FORCE_IDX64");
        g_n_strings = ns; g_n_syms = n;
        stats.max_sym = mx; stats.max_sym_freq = F;
        stats.sb = (bitlen64(mx + 4) + 7) / 8;
        stats.fb = (bitlen64(F) + 7) / 8;
        cur_sigma = (u32)(mx + 1);
    }

    template <class cell_t, bool FIRST>
    void dist_round_t(const Comm &C, const cell_t *t, u64 n, u32 sigma, cell_t sep) {
        CellOps<cell_t, FIRST> ops{sep};
        prim::rt().tag = (int)levels.size();
        prim::rt().phase = 'p';
        LevelData L;
        L.sigma = sigma;
        L.info.sigma = sigma;
        LocalParse P;
        const int N = C.size, me = C.rank;
        // ---- my distinct phrases go to the ranks that merge them (owner = hash of the content) ---------------------
        DBuf<u32> order;                         // my phrases in owner order
        std::vector<u64> pc(N), cc(N), rpc(N), rcc(N);      // phrases / cells I send to every owner, and receive from every shard
        DBuf<u32> rlen, rcells;
        DBuf<u32> rfreq;
        u64 Dr = 0, Sr = 0, occ_total = 0, n_total = 0, maxp = 0;     // maxp: largest phrase block of the exchange (all ranks agree)
        {
            StageTimer st(&tm.hash, "hash");
            DBuf<u32> slen, scells;
            DBuf<u32> sfreq;
            std::vector<u64> mine(2 * (u64)N + 2, 0);
            // (a failure only this shard can have -- phrase table overflow, a parse beyond the supported range -- travels with the
            // counts: Comm::fail)
            try {
                // (GRLBWT_TEST_FAIL_RANK=<rank>: the tests make one rank fail here)
                if (test_fail_rank("GRLBWT_TEST_FAIL_RANK", me)) throw prim::Error(-28, "phrase hash table overflow (injected by the test)");
                // (levels above 0: partitioned naming as on one GPU -- short phrases are records in P.ph_key, not text positions)
                hash_local<cell_t, FIRST>(t, n, ops, P, L, !prim::dev_env("GRLBWT_DIST_NO_PART"));
                if (P.n_occ >= 0xFFFFFFF0ull) throw prim::Error(-75, "a shard's parse has >= 2^32 phrases (its frequencies travel as u32): use more ranks");
                DBuf<u32> owner(P.D), owner2(P.D), idx(P.D), idx2(P.D), soff(P.D + 1);
                DBuf<u64> bound(2 * ((u64)N + 1));
                prim::for_each(P.D, PhraseOwnerFn<cell_t, FIRST>{t, ops, P.ph_pos.p, P.ph_len.p, (u32)N, owner.p, idx.p, P.ph_key.p, P.Ds, P.rec_b},
                               "dist.phrase_owner");
                int obits = (int)bitlen64((u64)N - 1);
                if (obits < 1) obits = 1;
                const int res = prim::sort_pairs<u32, u32>(owner.p, idx.p, owner2.p, idx2.p, P.D, 0, obits, "dist.owner_sort");
                order = std::move(res ? idx2 : idx);
                const u32 *okey = res ? owner2.p : owner.p;
                slen.alloc(P.D); sfreq.alloc(P.D);
                prim::for_each(P.D, SendPhraseFn{order.p, P.ph_len.p, P.ph_freq.p, slen.p, sfreq.p}, "dist.send_phrases");
                const u64 chk = prim::exclusive_scan<u32>(P.D, LenIn{slen.p}, soff.p, true, "dist.send_offsets");
                if (chk != P.S) throw prim::Error(-71, "dictionary exchange: cell count mismatch");
                scells.alloc(P.S);
                {
                    RankBits sbits;
                    build_rankbits32(sbits, soff.p, P.D, P.S + 1, "dist.send_cells");
                    prim::for_each((P.S + 15) / 16, SendCellsFn<cell_t, FIRST>{t, ops, order.p, P.ph_pos.p, soff.p, P.D, P.S, scells.p, sbits.words.p, sbits.base.p,
                                                                              P.ph_key.p, P.Ds, P.rec_b},
                                   "dist.send_cells");
                }
                prim::for_each((u64)N + 1, KeyBoundFn{okey, P.D, soff.p, bound.p}, "dist.owner_bounds");
                std::vector<u64> bh = bound.to_host(2 * ((u64)N + 1));
                for (int d = 0; d < N; d++) { pc[d] = bh[2 * (d + 1)] - bh[2 * d]; cc[d] = bh[2 * (d + 1) + 1] - bh[2 * d + 1]; mine[d] = pc[d]; mine[N + d] = cc[d]; }
                mine[2 * N] = P.n_occ; mine[2 * N + 1] = n;
            } catch (const prim::Error &e) {
                C.fail(e);
                std::fill(mine.begin(), mine.end(), 0);
                std::fill(pc.begin(), pc.end(), 0); std::fill(cc.begin(), cc.end(), 0);
            }
            const u64 w = mine.size();
            std::vector<u64> mat = C.allgather_u64(mine);
            u64 maxc = 0;
            for (int g = 0; g < N; g++) {
                rpc[g] = mat[g * w + me]; rcc[g] = mat[g * w + N + me];
                Dr += rpc[g]; Sr += rcc[g];
                occ_total += mat[g * w + 2 * N]; n_total += mat[g * w + 2 * N + 1];
                for (int d = 0; d < N; d++) { maxp = std::max(maxp, mat[g * w + d]); maxc = std::max(maxc, mat[g * w + N + d]); }
            }
            rlen.alloc(Dr); rfreq.alloc(Dr); rcells.alloc(Sr);
            C.named("dict.phrase_len").alltoall(slen.p, pc, rlen.p, rpc, 4, maxp);
            C.named("dict.phrase_freq").alltoall(sfreq.p, pc, rfreq.p, rpc, 4, maxp);
            C.named("dict.phrase_cells").alltoall(scells.p, cc, rcells.p, rcc, 4, maxc);
        }
        L.info.n_in = n_total;
        L.info.parse_size = occ_total;
        // ---- merge what I own: one table over the received lists --------------------------------------------------
        DBuf<u32> list_slot, slot_min, rep_ex;
        try { list_slot.alloc(Dr); rep_ex.alloc(Dr + 1); } catch (const prim::Error &e) { C.fail(e); }
        DBuf<u32> gcells, ph_len, ph_off;
        DBuf<idx_t> ph_freq;
        DBuf<u8> ph_lastT;
        DBuf<u64> ph_pos;
        std::vector<u64> dbase(N + 1, 0), sbase(N + 1, 0);
        u64 D, S;
        u32 maxlen;
        bool sharded_dict = false;               // the merged dictionary stays sharded by owner through the dictionary stage
        u64 maxfreq = 0;                         // the largest merged phrase frequency
        {
            StageTimer st(&tm.hash, "hash");
            // (rank-local, sized by what THIS rank received: a failure here -- memory, table overflow -- is recorded and
            // travels with the counter exchange below, where every rank raises)
            u64 Do = 0, So64 = 0;
            DBuf<u64> goff, o_pos; DBuf<idx_t> o_freq; DBuf<u32> o_len, o_off; DBuf<u8> o_lastT;
            try {
                // (GRLBWT_TEST_FAIL_RANK_MERGE=<rank>: the tests make one rank fail here)
                if (test_fail_rank("GRLBWT_TEST_FAIL_RANK_MERGE", me)) throw prim::Error(-28, "merged phrase table overflow (injected by the test)");
                goff.alloc(Dr + 1);
                const u64 chk = prim::exclusive_scan<u64>(Dr, LenIn{rlen.p}, goff.p, true, "dist.list_offsets");
                if (chk != Sr) throw prim::Error(-71, "dictionary exchange: cell count mismatch");
                u64 cap = 1024;
                while (cap < 2 * Dr) cap <<= 1;
                DBuf<u64> keys(cap);
                DBuf<idx_t> counts(cap);
                DBuf<u32> scal(4);
                keys.zero(); counts.zero(); scal.zero();
                DBuf<u8> is_rep(Dr);
                is_rep.zero();
                prim::for_each(Dr, ListInsertFn{rcells.p, goff.p, rlen.p, rfreq.p, keys.p, counts.p, cap - 1, list_slot.p, scal.p, is_rep.p}, "dist.merge_phrases");
                if (scal.to_host(2)[1]) throw prim::Error(-28, "merged phrase table overflow");
                // representatives = the entries that created their slots, phrases in representative (list) order
                slot_min.alloc(cap);
                prim::for_each(cap, SlotWinnerFn{keys.p, slot_min.p}, "dist.slot_min");
                keys.release();
                Do = prim::exclusive_scan<u32>(Dr, ByteIn{is_rep.p}, rep_ex.p, false, "dist.merge_compact");
                o_pos.alloc(Do); o_freq.alloc(Do); o_len.alloc(Do); o_off.alloc(Do + 1); o_lastT.alloc(Do);
                prim::for_each(Dr, ListPhraseFn{rcells.p, goff.p, rlen.p, list_slot.p, is_rep.p, rep_ex.p, counts.p, o_pos.p, o_freq.p,
                                                o_len.p, o_lastT.p}, "dist.merge_compact");
                So64 = prim::reduce_sum<u64>(Do, LenIn{o_len.p}, "dist.dict_syms");
            } catch (const prim::Error &e) { C.fail(e); Do = 0; So64 = 0; }
            // ---- the merged dictionary, replicated: owner parts in rank order -----------------------------------
            std::vector<u64> cnt = C.allgather_u64({Do, So64});
            // (GRLBWT_TEST_DICT_PART_PAD=<symbols>: the tests put that many unused positions behind every rank's part, so that the
            // global numbering passes 2^32 on a small dictionary -- a form that still used it would wrap)
            const u64 part_pad = test_dict_part_pad();
            for (int g = 0; g < N; g++) {
                dbase[g + 1] = dbase[g] + cnt[2 * g]; sbase[g + 1] = sbase[g] + cnt[2 * g + 1] + part_pad;
                if (cnt[2 * g + 1] >= 0xFFFFFFF0ull) throw prim::Error(-75, "a rank's part of the dictionary has >= 2^32 symbols: use more ranks");
            }
            D = dbase[N];
            // (a dictionary of 2^32 symbols and more in all is taken in the sharded form, where positions are (owner, offset): below)
            const bool wide_dict = sbase[N] >= 0xFFFFFFF0ull;
            prim::exclusive_scan_nosync<u32>(Do, LenIn{o_len.p}, o_off.p, true, "dist.dict_offsets");
            DBuf<u32> ocells(So64);
            {
                RankBits obits;
                build_rankbits32(obits, o_off.p, Do, So64 + 1, "dist.owner_cells");
                prim::for_each((So64 + 15) / 16, ListCellsFn{o_pos.p, o_off.p, Do, So64, rcells.p, ocells.p, obits.words.p, obits.base.p}, "dist.owner_cells");
            }
            rcells.release(); rlen.release(); rfreq.release();
            // DICTIONARY SHARDED BY OWNER (round 5; dict_stage): every rank keeps the phrases it merged and nobody gets anybody
            // else's -- no all-gather of cells, lengths, frequencies and flags, no O(D) pass over the whole dictionary on every rank.
            // Levels with very long phrases (run-aware suffix keys, >= GRLBWT_RUN_KEYS_MIN cells) take the gathered form below, as
            // does GRLBWT_DIST_GATHERED_DICT=1 (rounds 1-4).
            {
                u64 ml = 0, fl = 0, fm = 0;
                if (!C.pending) { try {
                    ml = Do ? (u64)prim::reduce_max<u32>(Do, LenIn{o_len.p}, "dist.maxlen") : 0;
                    fl = prim::reduce_sum<u64>(Do, IdxIn<idx_t>{o_freq.p}, "dist.freq_check");
                    fm = Do ? (u64)prim::reduce_max<u64>(Do, IdxIn<idx_t>{o_freq.p}, "dist.freq_check") : 0;
                } catch (const prim::Error &e) { C.fail(e); } }
                std::vector<u64> mf = C.allgather_u64({ml, fl, fm});
                u64 mx = 0, fs = 0;
                for (int g = 0; g < N; g++) { mx = std::max(mx, mf[3 * g]); fs += mf[3 * g + 1]; maxfreq = std::max(maxfreq, mf[3 * g + 2]); }
                if (fs != occ_total) throw prim::Error(-71, "merged phrase frequencies do not add up to the global parse size");
                static const u64 run_min = getenv("GRLBWT_RUN_KEYS_MIN") ? (u64)atoll(getenv("GRLBWT_RUN_KEYS_MIN")) : 512;
                static const bool gathered = getenv("GRLBWT_DIST_GATHERED_DICT") != nullptr || prim::test_env("GRLBWT_DIST_REPLICATED_DICT") != nullptr;
                // (GRLBWT_DIST_SHARDED_DICT_MIN=<ranks>: from how many ranks on -- see the figures in DESIGN.md section 6)
                static const int sd_min = getenv("GRLBWT_DIST_SHARDED_DICT_MIN") ? atoi(getenv("GRLBWT_DIST_SHARDED_DICT_MIN")) : 4;
                // (... and from how many dictionary symbols on: a small dictionary's all-gather costs less than the collectives of the
                // refinement rounds -- the 1 GB collection at N = 8: 35.7 ms gathered, 40.3 ms sharded, 216 vs 349 collectives)
                static const u64 sd_syms = getenv("GRLBWT_DIST_SHARDED_DICT_MIN_SYMS") ? (u64)atoll(getenv("GRLBWT_DIST_SHARDED_DICT_MIN_SYMS")) : ((u64)1 << 27);
                if (!gathered && mx < run_min && ((N >= sd_min && sbase[N] >= sd_syms) || wide_dict)) {
                    sharded_dict = true;
                    maxlen = (u32)mx;
                    S = sbase[N];
                    gcells = std::move(ocells);              // MY cells, phrases, frequencies, flags
                    ph_off = std::move(o_off);
                    ph_freq = std::move(o_freq);
                    ph_lastT = std::move(o_lastT);
                    ph_pos.alloc(Do);
                    prim::for_each(Do, OffToPosFn{ph_off.p, ph_pos.p}, "dist.dict_offsets");
                }
            }
            if (!sharded_dict && wide_dict) throw prim::Error(-75, "dictionary too large (>= 2^32 symbols with phrases of >= GRLBWT_RUN_KEYS_MIN cells, or GRLBWT_DIST_GATHERED_DICT)");
            if (!sharded_dict) {
            gcells = C.named("dict.merged_cells").allgather_v<u32>(ocells.p, So64, sbase, true);
            ph_len = C.named("dict.merged_len").allgather_v<u32>(o_len.p, Do, dbase, true);
            if (sizeof(idx_t) == 8 && occ_total < 0xFFFFFFFFull) {      // (every merged frequency fits u32: 4 instead of 8 bytes per phrase on the wire)
                DBuf<u32> of32(Do);
                prim::for_each(Do, NarrowIdxFn{o_freq.p, of32.p}, "dist.dict_offsets");
                DBuf<u32> all32 = C.named("dict.merged_freq").allgather_v<u32>(of32.p, Do, dbase, true);
                ph_freq.alloc(D);
                prim::for_each(D, WidenLenFn{all32.p, ph_freq.p}, "dist.dict_offsets");
            } else ph_freq = C.named("dict.merged_freq").allgather_v<idx_t>(o_freq.p, Do, dbase, true);
            ph_lastT = C.named("dict.merged_lastT").allgather_v<u8>(o_lastT.p, Do, dbase, true);
            const u64 fsum = prim::reduce_sum<u64>(D, IdxIn<idx_t>{ph_freq.p}, "dist.freq_check");
            if (fsum != occ_total) throw prim::Error(-71, "merged phrase frequencies do not add up to the global parse size");
            maxlen = prim::reduce_max<u32>(D, LenIn{ph_len.p}, "dist.maxlen");
            ph_off.alloc(D + 1); ph_pos.alloc(D);
            S = prim::exclusive_scan<u32>(D, LenIn{ph_len.p}, ph_off.p, true, "dist.dict_offsets");
            if (S != sbase[N]) throw prim::Error(-71, "dictionary exchange: cell count mismatch");
            prim::for_each(D, OffToPosFn{ph_off.p, ph_pos.p}, "dist.dict_offsets");
            }
        }
        // ---- dictionary stage: suffix sort + group stage sharded by key range, grammar passes and dictionary by owner ----
        DBuf<u32> gval;
        dict_stage<u32, false>(prim::test_env("GRLBWT_DIST_REPLICATED_DICT") ? nullptr : &C, gcells.p, CellOps<u32, false>{0u}, D, S, maxlen, ph_pos.p, ph_freq.p,
                               ph_off.p, ph_lastT.p, sigma, L, gval, nullptr, nullptr, nullptr, 0, 0, 0, &dbase, &sbase, sharded_dict, maxfreq);
        // ---- back to the shards: the value of every phrase I merged returns to its sender, in the order it came ----
        DBuf<u32> lval(P.D);
        {
            StageTimer st(&tm.emit, "emit");
            DBuf<u32> rval(Dr), sval(P.D);
            prim::for_each(Dr, ListValFn{list_slot.p, slot_min.p, rep_ex.p, gval.p, sharded_dict ? 0 : dbase[me], rval.p}, "dist.list_values");      // (sharded dictionary: gval holds my phrases only)
            C.named("emit.phrase_values").alltoall(rval.p, rpc, sval.p, pc, 4, maxp);
            prim::for_each(P.D, ScatterU32Fn{order.p, sval.p, lval.p}, "dist.local_values");
        }
        emit_local(P, lval.p);
        finish_round(P, L, g_n_strings, occ_total);
    }

    bool dist_parse_round(const Comm &C) {
        if (parse_done) return true;
        if (levels.empty()) {
            switch (cell_bytes) {
                case 1: dist_round_t<u8, true>(C, (const u8 *)text0, n0, cur_sigma, (u8)stats.min_sym); break;
                case 2: dist_round_t<u16, true>(C, (const u16 *)text0, n0, cur_sigma, (u16)stats.min_sym); break;
                case 4: dist_round_t<u32, true>(C, (const u32 *)text0, n0, cur_sigma, (u32)stats.min_sym); break;
                default: dist_round_t<u64, true>(C, (const u64 *)text0, n0, cur_sigma, (u64)stats.min_sym); break;
            }
        } else {
            DBuf<u32> t = std::move(cur_text);
            dist_round_t<u32, false>(C, t.p, cur_n, cur_sigma, 0u);
        }
        if (levels.size() > 64) throw prim::Error(-75, "too many parsing rounds");
        return parse_done;
    }

    // ---- distributed induction --------------------------------------------------------------
    // Invariant: every rank holds a contiguous slice of BWT_{level} (runs), slices in rank order.
    // Per level: (1) the shards agree on the cell layout (longest run) and every rank runs passes A+B over its slice: chain
    // expansion + stable bucket split, the single-GPU kernels; (2) the level's output is cut into one piece per rank at
    // pre-BWT run boundaries; a piece owns the buckets inside it; (3) all-to-all #1: every rank sends each owner the part of
    // the rewritten BWT_{r+1} (runs, clipped) that the owner's piece consumes and that lies in its slice -- consumption is
    // monotone in output order, so a piece needs ONE contiguous window; (4) all-to-all #2: the cells, already bucket-major,
    // go to the owners of their buckets (8 bytes per cell in the usual layout) and are merged by one stable split on the
    // bucket bits (rank order inside a bucket = slice order); (5) every rank runs the single-GPU pass C on its piece and
    // ends up holding its slice of BWT_r.  Nothing is replicated and no symbol crosses the fabric more than once per level.
    void dist_first_bwt(const Comm &C) {
        try { first_bwt(); }     // local: my strings' final symbols, in collection order
        catch (const prim::Error &e) {            // (raised by every rank at the first counter exchange of the level below)
            C.fail(e);
            bwt = Runs();
            bwt.sym.alloc(0); bwt.len.alloc(0);
            bwt_level = (int)levels.size();
            linfo.assign(levels.size() + 1, LevelInfo());
        }
    }

    void dist_induce_level(const Comm &C) {
        if (bwt_level <= 0) throw prim::Error(-22, "no level left to induce");
        const int r = bwt_level - 1;
        prim::rt().tag = r;
        prim::rt().phase = 'i';
        LevelData &L = levels[r];
        const u32 bwt_code = L.sigma + 1, hocc_code = L.sigma + 2, take_code = bwt_code;
        const u64 P = L.prebwt.R, M = L.M, R = bwt.R;
        const int N = C.size, me = C.rank;
        LevelInfo &I = linfo[r];
        I.R_next = R; I.P = P;
        // (1) one cell layout for all shards, then passes A+B over my slice (the single-GPU kernels)
        // (local failures -- out of memory, an internal check -- are recorded with C.fail and travel with the next counter
        // exchange, where every rank raises: nobody is left waiting for a rank that gave up)
        DBuf<idx_t> Tpos;
        u64 Tlocal = 0, maxrun = 0, Toff = 0, Ttotal = 0;
        std::vector<u64> piece;                  // (pre_local) per rank boundary: -, first metasymbol, symbols and BWT-marker symbols in front
        {
            u64 mr = 0;
            try {
                StageTimer st(&tm.ind_expand, "ind_expand");
                need_len();                      // (pass C leaves the prefix only; the passes below read lengths after the prefix has moved on)
                if (bwt.pos.p) { Tpos = std::move(bwt.pos); Tlocal = bwt.n; }      // pass C of the level above left my slice's prefix behind
                else {
                    Tpos.alloc(R + 1);
                    Tlocal = (u64)prim::exclusive_scan<idx_t>(R, IdxIn<idx_t>{bwt.len.p}, Tpos.p, true, "dist.Tpos");
                }
                mr = level_maxrun();
            } catch (const prim::Error &e) { C.fail(e); Tlocal = 0; mr = 0; }
            // (pre-BWT kept by key range: my piece's symbols, BWT-marker symbols, metasymbols and runs travel with this exchange)
            u64 pn = 0, pb = 0;
            if (L.pre_local) {
                try {
                    pn = prim::reduce_sum<u64>(P, IdxIn<idx_t>{L.prebwt.len.p}, "dist.piece_sums");
                    pb = prim::reduce_sum<u64>(P, PreBwtLenIn{L.prebwt.sym.p, L.prebwt.len.p, bwt_code}, "dist.piece_sums");
                } catch (const prim::Error &e) { C.fail(e); }
            }
            std::vector<u64> g1 = C.allgather_u64({mr, Tlocal, pn, pb, (u64)L.Ml, P});
            for (int g = 0; g < N; g++) {
                if (g1[6 * g] > maxrun) maxrun = g1[6 * g];
                if (g < me) Toff += g1[6 * g + 1];
                Ttotal += g1[6 * g + 1];
            }
            if (L.pre_local) {
                piece.assign(4 * ((u64)N + 1), 0);
                u64 Psum = 0;
                for (int g = 0; g < N; g++) {
                    piece[4 * (g + 1) + 1] = piece[4 * g + 1] + g1[6 * g + 4];
                    piece[4 * (g + 1) + 2] = piece[4 * g + 2] + g1[6 * g + 2];
                    piece[4 * (g + 1) + 3] = piece[4 * g + 3] + g1[6 * g + 3];
                    Psum += g1[6 * g + 5];
                }
                if (piece[4 * (u64)N + 1] != M) throw prim::Error(-71, "dist induction: the pieces' metasymbols do not add up (level " + std::to_string(r) + ")");
                I.P = Psum;
            }
        }
        DBuf<u32> term;
        int kb = 1, lb = 1;
        u64 E = 0;
        u32 p0, p1, u0, u1;
        u64 n_out, n_r = 0;
        std::vector<u64> sp(4 * ((u64)N + 1), 0), cbh, v(2 * (u64)N, 0);
        DBuf<u64> split;
        try {
            term.alloc(R);
            split.alloc(4 * ((u64)N + 1));
            // (GRLBWT_TEST_FAIL_RANK_INDUCE=<rank>: the tests make one rank fail here)
            if (test_fail_rank("GRLBWT_TEST_FAIL_RANK_INDUCE", me)) throw prim::Error(-12, "out of device memory (injected by the test)");
            E = expand_split(L, term, maxrun, kb, lb);
            I.E = E;
            StageTimer st(&tm.ind_assemble, "ind_assemble");
            // (2) owners of the output: pre-BWT run ranges of about n_r / size symbols, and the buckets inside them
            if (L.pre_local) {                   // the pieces are the ranks' own parts of the pre-BWT: nothing to look up
                sp = piece;
                sp[4 * (u64)me] = 0; sp[4 * ((u64)me + 1)] = P;          // (run indices are relative to my part)
                n_r = piece[4 * (u64)N + 2];
                if (n_r != L.info.n_in) throw prim::Error(-71, "dist induction: the pre-BWT of level " + std::to_string(r) + " does not describe the level");
                prim::h2d(split.p, sp.data(), 4 * ((u64)N + 1) * 8);
            } else {
                DBuf<idx_t> Ppos(P + 1);
                DBuf<HoccBwt> PHB(P + 1);
                n_r = (u64)prim::exclusive_scan<idx_t>(P, IdxIn<idx_t>{L.prebwt.len.p}, Ppos.p, true, "dist.Ppos");
                prim::exclusive_scan_nosync<HoccBwt>(P, PreScanIn{L.prebwt.sym.p, L.prebwt.len.p, hocc_code, bwt_code}, PHB.p, true, "dist.pre_scan");
                if (n_r != L.info.n_in) throw prim::Error(-71, "dist induction: the pre-BWT of level " + std::to_string(r) + " does not describe the level");
                prim::for_each((u64)N + 1, OwnerSplitFn{Ppos.p, PHB.p, L.u_to_p.p, P, M, n_r, N, split.p}, "dist.owners");
                sp = split.to_host(4 * ((u64)N + 1));
            }
            // my cells and my TAKE symbols per owner
            {
                const CellView mine = cell_view(kb, lb);
                DBuf<u64> cb((u64)N + 1);
                prim::for_each((u64)N + 1, CellBoundsFn{mine, E, split.p, cb.p}, "dist.cell_bounds");
                cbh = cb.to_host((u64)N + 1);
                for (int d = 0; d < N; d++) {
                    const u64 cnt = cbh[d + 1] - cbh[d];
                    v[d] = cnt;
                    v[N + d] = cnt ? prim::reduce_sum<u64>(cnt, CellTakeOffIn{mine, take_code, cbh[d]}, "dist.take_sums") : 0;
                }
            }
        } catch (const prim::Error &e) { C.fail(e); std::fill(v.begin(), v.end(), 0); }
        {
            StageTimer st(&tm.ind_assemble, "ind_assemble");
            std::vector<u64> mat = C.allgather_u64(v);           // mat[s*2N + d] cells, mat[s*2N + N + d] TAKE symbols of rank s for owner d
            // symbols of the rewritten BWT_{r+1} consumed in front of every owner's piece (output order = consumption order)
            std::vector<u64> Tc((u64)N + 1);
            {
                u64 acc = 0;
                for (int d = 0; d <= N; d++) {
                    Tc[d] = sp[4 * d + 3] + acc;
                    if (d < N) for (int g = 0; g < N; g++) acc += mat[(u64)g * 2 * N + N + d];
                }
            }
            if (Tc[N] != Ttotal) throw prim::Error(-71, "dist induction: BWT_{r+1} consumption mismatch (level " + std::to_string(r) + ": " +
                                                            std::to_string(Tc[N]) + " vs " + std::to_string(Ttotal) + ")");
            // (3) every owner's window of the rewritten BWT_{r+1}: I send the part of it that lies in my slice
            DBuf<u32> wsym; DBuf<idx_t> wlen;
            u64 Rw = 0;
            {
                std::vector<u64> ab(2 * (u64)N), scnt(N), soff((u64)N + 1, 0), rcnt(N);
                for (int d = 0; d < N; d++) {
                    const u64 lo = Tc[d] > Toff ? Tc[d] - Toff : 0, hi = Tc[d + 1] > Toff ? Tc[d + 1] - Toff : 0;
                    ab[2 * d] = lo < Tlocal ? lo : Tlocal;
                    ab[2 * d + 1] = hi < Tlocal ? hi : Tlocal;
                }
                DBuf<u64> abd(2 * (u64)N), kc(2 * (u64)N), sod((u64)N + 1);
                try {
                    prim::h2d(abd.p, ab.data(), 2 * (u64)N * 8);
                    prim::for_each((u64)N, WindowRunsFn{Tpos.p, R, abd.p, kc.p}, "dist.window_runs");
                    std::vector<u64> kch = kc.to_host(2 * (u64)N);
                    for (int d = 0; d < N; d++) { scnt[d] = kch[2 * d + 1]; soff[d + 1] = soff[d] + scnt[d]; }
                } catch (const prim::Error &e) { C.fail(e); std::fill(scnt.begin(), scnt.end(), 0); std::fill(soff.begin(), soff.end(), 0); }
                std::vector<u64> rc = C.allgather_u64(scnt);     // rc[s*N + d]
                for (int g = 0; g < N; g++) { rcnt[g] = rc[(u64)g * N + me]; Rw += rcnt[g]; }
                const u64 maxw = *std::max_element(rc.begin(), rc.end());
                // (the lengths travel as u32 whenever the level's longest run fits -- the ranks agreed on `maxrun` for the cell layout)
                const bool narrow = sizeof(idx_t) == 8 && maxrun < 0xFFFFFFFFull;
                DBuf<u32> ssym, slen32, wlen32; DBuf<idx_t> slen;
                try {
                    ssym.alloc(soff[N]);
                    prim::h2d(sod.p, soff.data(), ((u64)N + 1) * 8);
                    if (narrow) {
                        slen32.alloc(soff[N]); wlen32.alloc(Rw);
                        prim::for_each(soff[N], WindowSendFn<u32>{Tpos.p, term.p, abd.p, kc.p, sod.p, N, ssym.p, slen32.p}, "dist.window_send");
                    } else {
                        slen.alloc(soff[N]);
                        prim::for_each(soff[N], WindowSendFn<idx_t>{Tpos.p, term.p, abd.p, kc.p, sod.p, N, ssym.p, slen.p}, "dist.window_send");
                    }
                    wsym.alloc(Rw); wlen.alloc(Rw);
                } catch (const prim::Error &e) { C.fail(e); }
                C.allgather_u64({});                             // (nothing but the failure flag: the bulk exchanges below have no way back)
                C.named("induce.window_sym").alltoall(ssym.p, scnt, wsym.p, rcnt, sizeof(u32), maxw);
                if (narrow) {
                    C.named("induce.window_len").alltoall(slen32.p, scnt, wlen32.p, rcnt, sizeof(u32), maxw);
                    prim::for_each(Rw, WidenLenFn{wlen32.p, wlen.p}, "dist.window_send");
                } else C.named("induce.window_len").alltoall(slen.p, scnt, wlen.p, rcnt, sizeof(idx_t), maxw);
            }
            Tpos.release(); term.release();
            bwt.sym.release(); bwt.len.release(); bwt.pos.release();
            // (4) the cells of my buckets, from every rank; rank order inside a bucket = order of the slices
            {
                std::vector<u64> scnt(N), rcnt(N);
                u64 Er = 0;
                for (int d = 0; d < N; d++) scnt[d] = v[d];
                for (int g = 0; g < N; g++) { rcnt[g] = mat[(u64)g * 2 * N + me]; Er += rcnt[g]; }
                u64 maxc = 0;
                for (int g = 0; g < N; g++) for (int d = 0; d < N; d++) maxc = std::max(maxc, mat[(u64)g * 2 * N + d]);
                const int bits = kb;
                // (the received blocks are bucket-sorted: merged by block offsets; GRLBWT_MERGE_CELLS=sort keeps the stable radix sort)
                static const bool merge_by_blocks = !(prim::test_env("GRLBWT_MERGE_CELLS") && prim::test_env("GRLBWT_MERGE_CELLS")[0] == 's');
                const u32 mu0 = (u32)sp[4 * me + 1], mu1 = (u32)sp[4 * (me + 1) + 1];
                if (c_sfused32.p) {
                    DBuf<u32> rf(Er);
                    C.named("induce.cells").alltoall(c_sfused32.p, scnt, rf.p, rcnt, 4, maxc);
                    c_sfused32 = std::move(rf);
                    if (N > 1 && Er) {
                        if (merge_by_blocks) merge_cell_blocks<u32>(c_sfused32, rcnt, kb, lb, mu0, (u64)(mu1 - mu0), Er);
                        else {
                            DBuf<u32> tmp(Er);
                            if (prim::sort_keys<u32, 1>(c_sfused32.p, tmp.p, Er, 0, bits, "dist.merge_cells")) c_sfused32 = std::move(tmp);
                        }
                    }
                } else if (c_sfused.p) {
                    DBuf<u64> rf(Er);
                    C.named("induce.cells").alltoall(c_sfused.p, scnt, rf.p, rcnt, 8, maxc);
                    c_sfused = std::move(rf);
                    if (N > 1 && Er) {
                        if (merge_by_blocks) merge_cell_blocks<u64>(c_sfused, rcnt, kb, lb, mu0, (u64)(mu1 - mu0), Er);
                        else {
                            DBuf<u64> tmp(Er);
                            if (prim::sort_keys<u64, 1>(c_sfused.p, tmp.p, Er, 0, bits, "dist.merge_cells")) c_sfused = std::move(tmp);
                        }
                    }
                } else if (c_spack.p) {
                    DBuf<u32> rk(Er); DBuf<u64> rp(Er);
                    C.named("induce.cells_key").alltoall(c_skey.p, scnt, rk.p, rcnt, 4, maxc);
                    C.named("induce.cells_pack").alltoall(c_spack.p, scnt, rp.p, rcnt, 8, maxc);
                    c_skey = std::move(rk); c_spack = std::move(rp);
                    if (N > 1 && Er) {
                        DBuf<u32> k2(Er); DBuf<u64> p2(Er);
                        if (prim::sort_pairs<u32, u64>(c_skey.p, c_spack.p, k2.p, p2.p, Er, 0, bits, "dist.merge_cells")) { c_skey = std::move(k2); c_spack = std::move(p2); }
                    }
                } else {
                    DBuf<u32> rk(Er), rs(Er); DBuf<idx_t> rl(Er);
                    C.named("induce.cells_key").alltoall(c_skey.p, scnt, rk.p, rcnt, 4, maxc);
                    C.named("induce.cells_sym").alltoall(c_ssym.p, scnt, rs.p, rcnt, 4, maxc);
                    C.named("induce.cells_len").alltoall(c_slen.p, scnt, rl.p, rcnt, sizeof(idx_t), maxc);
                    c_skey = std::move(rk);
                    if (N > 1 && Er) {
                        DBuf<u32> k2(Er); DBuf<idx_t> ix(Er), ix2(Er);
                        prim::for_each(Er, IotaIdxFn{ix.p}, "dist.merge_cells");
                        int res = prim::sort_pairs<u32, idx_t>(c_skey.p, ix.p, k2.p, ix2.p, Er, 0, bits, "dist.merge_cells");
                        c_ssym.alloc(Er); c_slen.alloc(Er);
                        prim::for_each(Er, GatherCellFn{res ? ix2.p : ix.p, rs.p, rl.p, c_ssym.p, c_slen.p}, "dist.merge_cells");
                        if (res) c_skey = std::move(k2);
                    } else { c_ssym = std::move(rs); c_slen = std::move(rl); }
                }
                E = Er;
            }
            p0 = (u32)sp[4 * me]; p1 = (u32)sp[4 * (me + 1)];
            u0 = (u32)sp[4 * me + 1]; u1 = (u32)sp[4 * (me + 1) + 1];
            n_out = sp[4 * (me + 1) + 2] - sp[4 * me + 2];
            // my window becomes "the BWT_{r+1}" of my piece
            bwt.sym.alloc(0);
            bwt.len = std::move(wlen);
            bwt.R = Rw;
            bwt.n = Tc[me + 1] - Tc[me];
            term = std::move(wsym);
        }
        // (5) pass C on my piece (the single-GPU kernels)
        try {
            if (n_out == 0) {
                if (E || bwt.R) throw prim::Error(-71, "dist induction: cells or BWT_{r+1} symbols for an empty piece (level " + std::to_string(r) + ")");
                release_cells();
                bwt = Runs();
                bwt.sym.alloc(0); bwt.len.alloc(0);
            } else {
                const u64 Pm = p1 - p0, Mm = u1 - u0;
                if (L.pre_local) {               // my part's maps are relative to it already
                    if (Mm != (u64)L.Ml) throw prim::Error(-71, "dist induction: piece and key range disagree (level " + std::to_string(r) + ")");
                    assemble(AsmIn{L.prebwt.sym.p, L.prebwt.len.p, Pm, L.u_to_p.p, L.p_to_u.p, Mm, L.sigma, n_out}, I, cell_view(kb, lb, u0), E, term, r);
                } else {
                DBuf<u32> u2p(Mm), p2u(Pm);
                prim::for_each(Mm, RebaseFn{L.u_to_p.p + u0, p0, u2p.p}, "dist.piece_maps");
                if (L.p_to_u.p) prim::for_each(Pm, RebaseFn{L.p_to_u.p + p0, u0, p2u.p}, "dist.piece_maps");
                else prim::for_each(Pm, PieceMetaFn{L.u_to_p.p, M, p0, u0, p2u.p}, "dist.piece_maps");
                assemble(AsmIn{L.prebwt.sym.p + p0, L.prebwt.len.p + p0, Pm, u2p.p, p2u.p, Mm, L.sigma, n_out}, I, cell_view(kb, lb, u0), E, term, r);
                }
            }
        } catch (const prim::Error &e) {          // the next counter exchange (next level, or the image assembly) raises on every rank
            C.fail(e);
            release_cells();
            bwt = Runs();
            bwt.sym.alloc(0); bwt.len.alloc(0);
        }
        bwt_level = r;
        I.R = bwt.R; I.n = n_r;
        release_level(L);
    }

    // the image from the slices of BWT_0: every rank packs the records of its own slice, the packed parts are all-gathered
    // behind the header.  Maximal runs across slice boundaries: a first run that continues the last run of the slices in
    // front is dropped and its length goes to the slice where that run began (decided from every slice's two edge runs).
    void dist_finish(const Comm &C) {
        if (bwt_level != 0) throw prim::Error(-22, "induction not finished");
        prim::rt().tag = -1; prim::rt().phase = 0;
        StageTimer st(&tm.finish, "finish");
        const int N = C.size, me = C.rank;
        const u64 R = bwt.R;
        need_len();
        std::vector<u64> mine(5, 0);             // runs, first (sym, len), last (sym, len)
        mine[0] = R;
        if (R) { mine[1] = bwt.sym.get(0); mine[2] = (u64)bwt.len.get(0); mine[3] = bwt.sym.get(R - 1); mine[4] = (u64)bwt.len.get(R - 1); }
        std::vector<u64> all = C.allgather_u64(mine);
        std::vector<u64> extra(N, 0);
        std::vector<int> drop(N, 0);
        int open = -1;
        u64 open_sym = 0;
        for (int g = 0; g < N; g++) {
            const u64 Rg = all[5 * g];
            if (!Rg) continue;
            if (open >= 0 && all[5 * g + 1] == open_sym) {
                drop[g] = 1;
                extra[open] += all[5 * g + 2];
                if (Rg == 1) continue;           // the whole slice was the tail of that run
            }
            open = g;
            open_sym = all[5 * g + 3];
        }
        const u64 first = drop[me], Rm = R - first;
        if (extra[me]) prim::for_each(1, AddLenFn{bwt.len.p + (R - 1), extra[me]}, "pack_rl_bwt");
        const u32 sb = (u32)stats.sb, fb = (u32)stats.fb, rec = sb + fb;
        u8 hdr[16] = {0};
        for (int i = 0; i < 8; i++) { hdr[i] = (u8)((u64)sb >> (8 * i)); hdr[8 + i] = (u8)((u64)fb >> (8 * i)); }
        std::vector<u64> cnt = C.allgather_u64({Rm * rec}), base(N + 1, 0);
        for (int g = 0; g < N; g++) base[g + 1] = base[g] + cnt[g];
        image_bytes = 16 + base[N];
        image_runs = base[N] / rec;
        if (C.keep_parts) {
            // The image stays in parts (Comm::keep_parts): my runs are bytes [16 + base[me], 16 + base[me + 1]) of it, rank 0's part
            // starts with the header.  Nothing crosses the fabric -- every rank sends its part to N - 1 peers otherwise, 13 GB per
            // rank of the 10 GB collection's 8.3 GB image at N = 8 -- and N ranks write one file at N offsets (grlbwt_result_write_part).
            const u64 h = me == 0 ? 16 : 0;
            image_part_off = me == 0 ? 0 : 16 + base[me];
            image_part_bytes = h + Rm * rec;
            image.alloc(image_part_bytes);
            if (h) prim::h2d(image.p, hdr, 16);
            prim::for_each(Rm, PackRunsFn{bwt.sym.p + first, RunLen{bwt.len.p + first, nullptr}, sb, fb, image.p + h, 0u}, "pack_rl_bwt");
        } else {
            DBuf<u8> part(Rm * rec);
            prim::for_each(Rm, PackRunsFn{bwt.sym.p + first, RunLen{bwt.len.p + first, nullptr}, sb, fb, part.p, 0u}, "pack_rl_bwt");
            image.alloc(image_bytes);
            prim::h2d(image.p, hdr, 16);
            C.named("image.parts").allgather_v<u8>(part.p, Rm * rec, base, true, image.p + 16);
            image_part_off = 0; image_part_bytes = image_bytes;
        }
        stats.n_strings = g_n_strings;
        stats.n_syms = g_n_syms;
        linfo[0].R = image_runs;
        prim::sync();
    }

    void dist_induce(const Comm &C) {
        dist_first_bwt(C);
        while (bwt_level > 0) dist_induce_level(C);
        dist_finish(C);
    }

    // Fallback kept for comparison: the deepest parse (one cell per string) is all-gathered and
    // every rank induces the whole collection's BWT from the replicated grammar.
    void dist_induce_replicated(const Comm &C) {
        std::vector<u64> base;
        DBuf<u32> all = C.named("replicated.text").allgather_v<u32>(cur_text.p, cur_n, base);
        cur_n = base[C.size];
        if (cur_n != g_n_strings) throw prim::Error(-71, "deepest parse does not have one cell per string");
        cur_text = std::move(all);
        stats.n_strings = g_n_strings;
        stats.n_syms = g_n_syms;
        induce_phase();
        finish();
    }

    void dist_build(const Comm &C) {
        dist_stats(C);
        while (!dist_parse_round(C)) {}
        if (prim::test_env("GRLBWT_DIST_REPLICATED_INDUCTION")) dist_induce_replicated(C);
        else dist_induce(C);
    }

    // ---- consumers of the .rl_bwt image (validation): grl2plain + reverse_bwt on the device ----------
    // Rebuilds the collection (strings in input order) from an image in device memory.
    template <class cell_t>
    static u64 invert_t(const u8 *img, u64 R, u32 sb, u32 fb, cell_t *text_out, u64 capacity) {
        DBuf<u32> rsym(R);
        DBuf<idx_t> rlen(R), rpos(R + 1);
        prim::for_each(R, UnpackRunsFn{img, sb, fb, rsym.p, rlen.p}, "inv.unpack");
        u64 n = (u64)prim::exclusive_scan<idx_t>(R, IdxIn<idx_t>{rlen.p}, rpos.p, true, "inv.positions");
        if (n > capacity) throw prim::Error(-22, "inversion: output buffer too small");
        u32 sep = prim::reduce_min<u32>(R, PtrU32In{rsym.p}, "inv.sep");
        u32 mx = prim::reduce_max<u32>(R, PtrU32In{rsym.p}, "inv.max");
        RankBits rb;
        build_rankbits(rb, rpos.p, R, n + 1, "inv.runbits");
        DBuf<u32> bwt(n), k2(n);
        DBuf<idx_t> ia(n), ib(n), lf(n);
        prim::for_each(n, ExpandRunsFn{rsym.p, rb.words.p, rb.base.p, bwt.p, ia.p}, "inv.expand");          // grl2plain
        d2d_copy(k2.p, bwt.p, n);
        DBuf<u32> k3(n);
        int bits = (int)bitlen64(mx);
        if (bits < 1) bits = 1;
        int res = prim::sort_pairs<u32, idx_t>(k2.p, ia.p, k3.p, ib.p, n, 0, bits, "inv.lf_sort");
        prim::for_each(n, LfScatterFn{res ? ib.p : ia.p, lf.p}, "inv.lf");
        k2.release(); k3.release();
        u64 k = 0;                       // number of strings = occurrences of the separator
        {
            k = prim::reduce_sum<u64>(R, SepLenIn{rsym.p, rlen.p, sep}, "inv.nstrings");
        }
        DBuf<idx_t> slen(k + 1);
        prim::for_each(k, InvertLenFn{bwt.p, lf.p, sep, slen.p}, "inv.lengths");
        u64 tot = (u64)prim::exclusive_scan<idx_t>(k, IdxIn<idx_t>{slen.p}, slen.p, true, "inv.offsets");
        if (tot != n) throw prim::Error(-71, "inversion: string lengths do not add up to the BWT length");
        prim::for_each(k, InvertWriteFn<cell_t>{bwt.p, lf.p, slen.p, sep, text_out}, "inv.write");
        prim::sync();
        return n;
    }
    template <class T>
    static void d2d_copy(T *dst, const T *src, u64 n) { prim::d2d(dst, src, n * sizeof(T)); }
    // the run-indexed form (functor block "the same inversion indexed by RUNS")
    template <class cell_t>
    static u64 invert_runs_t(const u8 *img, u64 R, u32 sb, u32 fb, cell_t *text_out, u64 capacity, u64 tail = 0, u64 *n_strings_out = nullptr) {
        DBuf<u32> rsym(R);
        DBuf<idx_t> rlen(R), rpos(R + 1);
        prim::for_each(R, UnpackRunsFn{img, sb, fb, rsym.p, rlen.p}, "inv.unpack");
        const u64 n = (u64)prim::exclusive_scan<idx_t>(R, IdxIn<idx_t>{rlen.p}, rpos.p, true, "inv.positions");
        if (!tail && n > capacity) throw prim::Error(-22, "inversion: output buffer too small");
        const u32 sep = prim::reduce_min<u32>(R, PtrU32In{rsym.p}, "inv.sep");
        const u32 mx = prim::reduce_max<u32>(R, PtrU32In{rsym.p}, "inv.max");
        const u64 k = prim::reduce_sum<u64>(R, SepLenIn{rsym.p, rlen.p, sep}, "inv.nstrings");      // strings = separators
        // run-start bit-vector with the rank of every word beside it
        DBuf<RankCell> rc;
        {
            RankBits rb;
            build_rankbits(rb, rpos.p, R, n + 1, "inv.runbits");
            const u64 nw = (n + 1) / 64 + 2;
            rc.alloc(nw);
            prim::for_each(nw, RankCellFn{rb.words.p, rb.base.p, rc.p}, "inv.rankcells");
        }
        // LF of every run start: runs in (symbol, position) order, scan of their lengths
        DBuf<RunRec> rec(R);
        {
            DBuf<u32> k2(R), k3(R);
            DBuf<idx_t> ia(R), ib(R);
            d2d_copy(k2.p, rsym.p, R);
            prim::for_each(R, IotaFn{ia.p}, "inv.iota");
            int bits = (int)bitlen64(mx);
            if (bits < 1) bits = 1;
            const int res = prim::sort_pairs<u32, idx_t>(k2.p, ia.p, k3.p, ib.p, R, 0, bits, "inv.lf_sort");
            const idx_t *order = res ? ib.p : ia.p;
            const u64 tot = (u64)prim::exclusive_scan_emit<idx_t>(R, RunOrderLenIn{rlen.p, order}, RunLfEmitFn{order, rpos.p, rsym.p, rec.p}, "inv.lf");
            if (tot != n) throw prim::Error(-71, "inversion: run lengths do not add up");
        }
        rsym.release(); rlen.release(); rpos.release();
        if (tail) {
            // the ends of the strings only (collections whose strings are too long to walk in a test's time -- 100
            // strings of 249 M cells are 249 M dependent steps each): `tail` LF steps per string from its terminator's row
            if (k * tail > capacity) throw prim::Error(-22, "inversion: output buffer too small");
            DBuf<idx_t> got(k);
            prim::for_each(k, RunInvertTailFn<cell_t>{rc.p, rec.p, sep, tail, text_out, got.p}, "inv.tails");
            prim::sync();
            if (n_strings_out) *n_strings_out = k;
            return (u64)prim::reduce_sum<u64>(k, IdxIn<idx_t>{got.p}, "inv.tails");
        }
        DBuf<idx_t> slen(k + 1);
        prim::for_each(k, RunInvertLenFn{rc.p, rec.p, sep, slen.p}, "inv.lengths");
        const u64 tot = (u64)prim::exclusive_scan<idx_t>(k, IdxIn<idx_t>{slen.p}, slen.p, true, "inv.offsets");
        if (tot != n) throw prim::Error(-71, "inversion: string lengths do not add up to the BWT length");
        prim::for_each(k, RunInvertWriteFn<cell_t>{rc.p, rec.p, slen.p, sep, text_out}, "inv.write");
        prim::sync();
        return n;
    }

    // the last `tail` cells of every string (slot i of `tail` cells holds string i's end, right-aligned); returns the cells written
    static u64 invert_image_tails(const void *dev_image, u64 image_bytes, int cell_bytes, u64 tail, void *dev_out, u64 capacity_cells, u64 *n_strings_out) {
        if (image_bytes < 16 || tail == 0) throw prim::Error(-22, "not an .rl_bwt image, or no tail length");
        u64 hdr[2];
        prim::d2h(hdr, dev_image, 16);
        const u64 sb = hdr[0], fb = hdr[1];
        if (sb == 0 || sb > 8 || fb == 0 || fb > 8 || (image_bytes - 16) % (sb + fb)) throw prim::Error(-22, "bad .rl_bwt header");
        const u64 R = (image_bytes - 16) / (sb + fb);
        const u8 *img = (const u8 *)dev_image;
        switch (cell_bytes) {
            case 1: return invert_runs_t<u8>(img, R, (u32)sb, (u32)fb, (u8 *)dev_out, capacity_cells, tail, n_strings_out);
            case 2: return invert_runs_t<u16>(img, R, (u32)sb, (u32)fb, (u16 *)dev_out, capacity_cells, tail, n_strings_out);
            case 4: return invert_runs_t<u32>(img, R, (u32)sb, (u32)fb, (u32 *)dev_out, capacity_cells, tail, n_strings_out);
            case 8: return invert_runs_t<u64>(img, R, (u32)sb, (u32)fb, (u64 *)dev_out, capacity_cells, tail, n_strings_out);
            default: throw prim::Error(-22, "bad cell width");
        }
    }

    static u64 invert_image(const void *dev_image, u64 image_bytes, int cell_bytes, void *dev_text_out, u64 capacity_cells, u64 n_total_hint = 0) {
        if (image_bytes < 16) throw prim::Error(-22, "not an .rl_bwt image");
        u64 hdr[2];
        prim::d2h(hdr, dev_image, 16);
        u64 sb = hdr[0], fb = hdr[1];
        if (sb == 0 || sb > 8 || fb == 0 || fb > 8 || (image_bytes - 16) % (sb + fb)) throw prim::Error(-22, "bad .rl_bwt header");
        u64 R = (image_bytes - 16) / (sb + fb);
        const u8 *img = (const u8 *)dev_image;
        // Two index forms.  Per POSITION (LF array: one gather per symbol of a string, (12 + 3 * sizeof(idx_t)) bytes per symbol
        // while it is built) when that fits the device comfortably; per RUN (two dependent gathers per symbol, 30-50 bytes per
        // run) otherwise -- the 10 GB headline image needs 360 GB in the first form and ~100 GB in the second.
        // GRLBWT_INVERT=runs|positions forces one (the tests take both on small inputs).
        bool by_runs = (n_total_hint ? n_total_hint : capacity_cells) * (12 + 3 * (u64)sizeof(idx_t)) > prim::mem_available() / 2;
        if (const char *f = getenv("GRLBWT_INVERT")) by_runs = f[0] == 'r';
        if (by_runs) {
            switch (cell_bytes) {
                case 1: return invert_runs_t<u8>(img, R, (u32)sb, (u32)fb, (u8 *)dev_text_out, capacity_cells);
                case 2: return invert_runs_t<u16>(img, R, (u32)sb, (u32)fb, (u16 *)dev_text_out, capacity_cells);
                case 4: return invert_runs_t<u32>(img, R, (u32)sb, (u32)fb, (u32 *)dev_text_out, capacity_cells);
                case 8: return invert_runs_t<u64>(img, R, (u32)sb, (u32)fb, (u64 *)dev_text_out, capacity_cells);
                default: throw prim::Error(-22, "bad cell width");
            }
        }
        switch (cell_bytes) {
            case 1: return invert_t<u8>(img, R, (u32)sb, (u32)fb, (u8 *)dev_text_out, capacity_cells);
            case 2: return invert_t<u16>(img, R, (u32)sb, (u32)fb, (u16 *)dev_text_out, capacity_cells);
            case 4: return invert_t<u32>(img, R, (u32)sb, (u32)fb, (u32 *)dev_text_out, capacity_cells);
            case 8: return invert_t<u64>(img, R, (u32)sb, (u32)fb, (u64 *)dev_text_out, capacity_cells);
            default: throw prim::Error(-22, "bad cell width");
        }
    }

    // ---- the other .rl_bwt consumers of scripts/ (SURVEY 8f-2), on an image in device memory --------------
    struct ImageHeader { u64 sb, fb, R; };
    static ImageHeader image_header(const void *dev_image, u64 image_bytes) {
        if (image_bytes < 16) throw prim::Error(-22, "not an .rl_bwt image");
        u64 hdr[2];
        prim::d2h(hdr, dev_image, 16);
        if (hdr[0] == 0 || hdr[0] > 8 || hdr[1] == 0 || hdr[1] > 8 || (image_bytes - 16) % (hdr[0] + hdr[1]))
            throw prim::Error(-22, "bad .rl_bwt header");
        return ImageHeader{hdr[0], hdr[1], (image_bytes - 16) / (hdr[0] + hdr[1])};
    }
    // ---- f3: FASTA/FASTQ text in device memory -> one string per line (see the functor block) ------------------------
    struct FastxInfo { u64 n_out, n_strings; int fastq; };
    static constexpr int kErrNotFastx = -22, kErrNotDna = -86;
    static FastxInfo fastx_to_text(const u8 *in, u64 n, bool rc, u8 *out, u64 capacity) {
        if (n == 0) throw prim::Error(kErrNotFastx, "FASTA/Q input is empty");
        u8 first, lastb;
        prim::d2h(&first, in, 1);
        prim::d2h(&lastb, in + n - 1, 1);
        if (first != '>' && first != '@') throw prim::Error(kErrNotFastx, "the input is not in FASTA/Q format (first byte is neither '>' nor '@')");
        // line table
        const u64 nw = n / 64 + 2;
        DBuf<u64> words(nw);
        DBuf<idx_t> base(nw + 1);
        words.zero();
        prim::for_each((n + 63) / 64, FxNlWordsFn{in, n, words.p}, "fastx.newlines");
        const u64 n_nl = (u64)prim::exclusive_scan<idx_t>(nw, PopcIn{words.p}, base.p, true, "fastx.newlines");
        const bool tail = lastb != 10;                               // a last line without '\n'
        const u64 m = n_nl + (tail ? 1 : 0);
        DBuf<u64> nlpos(m);
        prim::for_each((n + 63) / 64, FxNlPosFn{words.p, base.p, nlpos.p}, "fastx.line_ends");
        if (tail) prim::h2d(nlpos.p + (m - 1), &n, 8);
        // layout: FASTA (no '+' line) or strict four-line FASTQ take the parallel classification; anything else is walked
        // record by record like kseq does
        const bool fastq = prim::reduce_sum<u64>(m, FxPlusIn{in, nlpos.p}, "fastx.plus_lines") != 0;
        u64 fq_lines = 0;
        bool regular = true;
        if (fastq) {
            fq_lines = prim::reduce_max<u64>(m, FxLastNonEmptyIn{nlpos.p}, "fastx.last_line");
            regular = fq_lines % 4 == 0 && prim::reduce_sum<u64>(fq_lines / 4, FxFastqBadIn{in, nlpos.p}, "fastx.fastq_check") == 0;
        }
        DBuf<u8> cls(m);
        DBuf<u64> raw(m), hrank(m + 1), praw(m + 1), hline, kept(m), pk(m + 1);
        u64 R, K;
        if (regular) {
            prim::for_each(m, FxClassFn{in, nlpos.p, fq_lines, fastq, cls.p, raw.p}, "fastx.classify");
            R = prim::exclusive_scan<u64>(m, FxIsHdrIn{cls.p}, hrank.p, true, "fastx.records");
            prim::exclusive_scan_nosync<u64>(m, IdxIn<u64>{raw.p}, praw.p, true, "fastx.raw_offsets");
            hline.alloc(R + 1);
            prim::for_each(m, FxHdrLinesFn{cls.p, hrank.p, m, R, hline.p}, "fastx.header_lines");
            prim::for_each(m, FxKeptFn{in, nlpos.p, cls.p, hrank.p, hline.p, praw.p, kept.p}, "fastx.kept");
        } else {
            DBuf<u64> res(1);
            cls.zero(); kept.zero();
            prim::for_each(1, FxSeqClassFn{in, n, nlpos.p, m, tail, cls.p, kept.p, res.p}, "fastx.record_walk");
            prim::for_each(m, FxStopFn{res.get(0), cls.p, kept.p}, "fastx.record_walk");
            R = prim::exclusive_scan<u64>(m, FxIsHdrIn{cls.p}, hrank.p, true, "fastx.records");
            hline.alloc(R + 1);
            prim::for_each(m, FxHdrLinesFn{cls.p, hrank.p, m, R, hline.p}, "fastx.header_lines");
        }
        K = prim::exclusive_scan<u64>(m, IdxIn<u64>{kept.p}, pk.p, true, "fastx.offsets");
        raw.release(); praw.release();
        // a header character that ends the input ("...\n>") opens no record (kseq.h:188: nothing to read after it)
        if (regular && R && tail) {
            const u64 hl = hline.get(R - 1);
            if (hl == m - 1 && n - (hl ? nlpos.get(hl - 1) + 1 : 0) == 1) R--;
        }
        FastxInfo info;
        info.fastq = fastq ? 1 : 0;
        info.n_strings = rc ? 2 * R : R;
        info.n_out = rc ? 2 * K + 2 * R : K + R;
        if (info.n_out > capacity) throw prim::Error(-22, "fastx: output buffer too small");
        DBuf<FxLineRec> rec(m);
        prim::for_each(m, FxLineRecFn{nlpos.p, cls.p, hrank.p, hline.p, kept.p, pk.p, rc, rec.p}, "fastx.line_records");
        DBuf<u64> bad(2);
        const u64 none[2] = {~0ull, 0};
        prim::h2d(bad.p, none, 16);
        prim::for_each(n, FxCopyFn{in, n, words.p, base.p, rec.p, hrank.p, rc, out, bad.p}, "fastx.copy");
        prim::for_each(R, FxSepFn{hline.p, pk.p, rc, (u8)10, out}, "fastx.separators");
        if (rc) {
            const u64 br = bad.get(0);
            if (br != ~0ull) {
                const u64 l0 = hline.get(br), l1 = hline.get(br + 1);
                const u64 x0 = l0 ? nlpos.get(l0 - 1) + 1 : 0, x1 = l1 < m ? nlpos.get(l1 - 1) + 1 : n;
                prim::for_each(x1 - x0, FxBadPosFn{in, words.p, base.p, rec.p, hrank.p, br, x0, bad.p + 1}, "fastx.bad_symbol");
                const u64 bp = bad.get(1);
                u8 c = '?';
                if (bp) prim::d2h(&c, in + (bp - 1), 1);
                throw prim::Error(kErrNotDna, std::string("The input seems not to be DNA (invalid symbol:") + (char)c + ")");
            }
        }
        prim::sync();
        return info;
    }

    // number of symbols an image describes, summed in 64 bits whatever the index width of the caller
    static u64 image_total_symbols(const void *dev_image, u64 image_bytes) {
        ImageHeader h = image_header(dev_image, image_bytes);
        return prim::reduce_sum<u64>(h.R, ImageLenIn{(const u8 *)dev_image, (u32)h.sb, (u32)h.fb}, "image.total");
    }
    // grl2plain (scripts/grl2plain.cpp): the plain BWT, one byte per symbol; null_char >= 0 replaces symbol 0
    static u64 image_plain(const void *dev_image, u64 image_bytes, u8 *dev_out, u64 capacity, int null_char) {
        ImageHeader h = image_header(dev_image, image_bytes);
        DBuf<u32> rsym(h.R);
        DBuf<idx_t> rlen(h.R), rpos(h.R + 1);
        prim::for_each(h.R, UnpackRunsFn{(const u8 *)dev_image, (u32)h.sb, (u32)h.fb, rsym.p, rlen.p}, "plain.unpack");
        u64 n = (u64)prim::exclusive_scan<idx_t>(h.R, IdxIn<idx_t>{rlen.p}, rpos.p, true, "plain.positions");
        if (n > capacity) throw prim::Error(-22, "grl2plain: output buffer too small");
        RankBits rb;
        build_rankbits(rb, rpos.p, h.R, n + 1, "plain.runbits");
        prim::for_each(n, PlainRunsFn{rsym.p, rb.words.p, rb.base.p, null_char, dev_out}, "plain.expand");
        prim::sync();
        return n;
    }
    // grlbwt2rle (scripts/grlbwt2rle.cpp): the two RLE arrays
    static u64 image_rle(const void *dev_image, u64 image_bytes, u8 *dev_syms, u32 *dev_lens, u64 capacity_runs) {
        ImageHeader h = image_header(dev_image, image_bytes);
        if (h.R > capacity_runs) throw prim::Error(-22, "grlbwt2rle: output buffers too small");
        DBuf<u32> rsym(h.R);
        DBuf<idx_t> rlen(h.R);
        prim::for_each(h.R, UnpackRunsFn{(const u8 *)dev_image, (u32)h.sb, (u32)h.fb, rsym.p, rlen.p}, "rle.unpack");
        prim::for_each(h.R, RleExportFn{rsym.p, rlen.p, dev_syms, dev_lens}, "rle.export");
        prim::sync();
        return h.R;
    }
    // bwt_stats (scripts/bwt_stats.cpp:18-98), byte alphabets (the reference indexes 256-entry tables by the symbol)
    struct ImageStats {
        u64 n_runs, sigma, text_size, min_run, max_run, fit1, fit2, fit3;
        u64 runs_of[256], freq_of[256];
        u64 deciles[9];
        u64 non_maximal;
    };
    static void image_stats(const void *dev_image, u64 image_bytes, ImageStats &st) {
        ImageHeader h = image_header(dev_image, image_bytes);
        if (h.sb != 1) throw prim::Error(-22, "bwt_stats: byte alphabets only (sb == 1)");
        if (h.R == 0) throw prim::Error(-22, "bwt_stats: empty BWT");
        const u64 R = h.R;
        DBuf<u32> rsym(R);
        DBuf<idx_t> rlen(R);
        prim::for_each(R, UnpackRunsFn{(const u8 *)dev_image, (u32)h.sb, (u32)h.fb, rsym.p, rlen.p}, "stats.unpack");
        st.n_runs = R;
        st.min_run = prim::reduce_min<u64>(R, IdxIn64{rlen.p}, "stats.min_run");
        st.max_run = prim::reduce_max<u64>(R, IdxIn64{rlen.p}, "stats.max_run");
        st.text_size = prim::reduce_sum<u64>(R, IdxIn64{rlen.p}, "stats.text_size");
        st.fit1 = prim::reduce_sum<u64>(R, LenFitIn{rlen.p, 1}, "stats.fit1");
        st.fit2 = prim::reduce_sum<u64>(R, LenFitIn{rlen.p, 2}, "stats.fit2");
        st.fit3 = R - st.fit1 - st.fit2;
        st.non_maximal = R - prim::reduce_sum<u64>(R, SymHeadIn64{rsym.p}, "stats.maximal");
        DBuf<u64> bins(256);
        bins.zero();
        prim::for_each_agg(R, RunSymFn{rsym.p}, BinAddFn{bins.p}, true, "stats.runs_per_symbol");
        std::vector<u64> hb = bins.to_host(256);
        st.sigma = 0;
        for (int c = 0; c < 256; c++) {
            st.runs_of[c] = hb[c];
            st.freq_of[c] = hb[c] ? prim::reduce_sum<u64>(R, SepLenIn{rsym.p, rlen.p, (u32)c}, "stats.freq") : 0;
            if ((hb[c] & 0xFFu) != 0) st.sigma++;       // the reference iterates the table as unsigned char (:57-62)
        }
        // deciles of the sorted run lengths (uint32 like the reference's vector), index ceil(R * k/10) with the
        // reference's accumulated double (:83-89); an index past the end (R < 10) is clamped to the last run
        DBuf<u32> ka(R), kb(R), va(R), vb(R);
        prim::for_each(R, LenKeyFn{rlen.p, ka.p, va.p}, "stats.len_keys");
        int res = prim::sort_pairs<u32, u32>(ka.p, va.p, kb.p, vb.p, R, 0, 32, "stats.sort");
        const DBuf<u32> &sorted = res ? kb : ka;
        double l = (double)R, prop = 0.1;
        for (int i = 0; i < 9; i++) {
            u64 q = (u64)std::ceil(l * prop);
            if (q >= R) q = R - 1;
            st.deciles[i] = sorted.get(q);
            prop += 0.1;
        }
    }

    // split_runs (scripts/split_runs.cpp): re-encode an image with run lengths of at most 2^bits - 1 and, for
    // block_size > 0, no run crossing a multiple of block_size; run-length field of ceil(bits/8) bytes.  block_size 0
    // = no partition (what the reference's usage text promises; its own code asserts there).
    struct SplitInfo { u64 runs_before, runs_after, overflow_splits, block_splits, n_syms, out_bytes; };
    static SplitInfo image_split_runs(const void *dev_image, u64 image_bytes, int bits, u64 block_size, u8 *dev_out, u64 capacity_bytes) {
        ImageHeader h = image_header(dev_image, image_bytes);
        if (bits < 1 || bits > 63) throw prim::Error(-22, "split_runs: bits must be in [1, 63]");
        const u64 L = (1ull << bits) - 1, B = block_size, R = h.R;
        const u32 fb2 = (u32)((bits + 7) / 8);
        DBuf<u32> rsym(R);
        DBuf<idx_t> rlen(R), rpos(R + 1), jbase(R + 1);
        prim::for_each(R, UnpackRunsFn{(const u8 *)dev_image, (u32)h.sb, (u32)h.fb, rsym.p, rlen.p}, "split.unpack");
        const u64 n = (u64)prim::exclusive_scan<idx_t>(R, IdxIn<idx_t>{rlen.p}, rpos.p, true, "split.positions");
        const u64 J = (u64)prim::exclusive_scan<idx_t>(R, PieceCountFn{rlen.p, L}, jbase.p, true, "split.pieces");
        const u64 cuts = (B && n > 0) ? (n - 1) / B : 0;
        SplitInfo si{R, J + cuts, J - R, cuts, n, 16 + (J + cuts) * (u64)(h.sb + fb2)};
        if (si.out_bytes > capacity_bytes) throw prim::Error(-22, "split_runs: output buffer too small");
        RankBits jb;
        build_rankbits(jb, jbase.p, R, J + 1, "split.piecebits");
        DBuf<u32> osym(si.runs_after);
        DBuf<idx_t> olen(si.runs_after);
        prim::for_each(J, SplitRunsFn{rsym.p, rlen.p, rpos.p, jbase.p, jb.words.p, jb.base.p, L, B, osym.p, olen.p}, "split.emit");
        u8 hdr[16] = {0};
        for (int i = 0; i < 8; i++) { hdr[i] = (u8)(h.sb >> (8 * i)); hdr[8 + i] = (u8)((u64)fb2 >> (8 * i)); }
        prim::h2d(dev_out, hdr, 16);
        prim::for_each(si.runs_after, PackRunsFn{osym.p, RunLen{olen.p, nullptr}, (u32)h.sb, fb2, dev_out, 16u}, "split.pack");
        prim::sync();
        return si;
    }

    void run_all() {
        parse_phase();
        induce_phase();
        finish();
    }
};

}   // namespace GRL_NS

/*
 * grlbwt_hip.h -- C-ABI of the MI355X-native BCR-BWT engine (libgrlbwt_hip.so).
 *
 * Drop-in boundary for the parse-then-induce hot path of ddiazdom/grlBWT.  The
 * reference has no FFI/plugin interface: its seam for this
 * path is the pair of C++ entry points called by grl_bwt_algo
 * (include/grl_bwt.hpp:23-79):
 *
 *     size_t exact_algo::par_phase<sym_type>(i_file, n_threads, hbuff_frac, ws)   lib/exact_algo/exact_par_phase.cpp:285-372
 *     void   exact_algo::ind_phase<b_f_r>(ws, p_round)                            lib/exact_algo/exact_ind_phase.cpp:674-697
 *
 * plus the file formats at both ends (raw cells in, `.rl_bwt` out, include/bwt_io.h).
 * Each entry point below names the reference function(s) it replaces.  Plain C
 * types only; one context per GPU; calls on one context must be serialised by the
 * caller; every function returns 0 or a negative GRLBWT_E* code, never throws and
 * never exits; host buffers are borrowed for the duration of the call; device
 * buffers are owned by the context and released by grlbwt_ctx_destroy.
 *
 * There is no CPU fallback: without a usable HIP device grlbwt_ctx_create fails
 * with GRLBWT_EDEVICE.
 *
 * Process model: ONE context (= one GPU) per process, as in the one-process-per-GPU launch of
 * torch.distributed; the device, the engine's HIP stream and its slab allocator are process-wide.
 * The engine runs on its own non-blocking stream: data handed over with grlbwt_text_attach_device must
 * be complete (synchronise the producing stream first), results are complete when a call returns.
 */
#ifndef GRLBWT_HIP_H
#define GRLBWT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRLBWT_ABI_VERSION 3

#define GRLBWT_OK 0
#define GRLBWT_EINVAL (-22)     /* bad argument / call out of order                                  */
#define GRLBWT_EDEVICE (-5)     /* HIP runtime error (no device, launch failure, ...)                */
#define GRLBWT_ENOMEM (-12)     /* device or host allocation failed                                  */
#define GRLBWT_EILLFORMED (-84) /* reference: "Error: the file is ill formed", exit(1) (utils.cpp:177-180) */
#define GRLBWT_ERANGE (-75)     /* input beyond what this build supports -- the limits, all checked:
                                 *   collection            < 2^40 cells (64-bit positions from 2^32 - 256 cells on);
                                 *   symbols, and the alphabet of every level (metasymbols of a round)       < 2^30;
                                 *   distinct phrases of one round < 2^32, their symbols (the round's dictionary) < 2^32
                                 *   (collection-level mode: per rank's part of the merged dictionary and per key range of its
                                 *   suffixes, while every phrase frequency is < 2^32 and no phrase has >= 512 cells);
                                 *   phrase-table slots of one round <= 2^31 (2^30 with the hot table of level 0);
                                 *   phrase OCCURRENCES of one round: no bound of their own in the 64-bit build (level 0 of a
                                 *   24.9 GB collection has 7.3 G); a shard of the collection-level mode < 2^32 per round.
                                 * The reference takes dictionaries up to 64-bit suffix-array cells (exact_par_phase.cpp:244-263). */
#define GRLBWT_ENOSPC (-28)     /* phrase table overflow                                              */
#define GRLBWT_EINTERNAL (-71)  /* internal consistency check failed                                  */
#define GRLBWT_ENOTDNA (-86)    /* reference: "The input seems not to be DNA (invalid symbol:X)", exit(1) (fastx_handler.cpp:30-33) */

/* ctx flags */
#define GRLBWT_FLAG_KEEP_LEVELS 1u   /* keep every level's text and BWT for parity inspection        */
#define GRLBWT_FLAG_SYNC_DEBUG 2u    /* synchronise after every kernel launch (fault localisation)   */
#define GRLBWT_FLAG_FORCE_IDX64 4u   /* 64-bit positions/lengths whatever the input size (sharded collections, tests) */
#define GRLBWT_FLAG_CLASSIC_POOL 8u  /* device memory from hipMalloc slabs, not from the on-demand virtual-memory arena:
                                      * for buffers that are handed to a communication library (multi-process RCCL) */

typedef struct grlbwt_ctx grlbwt_ctx;

/* collection_stats<T> result (external/cdt/lib/utils.cpp:100-189, include/utils.h:20-28) + final header */
typedef struct grlbwt_stats {
    uint64_t n_strings, n_syms, min_sym, max_sym, max_sym_freq;
    uint64_t sb, fb;            /* bytes per run symbol / run length in the final .rl_bwt */
} grlbwt_stats;

/* the "Stats:" block of par_round (lib/exact_algo/exact_par_phase.cpp:484-488) */
typedef struct grlbwt_round_info {
    uint64_t n_in;              /* cells of the round's input text                  */
    uint64_t n_phrases;         /* "Parsing phrases" (distinct)                     */
    uint64_t dict_syms;         /* "Number of symbols in the phrases"               */
    uint64_t n_metasyms;        /* "Number of unsolved BWT blocks" (tot_phrases)    */
    uint64_t parse_size;        /* "Parse size"                                     */
    uint64_t sigma;             /* alphabet size of the input text of the round     */
    uint64_t max_phrase_len, sort_iters;
} grlbwt_round_info;

/* the "Stats:" block of infer_lvl_bwt (lib/exact_algo/exact_ind_phase.cpp:372-384) + kernel shape */
typedef struct grlbwt_level_info {
    uint64_t n;                 /* "BWT size (n)"                                   */
    uint64_t n_runs;            /* "Number of runs (r)"                             */
    uint64_t runs_next;         /* runs of BWT_{r+1} scanned                        */
    uint64_t induced_cells;     /* chain steps + TAKE cells scattered (pass B)      */
    uint64_t prebwt_runs, segments, atoms;
    /* SURVEY 8d's E'_r (grammar-chain steps) and E_r (cells after the in-bucket merge = n_runs of compute_hocc_size,
     * exact_ind_phase.cpp:42-109); counted only while grlbwt_profile_enable(1) is on, else 0 */
    uint64_t chain_steps, merged_cells;
} grlbwt_level_info;

/* wall-clock seconds per stage (stream-synchronised) and algorithmic byte counts */
typedef struct grlbwt_counters {
    double t_stats, t_classify, t_hash, t_dict_sort, t_dict_groups, t_emit;
    double t_ind_expand, t_ind_split, t_ind_assemble, t_finish;
    uint64_t bytes_classify_hash;   /* sum_r n_r*w_r  (one read of every level's text per scan) */
    uint64_t bytes_emit;            /* sum_r n_{r+1}*4 written + read back                                 */
    uint64_t bytes_induce_scatter;  /* sum_r R_{r+1}*(4+idx) + E_r*(4+idx)   (pass A+B formula) */
    uint64_t bytes_induce_assemble; /* sum_r P_r + E_r + R_{r+1} read + R_r written, (4+idx) B each        */
    uint64_t idx_bytes;             /* 4 or 8: width of positions/lengths in HBM for this input            */
} grlbwt_counters;

/* ---- lifetime ------------------------------------------------------------ */
int grlbwt_abi_version(void);
/* "hip-gfx950" for the product library.  (The serial stand-in that the CPU test-suite builds under
 * tests/hostsim answers "serial-test-standin"; the host mirrors refuse it unless a test asks for it.) */
const char *grlbwt_backend_name(void);
const char *grlbwt_strerror(int code);
/* last error message of this context (valid until the next call on it) */
const char *grlbwt_last_error(const grlbwt_ctx *ctx);
/* replaces tmp_workspace construction in run_int (main.cpp:80-96): all level files
 * of the reference's workspace become device buffers owned by the context.       */
int grlbwt_ctx_create(int device_id, uint32_t flags, grlbwt_ctx **out);
void grlbwt_ctx_destroy(grlbwt_ctx *ctx);
/* run the engine's kernels on a caller-provided hipStream_t (e.g. torch's current stream) */
int grlbwt_ctx_set_stream(grlbwt_ctx *ctx, void *hip_stream);

/* ---- input: replaces collection_stats<sym_type>(i_file) (utils.cpp:100-189) and
 * the i_file_stream reads of the first round (file_streams.hpp:93-105) ---------- */
/* copy n_cells cells of cell_bytes in {1,2,4,8} from host memory to HBM and scan them */
int grlbwt_text_upload(grlbwt_ctx *ctx, const void *host_cells, uint64_t n_cells, int cell_bytes);
/* the same from a file of raw cells (the reference's i_file_stream + collection_stats, file_streams.hpp:93-105,
 * utils.cpp:100-189): chunks are read into pinned staging buffers by reader threads and copied to HBM while the next
 * chunk is read; for byte cells the symbol histogram is taken per chunk on the device behind the copies.  A file whose
 * size is 0 or not a multiple of cell_bytes is ill formed (GRLBWT_EILLFORMED). */
int grlbwt_text_load_file(grlbwt_ctx *ctx, const char *path, int cell_bytes);
/* the same for bytes [offset_bytes, offset_bytes + n_bytes) of the file: a record shard of the collection-level mode
 * (the caller cuts at record boundaries; both numbers are multiples of cell_bytes) */
int grlbwt_text_load_file_range(grlbwt_ctx *ctx, const char *path, uint64_t offset_bytes, uint64_t n_bytes, int cell_bytes);
/* use cells already resident in HBM (borrowed until the context is reset/destroyed; 16-byte aligned).  The statistics are
 * taken here: the buffer must not change between this call and the end of the build (a build that finds a cell value in it
 * that was not there fails with GRLBWT_EINVAL; other changes give the image of neither text). */
int grlbwt_text_attach_device(grlbwt_ctx *ctx, const void *dev_cells, uint64_t n_cells, int cell_bytes);
int grlbwt_get_stats(const grlbwt_ctx *ctx, grlbwt_stats *out);

/* ---- FASTA/FASTQ ingestion: replaces is_fastx / check_gzip (external/cdt/lib/utils.cpp:13-30,53-60)
 * and fastx2plain_format (external/bioparsers/lib/fastx_handler.cpp:7-58, kseq.h:179-220), which the reference's
 * main.cpp:118-136 has switched off.  The file (gzip members are inflated on the host) goes to HBM as it is; the records
 * become the one-string-per-line text on the device: every record's sequence lines joined, '\n' behind it, and with
 * GRLBWT_FASTX_REVCOMP its reverse complement + '\n' as a second string (a symbol outside ACGT: GRLBWT_ENOTDNA, with the
 * reference's message naming the symbol).  FASTA with any line wrapping and four-line FASTQ are classified in parallel;
 * other layouts (FASTQ over several lines, damaged records) are walked record by record like kseq does, with kseq's
 * outcome (a record whose quality string has the wrong length ends the conversion). */
#define GRLBWT_FASTX_REVCOMP 1u
int grlbwt_fastx_probe(const char *path, int *is_fastx, int *is_gz);
int grlbwt_text_load_fastx(grlbwt_ctx *ctx, const char *path, uint32_t fastx_flags, uint64_t *n_strings);
/* the conversion alone, device buffer to device buffer (capacity >= n_in, or 2 * n_in with reverse complements) */
int grlbwt_fastx_convert_device(grlbwt_ctx *ctx, const void *dev_in, uint64_t n_in, uint32_t fastx_flags, void *dev_out,
                                uint64_t capacity, uint64_t *n_out, uint64_t *n_strings);

/* ---- parsing phase ------------------------------------------------------- */
/* one par_round (exact_par_phase.cpp:374-497): LMS breaks, phrase hashing, dictionary
 * sort, pre-BWT, grammar, parse emission.  *done = 1 after the last round (:496). */
int grlbwt_parse_round(grlbwt_ctx *ctx, grlbwt_round_info *info, int *done);
/* exact_algo::par_phase (exact_par_phase.cpp:285-372): all rounds; *n_rounds = rounds run */
int grlbwt_parse_phase(grlbwt_ctx *ctx, int *n_rounds);
int grlbwt_round_info_get(const grlbwt_ctx *ctx, int round, grlbwt_round_info *info);

/* ---- induction phase ------------------------------------------------------ */
/* parse2bwt (exact_ind_phase.cpp:603-672) */
int grlbwt_induce_first(grlbwt_ctx *ctx);
/* infer_lvl_bwt<b> (exact_ind_phase.cpp:111-386) for the next level down; *level = level produced.
 * The reference's -b/--run-len-bytes only sizes its bucket cells and never changes the
 * result; lengths here are full-width, so there is no b parameter. */
int grlbwt_induce_level(grlbwt_ctx *ctx, int *level, grlbwt_level_info *info);
/* exact_algo::ind_phase<b> (exact_ind_phase.cpp:674-697): parse2bwt + every level + .rl_bwt image */
int grlbwt_induce_phase(grlbwt_ctx *ctx);
int grlbwt_level_info_get(const grlbwt_ctx *ctx, int level, grlbwt_level_info *info);

/* ---- whole path: grl_bwt_algo<sym_type,false> (include/grl_bwt.hpp:23-79) -- */
int grlbwt_build(grlbwt_ctx *ctx);   /* parse_phase + induce_phase on the loaded text */

/* ---- output: replaces bwt_buff_writer + rename(bwt_lev_0 -> o_file)
 * (include/bwt_io.h:174-568, include/grl_bwt.hpp:77) ------------------------- */
int grlbwt_result_size(const grlbwt_ctx *ctx, uint64_t *image_bytes, uint64_t *n_runs);
/* device pointer of the .rl_bwt image (16-byte header + records), valid until reset/destroy */
int grlbwt_result_device_ptr(const grlbwt_ctx *ctx, const void **dev_ptr);
int grlbwt_result_download(const grlbwt_ctx *ctx, void *host_out, uint64_t capacity);
/* (pinned double-buffered download, the file is written while the next chunk comes down) */
int grlbwt_result_write_file(const grlbwt_ctx *ctx, const char *path);
/* What this context holds of the image: all of it -- (0, image_bytes) -- after grlbwt_build and grlbwt_dist_build, or this
 * rank's part after a grlbwt_dist_build with GRLBWT_COMM_KEEP_PARTS (bytes may be 0 on a rank whose slice merged into a
 * neighbour's run). */
int grlbwt_result_part(const grlbwt_ctx *ctx, uint64_t *offset, uint64_t *bytes);
/* The part at its offset of `path` (created if missing; the rank whose part ends the image sets the file's size, so a longer
 * file that was there before keeps no tail).  The file is a valid image only once EVERY rank has returned GRLBWT_OK: the caller
 * writes to a temporary name and publishes the complete file -- a barrier and a rename -- as the grlbwt executable's --gpus does. */
int grlbwt_result_write_part(const grlbwt_ctx *ctx, const char *path);

/* ---- .rl_bwt consumers: scripts/grl2plain.cpp (expand the runs) + scripts/reverse_bwt.cpp with
 * scripts/fm_index.h:79-83 (LF walk), on the device.  Rebuilds the collection (strings in input order,
 * cell_bytes-wide cells) from an .rl_bwt image in device memory into dev_text_out; used as the
 * encode -> decode round-trip check at full benchmark sizes.  64-bit positions are used when
 * the image describes >= 2^32 - 256 symbols (pass n_hint = 0 if unknown: decided from the file size). */
int grlbwt_invert_image(grlbwt_ctx *ctx, const void *dev_image, uint64_t image_bytes, int cell_bytes,
                        void *dev_text_out, uint64_t capacity_cells, uint64_t *n_cells_out);

/* The same LF walk (scripts/reverse_bwt.cpp:36-52), stopped after tail_cells steps per string: slot i of dev_out (tail_cells cells wide)
 * receives the LAST min(length, tail_cells) cells of string i, separator included, right-aligned; the cells in front of them are
 * left as they were.  *n_strings_out = strings, *n_cells_out = cells written.  For collections whose strings are too long to walk
 * end to end in a test's time (100 strings of 249 M cells: 249 M dependent steps each): the ends of all strings check the ORDER of
 * the BWT, which a symbol-count comparison does not.  capacity_cells >= strings * tail_cells. */
int grlbwt_invert_image_tails(grlbwt_ctx *ctx, const void *dev_image, uint64_t image_bytes, int cell_bytes, uint64_t tail_cells,
                              void *dev_out, uint64_t capacity_cells, uint64_t *n_strings_out, uint64_t *n_cells_out);

/* The other .rl_bwt consumers of the reference's scripts/ (SURVEY 8f-2), on an image in device memory:
 * grl2plain  (scripts/grl2plain.cpp:18-50): plain BWT, one byte per symbol ((char)sym); null_char >= 0 replaces
 *            symbol 0, -1 leaves it.
 * grlbwt2rle (scripts/grlbwt2rle.cpp:17-33): run symbols as uint8 and run lengths as uint32.
 * bwt_stats  (scripts/bwt_stats.cpp:18-98): run statistics of a byte-alphabet image; sigma reproduces the script's
 *            count (table entries whose low byte is non-zero, :57-62); decile indices past the end are clamped. */
typedef struct grlbwt_image_stats {
    uint64_t n_runs, sigma, text_size, min_run, max_run;
    uint64_t fit1, fit2, fit3;            /* runs whose length fits 1 byte / 2 bytes / needs 3 or more */
    uint64_t runs_of[256], freq_of[256];  /* per symbol: number of runs, total length */
    uint64_t deciles[9];                  /* sorted run lengths at ceil(n_runs * k/10), k = 1..9 */
    uint64_t non_maximal;                 /* runs with the same symbol as the run before them (0 for a grlBWT output) */
} grlbwt_image_stats;
int grlbwt_image_plain(grlbwt_ctx *ctx, const void *dev_image, uint64_t image_bytes, void *dev_out_u8,
                       uint64_t capacity, int null_char, uint64_t *n_out);
int grlbwt_image_rle(grlbwt_ctx *ctx, const void *dev_image, uint64_t image_bytes, void *dev_syms_u8, void *dev_lens_u32,
                     uint64_t capacity_runs, uint64_t *n_runs_out);
int grlbwt_image_stats_get(grlbwt_ctx *ctx, const void *dev_image, uint64_t image_bytes, grlbwt_image_stats *out);
/* split_runs (scripts/split_runs.cpp:44-125; its argv order is `file.rlbwt bits n output_file`): re-encode the image so
 * that no run is longer than 2^bits - 1 and, for block_size > 0, no run crosses a multiple of block_size; the output is
 * an .rl_bwt image with the input's symbol width and ceil(bits/8) length bytes, byte-identical to the reference's output
 * file -- including the zero-length record the reference emits in front of a piece that starts exactly on a block
 * boundary (:87-90).  block_size 0 = no partition (the reference asserts on it).  1 <= bits <= 63.
 * dev_out needs 16 + runs_after * (sb + ceil(bits/8)) bytes; runs_after = runs + overflow_splits + block_splits. */
typedef struct grlbwt_split_info {
    uint64_t runs_before, runs_after;     /* "Number of runs before" / "Number of runs now" */
    uint64_t overflow_splits, block_splits;
    uint64_t n_syms, n_blocks;            /* symbols described; "Number of blocks" = ceil(n_syms / block_size) */
    uint64_t out_bytes;
} grlbwt_split_info;
int grlbwt_image_split_runs(grlbwt_ctx *ctx, const void *dev_image, uint64_t image_bytes, int bits, uint64_t block_size,
                            void *dev_out, uint64_t capacity_bytes, grlbwt_split_info *info);

/* ---- inspection (parity tests; need GRLBWT_FLAG_KEEP_LEVELS) --------------- */
/* text of level >= 1 as (rank<<1 | rep) cells, the reference's on-disk parse format */
int grlbwt_level_text_size(const grlbwt_ctx *ctx, int level, uint64_t *n_cells);
int grlbwt_level_text_download(const grlbwt_ctx *ctx, int level, uint64_t *cells_out);
int grlbwt_level_bwt_size(const grlbwt_ctx *ctx, int level, uint64_t *n_runs);
int grlbwt_level_bwt_download(const grlbwt_ctx *ctx, int level, uint64_t *sym_out, uint64_t *len_out);

/* the level's grammar (produce_grammar, exact_par_phase.cpp:14-95: two cells per metasymbol, g1 >= sigma+3 = nested
 * metasymbol + sigma+3; has_hocc = phrases_has_hocc) and pre-BWT runs (produce_pre_bwt, :136-242: BWT marker = sigma+1,
 * hocc marker = sigma+2), i.e. the contents of the reference's dict_lev_r / pre_bwt_lev_r files.  Available for every
 * level once the parsing phase is done and until that level has been induced. */
int grlbwt_level_grammar_size(const grlbwt_ctx *ctx, int level, uint64_t *n_metasyms, uint64_t *prebwt_runs);
int grlbwt_level_grammar_download(const grlbwt_ctx *ctx, int level, uint64_t *g0, uint64_t *g1, uint8_t *has_hocc,
                                  uint64_t *prebwt_sym, uint64_t *prebwt_len);

int grlbwt_get_counters(const grlbwt_ctx *ctx, grlbwt_counters *out);
/* device memory taken by the engine's slab allocator: peak bytes in use, bytes reserved from the runtime */
int grlbwt_memory_usage(const grlbwt_ctx *ctx, uint64_t *peak_live_bytes, uint64_t *reserved_bytes);

/* ---- collection-level multi-GPU ------------------------
 * One context per GPU/process; the collection is sharded by record: rank g loads (text_upload /
 * text_attach_device) a contiguous range of whole strings, ranks in collection order.  The engine
 * calls back into the host for the exchanges (the host implements them with torch.distributed:
 * RCCL over xGMI on GPUs, gloo in the CPU tests); all pointers are device pointers of this
 * engine's device.  Replaces the thread-range strategy of mt_parse_strat_t
 * (include/parsing_strategies.h:200-275,277-386): hashing and parse emission stay local to the
 * shard, the per-thread table merge (join_thread_phrases) becomes an all-gather + device merge.
 * The output is the BWT of the WHOLE collection, identical on every rank and for every N.   */
typedef struct grlbwt_comm {
    int rank, size;
    void *user;
    /* every rank contributes `bytes` bytes at `send`; `recv` gets size*bytes in rank order; 0 = ok */
    int (*allgather)(void *user, const void *send, void *recv, uint64_t bytes);
    /* variable all-to-all (MPI_Alltoallv shape, byte units, host arrays of `size` entries): send_bytes[g] bytes at
     * send + send_off[g] go to rank g; recv_bytes[g] bytes from rank g land at recv + recv_off[g].  The engine keeps every
     * block at or below 256 MiB (larger exchanges arrive as several calls; GRLBWT_A2A_BLOCK overrides the limit). */
    int (*alltoallv)(void *user, const void *send, const uint64_t *send_bytes, const uint64_t *send_off,
                     void *recv, const uint64_t *recv_bytes, const uint64_t *recv_off);
    /* GRLBWT_COMM_STREAM_ORDERED: the callbacks enqueue the exchange on the context's stream
     * (grlbwt_ctx_set_stream) and return without waiting; the engine then neither drains its stream
     * before a callback nor expects the data to be complete when it returns -- later work on the same
     * stream is ordered behind it (RCCL through torch.distributed with that stream current).
     * 0: the callback may use any stream or the host; the engine synchronises before calling it and
     * the callback returns only when `recv` is complete (gloo, host staging).                        */
    uint32_t flags;
} grlbwt_comm;
#define GRLBWT_COMM_STREAM_ORDERED 1u
/* GRLBWT_COMM_KEEP_PARTS: the image is NOT gathered at the end of the build; every rank keeps the part whose runs it induced --
 * bytes [offset, offset + bytes) of the image (grlbwt_result_part; rank 0's part starts with the 16-byte header), the parts in
 * rank order make up the file.  grlbwt_result_size still reports the whole image, grlbwt_result_device_ptr /
 * grlbwt_result_download give the part, grlbwt_result_write_part puts it at its offset of the output file: N ranks write one
 * file, nothing of the image crosses the fabric.  (The reference's writer is one stream: exact_ind_phase.cpp:287-361 fills
 * bwt_lev_0 front to back; the counterpart of its N threads here is N writers.) */
#define GRLBWT_COMM_KEEP_PARTS 2u
/* grl_bwt_algo over the sharded collection: par_phase with a dictionary merge per round, ind_phase, image */
int grlbwt_dist_build(grlbwt_ctx *ctx, const grlbwt_comm *comm);

/* The two callbacks over RCCL inside the library, for hosts without a communication framework of their own (the grlbwt
 * executable's --gpus N: one process per GPU, the counterpart of `-t N` reaching mt_parse_strat_t in the reference,
 * main.cpp:62, parsing_strategies.h:200-275).  librccl is loaded on first use.  ONE rank makes the 128-byte id and the
 * host hands it to the others (shared memory, a pipe, a file); comm_create is collective over all `size` ranks, each
 * with its own device, and fills *comm with stream-ordered callbacks: ncclAllGather, and one grouped ncclSend/ncclRecv
 * per peer for the all-to-all, enqueued on the context's stream. */
#define GRLBWT_RCCL_ID_BYTES 128
int grlbwt_rccl_unique_id(void *id128);
int grlbwt_rccl_comm_create(grlbwt_ctx *ctx, const void *id128, int rank, int size, grlbwt_comm *comm);
int grlbwt_rccl_comm_destroy(grlbwt_comm *comm);

/* per-kernel timing with HIP events on the engine's stream (bench.py's roofline leg).
 * enable(1) clears the table and starts recording; dump writes one line per kernel name:
 * "<name> <launches> <total_ms> <algorithmic_bytes>\n" (NUL terminated, truncated to capacity; bytes = 0 where
 * the launch site states none; names carry "#<level>"). */
int grlbwt_profile_enable(grlbwt_ctx *ctx, int on);
int grlbwt_profile_dump(grlbwt_ctx *ctx, char *buf, uint64_t capacity);

/* device self-test of the primitives (scan, radix sort, ballot bit-vectors) against
 * host loops on seeded data; returns 0 or the index (<0) of the failing check */
int grlbwt_selftest(grlbwt_ctx *ctx, uint64_t n, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif /* GRLBWT_HIP_H */

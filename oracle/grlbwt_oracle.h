/* grlbwt_oracle.h -- TEST INFRASTRUCTURE (see grlbwt_oracle.c header).
 * CPU restatement of grlBWT's parse-then-induce path used only as the
 * checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline. */
#ifndef GRLBWT_ORACLE_H
#define GRLBWT_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_OK 0
#define ORACLE_ERR_ARG (-22)
#define ORACLE_ERR_ILLFORMED (-84)   /* reference: "Error: the file is ill formed", exit(1) (utils.cpp:177-180) */

typedef struct oracle_result oracle_result;

/* cells: n cells of w in {1,2,4,8} bytes, little endian.  keep_trace != 0
 * keeps every level's text and BWT for stage-wise comparison.              */
oracle_result *oracle_run(const void *cells, uint64_t n, int w, int keep_trace);
int oracle_status(const oracle_result *R);
uint64_t oracle_out_size(const oracle_result *R);          /* bytes of the .rl_bwt image */
const uint8_t *oracle_out_bytes(const oracle_result *R);
int oracle_n_rounds(const oracle_result *R);
/* out = {n_strings, n_syms, min_sym, max_sym, max_sym_freq, longest_string, sb, fb} */
void oracle_stats(const oracle_result *R, uint64_t out[8]);
/* out = {n_in, D (phrases), S (dict symbols), M (metasymbols), parse_size, sigma} */
void oracle_round_counters(const oracle_result *R, int round, uint64_t out[6]);
uint64_t oracle_level_text(const oracle_result *R, int level, const uint64_t **sym, const uint8_t **rep);
uint64_t oracle_level_bwt(const oracle_result *R, int level, const uint64_t **sym, const uint64_t **len);
uint64_t oracle_level_prebwt(const oracle_result *R, int level, const uint64_t **sym, const uint64_t **len);
uint64_t oracle_level_grammar(const oracle_result *R, int level, const uint64_t **g0, const uint64_t **g1,
                              const uint8_t **has_hocc);
void oracle_free(oracle_result *R);

/* FASTA/FASTQ text (decompressed) -> one string per line, optionally with the reverse complements: the reference's
 * fastx2plain_format over kseq (fastx_oracle.c).  0 ok; 1 = a symbol without complement (*bad_sym); ORACLE_ERR_ARG = cap */
int oracle_fastx2plain(const uint8_t *in, uint64_t n, int rc, uint8_t sep, uint8_t *out, uint64_t cap, uint64_t *n_out,
                       uint64_t *n_strings, uint8_t *bad_sym);

#ifdef __cplusplus
}
#endif
#endif

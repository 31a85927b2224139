/* fastx_oracle.c -- TEST INFRASTRUCTURE.  Never linked by the product.
 *
 * CPU restatement of the reference's FASTA/FASTQ -> one-string-per-line converter on an in-memory buffer (the
 * decompressed file): fastx2plain_format (external/bioparsers/lib/fastx_handler.cpp:7-58) over kseq_read
 * (external/bioparsers/include/kseq.h:179-220) and ks_getuntil2 (kseq.h:93-144).  Pinned by the reference's own
 * converter built from its sources (oracle/Makefile target `ref`, oracle/ref_fastx_driver.cpp -> oracle/_ref/fastx2plain):
 * tests/golden/fastx_ref.json, tests/test_fastx.py. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "grlbwt_oracle.h"

typedef struct { const uint8_t *p; uint64_t n, i; } stream_t;
static int st_getc(stream_t *s) { return s->i < s->n ? (int)s->p[s->i++] : -1; }                 /* ks_getc, kseq.h:79-91 */

typedef struct { uint8_t *s; uint64_t l, m; } kstr_t;
static void ks_push(kstr_t *k, const uint8_t *src, uint64_t len) {
    if (k->m - k->l < len + 1) { k->m = (k->l + len + 1) * 2; k->s = (uint8_t *)realloc(k->s, k->m); }
    memcpy(k->s + k->l, src, len);
    k->l += len;
}
/* ks_getuntil2 with KS_SEP_LINE (kseq.h:93-144): rest of the line appended (or replacing); a trailing '\r' of the WHOLE
 * string is dropped when the string is longer than one byte (kseq.h:141); -1 if nothing could be read at end of input */
static int64_t getline_into(stream_t *s, kstr_t *k, int append) {
    if (!append) k->l = 0;
    if (s->i >= s->n) return -1;
    const uint8_t *b = s->p + s->i;
    const uint8_t *nl = (const uint8_t *)memchr(b, '\n', s->n - s->i);
    uint64_t len = nl ? (uint64_t)(nl - b) : s->n - s->i;
    ks_push(k, b, len);
    s->i += len + (nl ? 1 : 0);
    if (k->l > 1 && k->s[k->l - 1] == '\r') k->l--;
    return (int64_t)k->l;
}

/* kseq_read (kseq.h:179-220): >= 0 sequence length, -1 end of input, -2 truncated / mismatching quality */
typedef struct { kstr_t seq, qual; int last_char; } kseq_t;
static int64_t kseq_read(stream_t *s, kseq_t *q) {
    int c;
    if (q->last_char == 0) {                                       /* jump to the next header line (char by char) */
        while ((c = st_getc(s)) >= 0 && c != '>' && c != '@') {}
        if (c < 0) return -1;
        q->last_char = c;
    }
    q->seq.l = q->qual.l = 0;
    /* name: up to the first white space (ks_getuntil with delimiter 0 = isspace, kseq.h:116-118); comment: rest of the line */
    {
        int gotany = 0, d = -1;
        while (s->i < s->n) {
            gotany = 1;
            c = s->p[s->i++];
            if (c == ' ' || (c >= '\t' && c <= '\r')) { d = c; break; }
        }
        if (!gotany) return -1;                                    /* header char at the very end of the input */
        if (d != '\n' && d != -1) { while ((c = st_getc(s)) >= 0 && c != '\n') {} }
        /* (d == -1: the name ran into the end of the input: kseq then finds no comment and no sequence) */
    }
    while ((c = st_getc(s)) >= 0 && c != '>' && c != '+' && c != '@') {
        if (c == '\n') continue;                                   /* skip empty lines */
        uint8_t ch = (uint8_t)c;
        ks_push(&q->seq, &ch, 1);
        getline_into(s, &q->seq, 1);
    }
    if (c == '>' || c == '@') q->last_char = c;
    if (c != '+') return (int64_t)q->seq.l;                        /* FASTA */
    while ((c = st_getc(s)) >= 0 && c != '\n') {}                  /* rest of the '+' line */
    if (c == -1) return -2;
    while (getline_into(s, &q->qual, 1) >= 0 && q->qual.l < q->seq.l) {}
    q->last_char = 0;
    if (q->seq.l != q->qual.l) return -2;
    return (int64_t)q->seq.l;
}

/* dna_string::comp (external/bioparsers/lib/dna_string.cpp:6-14): A<->T, C<->G, 10 -> 10; everything else 0 = not DNA */
static uint8_t comp_of(uint8_t c) {
    switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; case 10: return 10;
                 case 149: return 168; case 151: return 155; case 155: return 151; case 168: return 149; default: return 0; }
}

/* fastx2plain_format (fastx_handler.cpp:7-58).  Returns 0, ORACLE_ERR_ARG when `cap` is too small, or 1 when a symbol has
 * no complement ("The input seems not to be DNA", exit(1)); then *bad_sym holds it and *n_out what had been written. */
int oracle_fastx2plain(const uint8_t *in, uint64_t n, int rc, uint8_t sep, uint8_t *out, uint64_t cap, uint64_t *n_out,
                       uint64_t *n_strings, uint8_t *bad_sym) {
    stream_t s = {in, n, 0};
    kseq_t q;
    memset(&q, 0, sizeof q);
    uint64_t o = 0, ns = 0;
    int ret = 0;
    while (ret == 0 && kseq_read(&s, &q) >= 0) {
        if (o + q.seq.l + 1 > cap) { ret = ORACLE_ERR_ARG; break; }
        memcpy(out + o, q.seq.s, q.seq.l);
        o += q.seq.l;
        out[o++] = sep;
        ns++;
        if (rc) {
            if (o + q.seq.l + 1 > cap) { ret = ORACLE_ERR_ARG; break; }
            for (uint64_t i = q.seq.l; i-- > 0;) {
                uint8_t c = (q.seq.s[i] < 170) ? comp_of(q.seq.s[i]) : 0;     /* (the reference indexes a 170-entry table) */
                if (c == 0) { if (bad_sym) *bad_sym = q.seq.s[i]; ret = 1; break; }
                out[o++] = c;
            }
            if (ret) break;
            out[o++] = sep;
            ns++;
        }
    }
    free(q.seq.s); free(q.qual.s);
    if (n_out) *n_out = o;
    if (n_strings) *n_strings = ns;
    return ret;
}

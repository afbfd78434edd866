/*
 * mf_oracle_wide.c -- CPU ORACLE, NO-REFERENCE EXTENSION for 32 <= k <= 63.  TEST INFRASTRUCTURE ONLY (see mf_oracle.h).
 *
 * The reference rejects k > 31 (src/tools/KmersCounterMain.java:66-73: one Java long per k-mer); BASELINE.json's config 4 has a
 * k = 63 leg.  This file is mf_oracle_core.inc -- the SAME text that, compiled with 64-bit keys in mf_oracle.c, is pinned by the
 * reference's golden matrix -- compiled with OKEY = unsigned __int128: counting (A2-A4), unitigs (A7), the cutter table (A9),
 * components (A10) and features (A12) on 2k-bit k-mers.  It replaces nothing in the reference and takes part in no parity claim;
 * it is the checker of metafast_amd/csrc/mf_wide.hip and mf_wgraph.hip.  It also runs at k <= 31, where
 * tests/test_oracle_wide_cpu.py requires it to agree with the pinned oracle on every output.
 *
 * Public names: or_* of the core become orw_* (the #defines below).  A 128-bit key crosses the C boundary as two uint64_t in memory
 * order (low word first, the layout of unsigned __int128 on x86-64 / numpy dtype [('lo','<u8'),('hi','<u8')]).
 */
#define _GNU_SOURCE
#include "mf_oracle.h"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static __thread char g_err[512];
const char *orw_last_error(void) { return g_err; }
static int fail(const char *fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
    return -1;
}
#define SHORT_MAX 32767

#define OKEY unsigned __int128
#define OKEY_BITS 128
#define OKEY_MAXK 63
/* the core's public names, wide */
#define or_table orw_table
#define or_seqs orw_seqs
#define or_comps orw_comps
#define or_revcomp orw_revcomp
#define or_canonical orw_canonical
#define or_table_new orw_table_new
#define or_table_free orw_table_free
#define or_table_size orw_table_size
#define or_table_get orw_table_get
#define or_table_add orw_table_add
#define or_table_export orw_table_export
#define or_count_buffer orw_count_buffer
#define or_unitig_census orw_unitig_census
#define or_build_unitigs orw_build_unitigs
#define or_seqs_free orw_seqs_free
#define or_seqs_count orw_seqs_count
#define or_seqs_total_len orw_seqs_total_len
#define or_seqs_get orw_seqs_get
#define or_seqs_write_fasta orw_seqs_write_fasta
#define or_write_distribution orw_write_distribution
#define or_count_seqs orw_count_seqs
#define or_cut_components orw_cut_components
#define or_comps_free orw_comps_free
#define or_comps_count orw_comps_count
#define or_comps_get orw_comps_get
#define or_features orw_features
#define or_features_selected orw_features_selected
#define or_features_reads orw_features_reads
#define or_features_reads_selected orw_features_reads_selected
typedef struct orw_table orw_table;
typedef struct orw_seqs orw_seqs;
typedef struct orw_comps orw_comps;
/* prototypes the core's functions are checked against (the narrow ones live in mf_oracle.h) */
OKEY orw_revcomp(OKEY x, int k);
OKEY orw_canonical(OKEY kmer, int k);
orw_table *orw_table_new(void);
void orw_table_free(orw_table *t);
uint64_t orw_table_size(const orw_table *t);
int64_t orw_table_get(const orw_table *t, OKEY key);
int orw_table_add(orw_table *t, OKEY key, int inc);
uint64_t orw_table_export(const orw_table *t, int threshold, OKEY *keys, int32_t *vals, uint64_t cap);
int orw_count_buffer(orw_table *t, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads, int k, int min_len);
void orw_unitig_census(uint64_t out[3]);
orw_seqs *orw_build_unitigs(const orw_table *t, int k, int thr, int min_len);
void orw_seqs_free(orw_seqs *s);
uint64_t orw_seqs_count(const orw_seqs *s);
uint64_t orw_seqs_total_len(const orw_seqs *s);
int orw_seqs_get(const orw_seqs *s, uint64_t i, const char **seq, uint64_t *len, int *avg_w, int *min_w, int *max_w);
int orw_seqs_write_fasta(const orw_seqs *s, const char *path);
int orw_write_distribution(const orw_table *t, const char *path);
int orw_count_seqs(orw_table *t, const orw_seqs *s, int k, int min_len);
orw_comps *orw_cut_components(const orw_table *t, int k, int b1, int b2);
void orw_comps_free(orw_comps *c);
uint64_t orw_comps_count(const orw_comps *c);
int orw_comps_get(const orw_comps *c, uint64_t i, uint64_t *size, int64_t *weight, int *thr, const OKEY **kmers);
int orw_features(const orw_comps *c, const orw_table *sample, int threshold, int64_t *vec, double *breadth);
int orw_features_selected(const orw_comps *c, const orw_table *sample, int threshold, const orw_table *selected, int64_t *vec, double *breadth);
int orw_features_reads(const orw_comps *c, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads, int k, int threshold, int64_t *vec, double *breadth);
int orw_features_reads_selected(const orw_comps *c, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads, int k, int threshold,
                                const orw_table *selected, int64_t *vec, double *breadth);

#include "mf_oracle_core.inc"

/* ---- by-value 128-bit arguments as two words (ctypes has no __int128) ---- */
int64_t orw_table_get2(const orw_table *t, uint64_t hi, uint64_t lo) { return orw_table_get(t, ((OKEY)hi << 64) | lo); }
int orw_table_add2(orw_table *t, uint64_t hi, uint64_t lo, int inc) { return orw_table_add(t, ((OKEY)hi << 64) | lo, inc); }
void orw_revcomp2(uint64_t hi, uint64_t lo, int k, uint64_t out[2]) { OKEY r = orw_revcomp(((OKEY)hi << 64) | lo, k); out[0] = (uint64_t)r; out[1] = (uint64_t)(r >> 64); }

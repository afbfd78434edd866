/*
 * mf_oracle.h -- CPU ORACLE for the MetaFast hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded restatement of the reference algorithm
 * (ctlab/metafast, pure Java) for the path
 *   kmer-counter(-many) -> seq-builder(-many) -> component-cutter ->
 *   features-calculator -> dist-matrix-calculator.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.  The product (metafast_amd/)
 * never links, imports or calls it.
 *
 * Parity pin: reproduces /root/reference/test_data/meta_test_matrix.txt
 * (the reference's only golden vector for this path) to all printed digits;
 * see tests/test_oracle_golden.py.  The reference itself (Java, no JDK/JRE in
 * the image, lib/itmo-assembler.jar missing) cannot be built or run here, so
 * there is no oracle/_ref.
 *
 * Citations:  src/...  = /root/reference/src/...
 *             itmo!/.. = /root/reference/lib/itmo-assembler-src.jar!/ru/ifmo/genetics/..
 */
#ifndef MF_ORACLE_H
#define MF_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct or_table or_table;   /* canonical k-mer -> value map            */
typedef struct or_seqs  or_seqs;    /* list of unitigs (+avg/min/max weights)   */
typedef struct or_comps or_comps;   /* list of connected components            */

const char *or_last_error(void);

/* ---- A1  readers (itmo!/io/readers/FastaReader.java:53-104, FastqReader.java:53-115,
 *      FastaReaderFromXQSource.java:66-70, ReadersUtils.java:27-102) ----
 * Parses a FASTA/FASTQ file (format by extension) into concatenated ASCII
 * bases (upper-cased ACGT) + offsets.  Reads containing N (FASTA) or any
 * phred-0 base / N / '.' (FASTQ) are dropped whole.  Returns 0 or <0.        */
int or_read_file(const char *path, uint8_t **bases, uint64_t **offsets,
                 uint64_t *n_reads, uint64_t *n_bases);
void or_free(void *p);

/* ---- A2-A4  counting (src/io/IOUtils.java:742-803; ShortKmer.java:54-71,104-150;
 *      Long2ShortHashMap.java:119-157; NumUtils.java:21-26) ---- */
or_table *or_table_new(void);
void      or_table_free(or_table *t);
/* adds every canonical k-mer of every read with len >= min_len, +1 saturating at 32767 */
int       or_count_buffer(or_table *t, const uint8_t *bases, const uint64_t *offsets,
                          uint64_t n_reads, int k, int min_len);
int       or_count_files(or_table *t, const char *const *files, int nfiles, int k, int min_len);
uint64_t  or_table_size(const or_table *t);
/* entries with value > threshold, ascending key order; returns count (call with cap=0 to size) */
uint64_t  or_table_export(const or_table *t, int threshold, uint64_t *keys, int32_t *vals, uint64_t cap);
int64_t   or_table_get(const or_table *t, uint64_t key);      /* -1 if absent (Long2ShortHashMap.get) */
int       or_table_add(or_table *t, uint64_t key, int inc);    /* addAndBound                         */

/* ---- A5/A6  .kmers.bin + .stat.txt (src/io/IOUtils.java:45-71, 369-401;
 *      src/io/KmersLoadWorker.java:16-34; QuickQuantitativeStatistics.java:38-72) ---- */
int       or_write_kmers(const or_table *t, int threshold, const char *kmers_bin,
                         const char *stat_txt, uint64_t *n_good);
int       or_load_kmers(or_table *t, const char *const *files, int nfiles, int freq_threshold);

/* ---- A7/A8  unitigs (src/algo/AddSequencesShiftingRightTask.java:40-123;
 *      src/algo/HashMapOperations.java:13-47; src/algo/SequencesFinders.java:13-31;
 *      src/structures/Sequence.java:26-37; FastaDedicatedWriter.java:15,33-49) ---- */
or_seqs  *or_build_unitigs(const or_table *t, int k, int freq_threshold, int min_len);
void      or_unitig_census(uint64_t out[3]);   /* of the last or_build_unitigs: walks started / long enough / emitted */
void      or_seqs_free(or_seqs *s);
uint64_t  or_seqs_count(const or_seqs *s);
uint64_t  or_seqs_total_len(const or_seqs *s);
/* i-th sequence: ASCII (not NUL-terminated) pointer + length + weights */
int       or_seqs_get(const or_seqs *s, uint64_t i, const char **seq, uint64_t *len,
                      int *avg_w, int *min_w, int *max_w);
int       or_seqs_write_fasta(const or_seqs *s, const char *path);
/* SeqBuilderMain.java:84-98,170-176 */
int       or_write_distribution(const or_table *t, const char *path);
/* count k-mers of all sequences with len >= min_len into t (A9, ComponentCutterMain.java:81) */
int       or_count_seqs(or_table *t, const or_seqs *s, int k, int min_len);

/* ---- A10/A11 components (src/algo/ComponentsBuilder.java:58-270;
 *      src/algo/KmerOperations.java:9-26; src/structures/ConnectedComponent.java:80-136) ---- */
or_comps *or_cut_components(const or_table *t, int k, int b1, int b2);
void      or_comps_free(or_comps *c);
uint64_t  or_comps_count(const or_comps *c);
int       or_comps_get(const or_comps *c, uint64_t i, uint64_t *size, int64_t *weight,
                       int *thr, const uint64_t **kmers /* ascending */);
int       or_comps_write(const or_comps *c, const char *components_bin, const char *stat_txt);
or_comps *or_comps_load(const char *components_bin);

/* ---- A12 features (src/tools/FeaturesCalculatorMain.java:97-103,137-162,169-236;
 *      src/io/IOUtils.java:577-597) ----
 * sample = table holding that sample's .kmers.bin content (count > b only).   */
int       or_features(const or_comps *c, const or_table *sample, int threshold,
                      int64_t *vec, double *breadth);

/* --use-reads-for-calculating-features (FeaturesCalculatorMain.java:117-131; src/io/IOUtils.java:806-834): the
 * features straight from the reads, with long (unsaturated) per-k-mer counts */
int       or_features_reads(const or_comps *c, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads,
                            int k, int threshold, int64_t *vec, double *breadth);

/* --selected (FeaturesCalculatorMain.java:55-57, 113-116, 193-203): `selected` = the table IOUtils.loadKmers(selectedKmers, 0, ..)
 * gives (or_load_kmers with freq_threshold 0); NULL = no selection */
int       or_features_selected(const or_comps *c, const or_table *sample, int threshold, const or_table *selected,
                               int64_t *vec, double *breadth);
int       or_features_reads_selected(const or_comps *c, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads,
                                     int k, int threshold, const or_table *selected, int64_t *vec, double *breadth);

/* ---- A13 Bray-Curtis (src/tools/DistanceMatrixCalculatorMain.java:140-152) ---- */
int       or_bray_curtis(const int64_t *vecs, int n_samples, int n_comp, double *out);
/* NO-REFERENCE EXTENSION (the reference rejects k > 31): canonical counts of 2k-bit k-mers, 32 <= k <= 63, ascending
 * (hi, lo); checker of metafast_amd/csrc/mf_wide.hip only */
int       or_count_wide(const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads, int k, int min_len, uint64_t *hi, uint64_t *lo,
                        int32_t *cnt, uint64_t cap, uint64_t *n_out, uint64_t *n_occ_out);

/* ---- helpers exposed for tests ---- */
uint64_t  or_revcomp(uint64_t kmer, int k);                  /* itmo!/utils/KmerUtils.java:12-22 */
uint64_t  or_canonical(uint64_t kmer, int k);

/* ---- multi-threaded CPU baseline of A2-A4 (bench.py cpu_baseline only):
 * same structure as the Java (IOUtils.java:772-803,838-865): P worker threads
 * over 32768-read batches, 2^(floor(log2 P)+4) lock-sharded linear-probing
 * maps (int64 key + int16 value, load 0.75, doubling rehash).  Returns the
 * number of distinct k-mers, fills *n_occ.                                    */
uint64_t  or_cpu_baseline_count(const uint8_t *bases, const uint64_t *offsets,
                                uint64_t n_reads, int k, int threads, uint64_t *n_occ);
/* the same with the reference's serial reader (FASTA file) and single-threaded dump; res[4], sec[2]: see mf_oracle.c */
int       or_cpu_baseline_file(const char *fasta, int k, int threads, int bcut, const char *kmers_bin, uint64_t *res, double *sec);

#ifdef __cplusplus
}
#endif
#endif

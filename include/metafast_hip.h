/*
 * metafast_hip.h -- C-ABI of the MI355X-native MetaFast hot path
 * (libmetafast_hip.so; hand-written HIP for gfx950 behind plain C entry points).
 *
 * The reference (ctlab/metafast) is pure Java and has NO FFI: its seams for this
 * path are Java static methods called from Tool.runImpl on the main thread.  Each
 * entry point below replaces one of those seams and cites it.  The reference-side
 * binding a maintainer would add (JNI natives forwarding 1:1) is in INTEGRATION.md.
 *
 * Citations:  src/...  = /root/reference/src/...
 *             itmo!/.. = /root/reference/lib/itmo-assembler-src.jar!/ru/ifmo/genetics/..
 *
 * Conventions
 *  - every call returns int: 0 = ok, <0 = error; mf_last_error() returns a
 *    thread-local message (mirrors the Java side's checked ExecutionFailedException,
 *    itmo!/utils/tool/Tool.java:450-463).
 *  - handles are opaque and CALLER-OWNED; free with the matching *_destroy.
 *  - calls are made from one host thread, block until the result is complete and
 *    parallelise internally on the GPU (reference: callee spawns P threads and joins
 *    on a latch, src/io/IOUtils.java:846-862).
 *  - "d_" pointers are device (HBM) pointers valid on the ctx's device; everything
 *    else is host memory.  No torch / C++ types cross this boundary.
 *  - there is NO CPU fallback: without a usable GPU mf_ctx_create fails.
 *
 * k-mer encoding (itmo!/dna/DnaTools.java:31,46-64; itmo!/dna/kmers/ShortKmer.java:54-71):
 *   A=0 G=1 C=2 T=3, first base in the most significant used bits, canonical =
 *   min(forward, reverse-complement); k in [1,31]; counts saturate at 32767.
 */
#ifndef METAFAST_HIP_H
#define METAFAST_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MF_OK 0
#define MF_ERR (-1)
#define MF_MAX_COUNT 32767        /* itmo!/utils/NumUtils.java:21-26 (short saturation) */

typedef struct mf_ctx   mf_ctx;    /* device + stream + workspace                           */
typedef struct mf_table mf_table;  /* canonical k-mer -> saturating count (BigLong2ShortHashMap) */
typedef struct mf_seqs  mf_seqs;   /* unitigs with weights (Deque<Sequence>)                */
typedef struct mf_comps mf_comps;  /* connected components (List<ConnectedComponent>)       */
typedef struct mf_dcc   mf_dcc;    /* one rank's part of the distributed component cutter   */
typedef struct mf_reads mf_reads;  /* the reads of a list of files, in HBM (NamedSource<Dna>) */

const char *mf_last_error(void);
const char *mf_version(void);

/* ---- context ---------------------------------------------------------------------- */
/* host_threads plays the role of -p/--available-processors (Tool.java:61-143) for the host
 * side parsers; GPU parallelism is fixed by the device. */
int  mf_ctx_create(int device, int host_threads, mf_ctx **out);
void mf_ctx_destroy(mf_ctx *ctx);
/* Devices this process sees (no context is made).  The reference's drivers loop over all libraries in one process
 * (src/tools/KmersCounterForManyFilesMain.java:80-108, SeqBuilderForManyFilesMain.java:82-94, FeaturesCalculatorMain.java:137-162);
 * a host that wants one library per GPU makes one context per device, each driven by a thread of its own. */
int  mf_device_count(void);
int  mf_device_memory(int device, uint64_t *total_bytes);
int  mf_ctx_device(const mf_ctx *ctx);
/* A context is used by ONE thread at a time; HIP's current device belongs to the thread, so a thread other than the one that made
 * the context calls this before its first call on it. */
int  mf_ctx_bind_thread(mf_ctx *ctx);
/* Use an existing HIP stream (hipStream_t as void*) for all launches instead of the ctx-owned one;
 * NULL = the device's default (null) stream. The caller keeps ownership. */
int  mf_ctx_set_stream(mf_ctx *ctx, void *hip_stream);
/* Tuning / test knobs, e.g. "l1_bits", "l2_bits", "part_target", "scatter_staged", "profile". */
int  mf_ctx_set_option(mf_ctx *ctx, const char *name, int64_t value);
int  mf_ctx_synchronize(mf_ctx *ctx);
/* Release cached workspace back to the driver. */
int  mf_ctx_trim(mf_ctx *ctx);
/* ... only `want` bytes of it, smallest idle regions first (a host that shares the device with the library and is short of
 * memory: a region the library has to allocate again costs about 35 ms per GiB).  *freed may be NULL. */
int  mf_ctx_trim_bytes(mf_ctx *ctx, uint64_t want, uint64_t *freed);
/* Per-kernel HIP-event timings accumulated while option "profile"=1.
 * Returns number of launches of `kernel` since the last reset; *total_ms = summed duration. */
int64_t mf_ctx_kernel_time(mf_ctx *ctx, const char *kernel, double *total_ms);
/* Writes "name\tlaunches\ttotal_ms\tmax_launch_ms\n" lines for every timed kernel into buf (NUL-terminated). */
int  mf_ctx_kernel_report(mf_ctx *ctx, char *buf, uint64_t cap);
int  mf_ctx_reset_timers(mf_ctx *ctx);
/* Counters and gauges by name: "slice_restarts" (counting runs that threw their slices away and started over with more because a
 * buffer found no place in HBM), "device_parsed_files" / "device_parser_stepped_back" (read files the device parser took / left to
 * the host readers), "unitig_doublings" (unitig runs whose long paths went through the doubled jump words), "wide_hashed_entries" /
 * "wide_big_entries" (k = 32..63: entries of large buckets ordered through the LDS hash tables / sorted aside), "hipmalloc_calls",
 * "hipmalloc_bytes", "hipmalloc_us", "arena_bytes", "arena_idle_bytes", "streamed_counts" / "streamed_counts_stepped_back" (counts of read
 * files that ran while the files crossed PCIe / that started so and were done again from the whole files).  < 0: unknown name. */
int64_t mf_ctx_stat(mf_ctx *ctx, const char *name);

/* ---- A1-A4  reads -> canonical k-mer counts ---------------------------------------- */
/* (round 6: plain FASTA / FASTQ files of 512 MB and more are counted WHILE they cross PCIe -- option "stream_count", mf_stream.hip: the reference's
 * reader feeds its workers while it reads, src/io/ReadsDispatcher.java:34-53 -- with the same table as a result) */
/* replaces IOUtils.loadReads (src/io/IOUtils.java:772-803), called from
 * KmersCounterMain.runImpl (src/tools/KmersCounterMain.java:77) with min_read_len=0 and from
 * ComponentCutterMain.runImpl (src/tools/ComponentCutterMain.java:81) with min_read_len=l.
 * Files are FASTA/FASTQ by extension, optionally compressed (.gz, .bz2; itmo!/io/ReadersUtils.java:27-54,
 * FastaGZReader.java, FastqGZReader.java, FastaBZ2Reader.java) or .binq (BinqReader.java); reads with N (FASTA)
 * or any phred-0 base (FASTQ) are dropped (FastaReader.java:53-76, FastaReaderFromXQSource.java:66-70).
 * All files go into ONE table (paired files are summed). */
int mf_count_reads(mf_ctx *ctx, const char *const *files, int nfiles, int k, int min_read_len,
                   mf_table **out);
/* The same load, keeping only the k-mers with count > threshold: KmersCounterMain.runImpl (src/tools/KmersCounterMain.java:77-99)
 * is loadReads followed by printKmers(hm, maximalBadFrequency, ...), which writes the entries with value > threshold only
 * (src/io/IOUtils.java:52-60).  The cut is made inside the counting kernels (mf_count_device_above), so a sample with more
 * than 2^32 distinct k-mers never exists as an uncut table; mf_table_write_kmers / mf_table_hist on the result give the same
 * .kmers.bin / .stat.txt as on the uncut table for every threshold >= this one.  *n_distinct_all (may be NULL) = distinct
 * k-mers before the cut (the "k-mers found" log line, KmersCounterMain.java:101-103). */
int mf_count_reads_above(mf_ctx *ctx, const char *const *files, int nfiles, int k, int min_read_len, int threshold,
                         mf_table **out, uint64_t *n_distinct_all);
/* The readers alone: ReadersUtils.readDnaLazy (itmo!/io/ReadersUtils.java:81-102) over a list of files -- FastaReader.java:53-104,
 * FastqReader.java:53-115 + FastaReaderFromXQSource.java:66-70, their .gz / .bz2 forms, BinqReader.java -- into the (bases, offsets)
 * layout mf_count_device takes (upper-case A / C / G / T; reads with N or a phred-0 base are not there).  Plain FASTA / FASTQ files
 * are parsed on the device (option "device_parse", mf_dparse.hip); the files that parser is not sure about, and every error, go
 * through the host readers. */
int  mf_reads_load(mf_ctx *ctx, const char *const *files, int nfiles, mf_reads **out);
void mf_reads_destroy(mf_reads *r);
int  mf_reads_stats(const mf_reads *r, uint64_t *n_reads, uint64_t *n_bases);
int  mf_reads_device_view(const mf_reads *r, const void **d_bases, const void **d_offsets);
int  mf_reads_export(const mf_reads *r, uint8_t *bases, uint64_t *offsets);      /* bases[n_bases], offsets[n_reads + 1] */
/* Same, for reads already resident in HBM: d_bases = concatenated ASCII bases (ACGT, either case,
 * no N), d_offsets = uint64[n_reads+1] with offsets[0]=0, offsets[n_reads]=n_bases.  d_bases must
 * be 16-byte aligned and readable up to the next multiple of 16 bytes.  This is the device half of
 * ReadsLoadWorker.process (src/io/IOUtils.java:756-768). */
int mf_count_device(mf_ctx *ctx, const void *d_bases, const void *d_offsets, uint64_t n_reads,
                    uint64_t n_bases, int k, int min_read_len, mf_table **out);
/* Same counting pass, but only the k-mers with count > threshold are kept (what KmersCounterMain hands on: IOUtils.printKmers
 * writes the entries with value > maximalBadFrequency, src/io/IOUtils.java:52-60; src/tools/KmersCounterMain.java:99) -- the
 * rejected entries are dropped inside the counting kernels instead of being written out and filtered afterwards.
 * *n_distinct_all (may be NULL) = number of distinct k-mers before the cut (BigLong2ShortHashMap.size()). */
int mf_count_device_above(mf_ctx *ctx, const void *d_bases, const void *d_offsets, uint64_t n_reads,
                          uint64_t n_bases, int k, int min_read_len, int threshold, mf_table **out,
                          uint64_t *n_distinct_all);
void mf_table_destroy(mf_table *t);
/* BigLong2ShortHashMap.size() and the sum of all (saturated) values */
int mf_table_stats(const mf_table *t, uint64_t *n_distinct, uint64_t *n_total);
/* k-mer occurrences fed into the table by the call that built it (N_occ) */
int mf_table_occurrences(const mf_table *t, uint64_t *n_occ);
/* Records the counting pass moved through its radix partitions to build this table: super-k-mer records of 16 bytes
 * (padding included) on the default path for k >= 20, else one 8-byte record per k-mer occurrence; 0 for tables that
 * were loaded or filtered.  Measurement only (bench.py prices the kernels' algorithmic bytes with it). */
int mf_table_records(const mf_table *t, uint64_t *n_records, int *record_bytes);
/* QuickQuantitativeStatistics of IOUtils.printKmers (src/io/IOUtils.java:45-71; itmo!/statistics/
 * QuickQuantitativeStatistics.java:38-72): hist[c] = number of distinct k-mers with count c over ALL k-mers that were
 * counted, c = 0 .. MF_MAX_COUNT (hist must hold MF_MAX_COUNT + 1 entries).  A table that comes out of
 * mf_count_device_above / mf_table_filter has remembered the k-mers the cut dropped, so this is what the .stat.txt of
 * the uncut table would hold. */
int mf_table_hist(const mf_table *t, uint64_t *hist);
/* Copies entries with count > threshold to host arrays in ASCENDING KEY order (the reference's
 * iteration order is thread-count dependent, BigLong2ShortHashMap.java:216-253, so callers must
 * not rely on it).  Call with cap=0 to get *n only. */
int mf_table_export(const mf_table *t, int threshold, uint64_t *keys, uint16_t *counts,
                    uint64_t cap, uint64_t *n);
/* Device view (unsorted, dense): uint64 keys[n], uint16 counts[n]. */
int mf_table_device_view(const mf_table *t, const void **d_keys, const void **d_counts, uint64_t *n);
/* Long2ShortHashMap.get (itmo!/structures/map/Long2ShortHashMap.java:160-175) for a batch of host
 * keys: values[i] = count or -1.  Builds the HBM open-addressed index on first use. */
int mf_table_lookup(mf_table *t, const uint64_t *keys, uint64_t n, int32_t *values);
/* Releases the table's lookup index (built on first use by the unitig builder, the features step and mf_table_lookup; rebuilt
 * when it is needed again).  The reference keeps one map per library alive at a time (KmersCounterForManyFilesMain.java:80-108);
 * a rank that holds several samples' tables drops each index between the sample's seq-builder and features steps.
 * Safe whenever no library call on this table is running (one calling thread): the views of a table that the library makes for
 * itself (a cut that keeps every entry) live inside the call that made them; a handle that came out of the context's file cache
 * (option file_cache: the same table under several handles, mf_table::refs) rebuilds the index when its next user needs it. */
int mf_table_drop_index(mf_table *t);

/* ---- NO-REFERENCE EXTENSION: 32 <= k <= 63 ----------------------------------------------------
 * The reference rejects k > 31 (src/tools/KmersCounterMain.java:66-73: one Java long per k-mer); BASELINE.json's config 4 has a
 * k = 63 leg.  These entry points replace nothing and take part in no parity claim: canonical counts of 2k-bit k-mers (two
 * 64-bit words, first base most significant, canonical = the smaller of the k-mer and its reverse complement, counts saturate
 * at 32767, reads shorter than max(k, min_read_len) give nothing) -- the k <= 31 definitions carried over; checked against
 * oracle/mf_oracle.c:or_count_wide (unsigned __int128).  Input as for mf_count_device. */
typedef struct mf_wtable mf_wtable;
int  mf_count_wide_device(mf_ctx *ctx, const void *d_bases, const void *d_offsets, uint64_t n_reads, uint64_t n_bases, int k,
                          int min_read_len, mf_wtable **out);
void mf_wtable_destroy(mf_wtable *t);
int  mf_wtable_stats(const mf_wtable *t, uint64_t *n_distinct, uint64_t *n_occ, int *k);
/* ascending k-mers as (high word, low word) + counts; NULL arrays: only *n */
int  mf_wtable_export(const mf_wtable *t, uint64_t *keys_hi, uint64_t *keys_lo, uint16_t *counts, uint64_t capacity, uint64_t *n);
/* the table where it is: it lies in HBM in *n_pieces pieces (one per pass over the reads), ascending inside a piece and piece after piece;
 * piece i: device pointers to its high words, low words (uint64) and counts (uint16), *n entries.  Valid until mf_wtable_destroy. */
int  mf_wtable_pieces(const mf_wtable *t, uint32_t *n_pieces);
int  mf_wtable_piece_view(const mf_wtable *t, uint32_t i, const void **d_keys_hi, const void **d_keys_lo, const void **d_counts, uint64_t *n);
/* ... "counted + graphed" for 32 <= k <= 63 (round 6; mf_wgraph.hip): the rest of the path on 2k-bit k-mers, same definitions as for
 * k <= 31 (the entry points named in brackets), checked against oracle/mf_oracle_wide.c = the pinned oracle's text compiled for 128-bit
 * keys.  A wide table is ascending, so a k-mer's place in it is a 32-bit vertex id that orders like the k-mer: only the neighbour
 * look-up sees 128-bit keys, everything after it is the k <= 31 machinery on ids. */
/* [mf_count_device_above] the count with the cut count > threshold inside the pass; *n_distinct_all (may be NULL) = distinct k-mers before it */
int  mf_count_wide_device_above(mf_ctx *ctx, const void *d_bases, const void *d_offsets, uint64_t n_reads, uint64_t n_bases, int k,
                                int min_read_len, int threshold, mf_wtable **out, uint64_t *n_distinct_all);
/* [mf_table_filter] the entries with count > threshold as a new table */
int  mf_wtable_filter(const mf_wtable *t, int threshold, mf_wtable **out);
/* [mf_table_drop_index] */
int  mf_wtable_drop_index(mf_wtable *t);
/* [mf_table_lookup] values[i] = count of the k-mer (keys_hi[i], keys_lo[i]) or -1; builds the lookup index on first use */
int  mf_wtable_lookup(mf_wtable *t, const uint64_t *keys_hi, const uint64_t *keys_lo, uint64_t n, int32_t *values);
/* [mf_build_unitigs_device] src/algo/AddSequencesShiftingRightTask.java:40-123 on 2k-bit k-mers; the result is an ordinary mf_seqs */
int  mf_build_unitigs_wide_device(mf_ctx *ctx, mf_wtable *t, int freq_threshold, int min_len, mf_seqs **out);
/* [mf_cut_components_device] src/algo/ComponentsBuilder.java:58-270; the cutter table = mf_count_wide_device over the unitigs with
 * min_read_len = l.  Components ordered as for k <= 31 (thr asc, weight desc, size desc, smallest k-mer asc), k-mers ascending inside */
typedef struct mf_wcomps mf_wcomps;
int  mf_cut_components_wide_device(mf_ctx *ctx, mf_wtable *cutter, int b1, int b2, mf_wcomps **out);
void mf_wcomps_destroy(mf_wcomps *c);
int  mf_wcomps_stats(const mf_wcomps *c, uint64_t *n_comp, uint64_t *n_kmers);
/* sizes[n], weights[n], thr[n], kmer_offsets[n + 1], the k-mers' high and low words [n_kmers]; any array may be NULL */
int  mf_wcomps_export(const mf_wcomps *c, uint64_t *sizes, int64_t *weights, int32_t *thr, uint64_t *kmer_offsets, uint64_t *kmers_hi,
                      uint64_t *kmers_lo);
/* [mf_features_device] vec[c] = sum of the sample's counts > threshold over the component's k-mers, breadth[c] = found / size */
int  mf_features_wide_device(mf_ctx *ctx, mf_wcomps *c, mf_wtable *sample, int threshold, int64_t *vec, double *breadth);

/* ---- A5/A6  .kmers.bin / .stat.txt --------------------------------------------------- */
/* replaces IOUtils.printKmers (src/io/IOUtils.java:45-71; KmersCounterMain.java:99): 10-byte
 * big-endian records (int64 k-mer, int16 count) for count > threshold, ascending key order;
 * histogram of ALL counts to stat_txt (may be NULL). */
int mf_table_write_kmers(const mf_table *t, int threshold, const char *kmers_bin,
                         const char *stat_txt, uint64_t *n_good);
/* replaces IOUtils.filterAndPrintKmers (src/io/IOUtils.java:101-123; KmersFilter.java:107): the
 * records of `t` with count > threshold whose k-mer has a value > filter_threshold in `filter`
 * (absent = 0), same record format and order as mf_table_write_kmers. */
int mf_table_write_kmers_filtered(const mf_table *t, int threshold, mf_table *filter, int filter_threshold,
                                  const char *kmers_bin, uint64_t *n_good);
/* replaces IOUtils.loadKmers (src/io/IOUtils.java:369-401; SeqBuilderMain.java:80): keeps records
 * with freq > freq_threshold; duplicate k-mers across files are summed with saturation. */
int mf_table_load_kmers(mf_ctx *ctx, const char *const *files, int nfiles, int freq_threshold, int k,
                        mf_table **out);
/* In-HBM equivalent of write_kmers + load_kmers: new table with the entries whose count > threshold. */
int mf_table_filter(const mf_table *t, int threshold, mf_table **out);
/* Build a table from host (key,count) arrays (insert-or-add with saturation). */
int mf_table_from_host(mf_ctx *ctx, const uint64_t *keys, const uint16_t *counts, uint64_t n, int k,
                       mf_table **out);

/* ---- A7/A8  unitigs ---------------------------------------------------------------- */
/* replaces SequencesFinders.thresholdStrategy (src/algo/SequencesFinders.java:13-31 ->
 * AddSequencesShiftingRightTask.java:40-123) over k-mers with count > freq_threshold, keeping
 * sequences with length >= min_len, including the reference's start/end emission rule
 * (:101-121) under which a path may be emitted 0, 1 or 2 times. */
int  mf_build_unitigs_device(mf_ctx *ctx, mf_table *t, int freq_threshold, int min_len, mf_seqs **out);
void mf_seqs_destroy(mf_seqs *s);
int  mf_seqs_stats(const mf_seqs *s, uint64_t *n_seqs, uint64_t *total_len);
/* Device view: ASCII bases concatenated + uint64 offsets[n+1] (same layout mf_count_device takes),
 * int32 avg/min/max weights per sequence. */
int  mf_seqs_device_view(const mf_seqs *s, const void **d_bases, const void **d_offsets,
                         const void **d_avg, const void **d_min, const void **d_max,
                         uint64_t *n_seqs, uint64_t *n_bases);
/* Host copy: bases[total_len], offsets[n+1], avg/min/max[n]; sequences ordered by
 * (canonical start k-mer, strand) for reproducibility (reference order is a thread race). */
int  mf_seqs_export(const mf_seqs *s, uint8_t *bases, uint64_t *offsets, int32_t *avg, int32_t *mn,
                    int32_t *mx);
/* Sequence.printSequences (src/structures/Sequence.java:26-37): ">i length=L av_weight=A
 * min_weight=m max_weight=M", 70 columns (FastaDedicatedWriter.java:15,33-49). */
int  mf_seqs_write_fasta(const mf_seqs *s, const char *seq_fasta);
/* One call = SeqBuilderMain.runImpl (src/tools/SeqBuilderMain.java:78-160): histogram file
 * `distribution` (:84-98,170-176; may be NULL) + unitigs to seq_fasta. */
int  mf_build_unitigs(mf_ctx *ctx, mf_table *t, int k, int freq_threshold, int min_len,
                      const char *seq_fasta, const char *distribution, uint64_t *n_seq);

/* ---- A9-A11  component cutter ------------------------------------------------------- */
/* replaces ComponentsBuilder.splitStrategy (src/algo/ComponentsBuilder.java:24-32,58-270) on the
 * cutter table (k-mers of every emitted unitig of every sample): connected components over the 8
 * canonical neighbours (src/algo/KmerOperations.java:9-26), size window [b1,b2], oversize
 * components re-split on value >= thr+1.  Components are ordered by (thr asc, weight desc,
 * size desc, min k-mer asc); k-mers inside a component ascending. */
int  mf_cut_components_device(mf_ctx *ctx, mf_table *cutter, int b1, int b2, mf_comps **out);
void mf_comps_destroy(mf_comps *c);
int  mf_comps_stats(const mf_comps *c, uint64_t *n_comp, uint64_t *n_kmers);
/* Host copy: sizes[n], weights[n], thr[n], kmer_offsets[n+1], kmers[n_kmers]. */
int  mf_comps_export(const mf_comps *c, uint64_t *sizes, int64_t *weights, int32_t *thr,
                     uint64_t *kmer_offsets, uint64_t *kmers);
/* ConnectedComponent.saveComponents (src/structures/ConnectedComponent.java:80-93) +
 * components-stat file (ComponentsBuilder.java:146-152; may be NULL). */
int  mf_comps_write(const mf_comps *c, const char *components_bin, const char *stat_txt);
/* ConnectedComponent.loadComponents (:95-122) */
int  mf_comps_load(mf_ctx *ctx, const char *components_bin, mf_comps **out);
/* One call = ComponentCutterMain.runImpl :92-108 */
int  mf_cut_components(mf_ctx *ctx, mf_table *cutter, int k, int b1, int b2,
                       const char *components_bin, const char *stat_txt, uint64_t *n_comp);

/* ---- A9-A11 on several GPUs: every rank owns a shard of the cutter table ---------------------
 * The cutter table and the components step join ALL samples (ComponentCutterMain.runImpl,
 * src/tools/ComponentCutterMain.java:78-114; ComponentsBuilder.run, src/algo/ComponentsBuilder.java:58-153).
 * With one process per GPU, rank r owns the k-mers whose minimizer-partition hash starts with r
 * (world = power of two <= 64).  The library does the per-rank work on device buffers; the caller moves
 * the buffers between the ranks (torch.distributed over RCCL in metafast_amd/pipeline.py; any
 * all-to-all / all-gather will do).  All d_* pointers are device memory of the ctx's GPU.
 *
 *   table:    all-gather of the unitigs -> mf_count_device_shard                   (the shard)
 *   once:     mf_dcc_create; mf_dcc_queries + _fill -> all-to-all -> mf_dcc_answer -> all-to-all back
 *             -> mf_dcc_set_answers
 *   level t:  mf_dcc_level_local + mf_dcc_pairs_fill -> all-to-all -> mf_dcc_pairs_complete
 *             -> all-gather -> mf_dcc_merge + mf_dcc_stats_fill -> all-gather -> mf_dcc_classify
 *             (+ mf_dcc_kept_fill: every rank derives ALL kept components and the number of oversize ones from
 *             the gathered records -- nothing more to exchange; stop when there is no oversize component)
 *   end:      mf_dcc_minkeys -> all-reduce (min); mf_dcc_members + _fill -> all-gather -> mf_dcc_finish: the same mf_comps on every rank,
 *             identical to mf_cut_components_device on the merged table.                              */
/* rank's shard of the table of ALL the given sequences (every rank passes the same input, the unitigs of all samples):
 * the k-mers whose minimizer-partition hash starts with `rank`.  k >= 20. */
int  mf_count_device_shard(mf_ctx *ctx, const void *d_bases, const void *d_offsets, uint64_t n_seqs, uint64_t n_bases,
                           int k, int min_len, int rank, int world, mf_table **out);
/* base[world + 1]: global id of each rank's first vertex (prefix sums of the shard sizes; total < 2^32).
 * The shard must outlive the handle. */
int  mf_dcc_create(mf_ctx *ctx, mf_table *shard, int rank, int world, const uint32_t *base, mf_dcc **out);
void mf_dcc_destroy(mf_dcc *d);
/* the world size the handle was created with (the length of the per-rank count arrays below) */
int  mf_dcc_world(const mf_dcc *d);
/* neighbours in other shards: counts[world] (host), then the 16-byte queries grouped by owner */
int  mf_dcc_queries(mf_dcc *d, uint64_t *counts);
int  mf_dcc_queries_fill(mf_dcc *d, void *d_queries);
/* owner side: n queries -> n 16-byte answers, same order */
int  mf_dcc_answer(mf_dcc *d, const void *d_queries, uint64_t n, void *d_answers);
/* the answers to this rank's queries, in the order the queries were written */
int  mf_dcc_set_answers(mf_dcc *d, const void *d_answers, uint64_t n);
/* one threshold level (ComponentsBuilder.java:86-150): union-find inside the shard; counts[world] = 8-byte half
 * pairs (edges to fragments of HIGHER ranks) for each owner, then the half pairs themselves */
int  mf_dcc_level_local(mf_dcc *d, uint64_t *counts);
int  mf_dcc_pairs_fill(mf_dcc *d, void *d_pairs);
/* owner side, in place: half pairs -> (own fragment root, other fragment root), global ids.  An entry (0xFFFFFFFF, 0xFFFFFFFF) is no pair: it is
 * skipped here and by mf_dcc_merge, so that exchanges of a fixed capacity can pad their slices instead of announcing their sizes first. */
int  mf_dcc_pairs_complete(mf_dcc *d, void *d_pairs, uint64_t n);
/* ALL ranks' completed pairs -> global root of each own fragment; *n_stats = components this rank has vertices of, then
 * its share of each: 16-byte records (global root u32, size u32, weight u64) */
int  mf_dcc_merge(mf_dcc *d, const void *d_pairs, uint64_t n, uint64_t *n_stats);
int  mf_dcc_stats_fill(mf_dcc *d, void *d_stats);
/* ALL ranks' records in rank order (seg_first[world + 1], host: rank r's are [seg_first[r], seg_first[r + 1]); this rank's own
 * are [own_first, own_first + own_n)) -> classification of every own vertex at threshold thr (size window [b1, b2]);
 * n_kept / n_big: the kept / oversize components of the level over ALL ranks -- the same on every rank, which holds every
 * component's records --; then the kept ones' 16-byte records (root u32, size u32, weight u64), all of them, in no
 * particular order (sort by root for an order every rank agrees on) */
int  mf_dcc_classify(mf_dcc *d, const void *d_stats, uint64_t n, const uint64_t *seg_first, uint64_t own_first, uint64_t own_n,
                     int b1, int b2, int thr, uint64_t *n_kept, uint64_t *n_big);
int  mf_dcc_kept_fill(mf_dcc *d, void *d_kept);
/* members of kept components among this rank's k-mers, all levels: k-mers u64[n], global roots u32[n] */
int  mf_dcc_members(mf_dcc *d, uint64_t *n);
int  mf_dcc_members_fill(mf_dcc *d, void *d_kmers, void *d_roots);
/* smallest member k-mer of each kept component among this rank's k-mers (0x7FFF...F: none) -> d_min u64[n_kept];
 * the caller takes the minimum over the ranks (it breaks ties in the components' order) */
int  mf_dcc_minkeys(mf_dcc *d, const uint32_t *kept_root, uint64_t n_kept, void *d_min);
/* The same members grouped by component: 8 bytes per member on the wire -- k-mers u64[n_members], sorted by component, and
 * (root u32, count u32) runs [n_runs] -- instead of 12; mf_dcc_finish_grouped takes ALL ranks' k-mers (rank order) and runs
 * (rank order) */
int  mf_dcc_members_grouped(mf_dcc *d, uint64_t *n_members, uint64_t *n_runs);
int  mf_dcc_members_grouped_fill(mf_dcc *d, void *d_kmers, void *d_runs);
int  mf_dcc_finish_grouped(mf_dcc *d, const void *d_kmers, uint64_t n_members, const void *d_runs, uint64_t n_runs,
                           const uint32_t *kept_root, const uint32_t *kept_size, const int64_t *kept_weight,
                           const int32_t *kept_thr, const uint64_t *kept_minkey, uint64_t n_kept, mf_comps **out);
/* ALL ranks' members + all levels' kept components (host arrays) -> components, ordered as above */
int  mf_dcc_finish(mf_dcc *d, const void *d_kmers, const void *d_roots, uint64_t n_members, const uint32_t *kept_root,
                   const uint32_t *kept_size, const int64_t *kept_weight, const int32_t *kept_thr,
                   const uint64_t *kept_minkey, uint64_t n_kept, mf_comps **out);

/* ---- the exchanges of the multi-GPU path, behind this boundary (round 6; mf_comm.hip) ----------------------
 * The reference is one JVM (ComponentCutterMain.runImpl, src/tools/ComponentCutterMain.java:78-114, joins all libraries in one map; IOUtils.run,
 * src/io/IOUtils.java:846-862, waits for its threads on a latch).  With a library per GPU the join is an exchange.  A communicator gives every rank
 * three primitives on buffers in HBM -- a gather of a few host integers, an all-gather and an all-to-all of slices whose sizes all ranks know --
 * over one of three transports, and the calls below run the path's exchange steps on it (on the context's stream and workspace):
 *   local     the ranks are THREADS of one process, a context each (metafast.sh --devices a,b,...): a rank copies its slice straight into every
 *             peer's receive buffer (peer access over xGMI; a device-to-device copy where two ranks share a GPU) -- the direct all-gather of a
 *             fully connected xGMI node, no ring;
 *   rccl      one process per GPU: librccl looked up at run time; slices travel as grouped ncclSend / ncclRecv pairs;
 *   external  the host's own primitives (MPI under a Java host; torch.distributed in this repository's tests).
 * One calling thread per communicator; every rank makes the same calls in the same order.  MF_ERR_TOGETHER: some rank could not do its part
 * and ALL ranks return this from the same call (mf_last_error names the ranks) -- the caller may take another route on all of them, e.g.
 * mf_comm_gather_sequences + mf_count_device + mf_cut_components_device everywhere. */
#define MF_ERR_TOGETHER (-2)
typedef struct mf_comm mf_comm;
/* n communicators for n threads of this process, out[r] for the thread that drives ctxs[r] */
int  mf_comm_create_local(mf_ctx *const *ctxs, int n, mf_comm **out);
/* one process per GPU: rank 0 makes an id (128 bytes) and hands it to the others by the host's own means; all ranks then create together */
int  mf_comm_rccl_id(void *id128);
int  mf_comm_create_rccl(mf_ctx *ctx, const void *id128, int rank, int world, mf_comm **out);
/* the host's primitives; all sizes in bytes, known to every rank; device pointers of the context's GPU; < 0 = failure.  The library has
 * synchronised its stream before a call and reads the result on that stream after it: the primitive returns when the data has arrived.
 *   gather_ints(user, vals[n], n, out[world * n])                       rank r's vals at out[r * n]
 *   all_gather(user, d_send, d_recv, bytes[world])                      rank r's bytes[r] bytes at d_recv + sum(bytes[:r]), on every rank
 *   all_to_all(user, d_send, send_bytes[world], d_recv, recv_bytes[world])   d_send grouped by destination, d_recv by source */
typedef struct mf_comm_ops {
    int (*gather_ints)(void *user, const int64_t *vals, int n, int64_t *out);
    int (*all_gather)(void *user, const void *d_send, void *d_recv, const uint64_t *bytes);
    int (*all_to_all)(void *user, const void *d_send, const uint64_t *send_bytes, void *d_recv, const uint64_t *recv_bytes);
} mf_comm_ops;
int  mf_comm_create_external(mf_ctx *ctx, int rank, int world, const mf_comm_ops *ops, void *user, mf_comm **out);
void mf_comm_destroy(mf_comm *c);
int  mf_comm_rank(const mf_comm *c);
int  mf_comm_world(const mf_comm *c);
const char *mf_comm_kind(const mf_comm *c);                  /* "local" / "rccl" / "external" */
/* "collectives", "bytes_in" (received), "us" (host time inside the exchanges) since the last reset; < 0: unknown name */
int64_t mf_comm_stat(const mf_comm *c, const char *name);
int  mf_comm_reset_stats(mf_comm *c);
/* the primitives themselves (tests; hosts that have more to exchange) */
int  mf_comm_gather_ints(mf_comm *c, const int64_t *vals, int n, int64_t *out);
int  mf_comm_all_gather(mf_comm *c, const void *d_send, void *d_recv, const uint64_t *bytes_per_rank);
int  mf_comm_all_to_all(mf_comm *c, const void *d_send, const uint64_t *send_bytes, void *d_recv, const uint64_t *recv_bytes);
/* Every rank's sequences (its libraries' unitigs: the layout of mf_seqs_device_view / mf_reads_device_view) on every rank, rank after rank:
 * IOUtils.loadReads over the .seq.fasta files of ALL libraries (src/tools/ComponentCutterMain.java:81).  The result is read with
 * mf_reads_stats / mf_reads_device_view and freed with mf_reads_destroy. */
int  mf_comm_gather_sequences(mf_comm *c, const void *d_bases, const void *d_offsets, uint64_t n_seqs, uint64_t n_bases, mf_reads **all);
/* One call = ComponentCutterMain.runImpl :81-108 over all ranks: the sequences are gathered, every rank counts the k-mers it owns
 * (mf_count_device_shard) and the exchange protocol of the sharded cutter above (mf_dcc_*) runs inside: the SAME components on every rank,
 * identical to mf_cut_components_device on the table of all sequences.  world = a power of two <= 64, 20 <= k <= 31. */
int  mf_cut_components_sharded(mf_comm *c, const void *d_bases, const void *d_offsets, uint64_t n_seqs, uint64_t n_bases, int k, int min_len,
                               int b1, int b2, mf_comps **out);
/* File form, every rank: its libraries' .seq.fasta files in (nfiles may be 0), rank 0 writes components.bin and the components-stat file
 * (ComponentCutterMain.runImpl :92-108), every rank keeps the components for the features step of its own libraries (option file_cache). */
int  mf_cut_components_sharded_files(mf_comm *c, const char *const *seq_files, int nfiles, int k, int min_len, int b1, int b2,
                                     const char *components_bin, const char *stat_txt, uint64_t *n_comp);
/* ... on a shard the caller has counted (NULL: this rank has none -- all ranks return MF_ERR_TOGETHER); info (may be NULL): [0] threshold levels,
 * [1] this rank's neighbour queries, [2] members over all ranks, [3] vertices over all ranks */
int  mf_cut_components_of_shard(mf_comm *c, mf_table *shard, int k, int b1, int b2, mf_comps **out, uint64_t *info);
/* The feature vectors of all ranks' samples, rank after rank, on every rank (the rows DistanceMatrixCalculatorMain reads back from the .vec files,
 * src/tools/DistanceMatrixCalculatorMain.java:125-138): rows = this rank's int64[n_rows][n_comp] (host); all_rows (host; NULL: only the count)
 * takes capacity_rows rows. */
int  mf_features_allgather(mf_comm *c, const int64_t *rows, uint64_t n_rows, uint64_t n_comp, int64_t *all_rows, uint64_t capacity_rows,
                           uint64_t *n_all_rows);

/* ---- A12  features ------------------------------------------------------------------ */
/* replaces FeaturesCalculatorMain: hm.put(kmer,0) for component k-mers (:97-103), presence pass
 * over the sample's records (IOUtils.calculatePresenceForKmers, src/io/IOUtils.java:577-597) and
 * buildAndPrintVector (:169-236): vec[c] = sum of counts > threshold, breadth[c] = found/size. */
int  mf_features_device(mf_ctx *ctx, mf_comps *c, const mf_table *sample, int threshold,
                        int64_t *vec, double *breadth);
/* File form: components.bin + <name>.kmers.bin -> <name>.vec / <name>.breadth (either may be NULL) */
int  mf_features(mf_ctx *ctx, const char *components_bin, const char *kmers_bin, int k, int threshold,
                 const char *vec_path, const char *breadth_path);

/* --use-reads-for-calculating-features (FeaturesCalculatorMain.java:117-131 + IOUtils.calculatePresenceForReads /
 * ReadsPresenceWorker, src/io/IOUtils.java:806-834): the features of a sample straight from its READS -- every k-mer of
 * every read that is a component k-mer adds 1 (64-bit counts, no saturation; reads shorter than k give nothing), then
 * vec[c] = sum of the counts > threshold, breadth[c] = found / size.  Device form: reads resident in HBM (layout of
 * mf_count_device); file form: FASTA / FASTQ (.gz) files of ONE library, output as mf_features. */
int mf_features_reads_device(mf_ctx *ctx, mf_comps *c, const void *d_bases, const void *d_offsets, uint64_t n_reads,
                             uint64_t n_bases, int k, int threshold, int64_t *vec, double *breadth);
int mf_features_reads(mf_ctx *ctx, const char *components_bin, const char *const *files, int nfiles, int k, int threshold,
                      const char *vec_path, const char *breadth_path);

/* --selected (FeaturesCalculatorMain.java:55-57 the parameter, :113-116 selected = IOUtils.loadKmers(selectedKmers, 0, ...),
 * :193-203 its use in buildAndPrintVector): with a table of selected k-mers only the component k-mers with
 * selected.getWithZero(kmer) > 0 enter vec[c], kmersFound AND kmersCount, so breadth[c] = found / (selected k-mers of c) and a
 * component without a selected k-mer gets 0.0 / 0.0 = NaN ("NaN" in the .breadth file, Double.toString).  `selected` is a table of
 * the same context, e.g. mf_table_load_kmers(ctx, files, n, 0, k, &selected); NULL = no selection (the calls above). */
int mf_features_device_selected(mf_ctx *ctx, mf_comps *c, const mf_table *sample, mf_table *selected, int threshold,
                                int64_t *vec, double *breadth);
int mf_features_selected(mf_ctx *ctx, const char *components_bin, const char *kmers_bin, int k, int threshold, mf_table *selected,
                         const char *vec_path, const char *breadth_path);
int mf_features_reads_device_selected(mf_ctx *ctx, mf_comps *c, const void *d_bases, const void *d_offsets, uint64_t n_reads,
                                      uint64_t n_bases, int k, mf_table *selected, int threshold, int64_t *vec, double *breadth);
int mf_features_reads_selected(mf_ctx *ctx, const char *components_bin, const char *const *files, int nfiles, int k, int threshold,
                               mf_table *selected, const char *vec_path, const char *breadth_path);

/* ---- A13  Bray-Curtis ---------------------------------------------------------------- */
/* replaces DistanceMatrixCalculatorMain.brayCurtisDistance (src/tools/DistanceMatrixCalculatorMain.java:
 * 140-152): d = sum|a-b| / sum(|a|+|b|) on raw vectors; vecs is row-major [n_samples][n_comp]. */
int  mf_bray_curtis(const int64_t *vecs, int n_samples, int n_comp, double *out_matrix);

/* ---- synthetic reads (bench / tests) ---------------------------------- */
/* Fills d_bases[n_reads*read_len] (ASCII) and d_offsets[n_reads+1] with the deterministic,
 * integer-only generator described in DESIGN.md (128-genome pool with shared repeats, log-normal
 * abundances, strand flips, 0.5 % substitutions, no N).  The host mirror mf_synth_reads_host
 * produces the identical bytes on the CPU. */
int  mf_synth_reads_device(mf_ctx *ctx, uint64_t seed, int sample, uint64_t first_read, uint64_t n_reads,
                           int read_len, uint64_t genome_scale_bp, void *d_bases, void *d_offsets);
int  mf_synth_reads_host(uint64_t seed, int sample, uint64_t first_read, uint64_t n_reads, int read_len,
                         uint64_t genome_scale_bp, uint8_t *bases, uint64_t *offsets);
/* the same with the substitution rate as a parameter: sub_per_16384 substitutions per 16384 bases (82 = the 0.5 % above,
 * 164 = 1 %: BASELINE.json config 5, "generator scaled: pool 2 Gbp, error 1 %") */
int  mf_synth_reads_device_ex(mf_ctx *ctx, uint64_t seed, int sample, uint64_t first_read, uint64_t n_reads,
                              int read_len, uint64_t genome_scale_bp, int sub_per_16384, void *d_bases, void *d_offsets);
int  mf_synth_reads_host_ex(uint64_t seed, int sample, uint64_t first_read, uint64_t n_reads, int read_len,
                            uint64_t genome_scale_bp, int sub_per_16384, uint8_t *bases, uint64_t *offsets);

#ifdef __cplusplus
}
#endif
#endif

// mf_common.h -- internal declarations shared by the HIP translation units of libmetafast_hip.so.
// gfx950 (MI355X) only: wave = 64 lanes, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <map>
#include <string>
#include <vector>
#include "../../include/metafast_hip.h"

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
int mf_set_error(const char *fmt, ...);   // returns MF_ERR

#define MF_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t e__ = (call);                                                                  \
        if (e__ != hipSuccess)                                                                    \
            return mf_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, \
                                __LINE__);                                                        \
    } while (0)
#define MF_TRY(call)                  \
    do {                              \
        int r__ = (call);             \
        if (r__ < 0) return r__;      \
    } while (0)

// ---------------------------------------------------------------------------------------------
// constants
// ---------------------------------------------------------------------------------------------
#define MF_WAVE 64
static constexpr uint64_t MF_EMPTY = 0xFFFFFFFFFFFFFFFFull;  // unreachable key: k<=31 keys are < 2^62
static constexpr int MF_MAX_DIGIT_BITS = 11;                 // 2048 staging lines of 64 B = 128 KiB LDS
static constexpr int MF_LINE = 8;                            // k-mers per 64-byte staging line
static constexpr int MF_COUNT_SLOTS = 4096;                  // LDS count table slots (32 KiB keys + 16 KiB counts): 3 workgroups per CU

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
struct mf_timer_rec {
    hipEvent_t a, b;
    std::string name;
};
struct mf_ctx {
    int device = 0;
    int host_threads = 1;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int n_cu = 256;
    // options
    int64_t opt_l1_bits = -1;      // -1 = auto
    int64_t opt_l2_bits = -1;
    int64_t opt_part_target = 6144;  // mean k-mer occurrences per final partition (k_skm_count2: ~800 distinct k-mers in 4096 slots)
    int64_t opt_part_target_long = 256;   // ... for assembled sequences (mean length >= 8k: nearly duplicate-free)
    int64_t opt_scatter_staged = 1;
    int64_t opt_profile = 0;
    int64_t opt_l1_blocks = 0;     // 0 = auto
    int64_t opt_verbose = 0;
    int64_t opt_skm = 1;           // super-k-mer counting path (mf_skm.hip) for k >= MF_SKM_MIN_K; 0 = always one record per k-mer
    int64_t opt_skm_dyn = 1;       // one-pass level-1 scatter with sampled region sizes: 0 never, 1 auto (large inputs), 2 always
    int64_t opt_stream_reader = 1; // plain FASTA/FASTQ files go through the pinned, double-buffered streaming reader (mf_io.hip)
    int64_t opt_sr_piece = 8 << 20, opt_sr_slack = 1 << 20;
    double t_hipmalloc = 0; uint64_t n_hipmalloc = 0, b_hipmalloc = 0;   // seconds / calls / bytes inside hipMalloc (diagnostics: MF_IO_TIMING)
    bool pin_pool_pinned = false;                                  // (hipHostMalloc'ed; else plain page-aligned host memory, option host_pinned = 0)
    void *pin_pool = nullptr; size_t pin_pool_bytes = 0;           // pinned staging chunks of the streaming reader (lazy, kept)
    void *up_pool = nullptr; size_t up_pool_bytes = 0; bool up_pool_pinned = false;   // staging chunks of mf_upload_file (mf_dparse.hip): small and PINNED (lazy, kept)
    int64_t opt_skm_slices = 0;    // digit-range slices of a counting run (0 = as many as the HBM budget asks for)
    int64_t opt_skm_dedupe = 1;    // k_skm_count: identical records of a unit are counted once, with their multiplicity (0 = every record for itself)
    int64_t opt_skm_shared = 1;    // slices behind one level 1 over all digits: 0 never, 1 when it fits, 2 whenever a run is sliced
    int64_t opt_arena_cap_gb = 0;  // pretend the device has this much memory when the slices are chosen (0 = what it has)
    int64_t opt_skm_batches = 0;   // partitions are counted + gathered in this many batches (0 = auto); tests force small values
    int64_t opt_union_samples = 0;     // hint: the sequences of the next count are the unitigs of this many samples (they share k-mers: partitions are planned twice as large from 4 on)
    int own_rank = 0, own_world = 1;   // mf_count_device_shard: only the k-mers this rank owns (level-1 digits [nd1 * rank / world, nd1 * (rank + 1) / world)) are counted
    int64_t opt_skm_pilot = 1;     // reads: a few level-1 digit regions are counted first to measure distinct k-mers per occurrence; the later levels are planned from it (0 = plan from the occurrences alone)
    #ifdef SKM_BIG_UNITS
    int64_t opt_skm_unit_distinct = 4400;
#elif defined(SKM_SMALL_UNITS)
    int64_t opt_skm_unit_distinct = 1100;
#else
    int64_t opt_skm_unit_distinct = 2200;
#endif   // ... so that a counting unit is expected to hold at most this many distinct k-mers (the LDS table takes C2_FILL = 3400 claims)
    int64_t opt_skm_dynq = 1;      // k_skm_count: units handed out from a counter as the workgroups get to them (0: fixed stride)
    int64_t opt_part_good = 220;   // ... and a TABLE partition at most this many k-mers that survive the cut (the graph kernels' LDS lookup table takes 352, mf_nbr.h)
    int64_t opt_unit_parts_long = 3;   // assembled sequences: log2 of the table partitions counted as one unit (k_gather_split_n cuts them apart)
    int64_t opt_skm_unit_records = 0;      // ... and at most this many super-k-mer records (the identical-record search of k_skm_count covers 2048); 0: 2000 for k >= 25, 4000 below (short k-mers make short records: mf_skm.hip)
    double last_pilot_rho = -1.0;
    double last_l1_per_occ = 0; int last_l1_k = 0;   // records (with padding) of the last run's level 1 per k-mer occurrence, for k = last_l1_k: the next sample's buffers are planned with it  // what the last pilot measured (diagnostics; < 0: none ran)
    int64_t opt_device_parse = 1;  // plain FASTA / FASTQ files are parsed on the device (mf_dparse.hip); files it is not sure about go to the host readers
    int64_t opt_device_parse_piece = 8 << 20, opt_device_parse_threads = 8;   // upload: piece size and host threads (each owns two staging chunks of a piece)
    int64_t opt_device_parse_min = 1 << 20;   // ... from this size on (bytes): a small file is not worth the kernels' launches
    int64_t opt_host_pinned = 0;   // staging buffers of the file readers / writers: 1 = hipHostMalloc (0.16 - 0.29 s per GB to get, 0.1 s to give back), 0 = plain host memory (copies to and from it run at the same 56 GB/s on this platform: tools/pin_alloc.hip)
    int64_t opt_file_cache_gb = 0; // > 0: tables / components written to files stay in HBM (up to this many GB) and are handed out when the same file is loaded again
    int64_t opt_gz_device_min = 32 << 20;   // .fa.gz / .fq.gz files of at least this size are inflated on many threads straight into HBM (mf_dparse_gz); tests: 0
    int64_t opt_gz_piece = 2 << 20;         // ... in pieces of at least this many compressed bytes (tests: 65536)
    int64_t opt_ut_double_after = 4;   // unitigs: walks still under way after this many chunked rounds (32, 128, 512, 4096 jumps) double the jump words instead (tests: 1)
    int64_t opt_ut_plain_rounds = 3;   // unitigs of a table without partitions (2k-bit tables, k < 20): rounds of doubling the one-hop jump words over all nodes before the walks (0: none)
    int64_t opt_cc_compress = 1;   // a pass that points every vertex at its root before the components' sizes are added up (mf_cc.hip, k_cc_compress)
    int64_t opt_stream_count = 1;  // mf_count_reads*: plain FASTA / FASTQ files are counted WHILE they cross PCIe (mf_stream.hip: upload || parse || level-1 scatter, piece by piece)
    int64_t opt_stream_count_min = 512 << 20, opt_stream_count_piece = 256 << 20;   // ... from this many bytes of files on; bytes per piece
    int64_t opt_stream_count_test_pct = 100;   // (tests: the digit regions get this share of what the sample says -- below 100 they overflow and the count steps back)
    uint64_t n_streamed = 0, n_stream_stepped_back = 0;     // counts that went that way / that started that way and were done again from whole files
    void *up_stream = nullptr;     // hipStream_t of the streamed count's uploads (lazy)
    int64_t opt_wide_skm = 1;      // mf_count_wide_device: super-k-mer records + LDS tables (mf_wskm.hip) instead of sorting every occurrence (0: the sort path, mf_wide.hip)
    int64_t opt_wide_skm_min = 1 << 20;     // ... from this many k-mer occurrences on (tests: 1)
    int64_t opt_wide_skm_lazy_order = 1;    // ... the table stays in the order of the counting units until an export / the cutter asks for ascending k-mers (0: ordered at once)
    int64_t opt_wide_skm_fine = 8;          // ... partitions of 1 / this of a unit's occurrences
    int64_t opt_wide_skm_pack = 1;          // ... the records are laid out in their units' order, identical ones counted once with a weight (k_wskm_pack; 0: off, 2: order only)
    int64_t opt_wide_skm_merge = 1;         // ... small neighbouring partitions share a counting unit (0: a unit per partition)
    int64_t opt_wide_skm_lead = 1;          // ... the kept entries are ordered by their leading 32 bits + a look at the runs of equal ones (0: all bits are sorted; tests)
    int64_t opt_wide_skm_unit = 2400;       // ... k-mer occurrences per counting unit (tests lower it: units that overflow the LDS table are counted in passes)
    int64_t opt_wide_finish = 1;   // mf_count_wide_device: radix passes over the leading 32 bits + the order inside the buckets in LDS (0: radix passes over all 2k bits)
    int64_t opt_wide_big_bucket = 256;   // ... buckets of more entries than this (<= 256) go through the LDS hash table instead of the walk (tests lower it)
    int64_t opt_wide_distinct = 1280;    // ... buckets of more distinct k-mers than this (<= 1280) are sorted aside (tests lower it)
    int64_t opt_wide_ablate = 0;         // ... TIMING ONLY, wrong tables: 1 = large buckets skipped, 2 = the representatives' walk skipped
    int64_t opt_wide_passes = 0;   // mf_count_wide_device: passes over the reads, each for one prefix class of the canonical k-mers (0 = as many as the memory asks for; tests force a number)
    int64_t opt_cc_sparse = 1;     // component cutter: threshold levels that few vertices reach run on a list of them (0: every level visits all vertices)
    int64_t opt_dcc_sparse = 0;    // sharded cutter, levels after the first: 1 = always the sparse set-up of the arrays over all vertex ids (tests)
    int64_t opt_dcc_test_fail = 0; int64_t dcc_test_calls[3] = {0, 0, 0};   // tests only: which * 1000 + n makes the n-th call of mf_dcc_merge (which = 1) / mf_dcc_level_local (2) on this context fail
    int64_t opt_nbr_global = 0;    // 1: neighbour lookups of the graph kernels through the HBM index only (A/B of mf_nbr.h)
    int64_t opt_scatter_fast = 1;  // k_skm_scatter: runs dealt evenly over the lanes through LDS where the level leaves room (0 = never)
    int64_t opt_ablate = 0;        // diagnostics only (tools/prof_count.py): results are WRONG when non-zero
    // workspace arena: a few large hipMalloc'd regions, sub-allocated with first-fit + coalescing free lists.
    // Everything runs on one stream, so a block can be handed out again as soon as it is released.
    struct span { size_t off, sz; };
    struct region { char *base; size_t size; std::vector<span> free_spans; };   // free_spans sorted by offset
    std::vector<region> regions;
    std::vector<struct mf_file_entry *> file_cache; size_t file_cache_bytes = 0; uint64_t file_cache_clock = 0;   // (mf_io.hip)
    size_t arena_bytes = 0;
    // counters a host can read (mf_ctx_stat): counting runs that started their slices over because a buffer found no place (mf_skm.hip);
    // read files the device parser took / handed to the host readers (mf_dparse.hip)
    uint64_t n_slice_restarts = 0, n_dparse_files = 0, n_dparse_stepped_back = 0, n_wide_big = 0, n_wide_hashed = 0, n_ut_doubled = 0, n_pilots = 0, n_gz_device = 0;
    // timers
    std::vector<mf_timer_rec> pending;
    std::vector<hipEvent_t> event_pool;
    struct ktime { int64_t n = 0; double total_ms = 0, max_ms = 0; };
    std::map<std::string, ktime> timings;
};

int  mf_alloc(mf_ctx *ctx, size_t bytes, void **out);   // cached hipMalloc
void mf_release(mf_ctx *ctx, void *p, size_t bytes);
size_t mf_arena_idle(const mf_ctx *ctx);     // back to the cache
int  mf_collect_timers(mf_ctx *ctx);
// debugging aid (option verbose >= 2): synchronise after a launch and report which kernel failed / hung
int  mf_debug_sync(mf_ctx *ctx, const char *what);
#define MF_DBG(ctx, what) do { if ((ctx)->opt_verbose >= 2) MF_TRY(mf_debug_sync((ctx), (what))); } while (0)

// RAII HIP-event timer around one kernel launch on ctx->stream (only when option profile=1)
struct mf_ktimer {
    mf_ctx *ctx; int idx;
    mf_ktimer(mf_ctx *c, const char *name);
    ~mf_ktimer();
};

// RAII roctx range around one stage of the path (count / unitigs / components / features and the file seams): what the reference
// prints as Timer lines at debug level (src/tools/KmersCounterMain.java:76,79; src/algo/ComponentsBuilder.java:60,88) shows up as
// named ranges in `rocprofv3 --marker-trace`.  The marker library (librocprofiler-sdk-roctx) is looked up at run time, once, when
// a profiler is attached (ROCP_TOOL_LIBRARIES set) or MF_ROCTX=1; otherwise a range costs one predictable branch.
struct mf_range {
    bool on;
    explicit mf_range(const char *name);
    ~mf_range();
};

template <typename T> struct mf_buf {   // RAII workspace buffer
    mf_ctx *ctx = nullptr; T *p = nullptr; size_t n = 0;
    bool owned = true;                   // false: a view of somebody else's buffer (borrow), never released here
    mf_buf() {}
    mf_buf(const mf_buf &) = delete;
    mf_buf &operator=(const mf_buf &) = delete;
    ~mf_buf() { reset(); }
    int alloc(mf_ctx *c, size_t count) {
        reset(); ctx = c; n = count; owned = true;
        void *q = nullptr;
        int r = mf_alloc(c, (count ? count : 1) * sizeof(T), &q);
        p = (T *)q;
        return r;
    }
    void reset() { if (p && owned) mf_release(ctx, p, (n ? n : 1) * sizeof(T)); p = nullptr; n = 0; owned = true; }
    void borrow(mf_ctx *c, T *q, size_t count) { reset(); ctx = c; p = q; n = count; owned = false; }
    void swap(mf_buf &o) { std::swap(ctx, o.ctx); std::swap(p, o.p); std::swap(n, o.n); std::swap(owned, o.owned); }
    T *take() { T *q = p; p = nullptr; return q; }   // ownership moves to the caller (who releases with mf_release)
    size_t bytes() const { return (n ? n : 1) * sizeof(T); }
};

// ---------------------------------------------------------------------------------------------
// opaque handle layouts
// ---------------------------------------------------------------------------------------------
struct mf_index {                 // open-addressed table in HBM: 16-byte slots {key, idx, val}
    void *slots = nullptr;        // ulonglong2-like: .x = key, .y = (uint64)idx | (uint64)val << 32
    uint64_t cap = 0;             // power of two
    // partitioned form (part_bits > 0): every partition of the dense table (top part_bits of mf_phash(key), or the
    // minimizer partition) has its own power-of-two region, dir[p] = (first slot << 6) | log2(region slots); linear
    // probing wraps inside the region.  Generic form (part_bits == 0): fmix64(key) & (cap-1).
    uint32_t part_bits = 0;
    uint64_t *dir = nullptr; size_t dir_bytes = 0;
    uint32_t skm_k = 0;           // != 0: partitions are MINIMIZER partitions of k-mers of this length (mf_skm_ph), else mf_phash
    // COMPACT partitioned form (round 3; what mf_table_ensure_index builds for a partitioned table): 4-byte slots
    // (12-bit tag << 20 | position of the key inside its partition; home slot and tag from the hash of the key -- of the key's canonical
    // INTERIOR for minimizer partitions, mf_cidx_hkey, so that the four neighbours of a side share a probe sequence), 0xFFFFFFFF = empty; a tag match is
    // confirmed against the dense key array, the value is read from the dense count array.  8 bytes of index per key at
    // load <= 0.5 instead of 32 (16-byte slots): the build writes a quarter, and a whole probe sequence sits in one 64-byte
    // line.  dir has TWO words per partition: (first slot << 6) | log2(region slots), first entry of the partition in the table.
    int compact = 0;
    const uint64_t *keys = nullptr; const uint16_t *counts = nullptr;
};
#define MF_CIDX_EMPTY 0xFFFFFFFFu
#define MF_CIDX_REL_BITS 20
struct mf_index_view { const void *slots; uint64_t mask; const uint64_t *dir; uint32_t part_bits, skm_k; int compact; const uint64_t *keys; const uint16_t *counts; };
static inline mf_index_view mf_view(const mf_index &ix) { return mf_index_view{ix.slots, ix.cap - 1, ix.dir, ix.part_bits, ix.skm_k, ix.compact, ix.keys, ix.counts}; }
struct mf_table {
    mf_ctx *ctx = nullptr;
    int refs = 1;                 // handles on this table (mf_table_destroy lets go of one): the file cache of the context holds one for a table it keeps
    int k = 0;
    uint64_t n = 0;               // distinct k-mers
    uint64_t n_occ = 0;           // occurrences fed in
    uint64_t n_records = 0; int record_bytes = 0;   // records the counting pass partitioned (mf_table_records)
    int cut_thr = -1;             // every entry has count > cut_thr (tables that went through a cut: no need to test again)
    std::vector<uint64_t> dropped_hist;   // cut inside the counting pass: [c] = distinct k-mers with count c <= cut_thr (not in the table)
    uint64_t *d_keys = nullptr;   // [n]
    uint16_t *d_counts = nullptr; // [n]
    size_t keys_bytes = 0, counts_bytes = 0;
    mf_index index;               // built lazily
    size_t index_bytes = 0;
    bool owns_arrays = true;      // false: d_keys / d_counts belong to another table (internal alias)
    // partition structure left by the counting path: entries of hash partition p (top part_bits of mf_phash) are
    // d_keys[d_part_off[p] .. d_part_off[p+1]); lets the index be built partition by partition without HBM atomics
    int part_bits = 0;
    int part_skm = 0;             // 1: the partitions are minimizer partitions (super-k-mer counting path), 0: mf_phash partitions
    uint64_t *d_part_off = nullptr; size_t part_off_bytes = 0;
};
struct mf_seqs {
    mf_ctx *ctx = nullptr;
    int k = 0;
    uint64_t n = 0, n_bases = 0;
    uint8_t *d_bases = nullptr;   // ASCII
    uint64_t *d_offsets = nullptr;  // [n+1]
    int32_t *d_avg = nullptr, *d_min = nullptr, *d_max = nullptr;
    uint64_t *d_startkey = nullptr;  // oriented start k-mer per sequence (for deterministic ordering)
    size_t bases_bytes = 0, offsets_bytes = 0, w_bytes = 0, sk_bytes = 0;
};
struct mf_reads {                 // the reads of a list of files, as the readers hand them on (mf_reads_load)
    mf_ctx *ctx = nullptr;
    uint64_t n = 0, n_bases = 0;
    uint8_t *d_bases = nullptr; uint64_t *d_offsets = nullptr;
    size_t bases_bytes = 0, offsets_bytes = 0;
};
struct mf_comps {
    mf_ctx *ctx = nullptr;
    int refs = 1;                 // (as mf_table::refs)
    int k = 0;
    uint64_t n = 0, n_kmers = 0;
    // per-component records on the host in final order; the member lists (offsets / kmers) are
    // materialised from the device arrays on first use (export / write)
    std::vector<uint64_t> sizes; std::vector<int64_t> weights; std::vector<int32_t> thr;
    std::vector<uint64_t> offsets; std::vector<uint64_t> kmers;
    bool host_ready = false;
    // device: component k-mers + component id per k-mer, and an index over them (for features)
    uint64_t *d_kmers = nullptr; uint32_t *d_comp = nullptr;
    size_t kmers_bytes = 0, comp_bytes = 0;
    mf_index index; size_t index_bytes = 0;
};

// ---- files this process has just written, kept as the objects they were written from (option file_cache, mf_io.hip): the next step of a
// matrix-builder run asks for the file and gets the table / the components that are still in HBM
struct mf_file_entry {
    std::string path; uint64_t size = 0; int64_t mtime_ns = 0;
    mf_table *t = nullptr; int thr = -1;       // every record of the file has count > thr
    mf_comps *c = nullptr;
    size_t bytes = 0; uint64_t stamp = 0;
};
void mf_file_cache_clear(mf_ctx *ctx);
mf_ctx *mf_comm_ctx(mf_comm *c);                 // (mf_comm.hip)
int mf_comm_agree(mf_comm *c, int ok);           // every rank says whether it is fine: 0 when all are, else < 0 on every rank
int mf_ensure_pin_pool(mf_ctx *ctx, size_t want);
// a file's bytes to HBM as they are (pread into staging chunks by several threads + hipMemcpyAsync; mf_dparse.hip): 0 ok, 1 no staging memory, < 0 error
int mf_upload_file(mf_ctx *ctx, int fd, size_t fsize, uint8_t *d_dst);
int mf_dparse_file(mf_ctx *ctx, const char *path, int fmt, mf_buf<uint8_t> &bases, mf_buf<uint64_t> &offsets, uint64_t *n_reads, uint64_t *n_bases);
int mf_dparse_gz(mf_ctx *ctx, const char *path, const void *packed, size_t packed_n, int fmt, mf_buf<uint8_t> &bases, mf_buf<uint64_t> &offsets, uint64_t *n_reads, uint64_t *n_bases);
int mf_dparse_mem(mf_ctx *ctx, const char *path, const void *mem, size_t mem_n, int fmt, mf_buf<uint8_t> &bases, mf_buf<uint64_t> &offsets, uint64_t *n_reads, uint64_t *n_bases);
int mf_index_build(mf_ctx *ctx, const uint64_t *d_keys, const uint16_t *d_vals, uint64_t n, mf_index *out,
                   size_t *bytes);
int mf_table_ensure_index(mf_table *t);
int mf_table_count_hist(const mf_table *t, std::vector<uint64_t> &hist);   // all counts, dropped k-mers included
// entries with count > threshold; when every entry passes, a non-owning alias of `t` (no copy) -- internal use only
int mf_table_filter_or_alias(const mf_table *t, int threshold, mf_table **out);
#define MF_SKM_FALLBACK 1          /* mf_count_skm: input does not suit the super-k-mer path, nothing was produced */
int mf_comps_materialize(mf_comps *c);
int mf_table_adopt(mf_ctx *ctx, int k, uint64_t n, uint64_t n_occ, uint64_t *d_keys, size_t kb, uint16_t *d_counts,
                   size_t cb, mf_table **out);

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
#ifdef __HIPCC__

// 64-bit finaliser (murmur3 fmix64).  Partition digits come from the TOP bits, LDS / HBM table
// slots from the LOW bits, so the two are independent.
__host__ __device__ __forceinline__ uint64_t mf_hash64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

// Cheap hash of the counting path (5 evaluations per k-mer occurrence across the passes, and those kernels are
// instruction-issue bound): one xor-shift + one 64-bit multiply.  Partition digits = TOP bits of the product (they
// depend on every key bit), LDS slot = fold of the high and low halves.  Measured on 2e7 real canonical 31-mers the
// partition sizes are Poisson-like, same as with fmix64.
__host__ __device__ __forceinline__ uint64_t mf_phash(uint64_t x) {
    x ^= x >> 31;
    return x * 0x9E3779B97F4A7C15ULL;
}
__host__ __device__ __forceinline__ uint32_t mf_pslot(uint64_t h) { return (uint32_t)(h ^ (h >> 32)); }

// reverse complement of a 2-bit packed k-mer (A0 G1 C2 T3 -> complement = 3-n), k in [1,31]
__host__ __device__ __forceinline__ uint64_t mf_revcomp(uint64_t x, int k) {
#if defined(__HIP_DEVICE_COMPILE__)
    x = __brevll(x);                                                          // reverse all 64 bits
    x = ((x & 0x5555555555555555ULL) << 1) | ((x >> 1) & 0x5555555555555555ULL);  // restore bit order inside each pair
#else
    x = ((x & 0x3333333333333333ULL) << 2) | ((x & 0xccccccccccccccccULL) >> 2);
    x = ((x & 0x0f0f0f0f0f0f0f0fULL) << 4) | ((x & 0xf0f0f0f0f0f0f0f0ULL) >> 4);
    x = ((x & 0x00ff00ff00ff00ffULL) << 8) | ((x & 0xff00ff00ff00ff00ULL) >> 8);
    x = ((x & 0x0000ffff0000ffffULL) << 16) | ((x & 0xffff0000ffff0000ULL) >> 16);
    x = (x << 32) | (x >> 32);
#endif
    return (~x) >> (64 - 2 * k);
}
__host__ __device__ __forceinline__ uint64_t mf_canon(uint64_t x, int k) {
    uint64_t r = mf_revcomp(x, k);
    return x < r ? x : r;
}

__device__ __forceinline__ int mf_lane() { return (int)(threadIdx.x & 63); }

// exclusive prefix sum of v over the wave; *total = wave sum
__device__ __forceinline__ uint32_t mf_wave_excl_scan(uint32_t v, uint32_t *total) {
    // inclusive scan with DPP (row shifts inside the 16-lane rows, then the row totals broadcast to the following rows):
    // six VALU instructions, no LDS round trips (the __shfl_up version is seven dependent ds_bpermute, ~100 cycles each)
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);   // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);   // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2 and 3
    *total = (uint32_t)__builtin_amdgcn_readlane(x, 63);
    return (uint32_t)x - v;
}

// Reserve `mine` consecutive indices from a global cursor with ONE atomic per wave (an atomic per lane on a single address
// serialises at ~12 ns each).  EVERY lane of the wave must call it (no early return before it); returns this lane's first index.
__device__ __forceinline__ uint32_t mf_wave_reserve(unsigned int *counter, uint32_t mine) {
    uint32_t tot;
    const uint32_t ex = mf_wave_excl_scan(mine, &tot);
    uint32_t base = 0;
    if (mf_lane() == 0 && tot) base = atomicAdd(counter, tot);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)base) + ex;
}

// Block-wide exclusive scan of one value per thread (blockDim.x multiple of 64, <= 1024).
// `scratch` must hold 17 uint32 in LDS.  Returns exclusive prefix; *block_total = sum.
__device__ __forceinline__ uint32_t mf_block_excl_scan(uint32_t v, uint32_t *scratch, uint32_t *block_total) {
    uint32_t wtot;
    uint32_t ex = mf_wave_excl_scan(v, &wtot);
    int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (mf_lane() == 0) scratch[wave] = wtot;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (int i = 0; i < nw; i++) { uint32_t t = scratch[i]; scratch[i] = acc; acc += t; }
        scratch[16] = acc;
    }
    __syncthreads();
    uint32_t r = ex + scratch[wave];
    *block_total = scratch[16];
    return r;
}

// Reserve `mine` consecutive indices from a global cursor with ONE atomic per WORKGROUP (every thread of the block calls it;
// scratch: 18 uint32 in LDS).  A cursor that every wave of a large grid hits costs ~12 ns per atomic, serialised.
__device__ __forceinline__ uint32_t mf_block_reserve(unsigned int *counter, uint32_t mine, uint32_t *scratch) {
    uint32_t tot;
    const uint32_t ex = mf_block_excl_scan(mine, scratch, &tot);
    if (threadIdx.x == 0) scratch[17] = tot ? atomicAdd(counter, tot) : 0u;
    __syncthreads();
    return scratch[17] + ex;
}

// ---- HBM open-addressed index lookup (16-byte slots) ----
// ---- minimizer partitions (super-k-mer counting path, mf_skm.hip) ----
// A k-mer's partition is decided by its MINIMIZER: the smallest mf_mmer_hash over the canonical forms of its k-M+1 M-mers.
// Consecutive k-mers of a read mostly share it, so a read is cut into a few super-k-mers (runs of k-mers with one
// minimizer) that travel through the radix passes as ONE 16-byte record instead of 8 bytes per k-mer, and graph
// neighbours mostly live in the same partition.  Orientation-independent: rc(x) has the reverse-complemented M-mers.
// The minimizer length follows k (round 5): M = 13 for k <= 25, M = 15 above.  A k-mer has k - M + 1 M-mers; the fewer, the shorter the runs of
// k-mers that share a minimizer (more records per k-mer occurrence) and the more often a graph neighbour lives in another partition (2 in
// k - M + 2: 25 % at k = 21 with M = 15, 20 % with M = 13 -- tools/nbr_locality.py, profiles/r05h_nbr_locality.txt).  Measured on 50 M reads,
// whole step (profiles/r05m_minimizer_length.txt): k = 21: 148.9 -> 127.6 ms, k = 23: 126.2 -> 117.7, k = 25: 116.6 -> 105.5 with M = 13; k = 27:
// 99.6 -> 101.2, k = 29: 95.1 -> 102.5 (too few distinct M-mers per partition: units overflow the LDS table and are counted in several
// passes); M = 11 loses everywhere (k_skm_count 68 - 125 ms instead of 20); M = 14 is within 1.6 % of 15 for k = 27 ... 30.  -DMF_SKM_M=<n> fixes one length for every k (experiments).
#ifdef MF_SKM_M
__host__ __device__ constexpr int mf_skm_m(int) { return MF_SKM_M; }
#else
__host__ __device__ constexpr int mf_skm_m(int k) { return k <= 25 ? 13 : 15; }
#endif
#define MF_SKM_MIN_K 20           // shorter k: too few M-mers per k-mer for runs worth packing -> one-record-per-k-mer path
#define MF_SKM_BASES 50           // bases a record can hold: x = bases 0..31, y = bases 32..49 | 22 digit bits | 6-bit k-mer count
__device__ __forceinline__ uint32_t mf_mmer_rc(uint32_t f, int M) {      // reverse complement of a 2 M-bit M-mer
    uint32_t r = __brev(~f);
    r = ((r & 0xAAAAAAAAu) >> 1) | ((r & 0x55555555u) << 1);
    return r >> (32 - 2 * M);
}
// The seed keeps low-complexity M-mers away from the bottom of the order: with a plain multiply A..A hashes to 0, the
// global minimum, so EVERY k-mer that touches a poly-A / poly-T stretch would meet in one partition (hundreds of thousands
// of distinct k-mers in real reads: the LDS table overflows).  0x051E6720 puts all 38 canonical repeats of period 1-3
// (homopolymers, di- and trinucleotide microsatellites) of length 15 above the 77th percentile, 0x00B9107F those of length 13 above the
// 73rd (with the 15-mers' seed the worst 13-mer repeat sits at the 2nd percentile), so they are almost never the minimum.
// (No xor-shift after the multiply: the hash only ORDERS the M-mers, and the order is decided by the well-mixed top bits.)
__host__ __device__ constexpr uint32_t mf_mmer_seed(int M) { return M == 13 ? 0x00B9107Fu : 0x051E6720u; }
__device__ __forceinline__ uint32_t mf_mmer_hash(uint32_t canon, int M) { return (canon ^ mf_mmer_seed(M)) * 0x9E3779B1u; }
// minimizer hash -> partition hash (the minimum of several uniform values is not uniform: mix again)
__device__ __forceinline__ uint32_t mf_remix32(uint32_t h) { h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; return h ^ (h >> 16); }
__device__ __forceinline__ uint32_t mf_skm_ph(uint64_t key, int k) {
    const int M = mf_skm_m(k);
    const uint32_t mm = (1u << (2 * M)) - 1u;
    uint32_t f = (uint32_t)(key >> (2 * (k - M))) & mm, r = mf_mmer_rc(f, M);
    uint32_t best = mf_mmer_hash(f < r ? f : r, M);
    for (int j = k - M - 1; j >= 0; j--) {
        const uint32_t b = (uint32_t)(key >> (2 * j)) & 3u;
        f = ((f << 2) | b) & mm;
        r = (r >> 2) | ((3u - b) << (2 * M - 2));
        const uint32_t h = mf_mmer_hash(f < r ? f : r, M);
        best = h < best ? h : best;
    }
    return mf_remix32(best);
}
// The 8 graph neighbours of x share all but one of its M-mers, so their minimizers need ONE pass over x, not eight:
// smallest M-mer hash of x without its first M-mer (right neighbours drop it) and without its last (left neighbours).
// (*own: the k-mer's own minimizer hash, before the re-mix: equal minimizer hashes mean equal partitions)
__device__ __forceinline__ void mf_skm_nbr_mins(uint64_t x, int k, uint32_t *no_first, uint32_t *no_last, uint32_t *own = nullptr) {
    const int M = mf_skm_m(k);
    const uint32_t mm = (1u << (2 * M)) - 1u;
    uint32_t f = (uint32_t)(x >> (2 * (k - M))) & mm, r = mf_mmer_rc(f, M);
    uint32_t h = mf_mmer_hash(f < r ? f : r, M);
    uint32_t a = 0xFFFFFFFFu, b = h;
    for (int j = k - M - 1; j >= 0; j--) {
        const uint32_t c = (uint32_t)(x >> (2 * j)) & 3u;
        f = ((f << 2) | c) & mm;
        r = (r >> 2) | ((3u - c) << (2 * M - 2));
        h = mf_mmer_hash(f < r ? f : r, M);
        a = h < a ? h : a;
        if (j > 0) b = h < b ? h : b;
    }
    *no_first = a; *no_last = b;
    if (own) *own = h < b ? h : b;                    // (h: the last M-mer's)
}
// partition hash of the neighbour y = x[1..]+c (right) / c+x[..k-2] (left) from the matching minimum of x
// (_mn: the neighbour's minimizer hash itself; the partition hash is its re-mix)
__device__ __forceinline__ uint32_t mf_skm_mn_right(uint64_t y, int k, uint32_t no_first_of_x) {
    const int M = mf_skm_m(k);
    const uint32_t f = (uint32_t)y & ((1u << (2 * M)) - 1u), r = mf_mmer_rc(f, M);
    const uint32_t h = mf_mmer_hash(f < r ? f : r, M);
    return h < no_first_of_x ? h : no_first_of_x;
}
__device__ __forceinline__ uint32_t mf_skm_mn_left(uint64_t y, int k, uint32_t no_last_of_x) {
    const int M = mf_skm_m(k);
    const uint32_t f = (uint32_t)(y >> (2 * (k - M))) & ((1u << (2 * M)) - 1u), r = mf_mmer_rc(f, M);
    const uint32_t h = mf_mmer_hash(f < r ? f : r, M);
    return h < no_last_of_x ? h : no_last_of_x;
}
__device__ __forceinline__ uint32_t mf_skm_ph_right(uint64_t y, int k, uint32_t no_first_of_x) { return mf_remix32(mf_skm_mn_right(y, k, no_first_of_x)); }
__device__ __forceinline__ uint32_t mf_skm_ph_left(uint64_t y, int k, uint32_t no_last_of_x) { return mf_remix32(mf_skm_mn_left(y, k, no_last_of_x)); }
struct mf_slot { uint64_t key; uint32_t idx; uint32_t val; };
// ascending (key, value) order (mf_sort.hip); select + sort of a table's entries with count > threshold (mf_table.hip)
int mf_sort_pairs(mf_ctx *ctx, const uint64_t *d_keys_in, const uint16_t *d_vals_in, uint64_t n, int key_bits, uint64_t *d_keys_out,
                  uint16_t *d_vals_out);
int mf_sort_u32_pairs(mf_ctx *ctx, const uint32_t *d_keys_in, const uint32_t *d_vals_in, uint64_t n, int bits, uint32_t *d_keys_out,
                      uint32_t *d_vals_out);
int mf_sort_u32_u64(mf_ctx *ctx, const uint32_t *d_keys_in, const uint64_t *d_vals_in, uint64_t n, int bits, uint32_t *d_keys_out,
                    uint64_t *d_vals_out);
int mf_sort_u64_u64(mf_ctx *ctx, const uint64_t *d_keys_in, const uint64_t *d_vals_in, uint64_t n, int bits, uint64_t *d_keys_out, uint64_t *d_vals_out);
int mf_sort_u64_u64_pingpong(mf_ctx *ctx, uint64_t *k0, uint64_t *v0, uint64_t n, int first_bit, int bits, uint64_t *k1, uint64_t *v1, int *in_second);
int mf_sort_u64_u64_range(mf_ctx *ctx, const uint64_t *d_keys_in, const uint64_t *d_vals_in, uint64_t n, int first_bit, int bits, uint64_t *d_keys_out, uint64_t *d_vals_out);
int mf_sort_kmers_by_comp(mf_ctx *ctx, const uint32_t *d_comp, const uint64_t *d_kmers, uint64_t n, int key_bits, uint32_t n_comps,
                          uint64_t *d_out);
// the canonical INTERIOR of a k-mer: its middle k-2 bases or their reverse complement, whichever is smaller (rcx = mf_revcomp(x, k)).
// The four k-mers that extend a (k-1)-mer on one side share it, and so do their reverse complements: an index hashed on the interior
// keeps the four neighbours of a side in ONE probe sequence (mf_nbr.h, mf_index_walk_side).
__host__ __device__ __forceinline__ uint64_t mf_interior(uint64_t x, uint64_t rcx, int k) {
    const uint64_t WM = (1ull << (2 * k - 4)) - 1ull;
    const uint64_t w = (x >> 2) & WM, rw = (rcx >> 2) & WM;
    return w < rw ? w : rw;
}
// what the compact index of a table hashes for home slot and tag: the interior for minimizer partitions of k-mers (skm_k = k), else the key
__host__ __device__ __forceinline__ uint64_t mf_cidx_hkey(uint64_t key, uint32_t skm_k) {
    return skm_k >= 3u ? mf_interior(key, mf_revcomp(key, (int)skm_k), (int)skm_k) : key;
}
// ph: the key's partition hash if the caller has it already (minimizer partitions only), see mf_index_find
__device__ __forceinline__ bool mf_index_find_ph(const mf_index_view &ix, uint64_t key, uint32_t ph, uint32_t *idx, uint32_t *val) {
    if (ix.compact) {
        const uint64_t h = mf_phash(mf_cidx_hkey(key, ix.skm_k));
        const uint64_t part = ix.skm_k ? (uint64_t)(ph >> (32 - ix.part_bits)) : (h >> (64 - ix.part_bits));
        const ulonglong2 d = *reinterpret_cast<const ulonglong2 *>(&ix.dir[2 * part]);
        const uint32_t *__restrict__ reg = reinterpret_cast<const uint32_t *>(ix.slots) + (d.x >> 6);
        const uint32_t rmask = (1u << (uint32_t)(d.x & 63ull)) - 1u, hs = mf_pslot(h), tag = hs >> MF_CIDX_REL_BITS;
        uint32_t s = hs & rmask;
        for (;;) {
            const uint32_t v = reg[s];
            if (v == MF_CIDX_EMPTY) return false;
            if ((v >> MF_CIDX_REL_BITS) == tag) {
                const uint64_t i = d.y + (uint64_t)(v & ((1u << MF_CIDX_REL_BITS) - 1u));
                if (ix.keys[i] == key) { *idx = (uint32_t)i; *val = ix.counts ? (uint32_t)ix.counts[i] : 0u; return true; }
            }
            s = (s + 1u) & rmask;
        }
    }
    const mf_slot *__restrict__ slots = reinterpret_cast<const mf_slot *>(ix.slots);
    uint64_t base = 0, rmask = ix.mask, s;
    if (ix.part_bits) {
        const uint64_t h = mf_phash(key);
        const uint64_t part = ix.skm_k ? (uint64_t)(ph >> (32 - ix.part_bits)) : (h >> (64 - ix.part_bits));
        const uint64_t d = ix.dir[part];
        base = d >> 6;
        rmask = (1ull << (d & 63ull)) - 1;
        s = mf_pslot(h) & rmask;
    } else s = mf_hash64(key) & rmask;
    for (;;) {
        const ulonglong2 raw = *reinterpret_cast<const ulonglong2 *>(&slots[base + s]);
        if (raw.x == key) { *idx = (uint32_t)raw.y; *val = (uint32_t)(raw.y >> 32); return true; }
        if (raw.x == MF_EMPTY) return false;
        s = (s + 1) & rmask;
    }
}
// (the two halves of mf_index_walk_side for a compact index: the directory entry of the partition, then the probe sequence -- a
// caller with other work to do issues the first, does the work, and walks when the entry has arrived)
__device__ __forceinline__ ulonglong2 mf_index_side_dir(const mf_index_view &ix, uint32_t ph) {
    return *reinterpret_cast<const ulonglong2 *>(&ix.dir[2 * (uint64_t)(ph >> (32 - ix.part_bits))]);
}
#ifndef MF_WALK_CAND
#define MF_WALK_CAND 2
#endif
// hs: the interior's slot hash (home slot = hs & region mask, tag = its top bits)
__device__ __forceinline__ uint32_t mf_index_side_hs(uint64_t pa, uint64_t pb, int k) {
    const uint64_t WM = (1ull << (2 * k - 4)) - 1ull;
    const uint64_t wa = pa & WM, wb = pb >> 2;           // the shared interior and its reverse complement
    return mf_pslot(mf_phash(wa < wb ? wa : wb));
}
__device__ __forceinline__ uint32_t mf_index_side_home(const mf_index_view &ix, const ulonglong2 d, uint32_t hs) {
    return (reinterpret_cast<const uint32_t *>(ix.slots) + (d.x >> 6))[hs & ((1u << (uint32_t)(d.x & 63ull)) - 1u)];
}
// v0: the home slot's word (mf_index_side_home), read by the caller ahead of time
__device__ __forceinline__ void mf_index_walk_side_v(const mf_index_view &ix, const ulonglong2 d, uint32_t hs, uint32_t v0, uint64_t pa, uint64_t pb, uint32_t side, uint32_t want,
                                                     int k, uint32_t (&out)[4], uint32_t *rev) {
    const uint64_t LM = (1ull << (2 * k - 2)) - 1ull;
    out[0] = out[1] = out[2] = out[3] = 0xFFFFFFFFu;
    uint32_t rv = 0;
    const uint32_t *__restrict__ reg = reinterpret_cast<const uint32_t *>(ix.slots) + (d.x >> 6);
    const uint32_t rmask = (1u << (uint32_t)(d.x & 63ull)) - 1u, tag = hs >> MF_CIDX_REL_BITS;
    // the probe sequence first (its slots sit next to each other: one line, seldom two), the keys behind the matching tags
    // afterwards and TOGETHER: two memory latencies per lookup, not one per slot and one per key in turn
    constexpr uint32_t NC = MF_WALK_CAND;                 // keys read together
    uint32_t s = hs & rmask, nc = 0, cand[NC];
#pragma unroll
    for (uint32_t q = 0; q < NC; q++) cand[q] = 0u;
    auto resolve = [&]() {
        uint64_t K[NC];
#pragma unroll
        for (uint32_t q = 0; q < NC; q++) K[q] = q < nc ? ix.keys[d.y + (uint64_t)cand[q]] : ~0ull;
#pragma unroll
        for (uint32_t q = 0; q < NC; q++) {
            if (q >= nc) continue;
            // the neighbour itself (K >> 2 == pa on the right side, K's low bases == pb on the left) or its reverse complement; a
            // palindrome is both and counts as itself
            const bool ma = (K[q] >> 2) == pa, mb = (K[q] & LM) == pb;
            const uint32_t ca = ((uint32_t)K[q] & 3u) ^ (side ? 3u : 0u), cb = (uint32_t)(K[q] >> (2 * k - 2)) ^ (side ? 0u : 3u);
            const bool fwd = side ? mb : ma;
            if ((ma || mb)) {
                const uint32_t c = fwd ? (side ? cb : ca) : (side ? ca : cb);
                if ((want >> c) & 1u) {
                    const uint32_t at = (uint32_t)(d.y + (uint64_t)cand[q]);
#pragma unroll
                    for (uint32_t r = 0; r < 4; r++) if (c == r) out[r] = at;
                    rv = (rv & ~(1u << c)) | ((fwd ? 0u : 1u) << c);
                }
            }
        }
        nc = 0;
    };
    for (uint32_t v = v0; v != MF_CIDX_EMPTY;) {
        if ((v >> MF_CIDX_REL_BITS) == tag) {
            const uint32_t rel = v & ((1u << MF_CIDX_REL_BITS) - 1u);
#pragma unroll
            for (uint32_t q = 0; q < NC; q++) if (nc == q) cand[q] = rel;
            if (++nc == NC) resolve();
        }
        s = (s + 1u) & rmask;
        v = reg[s];
    }
    if (nc) resolve();
    *rev = rv;
}
__device__ __forceinline__ void mf_index_walk_side_d(const mf_index_view &ix, const ulonglong2 d, uint64_t pa, uint64_t pb, uint32_t side, uint32_t want, int k, uint32_t (&out)[4],
                                                     uint32_t *rev) {
    const uint32_t hs = mf_index_side_hs(pa, pb, k);
    mf_index_walk_side_v(ix, d, hs, mf_index_side_home(ix, d, hs), pa, pb, side, want, k, out, rev);
}
// The neighbours of ONE side of a k-mer x in one probe sequence (minimizer-partitioned tables of k-mers).  The four k-mers y_c
// share k-1 bases with x: pa = what (K >> 2) of a stored key K is if K is y_c on the right side (x's last k-1 bases) or rc(y_c) on the
// left (rc(x)'s last k-1 bases); pb = what K's low k-1 bases are if K is rc(y_c) on the right / y_c on the left.  side 0: right
// (y_c = x[1..] + c), 1: left (y_c = c + x[..k-2]).  want: bit c set = look for neighbour c; ph: the partition hash they share (callers
// group by minimizer).  out[c] = table index or 0xFFFFFFFF; *rev: bit c set = the table holds neighbour c as its reverse complement.
__device__ __forceinline__ void mf_index_walk_side(const mf_index_view &ix, uint32_t ph, uint64_t pa, uint64_t pb, uint32_t side, uint32_t want, int k, uint32_t (&out)[4],
                                                   uint32_t *rev) {
    out[0] = out[1] = out[2] = out[3] = 0xFFFFFFFFu;
    *rev = 0;
    if (!ix.compact) {                                   // (a table with an oversized partition has the generic index: one lookup per neighbour)
#pragma unroll
        for (uint32_t c = 0; c < 4; c++) {
            if (!((want >> c) & 1u)) continue;
            const uint64_t y = side ? (((uint64_t)c << (2 * k - 2)) | pb) : ((pa << 2) | c);
            const uint64_t r = side ? ((pa << 2) | (3u - c)) : (pb | ((uint64_t)(3u - c) << (2 * k - 2)));
            uint32_t ii, val;
            if (mf_index_find_ph(ix, y < r ? y : r, ph, &ii, &val)) { out[c] = ii; if (r < y) *rev |= 1u << c; }
        }
        return;
    }
    mf_index_walk_side_d(ix, mf_index_side_dir(ix, ph), pa, pb, side, want, k, out, rev);
}
__device__ __forceinline__ bool mf_index_find(const mf_index_view &ix, uint64_t key, uint32_t *idx, uint32_t *val) {
    return mf_index_find_ph(ix, key, ix.skm_k ? mf_skm_ph(key, (int)ix.skm_k) : 0u, idx, val);
}
// =============================================================================================
// single-block exclusive scan (u32 in -> u64 out); PAD (a power of two) rounds every item up to a multiple of PAD
// =============================================================================================
template <int PAD>
static __global__ __launch_bounds__(1024) void k_scan(const uint32_t *__restrict__ in, uint64_t *__restrict__ out,
                                               uint64_t n, uint64_t *__restrict__ total) {
    __shared__ uint64_t sums[1024];
    uint64_t per = (n + 1023) / 1024;
    uint64_t lo = (uint64_t)threadIdx.x * per;
    uint64_t hi = lo + per < n ? lo + per : n;
    uint64_t s = 0;
    for (uint64_t i = lo; i < hi; i++) { uint64_t v = in[i]; v = (v + (PAD - 1)) & ~(uint64_t)(PAD - 1); s += v; }
    sums[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < 64) {                               // the 1024 partial sums: 16 per lane of one wave, then a wave scan
        const uint32_t lane = threadIdx.x;
        uint64_t loc[16], s16 = 0;
#pragma unroll
        for (int j = 0; j < 16; j++) { loc[j] = sums[lane * 16 + j]; s16 += loc[j]; }
        uint64_t inc = s16;
        for (int d = 1; d < 64; d <<= 1) { const uint64_t o = __shfl_up(inc, d, 64); if ((int)lane >= d) inc += o; }
        uint64_t acc = inc - s16;
#pragma unroll
        for (int j = 0; j < 16; j++) { const uint64_t t = loc[j]; sums[lane * 16 + j] = acc; acc += t; }
        if (lane == 63) { *total = inc; out[n] = inc; }
    }
    __syncthreads();
    uint64_t base = sums[threadIdx.x];
    for (uint64_t i = lo; i < hi; i++) { uint64_t v = in[i]; v = (v + (PAD - 1)) & ~(uint64_t)(PAD - 1); out[i] = base; base += v; }
}


// ---- multi-block exclusive scan (u32 in -> u64 out) for long arrays: tile sums, scan of the tile sums, add-back ----
#define MF_SCAN_TILE 4096
template <int PAD>
static __global__ __launch_bounds__(1024) void k_scan_tiles(const uint32_t *__restrict__ in, uint64_t *__restrict__ out, uint64_t n,
                                                            uint32_t *__restrict__ tile_sum) {
    __shared__ uint32_t scratch[17];
    const uint64_t base = (uint64_t)blockIdx.x * MF_SCAN_TILE + (uint64_t)threadIdx.x * 4;
    uint32_t v[4], s = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) { uint32_t x = base + j < n ? in[base + j] : 0u; x = (x + (uint32_t)(PAD - 1)) & ~(uint32_t)(PAD - 1); v[j] = x; s += x; }
    uint32_t tot;
    uint32_t ex = mf_block_excl_scan(s, scratch, &tot);
#pragma unroll
    for (int j = 0; j < 4; j++) { if (base + j < n) out[base + j] = ex; ex += v[j]; }
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = tot;
}
static __global__ __launch_bounds__(1024) void k_scan_addback(uint64_t *__restrict__ out, uint64_t n, const uint64_t *__restrict__ tile_off) {
    const uint64_t base = (uint64_t)blockIdx.x * MF_SCAN_TILE + (uint64_t)threadIdx.x * 4;
    const uint64_t o = tile_off[blockIdx.x];
#pragma unroll
    for (int j = 0; j < 4; j++) if (base + j < n) out[base + j] += o;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = tile_off[gridDim.x];
}
// out[0..n] = exclusive prefix of in (out[n] = total, also written to *total); tiles must hold < 2^32 each
template <int PAD>
static int mf_scan(mf_ctx *ctx, const uint32_t *in, uint64_t *out, uint64_t n, uint64_t *total) {
    hipStream_t st = ctx->stream;
    if (n <= 65536) { k_scan<PAD><<<1, 1024, 0, st>>>(in, out, n, total); return MF_OK; }
    const uint64_t nt = (n + MF_SCAN_TILE - 1) / MF_SCAN_TILE;
    mf_buf<uint32_t> ts; MF_TRY(ts.alloc(ctx, nt));
    mf_buf<uint64_t> to; MF_TRY(to.alloc(ctx, nt + 1));
    k_scan_tiles<PAD><<<(unsigned)nt, 1024, 0, st>>>(in, out, n, ts.p);
    k_scan<1><<<1, 1024, 0, st>>>(ts.p, to.p, nt, total);
    k_scan_addback<<<(unsigned)nt, 1024, 0, st>>>(out, n, to.p);
    return MF_OK;
}
#endif  // __HIPCC__

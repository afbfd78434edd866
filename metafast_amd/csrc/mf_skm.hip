// mf_skm.hip -- k-mer counting through SUPER-K-MERS (minimizer partitions), the default path for k >= MF_SKM_MIN_K.
//
// Same job as mf_count.hip (IOUtils.loadReads, src/io/IOUtils.java:756-768: every k-mer of every read ->
// BigLong2ShortHashMap.addAndBound(canonical, 1)), same result (dense (key,count) arrays), different traffic: the
// one-record-per-k-mer path moves 8 bytes per k-mer OCCURRENCE through two radix passes and the count pass (3 x 96 GB
// at 1.2e10 occurrences).  Here a read is cut into runs of consecutive k-mers that share their minimizer (mf_common.h),
// and a run of r k-mers travels as ONE 16-byte record holding its r+k-1 bases (about 7 k-mers per record at k=31):
//
//   S1 k_skm_hist     ASCII -> 2-bit -> M-mer hashes -> sliding-window minimum -> runs; per-block histogram of the
//                     records' level-1 digit (top bits of the partition hash of the run's minimizer).  On large inputs it
//                     runs over every 16th tile only and just SIZES the digit regions
//      k_scan         region starts / exact output ranges (padded to whole 64-byte lines)
//   S2 k_skm_scatter  the minimizer scan (again, where S1 ran over everything), records built and radix-partitioned through
//                     64-byte LDS staging lines; one-pass form: regions reserved chunk by chunk during the scatter
//      skm_pilot      (reads) a few hundred would-be units of level 1 are copied out and counted: distinct k-mers per occurrence
//                     and the share that survives the cut decide the bits of the following levels and the table partitions per
//                     unit -- the plan assumes no sequencing depth
//   S3 k_skm_split    further levels: the digit is read from the record (22 digit bits travel with it)
//   S4 k_skm_count    one counting unit per 512-thread workgroup at a time, units claimed from a counter three ahead: identical
//                     records are told apart first (a pass of the records through the empty table: the k-mers of a record that
//                     occurs n times are inserted once, with weight n); every wave deals the surviving records out as items of
//                     <= 2 k-mers -> open-addressed count table in LDS -> compacted (key,count) slices; a batch of units at a time
//      k_gather       slices of the batch -> dense arrays, every unit cut into 2^s table partitions by the next bits of the
//                     partition hash (the HBM index is built partition by partition, mf_table.hip)
//
// A partition holds ALL occurrences of its k-mers, so the counts are exact; only the grouping of the dense table
// differs from mf_count.hip (by minimizer partition instead of by hash partition).  If a partition has more distinct
// k-mers than the LDS table holds, the caller falls back to mf_count.hip's path.
#include "mf_common.h"
#include "mf_count_dev.h"
#include <algorithm>
#include <memory>
#include <vector>

typedef ulonglong2 skm_rec;
typedef uint32_t skm_v4 __attribute__((ext_vector_type(4)));
#define SKM_LINE 4                 // records per 64-byte staging line
#define SKM_QCAP 16
#define SKM_DIGIT_BITS 22          // digit bits stored in a record (levels after the first)
#ifdef SKM_BIG_UNITS              /* experiment (round 5): ONE workgroup of 1024 threads per CU with a table of 8192 slots -- units twice as large, half as many.
                                     Correct (77 parity tests) and SLOWER: k_skm_count 25.9 -> 27.9 ms on the benchmark, 34.9 -> 39.2 without the search for
                                     identical records, 61.1 -> 77.7 at 5-fold depth, 16.5 -> 18.4 at k = 21 (profiles/r05y_big_units.txt): sixteen waves wait
                                     longer at a unit's barriers than eight, and a CU with one workgroup has nothing to run while that one compacts.  Not built in. */
#define SKM_CT 1024
#elif defined(SKM_SMALL_UNITS)    /* experiment (round 6): workgroups of 256 threads with a table of 2048 slots, three per CU -- units half as large, twice as many */
#define SKM_CT 256
#else
#define SKM_CT 512                 // threads of k_skm_count
#endif
#define SKM_NW (SKM_CT / 64)       // its waves

__device__ __forceinline__ bool skm_rec_valid(const skm_rec &r) { return ((uint32_t)r.y & 63u) != 63u; }
__device__ __forceinline__ uint32_t skm_rec_n(const skm_rec &r) { return (uint32_t)r.y & 63u; }
__device__ __forceinline__ uint32_t skm_rec_digits(const skm_rec &r) { return (uint32_t)(r.y >> 6) & ((1u << SKM_DIGIT_BITS) - 1u); }

// =============================================================================================
// one lane = one 32-position word of the base stream: minimizer hash of each of its 32 k-mers, run starts
// =============================================================================================
template <int K> struct skm_word {
    static constexpr int W = K - mf_skm_m(K) + 1;       // M-mers per k-mer
    static constexpr int NM = 31 + W;                   // M-mers of the word's 32 k-mers
    static constexpr int RMAX = (MF_SKM_BASES - (K - 1)) < 20 ? (MF_SKM_BASES - (K - 1)) : 20;   // k-mers per record (<= 10 items of 2 in k_skm_count)
    uint32_t D[4];        // 64 bases from the word's first position, 2 bits each, first base in the top bits of D[0]
    uint32_t mh[32];      // minimizer hash of the k-mer at each position
    uint32_t valid;       // positions that start a k-mer (from the bitmap)
    uint32_t cut;         // positions where a run starts
};

// the second half of the scan: minimizer hash of each of the word's 32 k-mers from its M-mers' hashes, run starts
template <int K>
__device__ __forceinline__ void skm_scan_tail(skm_word<K> &S, uint32_t (&hs)[skm_word<K>::NM], uint32_t m) {
    constexpr int W = skm_word<K>::W, NM = skm_word<K>::NM;
    // sliding-window minimum over W M-mers with three-input minima (v_min3_u32): spans of 3, then of 9, then two (or
    // three) overlapping spans: 118 instructions for W = 17 instead of 214 with doubling
    static_assert(W >= 3 && W <= 18, "window");
    auto min3 = [](uint32_t a, uint32_t b, uint32_t c) { const uint32_t m = a < b ? a : b; return m < c ? m : c; };
#pragma unroll
    for (int i = 0; i + 2 < NM; i++) hs[i] = min3(hs[i], hs[i + 1], hs[i + 2]);                 // hs[i] = min of M-mers i .. i+2
    if (W >= 9) {
#pragma unroll
        for (int i = 0; i + 8 < NM; i++) hs[i] = min3(hs[i], hs[i + 3], hs[i + 6]);             // i .. i+8
#pragma unroll
        for (int j = 0; j < 32; j++) S.mh[j] = hs[j] < hs[j + W - 9] ? hs[j] : hs[j + W - 9];
    } else {
        constexpr int D = W - 3 < 3 ? W - 3 : 3;
#pragma unroll
        for (int j = 0; j < 32; j++) S.mh[j] = min3(hs[j], hs[j + D], hs[j + W - 3]);
    }
    // bit j: the k-mer at j has the minimizer of the one before it.  Built from the top with compare + add-with-carry (two
    // instructions per position instead of compare, select, or)
    uint32_t acc = 0;
#pragma unroll
    for (int j = 31; j >= 1; j--)
        asm("v_cmp_eq_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"(S.mh[j]), "v"(S.mh[j - 1]) : "vcc");
    const uint32_t same = acc << 1;
    S.cut = m & ~(same & (m << 1));
}

template <int K>
__device__ __forceinline__ void skm_scan_word(skm_word<K> &S, const uint8_t *__restrict__ bases, uint64_t n_bases, uint64_t w, uint32_t m) {
    constexpr int NM = skm_word<K>::NM, M = mf_skm_m(K);
    // four unconditional 16-byte loads (a guarded `cond ? p[i] : zero` compiles to sixteen predicated dword loads); chunks
    // that start beyond the buffer are re-pointed at the first one and zeroed afterwards
    const uint64_t b0 = w * 32;
    const uint4 *p = reinterpret_cast<const uint4 *>(bases + (b0 < n_bases ? b0 : 0));
    const bool in1 = b0 + 16 < n_bases, in2 = b0 + 32 < n_bases, in3 = b0 + 48 < n_bases;
    const uint4 c0 = p[0], c1 = p[in1 ? 1 : 0], c2 = p[in2 ? 2 : 0], c3 = p[in3 ? 3 : 0];
    S.D[0] = b0 < n_bases ? mf_dec16(c0) : 0u; S.D[1] = in1 ? mf_dec16(c1) : 0u; S.D[2] = in2 ? mf_dec16(c2) : 0u; S.D[3] = in3 ? mf_dec16(c3) : 0u;
    S.valid = m;
    uint32_t hs[NM];
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < NM; i++) {
        const int q = i >> 4, o = i & 15;       // bases i .. i+M-1 start in dword q at base offset o
        const uint32_t top = o ? __builtin_amdgcn_alignbit(S.D[q], S.D[q + 1 < 4 ? q + 1 : 3], 32 - 2 * o) : S.D[q];
        const uint32_t f = top >> (32 - 2 * M);
        r = (i == 0) ? mf_mmer_rc(f, M) : ((r >> 2) | ((3u - (f & 3u)) << (2 * M - 2)));
        hs[i] = mf_mmer_hash(f < r ? f : r, M);
    }
    skm_scan_tail<K>(S, hs, m);
}
// The same scan with the HALO taken from the next lane: a word's 32 k-mers need the M-mers of 48 positions and 64 bases, the
// last 16 / 32 of which are the next word's first ones -- which the next lane computes anyway.  v_mov_b32_dpp wave_shl:1 hands
// them over (18 moves instead of two 16-byte loads, their decoding and 16 M-mer hashes: ~170 of the scan's ~900 instructions).
// Lane 63 has no next lane: the caller gives it no k-mers of its own (it scans the word that lane 0 of the wave's NEXT batch
// owns), a wave covers 63 words.
__device__ __forceinline__ uint32_t skm_from_next_lane(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t skm_from_prev_lane(uint32_t v) {          // (lane 0: 0)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}
template <int K>
__device__ __forceinline__ void skm_scan_word_halo(skm_word<K> &S, const uint8_t *__restrict__ bases, uint64_t n_bases, uint64_t w, uint32_t m) {
    constexpr int NM = skm_word<K>::NM, M = mf_skm_m(K);
    const uint64_t b0 = w * 32;
    const uint4 *p = reinterpret_cast<const uint4 *>(bases + (b0 < n_bases ? b0 : 0));
    const bool in1 = b0 + 16 < n_bases;
    const uint4 c0 = p[0], c1 = p[in1 ? 1 : 0];
    S.D[0] = b0 < n_bases ? mf_dec16(c0) : 0u; S.D[1] = in1 ? mf_dec16(c1) : 0u;
    S.D[2] = skm_from_next_lane(S.D[0]); S.D[3] = skm_from_next_lane(S.D[1]);
    S.valid = m;
    uint32_t hs[NM];
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 32; i++) {
        const int q = i >> 4, o = i & 15;       // bases i .. i+M-1 start in dword q at base offset o
        const uint32_t top = o ? __builtin_amdgcn_alignbit(S.D[q], S.D[q + 1], 32 - 2 * o) : S.D[q];
        const uint32_t f = top >> (32 - 2 * M);
        r = (i == 0) ? mf_mmer_rc(f, M) : ((r >> 2) | ((3u - (f & 3u)) << (2 * M - 2)));
        hs[i] = mf_mmer_hash(f < r ? f : r, M);
    }
#pragma unroll
    for (int i = 32; i < NM; i++) hs[i] = skm_from_next_lane(hs[i - 32]);
    skm_scan_tail<K>(S, hs, m);
}

// lane-wise m ? b : a as ONE v_cndmask.  Written in C++ (`c ? x[2i+1] : x[2i]`) hipcc turns the select between two
// neighbouring array elements into a load with a run-time index, i.e. the whole register array goes to scratch memory and
// every select becomes a scratch round trip (k_skm_hist spent 40 % of its wave time in s_waitcnt on those).
__device__ __forceinline__ uint32_t skm_sel(uint32_t a, uint32_t b, unsigned long long m) {
    uint32_t r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(m));
    return r;
}
// S.mh[s] for a per-lane s: binary select tree over the register array
template <int K>
__device__ __forceinline__ uint32_t skm_mh_at(const skm_word<K> &S, uint32_t s) {
    uint32_t a[16], b[8], c[4], d[2];
    const unsigned long long s0 = __ballot((s & 1u) != 0), s1 = __ballot((s & 2u) != 0), s2 = __ballot((s & 4u) != 0),
                             s3 = __ballot((s & 8u) != 0), s4 = __ballot((s & 16u) != 0);
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = skm_sel(S.mh[2 * i], S.mh[2 * i + 1], s0);
#pragma unroll
    for (int i = 0; i < 8; i++) b[i] = skm_sel(a[2 * i], a[2 * i + 1], s1);
#pragma unroll
    for (int i = 0; i < 4; i++) c[i] = skm_sel(b[2 * i], b[2 * i + 1], s2);
#pragma unroll
    for (int i = 0; i < 2; i++) d[i] = skm_sel(c[2 * i], c[2 * i + 1], s3);
    return skm_sel(d[0], d[1], s4);
}

// takes the next run off `cut`: start s, number of k-mers len (<= RMAX; a longer run leaves its rest in `cut`)
template <int K>
__device__ __forceinline__ void skm_next_run(const skm_word<K> &S, uint32_t &cut, uint32_t &s, uint32_t &len) {
    s = (uint32_t)__builtin_ctz(cut);
    cut &= cut - 1u;
    const uint32_t stop = (cut | ~S.valid) & ~((2u << s) - 1u);
    const uint32_t e = stop ? (uint32_t)__builtin_ctz(stop) : 32u;
    len = e - s;
    if (len > (uint32_t)skm_word<K>::RMAX) { len = (uint32_t)skm_word<K>::RMAX; cut |= 1u << (s + len); }
}

// the record of the run [s, s+len): its len+K-1 bases left-aligned, the remaining bits zero, digit bits, k-mer count
template <int K>
__device__ __forceinline__ skm_rec skm_make_rec(const uint32_t (&D)[4], uint32_t s, uint32_t len, uint32_t digits, uint32_t D4 = 0u) {
    // D4: bases 64 .. 79 from the word's start (a run that goes on into the next word, k_skm_scatter)
    const uint32_t o = (2u * s) & 31u;
    const unsigned long long q = __ballot(s >= 16u);
    // (a run from s < 16 may need base 64, the first of D4; one from s >= 16 ends by base 79: the caller bounds its length)
    const uint32_t E0 = skm_sel(D[0], D[1], q), E1 = skm_sel(D[1], D[2], q), E2 = skm_sel(D[2], D[3], q), E3 = skm_sel(D[3], D4, q), E4 = skm_sel(D4, 0u, q);
    uint32_t T0 = E0, T1 = E1, T2 = E2, T3 = E3;
    if (o) {
        T0 = __builtin_amdgcn_alignbit(E0, E1, 32u - o);
        T1 = __builtin_amdgcn_alignbit(E1, E2, 32u - o);
        T2 = __builtin_amdgcn_alignbit(E2, E3, 32u - o);
        T3 = __builtin_amdgcn_alignbit(E3, E4, 32u - o);
    }
    const int kb = 2 * (int)(len + K - 1);                 // bits to keep (<= 100)
    auto keep = [](uint32_t v, int bits) { return bits >= 32 ? v : (bits <= 0 ? 0u : (v & ~(0xFFFFFFFFu >> bits))); };
    T0 = keep(T0, kb); T1 = keep(T1, kb - 32); T2 = keep(T2, kb - 64); T3 = keep(T3, kb - 96);
    skm_rec r;
    r.x = ((uint64_t)T0 << 32) | T1;
    r.y = ((uint64_t)T2 << 32) | (uint64_t)(T3 & 0xF0000000u) | ((uint64_t)digits << 6) | (uint64_t)len;
    return r;
}
// level-1 digit and the digit bits that travel in the record, from the run's minimizer hash
__device__ __forceinline__ void skm_route(uint32_t mh, int bits1, uint32_t &d1, uint32_t &digits) {
    const uint32_t ph = mf_remix32(mh);
    d1 = bits1 ? ph >> (32 - bits1) : 0u;
    digits = (bits1 ? (ph << bits1) : ph) >> (32 - SKM_DIGIT_BITS);
}

// =============================================================================================
// S1: per-block histogram of the records' level-1 digit (and of their k-mers, for plans without a split level)
// =============================================================================================
template <int K>
__global__ __launch_bounds__(1024) void k_skm_hist(const uint8_t *__restrict__ bases, uint64_t n_bases, const uint32_t *__restrict__ vmask,
                                                   uint64_t n_words, uint64_t words_per_block, int bits1,
                                                   uint32_t *__restrict__ blockhist, uint32_t *__restrict__ blockocc, int G, int stride,
                                                   uint32_t dlo, uint32_t dhi) {
    // [dlo, dhi): the level-1 digits of this SLICE of the run (skm_run); records of other digits belong to other slices
    // stride > 1: a SAMPLE (every stride-th tile of 1024 words), used to size the digit regions of the one-pass scatter
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nd = 1 << bits1;
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem), *occ = hist + nd;
    for (int i = threadIdx.x; i < 2 * nd; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const uint64_t wlo = (uint64_t)blockIdx.x * words_per_block;
    const uint64_t whi = wlo + words_per_block < n_words ? wlo + words_per_block : n_words;
    for (uint64_t wb = wlo; wb < whi; wb += (uint64_t)blockDim.x * (uint64_t)stride) {
        const uint64_t w = wb + threadIdx.x;
        const uint32_t m = w < whi ? vmask[w] : 0u;
        if (__ballot(m != 0u) == 0ull) continue;
        skm_word<K> S;
        skm_scan_word<K>(S, bases, n_bases, w < n_words ? w : 0, m);
        uint32_t cut = S.cut;
        while (__ballot(cut != 0u) != 0ull) {
            if (cut) {
                uint32_t s, len, d1, digits;
                skm_next_run<K>(S, cut, s, len);
                skm_route(skm_mh_at<K>(S, s), bits1, d1, digits);
                if (d1 >= dlo && d1 < dhi) {
                    atomicAdd(&hist[d1], 1u);
                    if (blockocc) atomicAdd(&occ[d1], len);
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nd; i += blockDim.x) {
        blockhist[(size_t)i * G + blockIdx.x] = hist[i];
        if (blockocc) blockocc[(size_t)i * G + blockIdx.x] = occ[i];
    }
}

// =============================================================================================
// LDS staging of 16-byte records: one 64-byte line (4 records) per digit; same protocol as mf_stage (mf_count.hip):
// reserve a slot, write it, commit; the 4th committer's line is written to HBM by four lanes with one store instruction
// =============================================================================================
struct skm_stage {
    skm_rec *line;     // [nd][4] + 64 dummy slots (one per lane)
    uint64_t *cur;     // [nd] next record index of this workgroup's range of digit d
    uint64_t *q_pos;   // [16 waves][SKM_QCAP]
    uint32_t *ctr;     // [nd] low 16 = reserved, high 16 = committed; + 64 dummy counters
    uint32_t *q_d;     // [16 waves][SKM_QCAP]
    int nd;
};
__device__ __forceinline__ skm_stage skm_stage_carve(unsigned char *smem, int nd) {
    skm_stage L;
    L.nd = nd;
    L.line = reinterpret_cast<skm_rec *>(smem);
    L.cur = reinterpret_cast<uint64_t *>(L.line + (size_t)nd * SKM_LINE + 64);
    L.q_pos = L.cur + nd;
    L.ctr = reinterpret_cast<uint32_t *>(L.q_pos + 16 * SKM_QCAP);
    L.q_d = L.ctr + nd + 64;
    return L;
}
__host__ __device__ static inline size_t skm_stage_bytes(int nd) {
    return ((size_t)nd * SKM_LINE + 64) * 16 + (size_t)nd * 8 + 16 * SKM_QCAP * 8 + ((size_t)nd + 64) * 4 + 16 * SKM_QCAP * 4;
}
// four 16-byte stores, then four returning adds (the commits); LDS operations of one wave execute in order
__device__ __forceinline__ void skm_lds_write4_add_rtn4(const uint32_t (&wa)[4], const skm_v4 (&wv)[4], const uint32_t (&a)[4],
                                                        const uint32_t (&inc)[4], uint32_t (&old)[4]) {
    asm volatile("ds_write_b128 %4, %8\n\tds_write_b128 %5, %9\n\tds_write_b128 %6, %10\n\tds_write_b128 %7, %11\n\t"
                 "ds_add_rtn_u32 %0, %12, %16\n\tds_add_rtn_u32 %1, %13, %17\n\tds_add_rtn_u32 %2, %14, %18\n\t"
                 "ds_add_rtn_u32 %3, %15, %19\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(old[0]), "=&v"(old[1]), "=&v"(old[2]), "=&v"(old[3])
                 : "v"(wa[0]), "v"(wa[1]), "v"(wa[2]), "v"(wa[3]), "v"(wv[0]), "v"(wv[1]), "v"(wv[2]), "v"(wv[3]),
                   "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(inc[0]), "v"(inc[1]), "v"(inc[2]), "v"(inc[3])
                 : "memory");
}
// DYN (one-pass level-1 scatter): a workgroup does not know its share of a digit in advance; it reserves the digit's region
// chunk by chunk (SKM_CH records) with a global atomic when its current chunk is used up.  Chunk starts are multiples of
// SKM_CH and a position is advanced right after it is taken, so "position % SKM_CH == 0" means "no room left".
#define SKM_CH 256
#define SKM_NONE (~0ull)
struct skm_dyn {
    unsigned long long *gcur;     // [nd] next free record of every digit's region
    const uint64_t *rend;         // [nd] end of the region
    unsigned int *overflow;       // set when a region was too small (the caller repeats the level with exact ranges)
    uint64_t dump;                // a scratch chunk of this workgroup: where lines go after an overflow
    unsigned long long *save;     // [G][nd] or null.  STREAMED level 1 (mf_stream.hip: the reads arrive piece by piece, a launch per piece): a workgroup's
                                  // position inside its current chunk of every digit outlives the launch, so that a piece leaves a partly filled LINE per
                                  // workgroup and digit behind, not a partly filled chunk (30 pieces x 256 workgroups x 1024 digits x 256 records = 30 GB of padding)
};
template <bool DYN>
__device__ __forceinline__ uint64_t skm_take_line(const skm_stage &L, uint32_t dg, const skm_dyn &Dy) {
    uint64_t pos = L.cur[dg];
    if (DYN && (pos == SKM_NONE || (pos & (uint64_t)(SKM_CH - 1)) == 0ull)) {
        pos = atomicAdd(&Dy.gcur[dg], (unsigned long long)SKM_CH);
        if (pos + SKM_CH > Dy.rend[dg]) { atomicExch(Dy.overflow, 1u); pos = Dy.dump; }
    }
    L.cur[dg] = pos + SKM_LINE;
    return pos;
}
// ALL 64 lanes of the wave must call this together (wave-uniform loop, wave-cooperative flush)
template <bool DYN = false>
__device__ __forceinline__ void skm_stage_insert(const skm_stage &L, skm_rec *__restrict__ out, const uint32_t (&d)[4],
                                                 const skm_rec (&rec)[4], bool (&pending)[4], const skm_dyn &Dy = skm_dyn()) {
    const uint32_t ctr0 = mf_lds_addr(L.ctr), line0 = mf_lds_addr(L.line);
    const uint32_t dummy_ctr = ctr0 + 4u * ((uint32_t)L.nd + (uint32_t)mf_lane());
    const uint32_t dummy_slot = line0 + 16u * ((uint32_t)L.nd * SKM_LINE + (uint32_t)mf_lane());
    skm_v4 wv[4];
#pragma unroll
    for (int b = 0; b < 4; b++) { wv[b].x = (uint32_t)rec[b].x; wv[b].y = (uint32_t)(rec[b].x >> 32); wv[b].z = (uint32_t)rec[b].y; wv[b].w = (uint32_t)(rec[b].y >> 32); }
    for (;;) {
        bool any = false;
#pragma unroll
        for (int b = 0; b < 4; b++) any |= pending[b];
        if (__ballot(any) == 0ull) break;
        uint32_t ca[4], w[4], inc[4], old[4], old2[4], wa[4];
        bool got[4];
#pragma unroll
        for (int b = 0; b < 4; b++) ca[b] = pending[b] ? ctr0 + 4u * d[b] : dummy_ctr;
        mf_lds_read4(ca, w);                                                     // peek: is the line open?
#pragma unroll
        for (int b = 0; b < 4; b++) inc[b] = (pending[b] && (w[b] & 0xFFFFu) < (uint32_t)SKM_LINE) ? 1u : 0u;
        mf_lds_add_rtn4(ca, inc, old);                                           // reserve a slot
#pragma unroll
        for (int b = 0; b < 4; b++) {
            got[b] = inc[b] && (old[b] & 0xFFFFu) < (uint32_t)SKM_LINE;
            wa[b] = got[b] ? line0 + 16u * (d[b] * SKM_LINE + (old[b] & 0xFFFFu)) : dummy_slot;
            ca[b] = got[b] ? ctr0 + 4u * d[b] : dummy_ctr;
            inc[b] = got[b] ? 0x10000u : 0u;
        }
        skm_lds_write4_add_rtn4(wa, wv, ca, inc, old2);                          // write the slot, commit
        uint64_t *qpos = L.q_pos + (threadIdx.x >> 6) * SKM_QCAP;
        uint32_t *qd = L.q_d + (threadIdx.x >> 6) * SKM_QCAP;
        const uint64_t lt_mask = (1ull << mf_lane()) - 1ull;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const bool fl = got[b] && (old2[b] >> 16) == (uint32_t)(SKM_LINE - 1);   // last committer of its line
            const unsigned long long F = __ballot(fl);
            if (got[b]) pending[b] = false;
            if (F == 0ull) continue;
            const uint32_t n = (uint32_t)__popcll(F), qi = (uint32_t)__popcll(F & lt_mask);
            uint64_t mypos = 0;
            if (fl) mypos = skm_take_line<DYN>(L, d[b], Dy);
            for (uint32_t e0 = 0; e0 < n; e0 += SKM_QCAP) {
                if (fl && qi >= e0 && qi < e0 + SKM_QCAP) { qpos[qi - e0] = mypos; qd[qi - e0] = d[b]; }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // queue visible to the wave (in-order LDS)
                const uint32_t e = e0 + ((uint32_t)mf_lane() >> 2), c = (uint32_t)mf_lane() & 3u;
                if (e < n) {
                    const uint32_t dd = qd[e - e0];
                    const uint64_t pp = qpos[e - e0];
                    const skm_rec v = L.line[dd * SKM_LINE + c];
                    out[pp + c] = v;
                    asm volatile("" ::: "memory");
                    if (c == 0) __hip_atomic_store(&L.ctr[dd], 0u, __ATOMIC_RELEASE, MF_WG);   // reopen (after the reads above)
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // before the queue is reused
            }
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
}
// after a barrier: write every partly filled line, padded with sentinel records (DYN: and the unused rest of the
// workgroup's last chunk of every digit, the reader of the region skips sentinels)
template <bool DYN = false>
__device__ __forceinline__ void skm_stage_flush_all(const skm_stage &L, skm_rec *__restrict__ out, int nd, const skm_dyn &Dy = skm_dyn()) {
    const skm_rec SENT = make_ulonglong2(~0ull, ~0ull);
    for (int d = threadIdx.x; d < nd; d += blockDim.x) {
        const uint32_t c = L.ctr[d] >> 16;
        if (c) {
            const uint64_t pos = skm_take_line<DYN>(L, (uint32_t)d, Dy);
            for (uint32_t s = 0; s < (uint32_t)SKM_LINE; s++) out[pos + s] = s < c ? L.line[d * SKM_LINE + s] : SENT;
            L.ctr[d] = 0;
        }
        if (DYN) {
            uint64_t pos = L.cur[d];
            if (Dy.save) Dy.save[(size_t)blockIdx.x * nd + d] = pos;         // (k_skm_close_chunks pads them after the last piece)
            else if (pos != SKM_NONE) for (; (pos & (uint64_t)(SKM_CH - 1)) != 0ull; pos++) out[pos] = SENT;
        }
    }
}
// after the last piece of a streamed level 1: the unused rest of every workgroup's current chunk becomes sentinels (positions in the dump
// area -- after an overflow: the level is thrown away -- are left alone)
__global__ void k_skm_close_chunks(const unsigned long long *__restrict__ save, uint64_t n, uint64_t cap, skm_rec *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t pos = save[i];
    if (pos == SKM_NONE || pos >= cap) return;
    const skm_rec SENT = make_ulonglong2(~0ull, ~0ull);
    for (; (pos & (uint64_t)(SKM_CH - 1)) != 0ull; pos++) out[pos] = SENT;
}

// =============================================================================================
// S2: records of every run, radix-partitioned by the level-1 digit
// =============================================================================================
// FAST (level 1 of <= 10 bits: the staging lines leave LDS for it): the runs of a wave's 64 words are dealt EVENLY over its
// lanes through LDS.  A word has 3.7 runs on average but the fullest of 64 has 8 or 9, so four runs per lane and iteration
// meant a second, nearly empty iteration for the whole wave (r02: ~580 of the kernel's 2120 instructions per word), and every
// run slot paid a 41-instruction select tree for the run's minimizer hash (a register array cannot be indexed per lane).  Now:
// every lane parks its word (bases, run starts, valid starts: 24 bytes) and writes one 8-byte descriptor per run -- the loop
// is over the 32 POSITIONS, so the hash is a fixed register, and lanes without a run at that position write to a dummy
// slot --; then lane l takes the runs l, l + 64, l + 128, l + 192 of the list: one four-wide iteration per 256 runs (237 on
// average per batch).  Rare batches (a run longer than RMAX that must be cut, more runs than the list holds) take the per-lane loop.
// RUNS GO ON ACROSS WORD BOUNDARIES here: a word used to start a new run whatever its first k-mer's minimizer -- a quarter of
// all records began at a word boundary.  Now a lane whose first k-mer continues the previous lane's last run (same minimizer
// hash, both valid: the same read) hands the head of its word over to that run, as far as the run's RMAX allows (`ext`
// k-mers; the rest of the head becomes a run of its own), and the lane that owns the run builds the record from its own 64
// bases + the next lane's (parked) -- 1.75e9 -> ~1.35e9 records at 100 M reads: less to write, to split and to unpack.
#define SKM_LCAP 384               // run descriptors per wave and batch
#define SKM_FAST_WAVE_BYTES (64 * 16 + 64 * 8 + (SKM_LCAP + 1) * 8)
template <int K, bool DYN, bool FAST>
__global__ __launch_bounds__(1024) void k_skm_scatter(const uint8_t *__restrict__ bases, uint64_t n_bases, const uint32_t *__restrict__ vmask,
                                                      uint64_t n_words, uint64_t words_per_block, int bits1,
                                                      const uint64_t *__restrict__ blockstart, int G, skm_rec *__restrict__ out, skm_dyn Dy,
                                                      uint32_t dlo, uint32_t dhi) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nd = 1 << bits1;
    skm_stage L = skm_stage_carve(smem, nd);
    if (DYN) Dy.dump += (uint64_t)blockIdx.x * SKM_CH;
    for (int i = threadIdx.x; i < nd; i += blockDim.x) {
        L.cur[i] = DYN ? (Dy.save ? (uint64_t)Dy.save[(size_t)blockIdx.x * nd + i] : SKM_NONE) : blockstart[(size_t)i * G + blockIdx.x];
        L.ctr[i] = 0;
    }
    if (threadIdx.x < 64) L.ctr[nd + threadIdx.x] = 0;
    __syncthreads();
    // FAST: this wave's parking area and run list, behind the staging area
    const uint32_t lane = (uint32_t)mf_lane();
    unsigned char *wbase = smem + ((skm_stage_bytes(nd) + 15) & ~(size_t)15) + (size_t)(threadIdx.x >> 6) * SKM_FAST_WAVE_BYTES;
    uint4 *pd = reinterpret_cast<uint4 *>(wbase);                               // [64] the words' bases
    uint2 *pc = reinterpret_cast<uint2 *>(wbase + 64 * 16);                     // [64] (positions where a run stops, k-mers of the word's head that belong to the previous lane's run)
    uint2 *rl = reinterpret_cast<uint2 *>(wbase + 64 * 16 + 64 * 8);            // [SKM_LCAP + 1] (minimizer hash, lane << 5 | position); the last one: dummy
    const uint64_t wlo = (uint64_t)blockIdx.x * words_per_block;
    const uint64_t whi = wlo + words_per_block < n_words ? wlo + words_per_block : n_words;
    // a wave takes 63 words per batch: lane 63 only provides lane 62's halo (skm_scan_word_halo), the word it scans is lane 0's
    // of the wave's next batch
    const uint64_t wstep = (uint64_t)(blockDim.x >> 6) * 63u;
    for (uint64_t wb = wlo; wb < whi; wb += wstep) {          // wave-uniform: all lanes reach skm_stage_insert together
        const uint64_t w = wb + (uint64_t)(threadIdx.x >> 6) * 63u + lane;
        const uint32_t m = (lane < 63u && w < whi) ? vmask[w] : 0u;
        if (__ballot(m != 0u) == 0ull) continue;
        skm_word<K> S;
        skm_scan_word_halo<K>(S, bases, n_bases, w < n_words ? w : 0, m);
        uint32_t cut = S.cut;
        bool fast = FAST;
        uint32_t NR = 0, roff = 0, cutf = cut, ext = 0;
        if (FAST) {
            // a run of more than RMAX k-mers (RMAX continuation positions in a row) has to be cut: the per-lane loop does that
            constexpr int R = skm_word<K>::RMAX;
            static_assert(R >= 16 && R <= 20, "run-length check");
            const uint32_t z = ~(cut | ~S.valid);
            const uint32_t z2 = z & (z >> 1), z4 = z2 & (z2 >> 2), z8 = z4 & (z4 >> 4), z16 = z8 & (z8 >> 8);
            const uint32_t zr = R == 16 ? z16 : (R == 17 ? (z16 & (z >> 16)) : (R == 18 ? (z16 & (z2 >> 16)) : (R == 19 ? (z16 & (z2 >> 16) & (z >> 18)) : (z16 & (z4 >> 16)))));
            // the head of this word joins the previous lane's last run?  (zr == 0 everywhere, or the batch takes the other path: then
            // the previous lane's last run starts at its highest cut and is at most R k-mers long inside its word)
            const uint32_t pcut = skm_from_prev_lane(cut), pvalid = skm_from_prev_lane(S.valid), pm31 = skm_from_prev_lane(S.mh[31]);
            const bool cont = (S.valid & 1u) && (pvalid >> 31) && S.mh[0] == pm31;
            const uint32_t hstop = (cut & ~1u) | ~S.valid;
            const uint32_t h = hstop ? (uint32_t)__builtin_ctz(hstop) : 32u;
            const uint32_t tail = pcut ? (uint32_t)__builtin_clz(pcut) + 1u : 32u;           // k-mers of that run in the previous word
            // (the owner builds the record from bases 0 .. 79 of its word: the last base of the run, 32 + ext + K - 2, must be one of them)
            constexpr uint32_t XMAX = 49 - K < R ? (uint32_t)(49 - K) : (uint32_t)R;
            uint32_t allowed = tail < (uint32_t)R ? (uint32_t)R - tail : 0u;
            if (allowed > XMAX) allowed = XMAX;
            // (only in the one-pass form: with exact output ranges -- small inputs, the retry after an overflow -- the histogram
            // pass has counted a run per word start, and the ranges must be filled exactly)
            ext = (DYN && cont) ? (h < allowed ? h : allowed) : 0u;
            cutf = ext ? ((cut & ~1u) | (ext < h ? (1u << ext) : 0u)) : cut;
            roff = mf_wave_excl_scan((uint32_t)__popc(cutf), &NR);
            fast = __ballot(zr != 0u) == 0ull && NR <= (uint32_t)SKM_LCAP;
        }
        if (FAST && fast) {
            pd[lane] = make_uint4(S.D[0], S.D[1], S.D[2], S.D[3]);
            pc[lane] = make_uint2(cutf | ~S.valid, ext);                 // (positions where a run stops, k-mers handed to the previous lane's run)
            {
                const uint32_t l0 = mf_lds_addr(rl), dummy = l0 + 8u * (uint32_t)SKM_LCAP;
                uint32_t at = l0 + 8u * roff;
                const uint32_t tagl = lane << 5;
#pragma unroll
                for (int j = 0; j < 32; j++) {
                    const uint32_t bit = (cutf >> j) & 1u;
                    const uint32_t addr = bit ? at : dummy;
                    const uint32_t v1 = tagl | (uint32_t)j;
                    asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" :: "v"(addr), "v"(S.mh[j]), "v"(v1) : "memory");
                    at += bit << 3;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            for (uint32_t g0 = 0; g0 < NR; g0 += 256) {                     // wave-uniform
                uint32_t d[4]; skm_rec rec[4]; bool pend[4];
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const uint32_t g = g0 + 64u * (uint32_t)b + lane;
                    pend[b] = g < NR;
                    const uint2 de = rl[pend[b] ? g : 0u];
                    const uint32_t src = (de.y >> 5) & 63u, s0 = de.y & 31u;
                    const uint4 Dw = pd[src]; const uint2 cv = pc[src];
                    const uint4 Dn = pd[(src + 1u) & 63u]; const uint2 cn = pc[(src + 1u) & 63u];        // (a run's lane is <= 62)
                    const uint32_t stop = cv.x & ~((2u << s0) - 1u);
                    const uint32_t e = stop ? (uint32_t)__builtin_ctz(stop) : 32u;
                    const uint32_t len = e - s0 + (e == 32u ? cn.y : 0u);                                // + the head of the next word
                    const uint32_t DD[4] = {Dw.x, Dw.y, Dw.z, Dw.w};
                    uint32_t digits;
                    skm_route(de.x, bits1, d[b], digits);
                    rec[b] = skm_make_rec<K>(DD, s0, len, digits, Dn.z);
                    if (!pend[b] || d[b] < dlo || d[b] >= dhi) { pend[b] = false; d[b] = 0; rec[b] = make_ulonglong2(~0ull, ~0ull); }
                }
                skm_stage_insert<DYN>(L, out, d, rec, pend, Dy);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // (the parked words are rewritten by the next batch)
            __builtin_amdgcn_wave_barrier();
            continue;
        }
        while (__ballot(cut != 0u) != 0ull) {
            uint32_t d[4]; skm_rec rec[4]; bool pend[4];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                pend[b] = cut != 0u;
                d[b] = 0; rec[b] = make_ulonglong2(~0ull, ~0ull);
                if (pend[b]) {
                    uint32_t s, len, digits;
                    skm_next_run<K>(S, cut, s, len);
                    skm_route(skm_mh_at<K>(S, s), bits1, d[b], digits);
                    rec[b] = skm_make_rec<K>(S.D, s, len, digits);
                    if (d[b] < dlo || d[b] >= dhi) { pend[b] = false; d[b] = 0; }      // (another slice's record)
                }
            }
            skm_stage_insert<DYN>(L, out, d, rec, pend, Dy);
        }
    }
    __syncthreads();
    skm_stage_flush_all<DYN>(L, out, nd, Dy);
}

// one-pass level 1: region size of every digit from a sampled histogram, and the directory afterwards
__global__ void k_skm_region_sizes(const uint32_t *__restrict__ blockhist, int G, int nd, uint32_t scale, uint32_t *__restrict__ rsize,
                                   uint32_t dlo, uint32_t dhi) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= nd) return;
    if ((uint32_t)d < dlo || (uint32_t)d >= dhi) { rsize[d] = 0; return; }
    uint64_t t = 0;
    for (int b = 0; b < G; b++) t += blockhist[(size_t)d * G + b];
    t *= scale;
    t += t / 16 + (uint64_t)G * SKM_CH + 8192;          // sampling error + a partly used chunk per workgroup
    t = (t + SKM_CH - 1) & ~(uint64_t)(SKM_CH - 1);
    rsize[d] = t > 0xFFFFFF00ull ? 0xFFFFFF00u : (uint32_t)t;
}
__global__ void k_skm_dir_dyn(const uint64_t *__restrict__ rstart, const unsigned long long *__restrict__ gcur, int nd,
                              uint64_t *__restrict__ pstart, uint32_t *__restrict__ plen) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= nd) return;
    pstart[d] = rstart[d];
    plen[d] = (uint32_t)(gcur[d] - rstart[d]);          // whole chunks; unused records are sentinels
}

// partition directory after level 1: start / padded length per digit; k-mers per digit (plans without a split level)
__global__ void k_skm_dir(const uint64_t *__restrict__ blockstart, const uint32_t *__restrict__ blockocc, int G, int nd,
                          uint64_t *__restrict__ pstart, uint32_t *__restrict__ plen, uint32_t *__restrict__ pocc) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= nd) return;
    const uint64_t s = blockstart[(size_t)d * G], e = blockstart[(size_t)(d + 1) * G];
    pstart[d] = s;
    plen[d] = (uint32_t)(e - s);
    if (blockocc) {
        uint64_t t = 0;
        for (int b = 0; b < G; b++) t += blockocc[(size_t)d * G + b];
        pocc[d] = t > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)t;
    }
}

// =============================================================================================
// S3: split every partition into 2^bits sub-partitions by the next digit bits stored in the records
// =============================================================================================
__global__ __launch_bounds__(1024) void k_skm_split(const skm_rec *__restrict__ in, const uint64_t *__restrict__ pstart,
                                                    const uint32_t *__restrict__ plen, uint32_t np, int shift, int bits,
                                                    skm_rec *__restrict__ out, uint64_t *__restrict__ ostart, uint32_t *__restrict__ olen,
                                                    uint32_t *__restrict__ oocc, unsigned long long *__restrict__ n_valid, uint64_t rebase) {
    // rebase: `in` is a buffer shared by several slices (skm_shared) and `out` belongs to this slice: its regions start at 0
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ uint32_t scratch[17];
    const int nd = 1 << bits;
    const uint32_t dmask = (uint32_t)nd - 1u;
    skm_stage L = skm_stage_carve(smem, nd);
    uint32_t *occ = reinterpret_cast<uint32_t *>(L.line);       // k-mers per digit: lives in the (idle) staging lines during the histogram
    const int ipt = (nd + (int)blockDim.x - 1) / (int)blockDim.x;
    for (uint32_t p = blockIdx.x; p < np; p += gridDim.x) {
        const uint64_t start = pstart[p];
        const uint32_t len = plen[p];
        const uint64_t obase = start - rebase + (uint64_t)p * (uint64_t)(SKM_LINE * nd);   // room for per-digit padding
        for (int i = threadIdx.x; i < nd; i += blockDim.x) { L.ctr[i] = 0; occ[i] = 0; }
        __syncthreads();
        const skm_rec *src = in + start;
        for (uint32_t jb = 0; jb < len; jb += 4 * blockDim.x) {
            skm_rec v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const uint32_t j = jb + u * blockDim.x + threadIdx.x; v[u] = j < len ? src[j] : make_ulonglong2(~0ull, ~0ull); }
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (skm_rec_valid(v[u])) {
                    const uint32_t dg = (skm_rec_digits(v[u]) >> shift) & dmask;
                    atomicAdd(&L.ctr[dg], 1u);
                    if (oocc) atomicAdd(&occ[dg], skm_rec_n(v[u]));
                }
        }
        __syncthreads();
        uint32_t mine = 0, raw = 0;
        const int b0 = threadIdx.x * ipt;
        for (int j = 0; j < ipt; j++) { const int b = b0 + j; if (b < nd) { mine += (L.ctr[b] + 3u) & ~3u; raw += L.ctr[b]; } }
        if (n_valid) {                                            // records without the padding (measurement: mf_table_records)
            for (int dd = 32; dd >= 1; dd >>= 1) raw += __shfl_down(raw, dd, 64);
            if (mf_lane() == 0 && raw) atomicAdd(n_valid, (unsigned long long)raw);
        }
        uint32_t tot;
        uint32_t ex = mf_block_excl_scan(mine, scratch, &tot);
        for (int j = 0; j < ipt; j++) {
            const int b = b0 + j;
            if (b < nd) {
                const uint32_t c = (L.ctr[b] + 3u) & ~3u;
                L.cur[b] = obase + ex;
                ostart[(size_t)p * nd + b] = obase + ex;
                olen[(size_t)p * nd + b] = c;
                if (oocc) oocc[(size_t)p * nd + b] = occ[b];
                ex += c;
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nd; i += blockDim.x) L.ctr[i] = 0;
        if (threadIdx.x < 64) L.ctr[nd + threadIdx.x] = 0;
        __syncthreads();
        // wave-uniform loop bounds: every lane reaches skm_stage_insert
        for (uint32_t jb = 0; jb < len; jb += 4 * blockDim.x) {
            skm_rec v[4]; uint32_t d[4]; bool pend[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t j = jb + u * blockDim.x + threadIdx.x;
                v[u] = j < len ? src[j] : make_ulonglong2(~0ull, ~0ull);
                pend[u] = skm_rec_valid(v[u]);
                d[u] = pend[u] ? ((skm_rec_digits(v[u]) >> shift) & dmask) : 0u;
            }
            skm_stage_insert(L, out, d, v, pend);
        }
        __syncthreads();
        skm_stage_flush_all(L, out, nd);
        __syncthreads();
    }
}

#define SKM_MAX_PASSES 64
// pass of a key when its partition is counted in several passes (bits independent of the slot hash's)
__device__ __forceinline__ uint32_t skm_pass_of(uint64_t key) {
    uint32_t g = ((uint32_t)(key >> 32) * 0xC2B2AE35u) ^ ((uint32_t)key * 0x27D4EB2Fu);
    g ^= g >> 15; g *= 0x165667B1u;
    return g >> 24;
}

// =============================================================================================
// S4: count one partition per workgroup.  Second generation: the first one (round 1, profiles/r01j_pmc_100M: 160 VALU lane-
// operations per k-mer occurrence, 76 ms at 100 M reads) probed four-wide with a separate read, CAS and claim-list scan per
// step and expanded records with 64-bit shifts.  What this one is built on (ablations: profiles/r02_count_ablation_*.txt):
//   * all arithmetic on 32-bit halves (v_alignbit funnel shifts, v_bfrev): the 64-bit variable shifts of the record
//     expansion and of mf_revcomp are multi-instruction sequences on CDNA;
//   * the probe IS the claim: one ds_cmpst_rtn_b64(slot, EMPTY, key) per key, four keys per lane.  It returns EMPTY (the
//     slot is ours now), the key (a hit) or another key (a collision, 5 %: those go to the wave's queue and are finished 64
//     at a time, one key per lane).  A separate read first (hits only, first sightings queued) looked cheaper on paper
//     -- a CAS costs its LDS cycles whether 8 lanes or 64 need it -- but all eight waves meet a partition's k-mers at the
//     same time: with the claims deferred, every copy of a new k-mer missed and went through the slow path (27 of 59 ms);
//   * lanes without work are switched off with EXEC inside the asm blocks, not parked on dummy slots (no address selects);
//   * the records are parked in LDS shifted right by ONE bit: the k-mer at base offset t then starts at the odd bit
//     2t+1, so the funnel-shift amount 32 - ((2t+1) & 31) is never 32 and the extraction needs no special case;
//   * the step is software-pipelined over LDS: the next step's item is fetched beside the probes and its record beside the
//     count updates, so one LDS round trip per step is exposed instead of three.
// Items carry everything the consumer lane needs (record slot, funnel-shift amount, word select, k-mers in the item) in
// 16 bits.  Two workgroup barriers per partition (flags are double-buffered by partition parity).
// The all-counts histogram of IOUtils.printKmers (src/io/IOUtils.java:45-71, the .stat.txt file) needs the entries the
// cut drops: their counts (<= thr) are tallied here (count 1 by ballot, the others in a small LDS histogram).
// =============================================================================================
#ifdef SKM_BIG_UNITS
#define C2_SLOTS 8192
#define C2_FILL 6800
#define C2_WG_PER_CU 1
#elif defined(SKM_SMALL_UNITS)
#define C2_SLOTS 2048
#define C2_FILL 1700
#define C2_WG_PER_CU 3
#else
#define C2_WG_PER_CU 2
#define C2_SLOTS 4096            // table slots: at the planned ~800 distinct k-mers per partition the table is 1/5 full.  (2048 slots
                                 // = three workgroups per CU instead of two, but an 80-VGPR budget: 54 ms against 45.)
#define C2_FILL 3400             // claims beyond which a partition is counted again in several passes
#endif
#define C2_QN 128                // queue entries per wave (keys): drained at 64, one push (<= 64 keys) between checks
#define C2_LH 64                 // bins of the workgroup's dropped-count histogram (a cut with thr >= C2_LH is made after the kernel)
#define C2_ITEMS 848             // item entries per wave: 64 records x 10 items + two steps of padding (the read-ahead of the last step runs past them, unused) + the dummy area of the item stores
#define C2_W 2                   // k-mers per item
#define DD_BITS (SKM_CT > 512 ? 12 : 11)   // bits of a record's index in its unit (the search for identical records: C2_DD * SKM_CT records)
#define C2_DD 4                  // records per thread of a unit whose identical records are counted once (k_skm_count): units of up to C2_DD * SKM_CT records
static constexpr size_t C2_LDS = (size_t)C2_SLOTS * 12 + (size_t)SKM_CT * 16 + (size_t)(SKM_CT / 64) * (C2_ITEMS * 2 + C2_QN * 8 + C2_QN * 2) + 2 * C2_LH * 4 + 48;
static_assert(C2_LDS <= 160 * 1024 / C2_WG_PER_CU, "workgroups per CU");

__device__ __forceinline__ uint32_t c2_swap_pairs(uint32_t x) {                 // exchanges the two bits of every base
    return ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
}
// slot of a key: full-rate instructions only (v_mul_lo_u32 is a quarter-rate instruction, and this is evaluated per k-mer
// occurrence): fold the halves, fold the top bits down, one 24-bit multiply, bits 11..22 of the product
__device__ __forceinline__ uint32_t c2_slot(uint32_t hi, uint32_t lo) {
    uint32_t x = lo ^ __builtin_amdgcn_alignbit(hi, hi, 19);
    x ^= x >> 15;
    uint32_t t;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(t) : "v"(x), "s"(0x9E3779u));      // (low 24 bits of x) * constant, low 32 bits of the product
    return (t >> 11) & (uint32_t)(C2_SLOTS - 1);
}
struct c2_wave {                 // LDS byte addresses of this wave's private areas + the table
    uint32_t tk0, tc0, rb0, items0, qk0, qw0;
};
// ---- lane masks straight from the compare (a C++ bool that meets __ballot costs a v_cndmask + v_cmp round trip per use)
__device__ __forceinline__ unsigned long long c2_eq_u64(uint64_t a, uint64_t b) {
    unsigned long long m;
    asm("v_cmp_eq_u64_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
    return m;
}
__device__ __forceinline__ unsigned long long c2_lt_u32(uint32_t a, uint32_t b) {      // lanes with a < b
    unsigned long long m;
    asm("v_cmp_lt_u32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
    return m;
}
template <int U> __device__ __forceinline__ unsigned long long c2_gt(uint32_t a) {       // lanes with a > U (U: inline constant)
    unsigned long long m;
    asm("v_cmp_lt_u32_e64 %0, %2, %1" : "=s"(m) : "v"(a), "n"(U));
    return m;
}
// Workgroup barrier for LDS traffic only.  __syncthreads() is a workgroup-scope fence + s_barrier, and the fence makes hipcc
// wait for vmcnt(0) first: every wave would sit out the HBM latency of the NEXT partition's record prefetch at each barrier
// (the prefetch is issued a whole partition ahead precisely so that nobody waits for it).  The data the barriers order
// lives in LDS; the global stores of the compaction are read by later kernels only.
__device__ __forceinline__ void c2_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// a wave-uniform 64-bit value that lives in VGPRs -> SGPRs (readfirstlane returns int: no sign extension on the way back)
__device__ __forceinline__ uint64_t c2_uniform64(uint64_t v) {
    return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32);
}
// ---- LDS operations of the lanes of a mask (EXEC is switched inside the block and restored)
__device__ __forceinline__ void c2_push64(unsigned long long m, uint32_t addr, uint64_t v) {
    unsigned long long save;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\tds_write_b64 %2, %3\n\ts_mov_b64 exec, %0"
                 : "=&s"(save) : "s"(m), "v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void c2_push16(unsigned long long m, uint32_t addr, uint32_t v) {
    unsigned long long save;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\tds_write_b16 %2, %3\n\ts_mov_b64 exec, %0"
                 : "=&s"(save) : "s"(m), "v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ uint64_t c2_cas_m(unsigned long long m, uint32_t addr, uint64_t key) {      // waited for
    unsigned long long save; uint64_t ret;
    asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, %2\n\tds_cmpst_rtn_b64 %0, %3, %4, %5\n\ts_mov_b64 exec, %1\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(ret), "=&s"(save) : "s"(m), "v"(addr), "v"(MF_EMPTY), "v"(key) : "memory");
    return ret;
}
__device__ __forceinline__ void c2_add_m(unsigned long long m, uint32_t addr, uint32_t one) {
    unsigned long long save;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\tds_add_u32 %2, %3\n\ts_mov_b64 exec, %0"
                 : "=&s"(save) : "s"(m), "v"(addr), "v"(one) : "memory");
}
__device__ __forceinline__ uint32_t c2_lds_u32(const uint32_t *p) {                    // a volatile C++ read of LDS becomes a flat load
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(mf_lds_addr(p)) : "memory");
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
// finish up to 64 queued keys (collisions), one per lane, by linear probing from the slot AFTER the home slot.  Four slots
// are read ahead per round (reads are cheap and go out together): the key is found (-> count), or an empty slot comes first
// (-> CAS there; if another key got in first, on from the next slot), or neither (-> four slots further).  Most keys need
// one round; one slot per round cost four to six dependent LDS round trips per drain.
__device__ __forceinline__ void c2_drain(const c2_wave &L, uint32_t &qn, uint32_t &won_acc, uint32_t *part_over) {
    const uint32_t lane = (uint32_t)mf_lane();
    constexpr uint32_t mask = (uint32_t)(C2_SLOTS - 1);
    const uint32_t c = qn < 64u ? qn : 64u;
    qn -= c;
    unsigned long long pend = c2_lt_u32(lane, c);
    uint64_t key; uint32_t one;                                                  // (one: the weight of the key's record)
    asm volatile("ds_read_b64 %0, %2\n\tds_read_u16 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(key), "=&v"(one) : "v"(L.qk0 + 8u * (qn + lane)), "v"(L.qw0 + 2u * (qn + lane)) : "memory");   // (idle lanes: an in-range entry)
    uint32_t s = (c2_slot((uint32_t)(key >> 32), (uint32_t)key) + 1u) & mask;
    for (uint32_t probes = 0; pend != 0ull; probes += 4) {
        if (probes >= 256u && ((probes & 255u) == 0u)) {
            if (probes > (uint32_t)C2_SLOTS) { if (lane == 0) *part_over = 1u; break; }
            if (c2_lds_u32(part_over)) break;
        }
        uint32_t a[4]; uint64_t v[4]; unsigned long long save;
#pragma unroll
        for (int j = 0; j < 4; j++) a[j] = L.tk0 + 8u * ((s + (uint32_t)j) & mask);
        asm volatile("s_mov_b64 %4, exec\n\ts_mov_b64 exec, %5\n\tds_read_b64 %0, %6\n\tds_read_b64 %1, %7\n\tds_read_b64 %2, %8\n\tds_read_b64 %3, %9\n\t"
                     "s_mov_b64 exec, %4\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&s"(save)
                     : "s"(pend), "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]) : "memory");
        // first slot among the four that holds the key or is empty
        unsigned long long eK[4], ev[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { eK[j] = c2_eq_u64(v[j], key); ev[j] = eK[j] | c2_eq_u64(v[j], MF_EMPTY); }
        const unsigned long long any = pend & (ev[0] | ev[1] | ev[2] | ev[3]);
        uint32_t j = skm_sel(3u, 2u, ev[2]); j = skm_sel(j, 1u, ev[1]); j = skm_sel(j, 0u, ev[0]);
        const unsigned long long f0 = ev[0], f1 = ev[1] & ~ev[0], f2 = ev[2] & ~(ev[0] | ev[1]), f3 = ev[3] & ~(ev[0] | ev[1] | ev[2]);
        unsigned long long ok = pend & ((f0 & eK[0]) | (f1 & eK[1]) | (f2 & eK[2]) | (f3 & eK[3]));      // found: count
        const unsigned long long need = any & ~ok;                                                          // an empty slot first: claim it
        const uint32_t sj = (s + j) & mask;
        unsigned long long lost = 0ull;
        if (need != 0ull) {
            const uint64_t ret = c2_cas_m(need, L.tk0 + 8u * sj, key);
            const unsigned long long won = need & c2_eq_u64(ret, MF_EMPTY);
            ok |= won | (need & c2_eq_u64(ret, key));
            lost = need & ~ok;                                              // another key got the slot: on from the next one
            won_acc += (uint32_t)__popcll(won);
        }
        c2_add_m(ok, L.tc0 + 4u * sj, one);
        pend &= ~ok;
        s = skm_sel((s + 4u) & mask, (sj + 1u) & mask, lost);
    }
}

// ---- the pipelined step.  State carried from step to step: the item and the parked record of THIS step (fetched during
// the previous one).  An item is up to TWO consecutive k-mers of one record: with four, the last item of a record is half
// empty on average and a step's instructions (VALU and LDS alike cost the same for 8 live lanes as for 64) ran at 71 %
// occupancy; with two at 93 %.  Item entry (16 bits): [0,2) k-mers in the item (0 = the padding after the list), [4,10)
// record slot, [10,16) = 63 - 4c for the record's item c: its low five bits are the funnel-shift amount, bit 15 says that the
// item starts in the record's FIRST word (c < 8) -- one field that falls by 4 from item to item, so the entries of a record
// are e0, e0 - 0x1000, e0 - 0x2000 ... and the list is written without a branch per item (k_skm_count).
template <int K>
__device__ __forceinline__ void c2_step(const c2_wave &L, uint32_t i0, uint32_t &it, skm_v4 &W, uint32_t &it1, uint32_t &qn,
                                        uint32_t &won_acc, uint32_t *part_over, uint32_t P, uint32_t pass) {
    constexpr int sh = 64 - 2 * K;              // 2 .. 24
    constexpr int nb = 2 * K - 34;              // bit of the high word where a new base enters the reverse complement
    const uint32_t lane = (uint32_t)mf_lane();
    const uint32_t nk = it & 3u;
    const uint32_t sa = it >> 10;                                          // funnel-shift amount in the low five bits
    unsigned long long q;                                                  // (VOP3 takes no 32-bit literal on gfx950: the bound sits in an SGPR)
    asm("v_cmp_lt_u32_e64 %0, %2, %1" : "=s"(q) : "v"(it), "s"(0x7FFFu));
    // two k-mers span 2K + 2 <= 64 bits from the item's first base: two funnel shifts over three words
    const uint32_t A = skm_sel(W.y, W.x, q), B = skm_sel(W.z, W.y, q), C = skm_sel(W.w, W.z, q);      // (q: first word)
    const uint32_t hr = __builtin_amdgcn_alignbit(A, B, sa), lr = __builtin_amdgcn_alignbit(B, C, sa);
    uint32_t fh[2], fl[2], ch[2], cl[2];
    fh[0] = hr >> sh; fl[0] = __builtin_amdgcn_alignbit(hr, lr, sh);
    { const uint32_t Vh = __builtin_amdgcn_alignbit(hr, lr, 30), Vl = lr << 2; fh[1] = Vh >> sh; fl[1] = __builtin_amdgcn_alignbit(Vh, Vl, sh); }
    uint32_t rh, rl;
    {
        const uint32_t H = c2_swap_pairs(__builtin_bitreverse32(~fl[0])), Lo = c2_swap_pairs(__builtin_bitreverse32(~fh[0]));
        rh = H >> sh;
        rl = __builtin_amdgcn_alignbit(H, Lo, sh);
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
        if (u) {
            rl = __builtin_amdgcn_alignbit(rh, rl, 2);
            rh = (rh >> 2) | (((~fl[u]) & 3u) << nb);
        }
        const bool lt = (((uint64_t)fh[u] << 32) | fl[u]) < (((uint64_t)rh << 32) | rl);      // one v_cmp_lt_u64
        ch[u] = lt ? fh[u] : rh;
        cl[u] = lt ? fl[u] : rl;
    }
    uint32_t s[2], ka[2]; uint64_t key[2], ret[2]; unsigned long long live[2];
    live[0] = c2_gt<0>(nk); live[1] = c2_gt<1>(nk);
#pragma unroll
    for (int u = 0; u < 2; u++) {
        key[u] = ((uint64_t)ch[u] << 32) | cl[u];
        s[u] = c2_slot(ch[u], cl[u]);
        ka[u] = L.tk0 + 8u * s[u];
    }
    if (P > 1u) {                                   // a partition counted in P passes: this pass takes the keys of its residue class
#pragma unroll
        for (int u = 0; u < 2; u++) live[u] &= c2_lt_u32((skm_pass_of(key[u]) & (P - 1u)) ^ pass, 1u);
    }
    // ---- probes (= claims) of this step, the NEXT step's record and the item of the step after that: one block, one wait.
    // (Everything an asm block reads ahead is waited for inside the same block: a register that is still in flight when the
    // block ends may be copied or spilled by the compiler before the data has landed.)
    uint32_t it2; skm_v4 W_n; unsigned long long save;
    const uint32_t ia2 = L.items0 + 2u * (i0 + 128u + lane);
    const uint32_t ra1 = L.rb0 | (it1 & 0x3F0u);                           // (rb0 is 1024-byte aligned)
    asm volatile("s_mov_b64 %4, exec\n\tds_read_u16 %2, %5\n\tds_read_b128 %3, %6\n\t"
                 "s_mov_b64 exec, %7\n\tds_cmpst_rtn_b64 %0, %9, %11, %12\n\t"
                 "s_mov_b64 exec, %8\n\tds_cmpst_rtn_b64 %1, %10, %11, %13\n\t"
                 "s_mov_b64 exec, %4\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(ret[0]), "=&v"(ret[1]), "=&v"(it2), "=&v"(W_n), "=&s"(save)
                 : "v"(ia2), "v"(ra1), "s"(live[0]), "s"(live[1]), "v"(ka[0]), "v"(ka[1]), "v"(MF_EMPTY), "v"(key[0]), "v"(key[1])
                 : "memory");
    unsigned long long won[2], ok[2], coll[2]; uint32_t aa[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
        won[u] = live[u] & c2_eq_u64(ret[u], MF_EMPTY);
        ok[u] = won[u] | (live[u] & c2_eq_u64(ret[u], key[u]));
        coll[u] = live[u] & ~ok[u];
        aa[u] = L.tc0 + 4u * s[u];
    }
    // ---- count updates of this step (nothing to wait for): + the weight of the item's record (the low half of the parked
    // record's last word: the number of identical records it stands for)
    const uint32_t one = W.w & 0xFFFFu;
    asm volatile("s_mov_b64 %0, exec\n\t"
                 "s_mov_b64 exec, %1\n\tds_add_u32 %3, %5\n\t"
                 "s_mov_b64 exec, %2\n\tds_add_u32 %4, %5\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(save)
                 : "s"(ok[0]), "s"(ok[1]), "v"(aa[0]), "v"(aa[1]), "v"(one)
                 : "memory");
    won_acc += (uint32_t)__popcll(won[0]) + (uint32_t)__popcll(won[1]);
#pragma unroll
    for (int u = 0; u < 2; u++) {
        if (coll[u] != 0ull) {
            const uint32_t at = __builtin_amdgcn_mbcnt_hi((uint32_t)(coll[u] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)coll[u], qn));
            c2_push64(coll[u], L.qk0 + 8u * at, key[u]);
            c2_push16(coll[u], L.qw0 + 2u * at, one);
            qn += (uint32_t)__popcll(coll[u]);
            while (qn >= 64u) c2_drain(L, qn, won_acc, part_over);       // (then qn < 64: + one push <= C2_QN)
        }
    }
    it = it1; W = W_n; it1 = it2;
}

// PROF: cycle counters per phase (diagnostics, option ablate & 32; s_memtime perturbs the kernel by about a tenth)
#define C2_TICK(i) do { if (PROF) { const long long t__ = clock64(); prof[i] += (unsigned long long)(t__ - tlast); tlast = t__; } } while (0)
template <int K, bool PROF>
__global__ __launch_bounds__(SKM_CT, 4) void k_skm_count(const skm_rec *__restrict__ recs, const uint64_t *__restrict__ pstart,
                                                           const uint32_t *__restrict__ plen, uint32_t np,
                                                           const uint64_t *__restrict__ toff, uint64_t *__restrict__ tkeys,
                                                           uint16_t *__restrict__ tcnt, uint32_t *__restrict__ dcount,
                                                           unsigned int *__restrict__ overflow, uint32_t p0, uint64_t tbase, int thr,
                                                           unsigned long long *__restrict__ n_all,
                                                           unsigned int *__restrict__ n_redo, unsigned long long *__restrict__ drop_hist,
                                                           unsigned long long *__restrict__ prof_out, uint64_t tcap, uint32_t dedupe,
                                                           unsigned int *__restrict__ unit_ctr) {
    // unit_ctr != nullptr (zero at launch): units are handed out as the workgroups get to them (round 4).  A workgroup's first three
    // units are blockIdx.x + {0, 1, 2} x gridDim.x; every further one is claimed from the counter THREE units ahead -- the directory entry
    // of a unit is asked for two units ahead, and the claim passes through LDS behind a barrier of the unit before that.  With the
    // fixed stride a workgroup's 1024 units add up to a total that differs by a few per cent from workgroup to workgroup (heavy
    // minimizers counted in several passes among them), and a launch ends when the slowest is done.
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long tlast = PROF ? clock64() : 0;
    // partitions [p0, np); slice of p in tkeys / tcnt starts at toff[p] - tbase; thr >= 0: entries with count <= thr are
    // dropped and tallied in drop_hist[count] (thr < C2_LH); *n_all += distinct k-mers before the cut.
    // (Two neighbouring partitions per table pass, told apart by a tag bit in the key, were tried in round 3: 48.3 ms against
    // 46.6 at 100 M reads -- a table twice as full costs more in collisions and extra passes than the halved per-partition
    // work saves.)
    // NO static __shared__ in this kernel: the dynamic array must start at LDS address 0 (the parked records are
    // addressed with `rb0 | offset`; an alignment attribute on the extern array is not honoured behind static variables)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t wave = threadIdx.x >> 6, lane = (uint32_t)mf_lane();
    skm_v4 *rbuf_all = reinterpret_cast<skm_v4 *>(smem);                        // [SKM_CT] parked records (1 KiB per wave, aligned)
    uint64_t *tk = reinterpret_cast<uint64_t *>(rbuf_all + SKM_CT);             // [C2_SLOTS]
    uint32_t *tc = reinterpret_cast<uint32_t *>(tk + C2_SLOTS);           // [C2_SLOTS]
    uint16_t *items_all = reinterpret_cast<uint16_t *>(tc + C2_SLOTS);    // [waves][C2_ITEMS]
    uint64_t *qk_all = reinterpret_cast<uint64_t *>(items_all + (SKM_CT / 64) * C2_ITEMS);   // [waves][C2_QN]
    uint16_t *qw_all = reinterpret_cast<uint16_t *>(qk_all + (SKM_CT / 64) * C2_QN);         // [waves][C2_QN] the queued keys' weights
    uint32_t *lhist = reinterpret_cast<uint32_t *>(qw_all + (SKM_CT / 64) * C2_QN);          // [C2_LH] dropped counts, committed
    uint32_t *lhist_try = lhist + C2_LH;                                                     // [C2_LH] ... of the running attempt at a partition
    uint32_t (*pflags)[2] = reinterpret_cast<uint32_t (*)[2]>(lhist_try + C2_LH);            // [parity][1] part_over
    uint32_t &out_cursor = lhist_try[C2_LH + 4], &blk_claims = lhist_try[C2_LH + 6];
    uint32_t &dd_gone = lhist_try[C2_LH + 5], &dd_all = lhist_try[C2_LH + 7];                // records that dropped out / that were looked at, of the last such unit
    uint32_t &claim = lhist_try[C2_LH + 8];                                                  // the unit claimed for three units ahead (unit_ctr)
    c2_wave L;
    L.tk0 = mf_lds_addr(tk); L.tc0 = mf_lds_addr(tc);
    L.rb0 = mf_lds_addr(rbuf_all + wave * 64);
    L.items0 = mf_lds_addr(items_all + wave * C2_ITEMS);
    L.qk0 = mf_lds_addr(qk_all + wave * C2_QN);
    L.qw0 = mf_lds_addr(qw_all + wave * C2_QN);
    uint32_t qn = 0, won_acc = 0;                                               // wave-uniform
    unsigned long long all_acc = 0;                                             // per wave: distinct k-mers of its share of the tables
    uint32_t ones_acc = 0;                                                      // per wave: dropped entries with count 1
    // the table is cleared ONCE; after that every partition leaves it clean (its compaction clears the slots it claimed)
    for (uint32_t i = threadIdx.x; i < (uint32_t)C2_SLOTS; i += (uint32_t)SKM_CT) { tk[i] = MF_EMPTY; tc[i] = 0; }
    for (uint32_t i = threadIdx.x; i < 2u * (uint32_t)C2_LH; i += (uint32_t)SKM_CT) lhist[i] = 0;
    if (threadIdx.x == 0) { out_cursor = 0; blk_claims = 0; dd_gone = 0; dd_all = 0; pflags[0][0] = pflags[0][1] = pflags[1][0] = pflags[1][1] = 0; }
    const skm_rec SENT = make_ulonglong2(~0ull, ~0ull);
    const uint32_t mine = (lane >> 3) * (uint32_t)(SKM_NW * 8) + wave * 8u + (lane & 7u);          // this lane's record within a round of SKM_CT
    const uint32_t nu = np - p0;                                                // unit u = partition p0 + u
    uint32_t ui = blockIdx.x;
    if (ui >= nu) return;
    // (an address select between the record and a sentinel object would turn the load into a flat load from scratch)
    // (a uniform base and a 32-bit lane offset: the load takes its address from an SGPR pair + one VGPR, and no 64-bit lane
    // addresses live across the unit loop)
    auto load_rec = [&](uint64_t first, uint32_t j, uint32_t n) -> skm_rec {
        const bool ok = j < n;
        const uint4 *const base = reinterpret_cast<const uint4 *>(recs) + (n ? first : 0ull);
        uint32_t jj = ok ? j : 0u;
        asm volatile("" : "+v"(jj));                       // (the address is made here, not ahead of time and kept -- or spilt -- until here)
        const uint4 w = base[jj];
        skm_rec v = make_ulonglong2(((uint64_t)w.y << 32) | w.x, ((uint64_t)w.w << 32) | w.z);
        if (!ok) v = SENT;
        return v;
    };
    // Directory entries are fetched TWO units ahead, with VECTOR loads: a scalar load (what hipcc makes of a uniform
    // address) is counted in lgkmcnt, and the next LDS wait would sit out its whole memory latency (and hipcc spilt the
    // scalars to VGPR lanes at once, with a wait after every load).  `vz` is a zero the compiler cannot see through.
    // (made anew for every entry: a register that lives across the unit loop is one the allocator may spill, and a reload from
    // scratch waits with vmcnt(0), i.e. for the compaction's stores to be acknowledged -- 13 % of the wave cycles at the loop's top)
    struct dirent { uint64_t start, toff0, toff1; uint32_t len; };
    auto load_dir = [&](uint32_t u) -> dirent {
        uint32_t vz; asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
        dirent d; const uint32_t i = p0 + u + vz;
        d.start = pstart[i]; d.len = plen[i]; d.toff0 = toff[i]; d.toff1 = toff[i + 1];
        return d;
    };
    dirent dc = load_dir(ui);
    uint64_t start = c2_uniform64(dc.start);
    uint32_t len = (uint32_t)__builtin_amdgcn_readfirstlane((int)dc.len);
    uint64_t o = c2_uniform64(dc.toff0) - tbase; uint32_t room = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(dc.toff1 - dc.toff0));
    // ---- identical records are counted once (round 3).  A minimizer partition of deeply sequenced reads holds the same
    // super-k-mer many times over (the same stretch of a genome read again and again: 100 M reads of the benchmark's community
    // have 27 % distinct records, holding 27 % of the k-mers).  A unit of up to C2_DD * SKM_CT records therefore goes through
    // the (empty) k-mer table once as RECORDS: a 64-bit compare-and-swap of (fingerprint, index in the unit) claims a slot;
    // a record that finds its fingerprint there fetches the claimant (L2: the unit has just been read) and, if the two are
    // the same bases, adds itself to the slot's weight and drops out.  The claimants take their weights, hand the slots
    // back, and only they (and the records whose slot was taken by somebody else) are dealt out as items, their weight in the
    // parked record's last word (c2_step adds it instead of 1).  Two more workgroup barriers per unit; exact, because a
    // record only drops out after a full comparison with the record that counts for it.
    // the same load in two halves: the raw words now, the record (or the sentinel) when they are wanted -- nothing looks at the
    // loaded registers in between
    auto load_raw = [&](uint64_t first, uint32_t j, uint32_t n) -> uint4 {
        const uint4 *const base = reinterpret_cast<const uint4 *>(recs) + (n ? first : 0ull);
        uint32_t jj = j < n ? j : 0u;
        asm volatile("" : "+v"(jj));
        return base[jj];
    };
    auto from_raw = [&](const uint4 &w, uint32_t j, uint32_t n) -> skm_rec {
        skm_rec v = make_ulonglong2(((uint64_t)w.y << 32) | w.x, ((uint64_t)w.w << 32) | w.z);
        if (!(j < n)) v = SENT;
        return v;
    };
    auto load4 = [&](uint64_t first, uint32_t n, skm_rec (&R)[C2_DD]) {
#pragma unroll
        for (int i = 0; i < C2_DD; i++) R[i] = load_rec(first, (uint32_t)i * (uint32_t)SKM_CT + mine, n);
    };
    // the record as the steps want it: shifted right by one bit, bases only; `low` = k-mers << 16 | weight
    auto park = [&](const skm_rec &c, uint32_t low, uint32_t slot) {
        const uint32_t x1 = (uint32_t)(c.x >> 32), x0 = (uint32_t)c.x, y1 = (uint32_t)(c.y >> 32), y0 = (uint32_t)c.y & 0xF0000000u;
        skm_v4 Pk;
        Pk.x = x1 >> 1; Pk.y = __builtin_amdgcn_alignbit(x1, x0, 1); Pk.z = __builtin_amdgcn_alignbit(x0, y1, 1); Pk.w = __builtin_amdgcn_alignbit(y1, y0, 1) | low;
        *reinterpret_cast<__attribute__((address_space(3))) skm_v4 *>((uintptr_t)(L.rb0 + 16u * slot)) = Pk;
    };
    // one round of a wave: the records parked in its slots (lane = slot, r = the slot's k-mers) are dealt out as items of two
    // k-mers and inserted; false: nothing to do
    auto run_round = [&](uint32_t r, uint32_t *part_over, uint32_t P, uint32_t pass) {
        const uint32_t nch = (r + (uint32_t)(C2_W - 1)) / (uint32_t)C2_W;
        uint32_t NI;
        const uint32_t ioff = mf_wave_excl_scan(nch, &NI);
        if (NI == 0) return;                                                // wave-uniform
        {   // the record's items: entry c = e0 - c * 0x1000 (two k-mers; see c2_step), lanes without an item c write to
            // their dummy slot -- no branch per item; a last item of ONE k-mer is written again afterwards
            static_assert(C2_W == 2 && C2_ITEMS >= 640 + 128 + 64 + 10, "item list layout");
            const uint32_t e0 = 2u | (lane << 4) | (63u << 10);
            const uint32_t ia = L.items0 + 2u * ioff, dummy = L.items0 + 2u * (uint32_t)(C2_ITEMS - 74) + 2u * lane;
#pragma unroll
            for (int c = 0; c < 10; c++) {
                const uint32_t addr = (uint32_t)c < nch ? ia : dummy;
                const uint32_t e = e0 - (uint32_t)c * 0x1000u;
                asm volatile("ds_write_b16 %0, %1 offset:%2" :: "v"(addr), "v"(e), "n"(2 * c) : "memory");
            }
            if (r & 1u) {
                const uint32_t e = e0 - (nch - 1u) * 0x1000u - 1u;
                *reinterpret_cast<__attribute__((address_space(3))) uint16_t *>((uintptr_t)(ia + 2u * (nch - 1u))) = (uint16_t)e;
            }
        }
        // padding after the list: items without k-mers for the idle lanes of the last step (the step after
        // that is only read ahead, never used)
        {
            const uint32_t z = 0u;
#pragma unroll
            for (int t = 0; t < 2; t++)
                *reinterpret_cast<__attribute__((address_space(3))) uint16_t *>((uintptr_t)(L.items0 + 2u * (NI + 64u * (uint32_t)t + lane))) = (uint16_t)z;
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t it, it1; skm_v4 W;
        // (LDS operations of one wave execute in order: the reads see the stores above without a wait in between)
        asm volatile("ds_read_u16 %0, %2\n\tds_read_u16 %1, %2 offset:128\n\ts_waitcnt lgkmcnt(0)" : "=&v"(it), "=&v"(it1) : "v"(L.items0 + 2u * lane) : "memory");
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(W) : "v"(L.rb0 | (it & 0x3F0u)) : "memory");
        C2_TICK(1);                                                         // round set-up (incl. the wait for the records)
        for (uint32_t i0 = 0; i0 < NI; i0 += 64) c2_step<K>(L, i0, it, W, it1, qn, won_acc, part_over, P, pass);      // wave-uniform
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // items / rbuf are rewritten in the next round
        __builtin_amdgcn_wave_barrier();
        C2_TICK(2);                                                         // steps (with the drains inside them)
    };
    // the records of a unit's first C2_DD rounds are in registers before the unit starts (fetched while the unit before it
    // is worked on); a longer unit fetches every further round while the round before it is worked on
    skm_rec R[C2_DD];
    uint32_t dd_skip = 0, dd_seen = 0;                                          // wave-uniform: units still to go without looking for identical records; units that looked
    load4(start, len, R);
    // (a directory entry lives in VGPRs only while its loads are in flight: once landed it moves to SGPRs)
    struct sdirent { uint64_t start, o; uint32_t len, room; };
    auto to_scalar = [&](const dirent &d) -> sdirent {
        sdirent q; q.start = c2_uniform64(d.start); q.o = c2_uniform64(d.toff0) - tbase;
        q.len = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.len); q.room = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(d.toff1 - d.toff0));
        return q;
    };
    dirent dn = {0, 0, 0, 0};
    uint32_t un = ui + gridDim.x, unn = ui + 2u * gridDim.x;                   // the next unit, and the one after it
    if (un < nu) dn = load_dir(un);
    // (settled before the loop: a load still pending on entry would make the compiler wait at the loop's top -- behind the
    // directory loads it has just issued there -- on every iteration)
    asm volatile("" :: "v"(R[0].x), "v"(R[0].y), "v"(R[1].x), "v"(R[1].y), "v"(R[2].x), "v"(R[2].y), "v"(R[3].x), "v"(R[3].y),
                       "v"(dn.start), "v"(dn.toff0), "v"(dn.toff1), "v"(dn.len));
    static_assert(C2_DD == 4, "the settle lists name R[0..3]");
    sdirent sn = to_scalar(dn);                                                 // the next unit
    __syncthreads();
    uint32_t parity = 0;
    for (;;) {
        bool abandon = false;
        uint32_t claimed = 0; bool claim_pending = false;
        uint32_t next_claim = 0; bool claim_latched = false;                      // the claim as every wave read it behind B1 of the first pass
        if (unit_ctr && threadIdx.x == 0) { claimed = atomicAdd(unit_ctr, 1u); claim_pending = true; }      // (in flight until the unit's first barrier B1)
        dirent dnn = {0, 0, 0, 0};
        bool dnn_asked = false;
        const uint64_t start_n = sn.start;
        const uint32_t len_n = sn.len;
        sdirent snn = {0, 0, 0, 0};                                             // the unit after the next (from dnn, once it has landed)
        C2_TICK(0);                                                             // unit top: directory
        // ---- identical records of the unit: weights wl[i] (0: no record, or counted by another one)
        // (reads without depth have nothing to tell apart: where fewer than a quarter of a unit's records dropped out, the next
        // 15 units of this workgroup go without the search; dedupe & 4: never skip -- tests)
        // A unit of more than C2_DD * SKM_CT records: the search covers its first records, the others follow round by round.
        const bool fast = dedupe != 0u && dd_skip == 0u;
        if (dd_skip) dd_skip--;
        uint32_t wl[C2_DD] = {0u, 0u, 0u, 0u};
        if (fast) {
            constexpr uint64_t YM = ~(((1ull << SKM_DIGIT_BITS) - 1ull) << 6);  // (the digit bits say where a record went, not what it holds)
            uint32_t sa[C2_DD], st[C2_DD];                                      // slot address; 1: claimed the slot, 2: found its fingerprint there
            uint64_t ret[C2_DD];
#pragma unroll
            for (int i = 0; i < C2_DD; i++) {
                const bool have = skm_rec_valid(R[i]) && skm_rec_n(R[i]) != 0u;
                wl[i] = have ? 1u : 0u; st[i] = 0; sa[i] = L.tk0; ret[i] = 0;
                if (have) {
                    const uint64_t ym = R[i].y & YM;
                    // (full-rate instructions and one multiply: what two different records share here only costs them their chance)
                    uint32_t h = (uint32_t)(R[i].x >> 32) ^ __builtin_amdgcn_alignbit((uint32_t)R[i].x, (uint32_t)R[i].x, 11)
                                 ^ __builtin_amdgcn_alignbit((uint32_t)(ym >> 32), (uint32_t)(ym >> 32), 21) ^ __builtin_amdgcn_alignbit((uint32_t)ym, (uint32_t)ym, 5);
                    h ^= h >> 16; h *= 0x9E3779B1u; h ^= h >> 15;
                    sa[i] = L.tk0 + 8u * (h & (uint32_t)(C2_SLOTS - 1));
                    const uint64_t word = (1ull << 32) | (uint64_t)(((h >> DD_BITS) << DD_BITS) | ((uint32_t)i * (uint32_t)SKM_CT + mine));
                    asm volatile("ds_cmpst_rtn_b64 %0, %1, %2, %3" : "=&v"(ret[i]) : "v"(sa[i]), "v"(MF_EMPTY), "v"(word) : "memory");
                    st[i] = h >> DD_BITS;                                       // (the fingerprint, until the answer is in)
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            static_assert(C2_DD * SKM_CT <= (1 << DD_BITS), "index in the unit");
            skm_rec G[C2_DD];
#pragma unroll
            for (int i = 0; i < C2_DD; i++) {
                const uint32_t fp = st[i];
                st[i] = 0;
                if (wl[i]) {
                    if (ret[i] == MF_EMPTY) st[i] = 1;
                    else if (((uint32_t)ret[i] >> DD_BITS) == fp) st[i] = 2;
                }
                G[i] = SENT;
                if (st[i] == 2) G[i] = load_rec(start, (uint32_t)ret[i] & ((1u << DD_BITS) - 1u), len);          // the claimant
            }
#pragma unroll
            for (int i = 0; i < C2_DD; i++) {
                if (st[i] == 2 && G[i].x == R[i].x && ((G[i].y ^ R[i].y) & YM) == 0ull) {
                    const uint32_t one = 1u;
                    asm volatile("ds_add_u32 %0, %1 offset:4" :: "v"(sa[i]), "v"(one) : "memory");
                    wl[i] = 0;
                }
            }
            const bool dd_look = (dd_seen++ & 3u) == 0u;                        // (every fourth such unit is looked at)
            if (dd_look) {   // how many dropped out (the next units' decision)
                uint32_t gone = 0, all = 0;
#pragma unroll
                for (int i = 0; i < C2_DD; i++) { gone += (uint32_t)__popcll(__ballot(st[i] == 2 && wl[i] == 0u)); all += (uint32_t)__popcll(__ballot(st[i] != 0u || wl[i] != 0u)); }
                if (lane == 0) { atomicAdd(&dd_gone, gone); atomicAdd(&dd_all, all); }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            c2_barrier();                                                       // ---- every record has claimed, added itself or stays as it is
            if (dd_look) {
                const uint32_t gone = c2_lds_u32(&dd_gone), all = c2_lds_u32(&dd_all);
                if (gone * 4u < all && !(dedupe & 4u)) dd_skip = 15;
            }
            {   // the claimants take their weights and hand the slots back: four exchanges under their lane masks, ONE wait
                static_assert(C2_DD == 4, "four exchanges");
                const unsigned long long m0 = __ballot(st[0] == 1), m1 = __ballot(st[1] == 1), m2 = __ballot(st[2] == 1), m3 = __ballot(st[3] == 1);
                uint64_t v0 = 0, v1 = 0, v2 = 0, v3 = 0; unsigned long long save;
                asm volatile("s_mov_b64 %4, exec\n\t"
                             "s_mov_b64 exec, %5\n\tds_wrxchg_rtn_b64 %0, %9, %13\n\t"
                             "s_mov_b64 exec, %6\n\tds_wrxchg_rtn_b64 %1, %10, %13\n\t"
                             "s_mov_b64 exec, %7\n\tds_wrxchg_rtn_b64 %2, %11, %13\n\t"
                             "s_mov_b64 exec, %8\n\tds_wrxchg_rtn_b64 %3, %12, %13\n\t"
                             "s_mov_b64 exec, %4\n\ts_waitcnt lgkmcnt(0)"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "=&s"(save)
                             : "s"(m0), "s"(m1), "s"(m2), "s"(m3), "v"(sa[0]), "v"(sa[1]), "v"(sa[2]), "v"(sa[3]), "v"(MF_EMPTY) : "memory");
                if (st[0] == 1) wl[0] = (uint32_t)(v0 >> 32);
                if (st[1] == 1) wl[1] = (uint32_t)(v1 >> 32);
                if (st[2] == 1) wl[2] = (uint32_t)(v2 >> 32);
                if (st[3] == 1) wl[3] = (uint32_t)(v3 >> 32);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            c2_barrier();                                                       // ---- the table is empty again
            if (dd_look && threadIdx.x == 0) { dd_gone = 0; dd_all = 0; }       // (read by everybody before this barrier; added to again a unit later)
        }
        C2_TICK(7);                                                             // identical records
        // A unit with more distinct k-mers than the table takes (a heavy minimizer: low-complexity sequence between many
        // different flanks) is counted again in P = 4, 16, 64 passes over its records, pass i inserting the k-mers with
        // skm_pass_of(key) mod P = i; the passes append to the same slices.  More than 64 passes: the global overflow flag,
        // and the caller falls back to the k-mer path.  What a failed attempt tallied is not committed.
        uint32_t P = 1, pass = 0, ones_try = 0, all_try = 0;
        bool have_cur = true;                                                   // R holds THIS unit's records (not yet the next unit's)
        for (;;) {
        P = (uint32_t)__builtin_amdgcn_readfirstlane((int)P); pass = (uint32_t)__builtin_amdgcn_readfirstlane((int)pass);      // (uniform: say so)
        uint32_t *part_over = &pflags[parity][1];
        const bool fetch_next = pass + 1u == P && un < nu;                      // (the pass that is meant to be the unit's last)
        // (Tried: the survivors DEALT evenly over the eight waves through the parked-record area before the barrier that ends the
        // search, so that every unit is one round of equal shares -- 34.2 ms per step round-robin, 34.5 in blocks, against 31.7
        // with every wave parking its own survivors: waves in lock step want the same unit of the CU at the same time.)
        // One loop of rounds for both kinds of unit.  A unit whose identical records have been told apart parks its surviving
        // records 64 to a round (wave by wave: `total` of them in this wave, numbered in the order i, lane); any other unit
        // parks round after round of its records as they come, each lane its own.
        uint32_t total = 0;
        constexpr uint32_t DDN = (uint32_t)(C2_DD * SKM_CT);
        if (fast) {
            if (!have_cur) { load4(start, len, R); have_cur = true; }          // (the pass before fetched the next unit and then overflowed)
#pragma unroll
            for (int i = 0; i < C2_DD; i++) total += (uint32_t)__popcll(__ballot(wl[i] != 0u));
        } else if (!have_cur) { R[0] = load_rec(start, mine, len); have_cur = true; }      // (any other unit: R[0] is the round's record)
        // rounds of surviving records, then rounds of the records as they come (from record `plain0` on)
        const uint32_t n_dense = fast ? (total + 63u) >> 6 : 0u;
        const uint32_t plain0 = fast ? DDN : 0u;
        const uint32_t n_plain = len > plain0 ? (len - plain0 + (uint32_t)SKM_CT - 1u) / (uint32_t)SKM_CT : 0u;
        bool gave_up = false;
        for (uint32_t rr = 0; rr < n_dense; rr++) {
            if (rr && c2_lds_u32(part_over)) { gave_up = true; break; }        // (abandoned)
            uint32_t before = 0;
#pragma unroll
            for (int i = 0; i < C2_DD; i++) {
                const unsigned long long m = __ballot(wl[i] != 0u);
                const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, before));
                before += (uint32_t)__popcll(m);
                if (wl[i] != 0u && (pos >> 6) == rr) park(R[i], (skm_rec_n(R[i]) << 16) | wl[i], pos & 63u);
            }
            __builtin_amdgcn_wave_barrier();
            uint32_t w3;
            asm volatile("ds_read_b32 %0, %1 offset:12\n\ts_waitcnt lgkmcnt(0)" : "=v"(w3) : "v"(L.rb0 + 16u * lane) : "memory");
            run_round(lane < total - rr * 64u ? (w3 >> 16) & 63u : 0u, part_over, P, pass);
        }
        if (fast && n_plain != 0u && !gave_up) R[0] = load_rec(start, plain0 + mine, len);      // (a long unit's other records; waited for on the spot)
        for (uint32_t rr = 0; rr < n_plain && !gave_up; rr++) {
            if ((rr || n_dense) && c2_lds_u32(part_over)) break;                // (abandoned)
            // hipcc waits with vmcnt(0) at the first use of a loaded register, i.e. for EVERY load in flight.  This round's
            // records (loaded a round ago) are "used" here, BEFORE the next round's load goes out: the wait the compiler puts
            // in front of this statement finds them landed, and nothing further down waits for the prefetch (with the use
            // after the prefetch every round sat out a full HBM round trip: profiles/r03_count_vmcnt.txt).
            asm volatile("" :: "v"(R[0].x), "v"(R[0].y));
            const uint32_t r = skm_rec_valid(R[0]) ? skm_rec_n(R[0]) : 0u;
            park(R[0], (r << 16) | 1u, lane);
            if (rr + 1u < n_plain) R[0] = load_rec(start, plain0 + (rr + 1u) * (uint32_t)SKM_CT + mine, len);      // the round after this one
            run_round(r, part_over, P, pass);
        }
        if (n_plain) have_cur = false;                                          // (R[0] has moved on, or the pass was abandoned on the way)
        // ---- the next unit's records and the directory entry after it, in ONE place and in straight-line code: they are in
        // flight during the drains, the wait for the other waves and the compaction, and are settled after it.  (Asked for
        // inside the round loops, the compiler joined the loaded registers with the loop's own copies of R right behind the
        // loads -- a copy is a use, and a use waits for the load: every unit sat out an HBM round trip before its first step.)
        // The raw words go into registers of their own (T) and become R only where they are settled: R is carried around the
        // pass loop and the unit loop, and where the loaded registers were R themselves the compiler copied them into R's loop
        // registers right behind the loads.
        uint4 T[C2_DD];
#pragma unroll
        for (int i = 0; i < C2_DD; i++) T[i] = make_uint4(0u, 0u, 0u, 0u);
        if (fetch_next) {
            // (a unit that will not search for identical records takes its rounds one by one: only the first is read ahead)
            const bool next_fast = dedupe != 0u && dd_skip == 0u;
            T[0] = load_raw(start_n, mine, len_n);
            if (next_fast) {
#pragma unroll
                for (int i = 1; i < C2_DD; i++) T[i] = load_raw(start_n, (uint32_t)i * (uint32_t)SKM_CT + mine, len_n);
            }
        }
        if (!dnn_asked) { if (unn < nu) dnn = load_dir(unn); dnn_asked = true; }
        while (qn) c2_drain(L, qn, won_acc, part_over);                        // wave-uniform
        // a crowded table probes slowly: past C2_FILL claims the unit is counted in (more) passes
        if (won_acc) { if (lane == 0) atomicAdd(&blk_claims, won_acc); won_acc = 0; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the LDS operations of the asm blocks are invisible to hipcc's waitcnt pass
        C2_TICK(3);                                                             // last drains
        if (claim_pending) { claim = 3u * gridDim.x + claimed; claim_pending = false; }
        c2_barrier();                                                           // ---- B1: every insert of the pass is done
        C2_TICK(4);                                                             // waiting for the other waves
        const bool over = c2_lds_u32(part_over) != 0u || c2_lds_u32(&blk_claims) > (uint32_t)C2_FILL;
        // (the claim is read HERE, one barrier behind thread 0's write and at least one -- B2 of this unit, or the barriers of
        // its further passes -- before thread 0 writes the next unit's; read at the bottom of the unit loop, a wave that lagged behind B2
        // could have met the newer value when the search for identical records is off and nothing else separates the two)
        if (unit_ctr && !claim_latched) { next_claim = c2_lds_u32(&claim); claim_latched = true; }
        if (threadIdx.x == 0) pflags[parity ^ 1u][1] = 0;                      // for the next pass / unit
        // the records and the directory entries read ahead are settled HERE, before the compaction's stores go out (they had the drains
        // and the wait for the other waves to land): behind the stores the wait would be for the stores' acknowledgements too
        asm volatile("" :: "v"(T[0].x), "v"(T[0].w), "v"(T[1].x), "v"(T[1].w), "v"(T[2].x), "v"(T[2].w), "v"(T[3].x), "v"(T[3].w),
                           "v"(dnn.start), "v"(dnn.toff0), "v"(dnn.toff1), "v"(dnn.len));
        snn = to_scalar(dnn);
        if (fetch_next) {
#pragma unroll
            for (int i = 0; i < C2_DD; i++) R[i] = from_raw(T[i], (uint32_t)i * (uint32_t)SKM_CT + mine, len_n);
            have_cur = false;
        }
        // ---- compaction: every wave sweeps ITS eighth of the table (consecutive slots, lane = slot: conflict-free),
        // 8 chunks of 64 in flight: keys are read and reset with one exchange, the counts of the occupied slots likewise.
        // (Claim lists -- visit only the slots that were won -- cost a list append per key slot in the hot loop and four
        // dependent LDS round trips per 64 entries here; with the table 2/5 full the sweep is cheaper on both counts.)
        {
            constexpr int NCH = 8;                                      // chunks in flight
            static_assert(C2_SLOTS % (SKM_NW * 64 * NCH) == 0, "sweep: whole groups of eight chunks per wave");
#pragma unroll 1
            for (uint32_t g = 0; g < (uint32_t)(C2_SLOTS / (SKM_NW * 64 * NCH)); g++) {
            uint64_t ck[NCH]; uint32_t cc[NCH]; unsigned long long have[NCH], keep[NCH];
            const uint32_t sl0 = wave * (uint32_t)(C2_SLOTS / SKM_NW) + g * (uint32_t)(64 * NCH) + lane;
            const uint32_t ka0 = L.tk0 + 8u * sl0, ca0 = L.tc0 + 4u * sl0;
            const uint64_t E = MF_EMPTY; const uint32_t zero = 0u;
            // (the counters of ALL lanes: an exchange costs the same for 8 lanes as for 64)
            asm volatile("ds_wrxchg_rtn_b64 %0, %16, %17\n\tds_wrxchg_rtn_b64 %1, %16, %17 offset:512\n\tds_wrxchg_rtn_b64 %2, %16, %17 offset:1024\n\t"
                         "ds_wrxchg_rtn_b64 %3, %16, %17 offset:1536\n\tds_wrxchg_rtn_b64 %4, %16, %17 offset:2048\n\tds_wrxchg_rtn_b64 %5, %16, %17 offset:2560\n\t"
                         "ds_wrxchg_rtn_b64 %6, %16, %17 offset:3072\n\tds_wrxchg_rtn_b64 %7, %16, %17 offset:3584\n\t"
                         "ds_wrxchg_rtn_b32 %8, %18, %19\n\tds_wrxchg_rtn_b32 %9, %18, %19 offset:256\n\tds_wrxchg_rtn_b32 %10, %18, %19 offset:512\n\t"
                         "ds_wrxchg_rtn_b32 %11, %18, %19 offset:768\n\tds_wrxchg_rtn_b32 %12, %18, %19 offset:1024\n\tds_wrxchg_rtn_b32 %13, %18, %19 offset:1280\n\t"
                         "ds_wrxchg_rtn_b32 %14, %18, %19 offset:1536\n\tds_wrxchg_rtn_b32 %15, %18, %19 offset:1792\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(ck[0]), "=&v"(ck[1]), "=&v"(ck[2]), "=&v"(ck[3]), "=&v"(ck[4]), "=&v"(ck[5]), "=&v"(ck[6]), "=&v"(ck[7]),
                           "=&v"(cc[0]), "=&v"(cc[1]), "=&v"(cc[2]), "=&v"(cc[3]), "=&v"(cc[4]), "=&v"(cc[5]), "=&v"(cc[6]), "=&v"(cc[7])
                         : "v"(ka0), "v"(E), "v"(ca0), "v"(zero) : "memory");
#pragma unroll
            for (int i = 0; i < NCH; i++) have[i] = ~c2_eq_u64(ck[i], MF_EMPTY) & __builtin_amdgcn_ballot_w64(true);
            if (!over) {
                // (straight-line code: eight chunks of branches -- is there a cut, did anything drop, is anything kept -- cost more
                // than the work they skipped; the cut is one signed compare, count > thr, which holds for every entry when thr = -1)
                uint32_t nkeep = 0, nhave = 0;
                unsigned long long odd = 0ull;                                  // chunks with dropped entries of a count other than 1
#pragma unroll
                for (int i = 0; i < NCH; i++) {
                    asm("v_min_u32 %0, %0, %1" : "+v"(cc[i]) : "s"((uint32_t)MF_MAX_COUNT));
                    unsigned long long gt;
                    asm("v_cmp_gt_i32_e64 %0, %1, %2" : "=s"(gt) : "v"(cc[i]), "s"(thr));
                    keep[i] = have[i] & gt;
                    const unsigned long long drop = have[i] & ~gt;
                    const unsigned long long d1 = drop & c2_lt_u32(cc[i], 2u);       // (count 1: nearly all of them)
                    ones_try += (uint32_t)__popcll(d1);
                    odd |= drop & ~d1;
                    nkeep += (uint32_t)__popcll(keep[i]); nhave += (uint32_t)__popcll(have[i]);
                }
                if (odd != 0ull) {                                              // (rare with the usual cut at 1: never)
#pragma unroll
                    for (int i = 0; i < NCH; i++) {
                        const unsigned long long dx = have[i] & ~keep[i] & ~c2_lt_u32(cc[i], 2u);
                        if ((dx >> lane) & 1ull) atomicAdd(&lhist_try[cc[i]], 1u);
                    }
                }
                all_try += nhave;
                uint32_t wb = 0;
                if (lane == 0 && nkeep) wb = atomicAdd(&out_cursor, nkeep);
                wb = (uint32_t)__builtin_amdgcn_readfirstlane((int)wb);
                if (nkeep && (wb + nkeep > room || o + wb + nkeep > tcap)) {                   // (only if a partition's k-mer count wrapped)
                    if (lane == 0) {
                        if (atomicExch(overflow, 2u) == 0u && prof_out) {
                            prof_out[8] = p0 + ui; prof_out[9] = o; prof_out[10] = room; prof_out[11] = wb; prof_out[12] = nkeep; prof_out[13] = tcap;
                            prof_out[14] = p0; prof_out[15] = np;
                        }
                    }
                }
                else {
                    // the kept entries of a chunk go out under the chunk's lane mask (EXEC is switched around the two stores: a
                    // C++ `if` on the lane's bit is a compare, a saved EXEC and a branch per chunk)
                    uint64_t *const kb = tkeys + o; uint16_t *const cb = tcnt + o;
#pragma unroll
                    for (int i = 0; i < NCH; i++) {
                        const uint32_t at = __builtin_amdgcn_mbcnt_hi((uint32_t)(keep[i] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)keep[i], wb));
                        const uint64_t ka = (uint64_t)(uintptr_t)(kb + at), ca = (uint64_t)(uintptr_t)(cb + at);
                        unsigned long long save;
                        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\tglobal_store_dwordx2 %2, %3, off\n\tglobal_store_short %4, %5, off\n\ts_mov_b64 exec, %0"
                                     : "=&s"(save) : "s"(keep[i]), "v"(ka), "v"(ck[i]), "v"(ca), "v"(cc[i]) : "memory");
                        wb += (uint32_t)__popcll(keep[i]);
                    }
                }
            }
        }
        }
        qn = 0;
        C2_TICK(5);                                                             // compaction
        c2_barrier();                                                           // ---- B2: the table is clean, the cursors final
        C2_TICK(6);
        parity ^= 1u;
        if (threadIdx.x == 0) blk_claims = 0;
        bool done = false;
        if (over) {
            // start again with four times the passes; nothing of this attempt counts
            ones_try = 0; all_try = 0; pass = 0;
            if (threadIdx.x == 0) out_cursor = 0;
            if (thr >= 0) for (uint32_t i = threadIdx.x; i < (uint32_t)C2_LH; i += (uint32_t)SKM_CT) lhist_try[i] = 0;
            if (P >= (uint32_t)SKM_MAX_PASSES) { if (threadIdx.x == 0) atomicExch(overflow, 1u); done = true; abandon = true; }
#ifndef C2_PASS_MULT
#define C2_PASS_MULT 4u          /* passes of a unit that overflowed its table: x 4 each time (x 2 measured in round 5: no difference -- 1209 of 2^20 units are redone at all; both unit bounds, distinct k-mers and records, ask for 2^20 units on the benchmark: profiles/r05x_pass_mult.txt) */
#endif
            P *= C2_PASS_MULT;
            if (threadIdx.x == 0 && n_redo && P == C2_PASS_MULT) atomicAdd(n_redo, 1u);    // (statistics: units counted in several passes)
        } else if (++pass == P) {
            done = true;
            // (wave-uniform sums, and told so: in SGPRs they cost nothing across the unit loop, as VGPRs the allocator spilt them to
            // scratch at the loop's top -- a reload waits with vmcnt(0) for the compaction's stores)
            ones_acc = (uint32_t)__builtin_amdgcn_readfirstlane((int)(ones_acc + ones_try)); all_acc = c2_uniform64(all_acc + all_try);
            if (thr >= 2) for (uint32_t i = threadIdx.x; i < (uint32_t)C2_LH; i += (uint32_t)SKM_CT) { const uint32_t v = lhist_try[i]; if (v) { atomicAdd(&lhist[i], v); lhist_try[i] = 0; } }
        }
        // (said to be uniform: behind an exit the compiler takes for divergent, everything the loop hands on -- the directory
        // entries read ahead among it -- is kept in VGPRs, six more across the unit loop)
        if (__builtin_amdgcn_readfirstlane((int)done)) break;
        c2_barrier();                                                           // (the cursors / the tallies are reset before the next pass appends)
        }
        if (threadIdx.x == 0) { dcount[p0 + ui] = out_cursor; out_cursor = 0; }
        // (a unit beyond SKM_MAX_PASSES: the run is void -- the host sees the flag and takes the k-mer path -- and this workgroup stops
        // here: the registers that were to hold the next unit's records still hold this one's)
        if (__builtin_amdgcn_readfirstlane((int)abandon)) break;
        if (un >= nu) break;
        ui = un; un = unn; unn = unit_ctr ? next_claim : unn + gridDim.x;
        start = start_n; len = len_n;
        o = sn.o; room = sn.room;
        sn = snn;
    }
    {   // distinct k-mers before the cut: one atomic per wave
        if (lane == 0 && n_all && all_acc) atomicAdd(n_all, all_acc);
    }
    if (PROF && lane == 0 && prof_out) for (int i = 0; i < 8; i++) atomicAdd(&prof_out[i], prof[i]);
    if (thr >= 0 && drop_hist) {
        if (lane == 0 && ones_acc) atomicAdd(&lhist[1], ones_acc);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < (uint32_t)C2_LH; i += (uint32_t)SKM_CT) { const uint32_t v = lhist[i]; if (v) atomicAdd(&drop_hist[i], (unsigned long long)v); }
    }
}

// Slices of a batch -> dense table, every COUNTING partition cut in two TABLE partitions on the way: the counting kernel likes
// its partitions twice as large as the graph kernels like theirs (100 M reads: 38.8 ms at 12288 occurrences per partition against
// 44.9 at 6144 -- fixed costs per partition --, while the neighbour lookup's LDS table (mf_nbr.h) wants the ~170 good k-mers of a
// 6144-occurrence partition).  The next bit of the partition hash comes from the k-mer itself (its minimizer: 17 M-mer hashes per
// kept k-mer, ~2 ms at 3.6e8 of them).  One wave per counting partition; the keys with bit 0 grow from the front of the
// partition's range in the dense table, the others from its back (no order inside a partition).
template <int K>
__global__ __launch_bounds__(256) void k_gather_split(const uint64_t *__restrict__ keys, const uint16_t *__restrict__ cnt,
                                                      const uint64_t *__restrict__ pstart, const uint32_t *__restrict__ dcount,
                                                      const uint64_t *__restrict__ coff, uint32_t np, uint64_t *__restrict__ dk,
                                                      uint16_t *__restrict__ dc, uint32_t p0, uint64_t sbase, uint64_t *__restrict__ dfine,
                                                      uint64_t dbase, int bit) {
    // partitions [p0, np) of the counting pass; coff[p]: offset of p's entries in dk / dc (this batch's part of the table, which
    // starts at entry dbase of the whole); dfine[2p], dfine[2p+1] (and dfine[2 np] by the last one): offsets of the table's partitions
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (uint32_t p = p0 + blockIdx.x * 4u + wave; p < np; p += gridDim.x * 4u) {
        const uint64_t s = pstart[p] - sbase, o = coff[p];
        const uint32_t d = dcount[p];
        uint32_t front = 0, back = d;                                       // wave-uniform
        for (uint32_t j0 = 0; j0 < d; j0 += 64) {
            const uint32_t j = j0 + lane;
            const bool have = j < d;
            const uint64_t key = have ? keys[s + j] : 0ull;
            const uint16_t c = have ? cnt[s + j] : (uint16_t)0;
            const bool hi = ((mf_skm_ph(key, K) >> bit) & 1u) != 0u;
            const unsigned long long m1 = __ballot(have && hi), m0 = __ballot(have && !hi);
            const uint32_t n1 = (uint32_t)__popcll(m1), n0 = (uint32_t)__popcll(m0);
            const uint32_t pos = hi ? back - n1 + __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u))
                                    : front + __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u));
            if (have) { dk[o + pos] = key; dc[o + pos] = c; }
            front += n0; back -= n1;
        }
        if (lane == 0) {
            dfine[2 * (size_t)p] = dbase + o;
            dfine[2 * (size_t)p + 1] = dbase + o + front;
            if (p + 1 == np) dfine[2 * (size_t)np] = dbase + coff[np];
        }
    }
}
// The same with every counting partition cut in 2^S table partitions, S = 0 ... 4 (round 4): what the counting kernel likes --
// units of ~2000 distinct k-mers, it pays four workgroup barriers and a sweep of its table per unit -- and what the graph kernels
// like -- ~170 k-mers per table partition, their LDS lookup table takes 352 -- are further apart than a factor of two for
// assembled sequences (the cutter's input: nearly every k-mer distinct, 256 occurrences per table partition) and for shallow
// reads (most distinct k-mers survive the cut).  Two passes over a partition's entries: the bins (the next S bits of each
// k-mer's partition hash) are counted and parked in LDS, then every entry goes to its bin's range; partitions of more than
// GS_CAP entries (counted in several passes: rare) hash twice instead.
#define GS_CAP 4096
template <int K>
__global__ __launch_bounds__(256) void k_gather_split_n(const uint64_t *__restrict__ keys, const uint16_t *__restrict__ cnt,
                                                        const uint64_t *__restrict__ pstart, const uint32_t *__restrict__ dcount,
                                                        const uint64_t *__restrict__ coff, uint32_t np, uint64_t *__restrict__ dk,
                                                        uint16_t *__restrict__ dc, uint32_t p0, uint64_t sbase, uint64_t *__restrict__ dfine,
                                                        uint64_t dbase, int bit, int S) {
    // bit: position of the lowest of the S bits in the partition hash; dfine[(p << S) + b]: first entry of table partition b of p
    __shared__ uint8_t bins_all[4][GS_CAP];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint8_t *const bins = bins_all[wave];
    const uint32_t nb = 1u << S, bmask = nb - 1u;
    for (uint32_t p = p0 + blockIdx.x * 4u + wave; p < np; p += gridDim.x * 4u) {
        const uint64_t s = pstart[p] - sbase, o = coff[p];
        const uint32_t d = dcount[p];
        const bool park = d <= (uint32_t)GS_CAP;
        uint32_t start[16];                                                 // wave-uniform
#pragma unroll
        for (int b = 0; b < 16; b++) start[b] = 0;
        for (uint32_t j0 = 0; j0 < d; j0 += 64) {
            const uint32_t j = j0 + lane;
            const bool have = j < d;
            const uint64_t key = have ? keys[s + j] : 0ull;
            const uint32_t bin = (mf_skm_ph(key, K) >> bit) & bmask;
            if (park && have) bins[j] = (uint8_t)bin;
#pragma unroll
            for (int b = 0; b < 16; b++)
                if ((uint32_t)b < nb) start[b] += (uint32_t)__popcll(__ballot(have && bin == (uint32_t)b));
        }
        {   // counts -> first entries
            uint32_t acc = 0;
#pragma unroll
            for (int b = 0; b < 16; b++) { const uint32_t c = start[b]; start[b] = acc; acc += c; }
        }
        if (lane < nb) {
            uint32_t mine = 0;
#pragma unroll
            for (int b = 0; b < 16; b++) if (lane == (uint32_t)b) mine = start[b];
            dfine[((size_t)p << S) + lane] = dbase + o + mine;
        }
        if (lane == 0 && p + 1 == np) dfine[(size_t)np << S] = dbase + coff[np];
        __builtin_amdgcn_wave_barrier();
        for (uint32_t j0 = 0; j0 < d; j0 += 64) {
            const uint32_t j = j0 + lane;
            const bool have = j < d;
            const uint64_t key = have ? keys[s + j] : 0ull;
            const uint16_t c = have ? cnt[s + j] : (uint16_t)0;
            const uint32_t bin = park ? (have ? (uint32_t)bins[j] : 0u) : ((mf_skm_ph(key, K) >> bit) & bmask);
            uint32_t pos = 0;
#pragma unroll
            for (int b = 0; b < 16; b++) {
                if ((uint32_t)b < nb) {
                    const unsigned long long m = __ballot(have && bin == (uint32_t)b);
                    if (bin == (uint32_t)b) pos = start[b] + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    start[b] += (uint32_t)__popcll(m);
                }
            }
            if (have) { dk[o + pos] = key; dc[o + pos] = c; }
        }
        __builtin_amdgcn_wave_barrier();
    }
}
__global__ void k_skm_add_base(uint64_t *__restrict__ v, uint64_t n, uint64_t base) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] += base;
}
// capacity of a partition's slice of the temporary (key,count) lists: it cannot hold more distinct k-mers than it has
// k-mers (a partition counted in several passes may hold more than the LDS table)
// ... and with the cut made inside the counting kernel (count > thr) every k-mer that is written stands for at least thr + 1
// occurrences: a partition of o occurrences writes at most o / (thr + 1) entries -- half the temporary lists at the usual thr = 1
// (12 instead of 24 GB for a 20 M-read sample; 26 GB less per slice where 8 x 200 M reads fight for the device)
__global__ void k_skm_cap(const uint32_t *__restrict__ pocc, uint32_t np, uint32_t *__restrict__ cap, uint32_t div) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < np) cap[p] = pocc[p] / div;
}

// =============================================================================================
// host orchestration
// =============================================================================================
template <typename KF> static int skm_set_lds(KF kern, size_t bytes) {
    MF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return MF_OK;
}

// =============================================================================================
// The PILOT (round 4): distinct k-mers per occurrence, measured before the levels after the first are planned.
// What a counting unit costs is decided by its DISTINCT k-mers (the LDS table takes C2_FILL claims; a unit beyond that is
// counted again in 4, 16, 64 passes), and a plan made from the number of occurrences alone has one sequencing depth in mind: with
// the benchmark's 7.4 occurrences per distinct k-mer built in, the same 100 M reads drawn from a pool sixteen times as large
// (5-fold instead of 83-fold depth, 3.4 occurrences per distinct k-mer) had every second unit counted in four passes -- 362 ms
// instead of 32.  The reference's map does not care: it doubles when it is full (Long2ShortHashMap.java:191-214).
// Once level 1 stands, the records of a few of its digit regions whose NEXT digits are among the smallest `nsel` values -- that
// is: a few hundred would-be units an eighth of the planned size, each with ALL the occurrences of its k-mers -- are copied out
// (two small passes over <= 8 regions: histogram, scatter) and counted with the counting kernel itself; nothing is kept but the
// number of distinct k-mers.  ~2 M occurrences, five small launches and one round trip to the host (0.2 ms at 100 M reads).
// =============================================================================================
#define SKM_PILOT_CH 8192          // records of a region per workgroup
__global__ __launch_bounds__(256) void k_skm_pilot(const skm_rec *__restrict__ recs, const uint64_t *__restrict__ pstart, const uint32_t *__restrict__ plen,
                                                   const uint32_t *__restrict__ regions, int shift, uint32_t nsel, uint32_t *__restrict__ cnt,
                                                   uint32_t *__restrict__ occ, const uint64_t *__restrict__ ostart, uint32_t *__restrict__ cur,
                                                   skm_rec *__restrict__ out) {
    // ostart == nullptr: histogram pass (cnt / occ [region][digit]: records, k-mers); else: the records go to out[ostart[bin] + ...]
    extern __shared__ uint32_t sh[];                                    // [2 * nsel] (histogram pass)
    const uint32_t rg = blockIdx.y, d1 = regions[rg];
    const uint64_t start = pstart[d1];
    const uint32_t len = plen[d1];
    const uint32_t j0 = blockIdx.x * (uint32_t)SKM_PILOT_CH;
    if (j0 >= len) return;
    const uint32_t j1 = min(len, j0 + (uint32_t)SKM_PILOT_CH);
    const bool hist = ostart == nullptr;
    if (hist) { for (uint32_t i = threadIdx.x; i < 2u * nsel; i += blockDim.x) sh[i] = 0; __syncthreads(); }
    for (uint32_t j = j0 + threadIdx.x; j < j1; j += blockDim.x) {
        const skm_rec v = recs[start + j];
        if (!skm_rec_valid(v)) continue;
        const uint32_t d = skm_rec_digits(v) >> shift;
        if (d >= nsel) continue;
        if (hist) { atomicAdd(&sh[d], 1u); atomicAdd(&sh[nsel + d], skm_rec_n(v)); }
        else { const uint32_t at = atomicAdd(&cur[rg * nsel + d], 1u); out[ostart[rg * nsel + d] + at] = v; }
    }
    if (hist) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nsel; i += blockDim.x) {
            if (sh[i]) { atomicAdd(&cnt[rg * nsel + i], sh[i]); atomicAdd(&occ[rg * nsel + i], sh[nsel + i]); }
        }
    }
}
__global__ void k_skm_pilot_dir(const uint32_t *__restrict__ cnt, uint32_t n, uint32_t *__restrict__ plen) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) plen[i] = (cnt[i] + 3u) & ~3u;
}
// -> *rho = distinct k-mers per occurrence among the sampled units (< 0: nothing sampled); bufA / pstart / plen: level 1 (indexed by digit)
template <int K>
static int skm_pilot(mf_ctx *ctx, const skm_rec *bufA, const uint64_t *pstart, const uint32_t *plen, uint32_t dlo, uint32_t dhi, uint64_t n_occ, int nd1,
                     int bits_planned, int kthr, double *rho, double *kept, double *rec_per_occ) {
    hipStream_t st = ctx->stream;
    *rho = -1.0; *kept = -1.0; *rec_per_occ = 0.0;
    if (dhi <= dlo) return MF_OK;
    const uint32_t R = std::min<uint32_t>(8u, dhi - dlo);
    const int pb = std::min((int)SKM_DIGIT_BITS, bits_planned + 3);                       // would-be units an eighth of the planned size
    const double per_unit = (double)n_occ / (double)nd1 / (double)(1u << pb);            // occurrences of one of them
    uint32_t nsel = (uint32_t)std::min<double>((double)(1u << pb), std::max(1.0, std::ceil(2.5e6 / (per_unit * R))));
    if (nsel > 2048u) nsel = 2048u;
    const uint32_t nb = R * nsel;
    std::vector<uint32_t> h_reg(R);
    for (uint32_t i = 0; i < R; i++) h_reg[i] = dlo + (uint32_t)(((uint64_t)(2 * i + 1) * (dhi - dlo)) / (2 * R));      // spread over the slice's digits
    std::vector<uint32_t> h_len(nd1);
    mf_buf<uint32_t> reg, cnt, occ, cur, uplen, dcount; mf_buf<uint64_t> ostart, toff; mf_buf<unsigned long long> ps;
    MF_TRY(reg.alloc(ctx, R)); MF_TRY(cnt.alloc(ctx, (size_t)3 * nb)); MF_TRY(uplen.alloc(ctx, nb)); MF_TRY(dcount.alloc(ctx, nb));
    MF_TRY(ostart.alloc(ctx, (size_t)nb + 1)); MF_TRY(toff.alloc(ctx, (size_t)nb + 1)); MF_TRY(ps.alloc(ctx, 24));
    uint32_t *const d_cnt = cnt.p, *const d_occ = cnt.p + nb, *const d_cur = cnt.p + 2 * (size_t)nb;
    MF_HIP(hipMemcpyAsync(reg.p, h_reg.data(), (size_t)R * 4, hipMemcpyHostToDevice, st));
    MF_HIP(hipMemsetAsync(cnt.p, 0, (size_t)3 * nb * 4, st));
    MF_HIP(hipMemsetAsync(ps.p, 0, 24 * 8, st));
    // (a region's length is not known on the host: the grid covers the longest a region can be, 4 x the mean)
    const uint64_t mean_len = std::max<uint64_t>(1, n_occ / (uint64_t)nd1 / 4);          // (records >= occurrences / 20; 4 is generous)
    const unsigned gx = (unsigned)std::min<uint64_t>(65535, (4 * mean_len + SKM_PILOT_CH - 1) / SKM_PILOT_CH + 1);
    const int shift = SKM_DIGIT_BITS - pb;
    mf_ktimer tm(ctx, "k_skm_pilot");
    k_skm_pilot<<<dim3(gx, R), 256, (size_t)2 * nsel * 4, st>>>(bufA, pstart, plen, reg.p, shift, nsel, d_cnt, d_occ, nullptr, nullptr, nullptr);
    k_skm_pilot_dir<<<(nb + 255) / 256, 256, 0, st>>>(d_cnt, nb, uplen.p);
    MF_TRY(mf_scan<1>(ctx, uplen.p, ostart.p, nb, (uint64_t *)&ps.p[0]));
    MF_TRY(mf_scan<1>(ctx, d_occ, toff.p, nb, (uint64_t *)&ps.p[1]));
    unsigned long long tot[2] = {0, 0};
    MF_HIP(hipMemcpyAsync(tot, ps.p, 16, hipMemcpyDeviceToHost, st));
    MF_HIP(hipStreamSynchronize(st));
    if (!tot[0] || !tot[1]) return MF_OK;
    mf_buf<skm_rec> pr; mf_buf<uint64_t> tkeys; mf_buf<uint16_t> tcnt;
    if (pr.alloc(ctx, tot[0] + 4) != MF_OK || tkeys.alloc(ctx, tot[1]) != MF_OK || tcnt.alloc(ctx, tot[1]) != MF_OK) return MF_OK;      // (no room: no pilot)
    MF_HIP(hipMemsetAsync(pr.p, 0xFF, (tot[0] + 4) * sizeof(skm_rec), st));               // padding = sentinels
    k_skm_pilot<<<dim3(gx, R), 256, 0, st>>>(bufA, pstart, plen, reg.p, shift, nsel, nullptr, nullptr, ostart.p, d_cur, pr.p);
    {
        const unsigned grid = (unsigned)std::min<uint64_t>(nb, (uint64_t)ctx->n_cu * C2_WG_PER_CU);
        // ps: [2] overflow [3] distinct before the cut [4] units redone [8..24) diagnostics
        k_skm_count<K, false><<<grid, SKM_CT, C2_LDS, st>>>(pr.p, ostart.p, uplen.p, nb, toff.p, tkeys.p, tcnt.p, dcount.p, (unsigned int *)&ps.p[2], 0u, (uint64_t)0, kthr,
                                                         &ps.p[3], (unsigned int *)&ps.p[4], nullptr, &ps.p[8], (uint64_t)tot[1], (uint32_t)ctx->opt_skm_dedupe, nullptr);
    }
    MF_TRY(mf_scan<1>(ctx, dcount.p, ostart.p, nb, (uint64_t *)&ps.p[5]));                // [5] k-mers that survive the cut
    unsigned long long res[4] = {0, 0, 0, 0};
    MF_HIP(hipMemcpyAsync(res, &ps.p[2], 32, hipMemcpyDeviceToHost, st));
    MF_HIP(hipStreamSynchronize(st));
    MF_HIP(hipGetLastError());
    const bool over = (res[0] & 0xFFFFFFFFull) != 0;
    *rho = over ? 1.0 : (double)res[1] / (double)tot[1];              // (a unit beyond 64 passes at an eighth of the size: plan as fine as it gets)
    if (!over) *kept = (double)res[3] / (double)tot[1];
    *rec_per_occ = (double)tot[0] / (double)tot[1];                    // (padded to fours per would-be unit: a few per cent high)
    if (ctx->opt_verbose)
        fprintf(stderr, "[mf] skm pilot: %u would-be units of %u regions (2^%d per region): %llu records, %llu occurrences, %llu distinct k-mers = %.4f per occurrence (%llu kept), %llu unit(s) redone\n",
                nb, R, pb, tot[0], tot[1], res[1], *rho, res[3], res[2] & 0xFFFFFFFFull);
    return MF_OK;
}

// lv: digit bits per level (lv[0] = level 1).  Returns MF_OK and *out, or MF_SKM_FALLBACK (nothing allocated) when the
// input does not suit this path (a partition too rich for the LDS table, too many levels, not enough memory).
// what the slices of one run add up to: the dense table (grouped by partition, slice after slice = partition order), the
// partition offsets, the tallies
struct skm_acc {
    mf_buf<uint64_t> dk; mf_buf<uint16_t> dc; uint64_t dused = 0, dcap = 0;
    mf_buf<uint64_t> doff;                 // [np_total + 1]: the TABLE's partitions (two per counting partition, k_gather_split)
    mf_buf<unsigned long long> dhist, c2p;
    unsigned long long n_records = 0, cap_l1 = 0;
    uint32_t np_total = 0;
    int split_bits = 1;                    // table partitions per counting partition = 2^split_bits (k_gather_split / k_gather_split_n)
    int table_bits = 0;                    // the caller's wish for the table (0: one bit more than the counting plan)
};
#define MF_SKM_NOMEM 2             /* skm_slice: a record buffer did not fit -- the caller tries again with more slices */
// Level 1 of a run that is counted in slices, done ONCE for all the run's digits (round 3): round 2's slices each scanned the reads
// again for their own digits because two sliced record buffers were budgeted -- at 380 M reads k_skm_scatter was 412 ms of a
// 970 ms sample.  One full level-1 buffer + one slice-sized buffer for the following levels fit as well (57 GB of reads +
// 87 GB + 87 / S GB), and the reads are scanned once; the slices then differ only in which level-1 partitions they split and count.
struct skm_shared {
    uint32_t lo = 0, hi = 0;           // the run's level-1 digits
    bool ready = false;
    mf_buf<skm_rec> buf; mf_buf<uint64_t> pstart; mf_buf<uint32_t> plen, pocc;
    std::vector<uint64_t> h_pstart;    // (host copy: where the digit regions start, for the slices' buffer sizes)
    unsigned long long cap = 0;
    void clear() { buf.reset(); pstart.reset(); plen.reset(); pocc.reset(); h_pstart.clear(); ready = false; cap = 0; }
};
// One SLICE of the run: the records whose level-1 digit lies in [dlo, dhi) go through scatter, split and count.  A run is one
// slice unless the records do not fit next to the reads (300 M reads and more): then the reads are scanned once per slice
// and only a 1 / S share of the records exists at any time.  Partitions are numbered by digit, so slice after slice fills
// the dense table in the same order as a single pass would.
template <int K>
static int skm_slice(mf_ctx *ctx, const uint8_t *d_bases, uint64_t n_bases, const uint32_t *vmask, uint64_t n_words,
                     uint64_t n_occ, std::vector<int> &lv, unsigned long long *scal, int kthr, uint32_t dlo, uint32_t dhi,
                     uint32_t slice, uint32_t n_slices, skm_acc &A, skm_shared *SH = nullptr, bool last_of_shared = false, bool *plan_open = nullptr) {
    // *plan_open: the levels after the first are still to be planned (skm_pilot): lv[1..] may change in here, once
    hipStream_t st = ctx->stream;
    const int bits1 = lv[0], nd1 = 1 << bits1;
    // (SH: level 1 covers all of the run's digits and is done by the first slice; dlo / dhi then only say which regions this slice takes)
    const uint32_t l1lo = SH ? SH->lo : dlo, l1hi = SH ? SH->hi : dhi;
    int total_bits = 0; for (int b : lv) total_bits += b;
    int G = ctx->opt_l1_blocks > 0 ? (int)ctx->opt_l1_blocks : ctx->n_cu;
    {
        uint64_t maxG = (n_words + 1023) / 1024;
        if ((uint64_t)G > maxG) G = (int)maxG;
        if (G < 1) G = 1;
    }
    const uint64_t wpb = (n_words + G - 1) / G;
    const bool l1_only = lv.size() == 1;
    // the scatter's run redistribution needs LDS behind the staging lines: levels of up to 10 bits (option scatter_fast = 0: never)
    size_t fast_lds = ((skm_stage_bytes(nd1) + 15) & ~(size_t)15) + (size_t)16 * SKM_FAST_WAVE_BYTES;
    if (fast_lds > (size_t)160 * 1024 || ctx->opt_scatter_fast == 0) fast_lds = 0;
    uint32_t np = (uint32_t)nd1;
    mf_buf<uint64_t> pstart; MF_TRY(pstart.alloc(ctx, np));
    mf_buf<uint32_t> plen; MF_TRY(plen.alloc(ctx, np));
    mf_buf<uint32_t> pocc; MF_TRY(pocc.alloc(ctx, np));
    mf_buf<skm_rec> bufA;
    unsigned long long cap = 0;
    // Three or more levels ping-pong between record buffers: all of them get the size of the LAST level's (every level
    // adds padding), so the third level re-uses the first level's buffer instead of asking for a third one that is a few
    // megabytes too large for it (at 200 M reads, k = 21 a buffer is 125 GB: there is no room for three, and a second
    // sample would find the arena's idle regions just too small again).
    const bool open = plan_open && *plan_open;
    bool headroom = open;
    auto final_cap = [&](unsigned long long c1) {
        // (plan still open: room for the padding of whatever the pilot decides -- at most 2 x n_occ / 1100 partitions of 4 records each)
        if (headroom) return c1 + n_occ / 128 + (unsigned long long)nd1 * SKM_LINE * (1ull << MF_MAX_DIGIT_BITS);
        if (lv.size() < 3) return c1;
        unsigned long long c = c1, p = (unsigned long long)nd1;
        for (size_t li = 1; li < lv.size(); li++) { c += p * SKM_LINE * (1ull << lv[li]); p <<= lv[li]; }
        return c;
    };
    // Level 1 in ONE pass over the reads where it pays (large inputs, a split level follows): the digit regions are sized
    // from a histogram of a sixteenth of the input and handed out chunk-wise during the scatter, instead of a full
    // minimizer pass just to count (k_skm_hist over everything costs 17 ms of 300 at 100 M reads).  If a region turns out
    // too small the level is repeated with exact ranges.  skm_dyn: 0 never, 1 auto, 2 always (tests).
    bool dyn = !l1_only && (ctx->opt_skm_dyn == 2 || (ctx->opt_skm_dyn == 1 && n_words >= (1ull << 24)));
    for (int attempt = dyn ? 0 : 1; attempt < 2 && !(SH && SH->ready); attempt++) {
        const bool D = attempt == 0;
        const int stride = D ? (ctx->opt_skm_dyn == 2 ? 3 : 16) : 1;
        mf_buf<uint32_t> blockhist; MF_TRY(blockhist.alloc(ctx, (size_t)nd1 * G));
        mf_buf<uint32_t> blockocc; if (l1_only) MF_TRY(blockocc.alloc(ctx, (size_t)nd1 * G));
        {
            mf_ktimer t(ctx, D ? "k_skm_hist_sample" : "k_skm_hist");
            k_skm_hist<K><<<G, 1024, (size_t)nd1 * 8, st>>>(d_bases, n_bases, vmask, n_words, wpb, bits1, blockhist.p, blockocc.p, G, stride, l1lo, l1hi);
        }
        MF_DBG(ctx, "k_skm_hist");
        if (D) {
            mf_buf<uint32_t> rsize; MF_TRY(rsize.alloc(ctx, np));
            mf_buf<uint64_t> rstart; MF_TRY(rstart.alloc(ctx, (size_t)np + 1));
            mf_buf<unsigned long long> gcur; MF_TRY(gcur.alloc(ctx, np));
            k_skm_region_sizes<<<(nd1 + 255) / 256, 256, 0, st>>>(blockhist.p, G, nd1, (uint32_t)stride, rsize.p, l1lo, l1hi);
            MF_TRY(mf_scan<1>(ctx, rsize.p, rstart.p, np, (uint64_t *)&scal[1]));
            MF_HIP(hipMemcpyAsync(gcur.p, rstart.p, (size_t)np * 8, hipMemcpyDeviceToDevice, st));
            MF_HIP(hipMemsetAsync(&scal[5], 0, 8, st));
            MF_HIP(hipMemcpyAsync(&cap, &scal[1], 8, hipMemcpyDeviceToHost, st));
            MF_HIP(hipStreamSynchronize(st));
            if (bufA.alloc(ctx, std::max<unsigned long long>(cap + (uint64_t)G * SKM_CH, SH ? 0ull : final_cap(cap))) != MF_OK) return MF_SKM_NOMEM;
            skm_dyn Dy; Dy.gcur = gcur.p; Dy.rend = rstart.p + 1; Dy.overflow = (unsigned int *)&scal[5]; Dy.dump = cap; Dy.save = nullptr;
            {
                mf_ktimer t(ctx, "k_skm_scatter");
                if (fast_lds) {
                    MF_TRY(skm_set_lds(k_skm_scatter<K, true, true>, fast_lds));
                    k_skm_scatter<K, true, true><<<G, 1024, fast_lds, st>>>(d_bases, n_bases, vmask, n_words, wpb, bits1, nullptr, G, bufA.p, Dy, l1lo, l1hi);
                } else {
                    const size_t lds = skm_stage_bytes(nd1);
                    MF_TRY(skm_set_lds(k_skm_scatter<K, true, false>, lds));
                    k_skm_scatter<K, true, false><<<G, 1024, lds, st>>>(d_bases, n_bases, vmask, n_words, wpb, bits1, nullptr, G, bufA.p, Dy, l1lo, l1hi);
                }
            }
            MF_DBG(ctx, "k_skm_scatter");
            k_skm_dir_dyn<<<(nd1 + 255) / 256, 256, 0, st>>>(rstart.p, gcur.p, nd1, pstart.p, plen.p);
            unsigned long long ovf = 0;
            MF_HIP(hipMemcpyAsync(&ovf, &scal[5], 8, hipMemcpyDeviceToHost, st));
            MF_HIP(hipStreamSynchronize(st));
            if (!ovf) break;
            if (ctx->opt_verbose) fprintf(stderr, "[mf] skm: a sampled digit region was too small, level 1 again with exact ranges\n");
            bufA.reset();
            continue;
        }
        mf_buf<uint64_t> blockstart; MF_TRY(blockstart.alloc(ctx, (size_t)nd1 * G + 1));
        MF_TRY(mf_scan<SKM_LINE>(ctx, blockhist.p, blockstart.p, (uint64_t)nd1 * G, (uint64_t *)&scal[1]));
        MF_HIP(hipMemcpyAsync(&cap, &scal[1], 8, hipMemcpyDeviceToHost, st));
        MF_HIP(hipStreamSynchronize(st));                       // padded number of records
        if (bufA.alloc(ctx, SH ? cap : final_cap(cap)) != MF_OK) return MF_SKM_NOMEM;
        {
            mf_ktimer t(ctx, "k_skm_scatter");
            if (fast_lds) {
                MF_TRY(skm_set_lds(k_skm_scatter<K, false, true>, fast_lds));
                k_skm_scatter<K, false, true><<<G, 1024, fast_lds, st>>>(d_bases, n_bases, vmask, n_words, wpb, bits1, blockstart.p, G, bufA.p, skm_dyn(), l1lo, l1hi);
            } else {
                const size_t lds = skm_stage_bytes(nd1);
                MF_TRY(skm_set_lds(k_skm_scatter<K, false, false>, lds));
                k_skm_scatter<K, false, false><<<G, 1024, lds, st>>>(d_bases, n_bases, vmask, n_words, wpb, bits1, blockstart.p, G, bufA.p, skm_dyn(), l1lo, l1hi);
            }
        }
        MF_DBG(ctx, "k_skm_scatter");
        k_skm_dir<<<(nd1 + 255) / 256, 256, 0, st>>>(blockstart.p, blockocc.p, G, nd1, pstart.p, plen.p, pocc.p);
    }

    uint64_t rebase = 0;
    if (SH) {
        if (!SH->ready) {
            // the first slice has made level 1 for all of them
            SH->buf.swap(bufA); SH->pstart.swap(pstart); SH->plen.swap(plen); SH->pocc.swap(pocc);
            SH->cap = cap;
            SH->h_pstart.resize(nd1);
            MF_HIP(hipMemcpyAsync(SH->h_pstart.data(), SH->pstart.p, (size_t)nd1 * 8, hipMemcpyDeviceToHost, st));
            MF_HIP(hipStreamSynchronize(st));
            SH->ready = true;
            if (l1lo == 0 && l1hi == (uint32_t)nd1 && n_occ) { ctx->last_l1_per_occ = (double)SH->cap / (double)n_occ; ctx->last_l1_k = K; }
        }
        // this slice's view: the shared buffer, its own copy of the directory (the levels swap it away), and as many records
        // as its digit regions span
        bufA.borrow(ctx, SH->buf.p, SH->buf.n);
        MF_TRY(pstart.alloc(ctx, np)); MF_TRY(plen.alloc(ctx, np)); MF_TRY(pocc.alloc(ctx, np));
        MF_HIP(hipMemcpyAsync(pstart.p, SH->pstart.p, (size_t)nd1 * 8, hipMemcpyDeviceToDevice, st));
        MF_HIP(hipMemcpyAsync(plen.p, SH->plen.p, (size_t)nd1 * 4, hipMemcpyDeviceToDevice, st));
        rebase = SH->h_pstart[dlo];
        cap = (dhi < (uint32_t)nd1 ? SH->h_pstart[dhi] : SH->cap) - rebase;
    }
    bool split_from_pilot = false;
    if (open) {
        // ---- the pilot: distinct k-mers per occurrence -> the levels after the first
        *plan_open = false;
        double rho = -1.0, kept = -1.0, rpo = 0.0;
        MF_TRY(skm_pilot<K>(ctx, bufA.p, pstart.p, plen.p, dlo, dhi, n_occ, nd1, total_bits - bits1, kthr, &rho, &kept, &rpo));
        ctx->last_pilot_rho = rho; ctx->n_pilots++;
        if (rho > 0.0) {
            double want_units = (double)n_occ * rho / (double)std::max<int64_t>(64, ctx->opt_skm_unit_distinct);
            // ... and a unit's RECORDS should fit the search for identical ones (its first C2_DD * SKM_CT = 2048 records go through the
            // table as records, the rest is inserted as it comes): 380 M reads at 315-fold depth planned by distinct k-mers alone had
            // 3170 records per unit and lost 6 % of k_skm_count to the records the search did not see
            // (round 5: a record of short k-mers holds few of them -- 4.8 per record at k = 21, 8.4 at k = 31 -- and 2000 such records are a small unit:
            // more than half of k_skm_count is per-unit overhead (profiles/r05k_count_phase_cycles.txt).  50 M reads, k_skm_count / the whole step in
            // ms with 2000 and with 4000 records: k = 21 20.2 -> 16.5 / 129.9 -> 125.7, k = 23 19.8 -> 15.7 / 118.0 -> 114.2, 200 M reads at k = 21
            // 87.1 -> 70.0 / 375.9 -> 358.7; k = 25, 27, 31: no difference -- the distinct k-mers bound those.  profiles/r05t_unit_records_sweep.txt)
            const int64_t unit_records = ctx->opt_skm_unit_records > 0 ? ctx->opt_skm_unit_records : (K >= 25 ? 2000 : 4000) * SKM_CT / 512;
            if (ctx->opt_skm_dedupe) want_units = std::max(want_units, (double)n_occ * rpo / (double)unit_records);
            int Bc = 0; while (Bc < 30 && (double)(1ull << Bc) < want_units) Bc++;
            const int r = std::max(1, std::min(Bc - bits1, std::min((int)SKM_DIGIT_BITS, 30 - bits1)));
            if (ctx->opt_verbose) fprintf(stderr, "[mf] skm pilot: %.4f distinct k-mers per occurrence -> %d bits after level 1 (planned from the occurrences alone: %d)\n", rho, r, total_bits - bits1);
            lv.resize(1);
            if (r <= MF_MAX_DIGIT_BITS) lv.push_back(r); else { lv.push_back((r + 1) / 2); lv.push_back(r / 2); }
            total_bits = 0; for (int b : lv) total_bits += b;
            if (kthr >= 0 && kept >= 0.0 && !A.doff.p) {
                // the table holds the k-mers that survive the cut: as many table partitions per counting unit as keep a partition
                // within the graph kernels' LDS lookup table (mf_nbr.h: 352 keys; 100 M reads of the benchmark: 343 per unit -> 2 x 172)
                const double per_unit = (double)n_occ * kept / (double)(1ull << total_bits);
                int sb = 0; while (sb < 4 && per_unit / (double)(1 << sb) > (double)ctx->opt_part_good) sb++;
                A.split_bits = sb; split_from_pilot = true;
                if (ctx->opt_verbose) fprintf(stderr, "[mf] skm pilot: %.4f k-mers per occurrence survive the cut: %.0f per counting unit -> %d table partition(s) per unit\n", kept, per_unit, 1 << sb);
            }
        }
    }
    if (!A.doff.p) {                                      // (the first slice of the run, now that the plan stands)
        if (!split_from_pilot) A.split_bits = A.table_bits > 0 ? std::max(0, std::min(4, A.table_bits - total_bits)) : 1;
        if (total_bits + A.split_bits > 30) A.split_bits = 30 - total_bits;
        A.np_total = (uint32_t)(1ull << (total_bits + A.split_bits));      // partitions of the table: 2^split_bits per counting partition
        MF_TRY(A.doff.alloc(ctx, (size_t)A.np_total + 1));
    }
    MF_HIP(hipMemsetAsync(&scal[6], 0, 8, st));           // [6] records without padding ([7] distinct k-mers before the cut: whole run)
    headroom = false;
    const unsigned long long cap_l1 = cap, cap_last = final_cap(cap);
    if (dlo != 0 || dhi != (uint32_t)nd1) {
        // the slice's digits only: the directory of the following levels starts at digit dlo
        np = dhi - dlo;
        mf_buf<uint64_t> ps; MF_TRY(ps.alloc(ctx, np)); mf_buf<uint32_t> pl, po; MF_TRY(pl.alloc(ctx, np)); MF_TRY(po.alloc(ctx, np));
        MF_HIP(hipMemcpyAsync(ps.p, pstart.p + dlo, (size_t)np * 8, hipMemcpyDeviceToDevice, st));
        MF_HIP(hipMemcpyAsync(pl.p, plen.p + dlo, (size_t)np * 4, hipMemcpyDeviceToDevice, st));
        MF_HIP(hipMemcpyAsync(po.p, pocc.p + dlo, (size_t)np * 4, hipMemcpyDeviceToDevice, st));
        pstart.swap(ps); plen.swap(pl); pocc.swap(po);
    }
    int used = 0;
    mf_buf<skm_rec> spare;                                // the buffer a level has read from: the next level writes into it
    for (size_t li = 1; li < lv.size(); li++) {
        const int bits = lv[li], nd = 1 << bits;
        const uint64_t cap2 = cap + (uint64_t)np * SKM_LINE * nd;
        const uint64_t np2 = (uint64_t)np * nd;
        if (np2 > 0xFFFFFFF0ull) return mf_set_error("too many partitions");
        const bool last = li + 1 == lv.size();
        mf_buf<skm_rec> bufB;
        if (spare.p && spare.n >= cap2) bufB.swap(spare);
        else if (bufB.alloc(ctx, std::max<unsigned long long>(cap2, cap_last)) != MF_OK) {
            if (ctx->opt_verbose) fprintf(stderr, "[mf] skm: no room for %.1f GB of level-%zu records (arena %.1f GB): %s\n", cap2 * 16 / 1e9, li + 1, ctx->arena_bytes / 1e9, mf_last_error());
            return MF_SKM_NOMEM;
        }
        mf_buf<uint64_t> ostart; MF_TRY(ostart.alloc(ctx, np2));
        mf_buf<uint32_t> olen; MF_TRY(olen.alloc(ctx, np2));
        mf_buf<uint32_t> oocc; if (last) MF_TRY(oocc.alloc(ctx, np2));
        const size_t lds = skm_stage_bytes(nd);
        const unsigned grid = (unsigned)std::min<uint64_t>(np, (uint64_t)ctx->n_cu * (lds > 72 * 1024 ? 1 : 2));
        {
            MF_TRY(skm_set_lds(k_skm_split, lds));
            mf_ktimer t(ctx, "k_skm_split");
            k_skm_split<<<grid, 1024, lds, st>>>(bufA.p, pstart.p, plen.p, np, SKM_DIGIT_BITS - used - bits, bits, bufB.p, ostart.p, olen.p, oocc.p,
                                                 li == 1 ? &scal[6] : nullptr, li == 1 ? rebase : 0);
        }
        MF_DBG(ctx, "k_skm_split");
        if (ctx->opt_verbose >= 2 && last) {
            // What a ONE-pass split would have to cope with ("measure, do not argue"): its sub-partitions would be sized from a
            // SAMPLE of the records (every 16th tile, as level 1 is) plus slack, and a record that finds its sub-partition full goes to an overflow
            // list.  From the exact sizes at hand: the records that overflow, in expectation over the sample (n' ~ Binomial(n, 1/16) as a normal
            // deviate, capacity = 16 n' (1 + slack) + 64), at 6 / 12 / 25 / 50 % slack.
            std::vector<uint32_t> h(np2);
            MF_HIP(hipMemcpyAsync(h.data(), olen.p, np2 * 4, hipMemcpyDeviceToHost, st));
            MF_HIP(hipStreamSynchronize(st));
            double tot = 0, tot2 = 0, mx = 0;
            for (uint32_t v : h) { tot += v; tot2 += (double)v * v; mx = std::max(mx, (double)v); }
            const double mean = tot / (double)np2, sd = std::sqrt(std::max(0.0, tot2 / (double)np2 - mean * mean));
            const double slack[4] = {0.06, 0.12, 0.25, 0.50};
            double over[4] = {0, 0, 0, 0}, parts[4] = {0, 0, 0, 0}, room[4] = {0, 0, 0, 0};
            for (uint32_t v : h) {
                const double n = v, mu = n / 16.0, sg = std::sqrt(n / 16.0 * 15.0 / 16.0);
                for (int q = 0; q < 4; q++) {
                    // overflow = n - cap(n') where cap = 16 n' (1 + s) + 64; integrate over n' = mu + z sg on a grid of z
                    double e = 0, pr = 0, rm = 0;
                    for (int zi = -40; zi <= 40; zi++) {
                        const double z = zi * 0.1, w = std::exp(-0.5 * z * z) * 0.1 / 2.5066282746;
                        const double cap = std::max(0.0, 16.0 * (mu + z * sg) * (1.0 + slack[q]) + 64.0);
                        if (n > cap) { e += w * (n - cap); pr += w; }
                        rm += w * cap;
                    }
                    over[q] += e; parts[q] += pr; room[q] += rm;
                }
            }
            fprintf(stderr, "[mf] skm split, one-pass what-if: %llu sub-partitions of %.0f records on average (sd %.0f, largest %.0f); sized from a 1/16 sample + slack:\n", (unsigned long long)np2, mean, sd, mx);
            for (int q = 0; q < 4; q++)
                fprintf(stderr, "[mf]     slack %2.0f %%: %.3f %% of the records overflow (%.0f records in %.0f sub-partitions), the regions take %.2f x the records\n", slack[q] * 100.0,
                        100.0 * over[q] / std::max(tot, 1.0), over[q], parts[q], room[q] / std::max(tot, 1.0));
        }
        bufA.swap(bufB);
        if (!last && bufB.owned) { spare.reset(); spare.swap(bufB); }        // (a borrowed level-1 buffer is the other slices' input too ...
        else if (!last && SH && last_of_shared && bufB.p == SH->buf.p) {    // ... unless this is the run's last slice: its third level writes there)
            bufB.reset(); spare.reset(); spare.swap(SH->buf); SH->ready = false;
        }
        pstart.swap(ostart); plen.swap(olen);
        if (last) pocc.swap(oocc);
        cap = cap2; np = (uint32_t)np2; used += bits;
    }

    spare.reset();
    // ---- count + gather, a batch of partitions at a time.  A partition's (key,count) slice is sized by its k-mer count
    // (it cannot hold more distinct k-mers than it has k-mers, nor more than the LDS table), i.e. all slices together
    // are as large as the k-mer stream itself: the buffer holds ONE batch of slices and is redused, the dense table grows
    // batch by batch (sized from the first batch's distinct / k-mer ratio; re-allocated if that was too optimistic).
    mf_buf<uint32_t> pcap; MF_TRY(pcap.alloc(ctx, np));
    mf_buf<uint64_t> toff; MF_TRY(toff.alloc(ctx, (size_t)np + 1));
    k_skm_cap<<<(np + 255) / 256, 256, 0, st>>>(pocc.p, np, pcap.p, kthr >= 0 ? (uint32_t)kthr + 1u : 1u);
    MF_TRY(mf_scan<1>(ctx, pcap.p, toff.p, np, (uint64_t *)&scal[4]));
    // As few batches as the memory allows (every batch is a launch, a scan, a gather and a round trip to the host: 8 -> 2
    // batches took 4 ms off a 200 ms step): the temporary lists of one batch may take a fifth of the device.
    uint32_t nbatch = 1;
    if (ctx->opt_skm_batches > 0) nbatch = (uint32_t)std::min<int64_t>(ctx->opt_skm_batches, np);
    else {
        unsigned long long tall = 0;
        MF_HIP(hipMemcpyAsync(&tall, &scal[4], 8, hipMemcpyDeviceToHost, st));
        MF_HIP(hipStreamSynchronize(st));
        size_t fr = 0, tot = 0;
        MF_HIP(hipMemGetInfo(&fr, &tot));
        double budget = (ctx->opt_arena_cap_gb > 0 ? (double)ctx->opt_arena_cap_gb * 1e9 : (double)tot) * 0.22;
        if (ctx->opt_arena_cap_gb <= 0) budget = std::min(budget, ((double)fr + (double)mf_arena_idle(ctx)) * 0.5);      // ... and half of what is free right now
        // (a small sample -- the drop-in's 20 M reads -- in one batch asked for 12 GB of lists beside 10 GB of records: memory a short process
        // pays 35 ms per GiB for when the driver has to clear it first; a batch more costs 0.7 ms)
        if (n_occ < (1ull << 32)) budget = std::min(budget, 3e9);
        while (nbatch < 64 && nbatch * 2 <= np && (double)tall * 10.0 / nbatch > budget) nbatch *= 2;
    }
    const uint32_t PB = (np + nbatch - 1) / nbatch;
    std::vector<unsigned long long> tb(nbatch + 1);
    for (uint32_t b = 0; b <= nbatch; b++)
        MF_HIP(hipMemcpyAsync(&tb[b], &toff.p[std::min<uint64_t>((uint64_t)b * PB, np)], 8, hipMemcpyDeviceToHost, st));
    MF_HIP(hipStreamSynchronize(st));
    pcap.reset(); pocc.reset();
    const unsigned long long tcap = tb[nbatch];
    unsigned long long tmax = 0;
    for (uint32_t b = 0; b < nbatch; b++) tmax = std::max(tmax, tb[b + 1] - tb[b]);
    mf_buf<uint64_t> tkeys; mf_buf<uint16_t> tcnt;
    if (tkeys.alloc(ctx, tmax) != MF_OK || tcnt.alloc(ctx, tmax) != MF_OK) {
        if (ctx->opt_verbose) fprintf(stderr, "[mf] skm: no room for %.1f GB of temporary lists\n", tmax * 10 / 1e9);
        return MF_SKM_NOMEM;
    }
    mf_buf<uint32_t> dcount; MF_TRY(dcount.alloc(ctx, np));
    mf_buf<uint64_t> coff; MF_TRY(coff.alloc(ctx, (size_t)np + 1));      // offsets of the counting partitions inside their batch
    // this slice's partitions in the numbering of the whole run (A.np_total, doffp: the table's partitions, two per counting partition)
    const uint32_t pbase = (uint32_t)(((uint64_t)dlo * A.np_total) >> bits1);
    uint64_t *const doffp = A.doff.p + pbase;
    mf_buf<uint64_t> &dk = A.dk; mf_buf<uint16_t> &dc = A.dc;
    uint64_t &dused = A.dused, &dcap = A.dcap;
    mf_buf<unsigned long long> &dhist = A.dhist, &c2p = A.c2p;
    const bool c2prof = K == 31 && (ctx->opt_ablate & 32);
    for (uint32_t b = 0; b < nbatch; b++) {
        const uint32_t p0 = (uint32_t)std::min<uint64_t>((uint64_t)b * PB, np), p1 = (uint32_t)std::min<uint64_t>((uint64_t)(b + 1) * PB, np);
        if (p0 == p1) continue;
        {
            const unsigned grid = (unsigned)std::min<uint64_t>(p1 - p0, (uint64_t)ctx->n_cu * C2_WG_PER_CU);      // resident workgroups
            MF_HIP(hipMemsetAsync(&scal[8], 0, 16, st));                     // ([8] units redone, [9] the unit counter)
            mf_ktimer t(ctx, "k_skm_count");
#define SKM_COUNT_ARGS bufA.p, pstart.p, plen.p, p1, toff.p, tkeys.p, tcnt.p, dcount.p, (unsigned int *)&scal[2], p0, (uint64_t)tb[b], kthr, &scal[7], \
                       (unsigned int *)&scal[8], dhist.p, c2p.p, (uint64_t)tmax, (uint32_t)ctx->opt_skm_dedupe, (ctx->opt_skm_dynq ? (unsigned int *)&scal[9] : nullptr)
            if (c2prof) k_skm_count<(K == 31 ? 31 : 20), true><<<grid, SKM_CT, C2_LDS, st>>>(SKM_COUNT_ARGS);
            else k_skm_count<K, false><<<grid, SKM_CT, C2_LDS, st>>>(SKM_COUNT_ARGS);
#undef SKM_COUNT_ARGS
        }
        MF_DBG(ctx, "k_skm_count");
        MF_TRY(mf_scan<1>(ctx, dcount.p + p0, coff.p + p0, p1 - p0, (uint64_t *)&scal[3]));      // offsets inside the batch
        unsigned long long res[2];
        MF_HIP(hipMemcpyAsync(res, &scal[2], 16, hipMemcpyDeviceToHost, st));
        MF_HIP(hipStreamSynchronize(st));
        if (ctx->opt_verbose) {
            unsigned long long nr = 0;
            MF_HIP(hipMemcpy(&nr, &scal[8], 8, hipMemcpyDeviceToHost));
            if (nr & 0xFFFFFFFFull) fprintf(stderr, "[mf] skm: batch %u: %llu partition(s) counted in several passes\n", b, nr & 0xFFFFFFFFull);
        }
        if (ctx->opt_verbose) {
            unsigned long long bad = 0;
            MF_HIP(hipMemcpy(&bad, &c2p.p[7], 8, hipMemcpyDeviceToHost));
            if (bad && !c2prof) fprintf(stderr, "[mf] skm: batch %u: %llu counter(s) found non-zero under an empty key so far\n", b, bad);
        }
        if (res[0]) {
            if (ctx->opt_verbose) fprintf(stderr, "[mf] skm: a minimizer partition overflows the LDS table (code %llu), using the k-mer path\n", res[0]);
            if (ctx->opt_verbose && res[0] == 2) {
                unsigned long long h[16];
                MF_HIP(hipMemcpy(h, c2p.p, 128, hipMemcpyDeviceToHost));
                fprintf(stderr, "[mf]   p=%llu o=%llu room=%llu wb=%llu nkeep=%llu tcap=%llu p0=%llu np=%llu (batch %u: tb=%llu..%llu)\n", h[8], h[9], h[10], h[11], h[12], h[13], h[14], h[15], b, tb[b], tb[b + 1]);
            }
            MF_HIP(hipMemsetAsync(&scal[2], 0, 8, st));
            return MF_SKM_FALLBACK;
        }
        const uint64_t d_b = res[1];
        if (dused + d_b > dcap) {
            // everything still to come is bounded by the slices of the remaining batches; expect the ratio seen so far
            // (+ the slices still to come, taken to be like this one)
            const uint64_t done_cap = tb[b + 1], rest_cap = tcap - tb[b + 1] + (uint64_t)(n_slices - 1 - slice) * tcap;
            const double ratio = done_cap ? (double)(dused + d_b) / (double)done_cap : 1.0;
            uint64_t want = dused + d_b + (uint64_t)std::min<double>((double)rest_cap, (double)rest_cap * ratio * 1.12 + 4096.0);
            if (want < dused + d_b) want = dused + d_b;
            mf_buf<uint64_t> nk; mf_buf<uint16_t> nc;
            if (nk.alloc(ctx, want) != MF_OK || nc.alloc(ctx, want) != MF_OK) return MF_SKM_NOMEM;
            if (dused) {
                MF_HIP(hipMemcpyAsync(nk.p, dk.p, dused * 8, hipMemcpyDeviceToDevice, st));
                MF_HIP(hipMemcpyAsync(nc.p, dc.p, dused * 2, hipMemcpyDeviceToDevice, st));
                MF_HIP(hipStreamSynchronize(st));
            }
            dk.swap(nk); dc.swap(nc);
            dcap = want;
        }
        {
            const unsigned grid = (unsigned)std::min<uint64_t>(((uint64_t)(p1 - p0) + 3) / 4, (uint64_t)ctx->n_cu * 32);
            mf_ktimer t(ctx, "k_gather");
            if (A.split_bits == 1)
                k_gather_split<K><<<grid, 256, 0, st>>>(tkeys.p, tcnt.p, toff.p, dcount.p, coff.p, p1, dk.p + dused, dc.p + dused, p0, (uint64_t)tb[b],
                                                        doffp, dused, 31 - total_bits);
            else
                k_gather_split_n<K><<<grid, 256, 0, st>>>(tkeys.p, tcnt.p, toff.p, dcount.p, coff.p, p1, dk.p + dused, dc.p + dused, p0, (uint64_t)tb[b],
                                                          doffp, dused, std::min(31, 32 - total_bits - A.split_bits), A.split_bits);
        }
        MF_DBG(ctx, "k_gather");
        dused += d_b;
    }
    MF_HIP(hipGetLastError());
    bufA.reset();
    {
        unsigned long long nv = 0;
        MF_HIP(hipMemcpyAsync(&nv, &scal[6], 8, hipMemcpyDeviceToHost, st));
        MF_HIP(hipStreamSynchronize(st));
        A.n_records += nv ? nv : cap_l1;                  // (plans without a split level: the padded level-1 count)
    }
    if (ctx->opt_verbose)
        fprintf(stderr, "[mf] count(skm): slice %u/%u digits [%u,%u): records=%llu levels=%zu bits=%d np=%u distinct so far=%llu\n", slice + 1, n_slices, dlo, dhi,
                (unsigned long long)cap, lv.size(), total_bits, np, (unsigned long long)dused);
    return MF_OK;
}

// lv: digit bits per level (lv[0] = level 1).  Returns MF_OK and *out, or MF_SKM_FALLBACK (nothing allocated) when the
// input does not suit this path (a partition too rich for the LDS table, too many levels, not enough memory).
template <int K>
static int skm_run(mf_ctx *ctx, const uint8_t *d_bases, uint64_t n_bases, const uint32_t *vmask, uint64_t n_words,
                   uint64_t n_occ, const std::vector<int> &lv0, bool adaptive, int table_bits, unsigned long long *scal, int thr, uint64_t *n_all, mf_table **out,
                   skm_shared *pre = nullptr) {
    // pre: level 1 stands already (a STREAMED one, skm_stream_* below: the reads are gone, d_bases == nullptr) -- whatever would scan them again
    // ends the run with MF_SKM_FALLBACK and the caller starts over from the files
    hipStream_t st = ctx->stream;
    std::vector<int> lv = lv0;                 // (the levels after the first may change once the pilot has measured the reads' depth)
    const int bits1 = lv[0], nd1 = 1 << bits1;
    int total_bits = 0; for (int b : lv) total_bits += b;
    if (total_bits > 30 || total_bits - bits1 > SKM_DIGIT_BITS) return MF_SKM_FALLBACK;      // (the table gets total_bits + 1 partition bits)
    // counts of the entries the cut drops (the .stat.txt histogram needs them): drop_hist[c], c <= thr.  The kernel tallies
    // them in a small LDS histogram: a cut above C2_LH - 1 is made afterwards (mf_count_skm)
    const int kthr = thr < C2_LH ? thr : -1;
    MF_TRY(skm_set_lds(k_skm_count<K, false>, C2_LDS));
    if (K == 31) MF_TRY(skm_set_lds(k_skm_count<(K == 31 ? 31 : 20), true>, C2_LDS));
    // Slices (HBM budget): the records of a run are about a third of the reads' bytes per radix level, two levels ping-pong.
    // Reads, table and index have to fit beside them.  Option skm_slices forces a number (tests); arena_cap_gb stands in for
    // a smaller device.
    uint32_t S = 1;
    // shared: the slices share ONE level 1 over all digits (skm_shared) instead of scanning the reads once each.  It needs
    // the whole level-1 buffer + the slice-sized buffers of the following levels beside the growing table, which is tried
    // when what is free right now says so (option skm_shared: 0 never, 1 auto, 2 whenever there are slices); if it then
    // does not fit after all, the level-1 buffer is kept for a second try with twice the slices before the run goes back
    // to slices that scan for themselves.
    bool shared = false;
    if (lv.size() >= 2) {
        size_t fr = 0, tot = 0;
        MF_HIP(hipMemGetInfo(&fr, &tot));
        if (ctx->opt_skm_slices > 0) S = (uint32_t)ctx->opt_skm_slices;
        else {
            double budget = (ctx->opt_arena_cap_gb > 0 ? (double)ctx->opt_arena_cap_gb * 1e9 : (double)tot * 0.92) - (double)n_bases * 1.15 - 8e9;
            const double table = (double)n_occ * 0.14 * 26.0;       // (distinct k-mers that survive ~ a seventh of the occurrences) x (entry + index)
            budget -= std::min(table, budget * 0.5);
            const double recs = (double)n_occ / 5.5 * 16.0 * 2.3;    // two buffers + slices' padding
            while (S < 64 && recs / S > budget) S *= 2;
        }
        while (S > 1 && (uint32_t)nd1 / S < 4) S /= 2;
        if (S > 1 && ctx->opt_skm_shared == 2) shared = true;
        else if (S > 1 && ctx->opt_skm_shared == 1 && ctx->opt_arena_cap_gb <= 0) {
            // free now (the reads are already resident) - the table (the k-mers that survive the cut; it is sized from the first
            // batch's ratio, so it rarely grows twice) - room for a batch of temporary lists (they adapt: more batches)
            const double avail = ((double)fr + (double)mf_arena_idle(ctx)) * 0.96 - (double)n_occ * (kthr >= 1 ? 0.04 : 0.10) * 10.0 * 1.3 - (double)tot * 0.05;
            // records: a minimizer run is (K - M + 2) / 2 k-mers long where reads and the record format do not cut it shorter
            const double run = std::min<double>((K - mf_skm_m(K) + 2) / 2.0, (double)skm_word<K>::RMAX) * 0.78;
            const double full = (double)n_occ / run * 16.0 * 1.10;
            const double nb = lv.size() >= 3 ? 2.0 : 1.0;
            auto need = [&](uint32_t s2) { return full * (1.0 + nb * 1.15 / s2); };
            uint32_t S2 = ctx->opt_skm_slices > 0 ? S : 2;                   // (a forced number of slices stays)
            // (beyond 16 slices a split launch has too few partitions to fill the device: then the slices scan for themselves)
            while (ctx->opt_skm_slices <= 0 && S2 < 16 && need(S2) > avail) S2 *= 2;
            while (S2 > 1 && (uint32_t)nd1 / S2 < 4) S2 /= 2;
            if (S2 > 1 && need(S2) <= avail) { shared = true; S = S2; }
            if (ctx->opt_verbose)
                fprintf(stderr, "[mf] skm: free %.1f GB + idle %.1f GB -> %.1f GB for records; one level 1 = %.1f GB + %.1f GB per slice at %u slices: %s\n", fr / 1e9,
                        mf_arena_idle(ctx) / 1e9, avail / 1e9, full / 1e9, (need(S2) - full) / 1e9, S2, shared ? "shared" : "every slice scans");
        }
    }
    // a shard of the table only (mf_count_device_shard): the owner's digits, sliced like a whole run
    const uint32_t W = (uint32_t)std::max(ctx->own_world, 1);
    if (W > 1 && (uint32_t)nd1 < W) return mf_set_error("count (shard): %d level-1 digits for %u ranks", nd1, W);
    const uint32_t own_lo = (uint32_t)((uint64_t)nd1 * (uint32_t)ctx->own_rank / W), own_hi = (uint32_t)((uint64_t)nd1 * ((uint32_t)ctx->own_rank + 1) / W);
    while (S > 1 && (own_hi - own_lo) / S < 1) S /= 2;
    if (S == 1) shared = false;
    skm_shared SH; SH.lo = own_lo; SH.hi = own_hi;
    if (pre) {
        if (W != 1 || lv.size() < 2 || !pre->ready) return MF_SKM_FALLBACK;
        shared = true;
        SH.buf.swap(pre->buf); SH.pstart.swap(pre->pstart); SH.plen.swap(pre->plen); SH.pocc.swap(pre->pocc); SH.h_pstart.swap(pre->h_pstart);
        SH.cap = pre->cap; SH.ready = true; pre->ready = false;
    }
    const uint32_t S_own = S;
    int shared_tries = 0;
    // ---- the PILOT (round 4, skm_pilot below): the plan the caller made sizes the counting units by OCCURRENCES; what a unit costs
    // is decided by its DISTINCT k-mers.  The first slice measures them once its level 1 stands and re-plans the later levels.
    bool plan_open = adaptive && ctx->opt_skm_pilot != 0 && lv.size() >= 2 && W == 1;
    for (;;) {
        if (pre && !(shared && SH.ready)) return MF_SKM_FALLBACK;
        skm_acc A;                                              // (A.np_total, A.doff: set by the first slice, once the plan stands)
        A.table_bits = table_bits;
        if (kthr >= 0) { MF_TRY(A.dhist.alloc(ctx, (size_t)MF_MAX_COUNT + 1)); MF_HIP(hipMemsetAsync(A.dhist.p, 0, A.dhist.bytes(), st)); }
        MF_TRY(A.c2p.alloc(ctx, 16)); MF_HIP(hipMemsetAsync(A.c2p.p, 0, 128, st));     // [0,8) phase cycles (profiling build), [8,16) overflow diagnostics
        MF_HIP(hipMemsetAsync(&scal[7], 0, 8, st));
        int rc = MF_OK;
        for (uint32_t sl = 0; sl < S && rc == MF_OK; sl++)
            rc = skm_slice<K>(ctx, d_bases, n_bases, vmask, n_words, n_occ, lv, scal, kthr, own_lo + (uint32_t)((uint64_t)(own_hi - own_lo) * sl / S),
                              own_lo + (uint32_t)((uint64_t)(own_hi - own_lo) * (sl + 1) / S), sl, S, A, shared ? &SH : nullptr, sl + 1 == S, &plan_open);
        total_bits = 0; for (int b : lv) total_bits += b;
        if (rc == MF_OK && W > 1) {
            // the other ranks' partitions are empty here: offsets 0 before the owner's range, the table's size after it
            const uint32_t pb0 = (uint32_t)(((uint64_t)own_lo * A.np_total) >> bits1), pb1 = (uint32_t)(((uint64_t)own_hi * A.np_total) >> bits1);
            if (pb0) MF_HIP(hipMemsetAsync(A.doff.p, 0, (size_t)pb0 * 8, st));
            if (A.np_total > pb1) {
                MF_HIP(hipMemsetAsync(A.doff.p + pb1 + 1, 0, (size_t)(A.np_total - pb1) * 8, st));
                k_skm_add_base<<<(A.np_total - pb1 + 255) / 256, 256, 0, st>>>(A.doff.p + pb1 + 1, (uint64_t)(A.np_total - pb1), A.dused);
            }
            if (!A.dused) MF_HIP(hipMemsetAsync(A.doff.p, 0, ((size_t)A.np_total + 1) * 8, st));
        }
        if (rc == MF_SKM_NOMEM) {
            ctx->n_slice_restarts++;                             // (what was counted so far is thrown away: mf_ctx_stat "slice_restarts", bench.py reports it)
            const bool can_double = lv.size() >= 2 && S < 64 && (own_hi - own_lo) / (2 * S) >= 1 && (W > 1 || (uint32_t)nd1 / (2 * S) >= 4);
            if (shared && SH.ready && can_double && shared_tries++ < 2) {
                if (ctx->opt_verbose) fprintf(stderr, "[mf] skm: %u slice(s) behind a shared level 1 do not fit, trying %u\n", S, 2 * S);
                S *= 2;
                continue;
            }
            if (shared) {
                if (ctx->opt_verbose) fprintf(stderr, "[mf] skm: a shared level 1 does not fit, every slice scans for itself\n");
                shared = false; SH.clear(); S = S_own;
                continue;
            }
            if (can_double) {
                if (ctx->opt_verbose) fprintf(stderr, "[mf] skm: %u slice(s) do not fit, trying %u\n", S, 2 * S);
                S *= 2;
                continue;
            }
            return MF_SKM_FALLBACK;
        }
        if (rc != MF_OK) return rc;
        if (K == 31 && (ctx->opt_ablate & 32)) {
            unsigned long long h[8];
            MF_HIP(hipMemcpyAsync(h, A.c2p.p, 64, hipMemcpyDeviceToHost, st));
            MF_HIP(hipStreamSynchronize(st));
            unsigned long long tot = 0; for (int i = 0; i < 8; i++) tot += h[i];
            static const char *nm[8] = {"partition top", "round set-up", "steps", "last drains", "barrier 1", "compaction", "barrier 2", "same records"};
            for (int i = 0; i < 8; i++) fprintf(stderr, "[mf] k_skm_count wave cycles: %-14s %5.1f %%\n", nm[i], 100.0 * (double)h[i] / (double)(tot ? tot : 1));
        }
        const uint64_t n_dist = A.dused;
        SH.clear();
        // (the record buffers stay in the arena for the next sample: hipMalloc costs 35 ms per GiB, 5 s for a 150 GiB level 1;
        // a host that needs the memory asks with mf_ctx_trim_bytes)
        if (ctx->opt_verbose)
            fprintf(stderr, "[mf] count(skm): n_occ=%llu records=%llu slices=%u%s levels=%zu bits=%d distinct=%llu\n", (unsigned long long)n_occ,
                    A.n_records, S, shared ? " (one level 1)" : "", lv.size(), total_bits, (unsigned long long)n_dist);
        const size_t kb = A.dk.bytes(), cb = A.dc.bytes();       // (capacity: may be a little larger than n_dist)
        MF_TRY(mf_table_adopt(ctx, K, n_dist, n_occ, A.dk.take(), kb, A.dc.take(), cb, out));
        {
            unsigned long long na = 0;
            MF_HIP(hipMemcpyAsync(&na, &scal[7], 8, hipMemcpyDeviceToHost, st));
            MF_HIP(hipStreamSynchronize(st));
            if (n_all) *n_all = na;
            (*out)->n_records = A.n_records;
            (*out)->cut_thr = kthr;
            if (kthr >= 0) {
                (*out)->dropped_hist.assign((size_t)kthr + 1, 0);
                MF_HIP(hipMemcpyAsync((*out)->dropped_hist.data(), A.dhist.p, ((size_t)kthr + 1) * 8, hipMemcpyDeviceToHost, st));
                MF_HIP(hipStreamSynchronize(st));
            }
            (*out)->record_bytes = 16;
        }
        if (total_bits + A.split_bits <= 30) {
            (*out)->part_bits = total_bits + A.split_bits;
            (*out)->part_skm = 1;
            (*out)->part_off_bytes = A.doff.bytes();
            (*out)->d_part_off = A.doff.take();
        }
        return MF_OK;
    }
}

// (a k-mer length whose window of M-mers the scan is not built for -- 3 <= k - M + 1 <= 17: the sliding minimum and the halo of 16 M-mers a
// lane takes from the next one -- goes to the one-record-per-k-mer path)
template <int K, typename... A> static int skm_run_if(A &&... a) {
    if constexpr (K - mf_skm_m(K) + 1 >= 3 && K - mf_skm_m(K) + 1 <= 17) return skm_run<K>(std::forward<A>(a)...);
    else return MF_SKM_FALLBACK;
}
// =============================================================================================
// STREAMED level 1 (round 6; the driver is mf_stream.hip): the reads of a library arrive piece by piece while the rest of the file is still crossing
// PCIe.  begin: the digit regions are sized from a SAMPLE of the files (chunks spread evenly over them, parsed like a small file) and the record
// buffer is made; piece: one launch of the one-pass scatter per piece, all pieces filling the same regions through the same cursors; finish: the
// directory, and the run goes on behind a level 1 that "the first slice has made" (skm_shared).  A region that turns out too small, a plan without
// a split level, anything else that would need the reads again: MF_SKM_FALLBACK, and the caller loads the files whole.
// =============================================================================================
struct mf_skm_stream {
    mf_ctx *ctx = nullptr; int k = 0;
    std::vector<int> lv; bool adaptive = false; int table_bits = 0;
    int G = 0; size_t fast_lds = 0;
    mf_buf<unsigned long long> scal, gcur, save;
    mf_buf<uint64_t> rstart;
    mf_buf<skm_rec> buf; unsigned long long cap = 0;
    uint64_t pieces = 0;
};
__global__ void k_skm_region_sizes_stream(const uint32_t *__restrict__ blockhist, int Gs, int nd, double scale, uint32_t G, uint32_t pct, uint32_t *__restrict__ rsize) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= nd) return;
    uint64_t t = 0;
    for (int b = 0; b < Gs; b++) t += blockhist[(size_t)d * Gs + b];
    const double e = (double)t * scale;
    // an eighth + four standard deviations of the sample's count, scaled; a partly used chunk per workgroup; the runs a piece border cuts
    uint64_t r = (uint64_t)(e + e / 8.0 + 4.0 * sqrt((double)t + 1.0) * scale) + (uint64_t)G * SKM_CH + 16384;
    if (pct != 100u) r = r * pct / 100u;                 // (tests: regions that overflow)
    r = (r + SKM_CH - 1) & ~(uint64_t)(SKM_CH - 1);
    rsize[d] = r > 0xFFFFFF00ull ? 0xFFFFFF00u : (uint32_t)r;
}
template <int K>
static int skm_stream_begin(mf_skm_stream *S, const uint8_t *s_bases, uint64_t s_nbases, const uint32_t *s_vmask, uint64_t s_nwords, double scale) {
    mf_ctx *ctx = S->ctx; hipStream_t st = ctx->stream;
    const int bits1 = S->lv[0], nd1 = 1 << bits1;
    int total_bits = 0; for (int b : S->lv) total_bits += b;
    if (S->lv.size() < 2 || total_bits > 30 || total_bits - bits1 > SKM_DIGIT_BITS) return MF_SKM_FALLBACK;
    S->G = ctx->opt_l1_blocks > 0 ? (int)ctx->opt_l1_blocks : ctx->n_cu;
    S->fast_lds = ((skm_stage_bytes(nd1) + 15) & ~(size_t)15) + (size_t)16 * SKM_FAST_WAVE_BYTES;
    if (S->fast_lds > (size_t)160 * 1024 || ctx->opt_scatter_fast == 0) S->fast_lds = 0;
    int Gs = S->G;
    { const uint64_t maxG = (s_nwords + 1023) / 1024; if ((uint64_t)Gs > maxG) Gs = (int)std::max<uint64_t>(maxG, 1); }
    const uint64_t wpb = (s_nwords + Gs - 1) / Gs;
    mf_buf<uint32_t> blockhist; MF_TRY(blockhist.alloc(ctx, (size_t)nd1 * Gs));
    {
        mf_ktimer t(ctx, "k_skm_hist_sample");
        k_skm_hist<K><<<Gs, 1024, (size_t)nd1 * 8, st>>>(s_bases, s_nbases, s_vmask, s_nwords, wpb, bits1, blockhist.p, nullptr, Gs, 1, 0u, (uint32_t)nd1);
    }
    MF_TRY(S->scal.alloc(ctx, 12)); MF_HIP(hipMemsetAsync(S->scal.p, 0, S->scal.bytes(), st));
    mf_buf<uint32_t> rsize; MF_TRY(rsize.alloc(ctx, (size_t)nd1));
    MF_TRY(S->rstart.alloc(ctx, (size_t)nd1 + 1)); MF_TRY(S->gcur.alloc(ctx, (size_t)nd1)); MF_TRY(S->save.alloc(ctx, (size_t)S->G * nd1));
    k_skm_region_sizes_stream<<<(nd1 + 255) / 256, 256, 0, st>>>(blockhist.p, Gs, nd1, scale, (uint32_t)S->G, (uint32_t)std::max<int64_t>(ctx->opt_stream_count_test_pct, 1), rsize.p);
    MF_TRY(mf_scan<1>(ctx, rsize.p, S->rstart.p, (uint64_t)nd1, (uint64_t *)&S->scal.p[1]));
    MF_HIP(hipMemcpyAsync(S->gcur.p, S->rstart.p, (size_t)nd1 * 8, hipMemcpyDeviceToDevice, st));
    MF_HIP(hipMemsetAsync(S->save.p, 0xFF, S->save.bytes(), st));
    MF_HIP(hipMemcpyAsync(&S->cap, &S->scal.p[1], 8, hipMemcpyDeviceToHost, st));
    MF_HIP(hipStreamSynchronize(st));
    if (S->buf.alloc(ctx, S->cap + (uint64_t)S->G * SKM_CH) != MF_OK) return MF_SKM_FALLBACK;
    if (ctx->opt_verbose) fprintf(stderr, "[mf] skm (streamed): level 1 of %d bits, regions for %.2f G records (%.1f GB) from a sample of %.1f M bases x %.1f\n", bits1, S->cap / 1e9, S->cap * 16 / 1e9, s_nbases / 1e6, scale);
    return MF_OK;
}
template <int K>
static int skm_stream_piece(mf_skm_stream *S, const uint8_t *d_bases, uint64_t n_bases, const uint32_t *vmask, uint64_t n_words) {
    mf_ctx *ctx = S->ctx; hipStream_t st = ctx->stream;
    const int bits1 = S->lv[0], nd1 = 1 << bits1, G = S->G;
    const uint64_t wpb = (n_words + G - 1) / G;
    skm_dyn Dy; Dy.gcur = S->gcur.p; Dy.rend = S->rstart.p + 1; Dy.overflow = (unsigned int *)&S->scal.p[5]; Dy.dump = S->cap; Dy.save = S->save.p;
    mf_ktimer t(ctx, "k_skm_scatter");
    if (S->fast_lds) {
        MF_TRY(skm_set_lds(k_skm_scatter<K, true, true>, S->fast_lds));
        k_skm_scatter<K, true, true><<<G, 1024, S->fast_lds, st>>>(d_bases, n_bases, vmask, n_words, wpb, bits1, nullptr, G, S->buf.p, Dy, 0u, (uint32_t)nd1);
    } else {
        const size_t lds = skm_stage_bytes(nd1);
        MF_TRY(skm_set_lds(k_skm_scatter<K, true, false>, lds));
        k_skm_scatter<K, true, false><<<G, 1024, lds, st>>>(d_bases, n_bases, vmask, n_words, wpb, bits1, nullptr, G, S->buf.p, Dy, 0u, (uint32_t)nd1);
    }
    S->pieces++;
    MF_HIP(hipGetLastError());
    return MF_OK;
}
static int skm_cut_above(int rc, int thr, mf_table **out) {
    if (rc == MF_OK && thr >= 0 && (*out)->cut_thr < thr) {
        // a cut the counting kernel does not make itself (thr >= C2_LH): as a pass over the finished table
        mf_table *all = *out, *good = nullptr;
        rc = mf_table_filter(all, thr, &good);
        if (rc == MF_OK) { good->n_occ = all->n_occ; good->n_records = all->n_records; good->record_bytes = all->record_bytes; }
        mf_table_destroy(all);
        *out = good;
    }
    return rc;
}
template <int K>
static int skm_stream_finish(mf_skm_stream *S, uint64_t n_occ, uint64_t n_bases, int thr, uint64_t *n_all, mf_table **out) {
    mf_ctx *ctx = S->ctx; hipStream_t st = ctx->stream;
    const int nd1 = 1 << S->lv[0];
    k_skm_close_chunks<<<(unsigned)(((uint64_t)S->G * nd1 + 255) / 256), 256, 0, st>>>(S->save.p, (uint64_t)S->G * nd1, S->cap, S->buf.p);
    skm_shared pre; pre.lo = 0; pre.hi = (uint32_t)nd1;
    MF_TRY(pre.pstart.alloc(ctx, (size_t)nd1)); MF_TRY(pre.plen.alloc(ctx, (size_t)nd1)); MF_TRY(pre.pocc.alloc(ctx, (size_t)nd1));
    k_skm_dir_dyn<<<(nd1 + 255) / 256, 256, 0, st>>>(S->rstart.p, S->gcur.p, nd1, pre.pstart.p, pre.plen.p);
    unsigned long long ovf = 0;
    pre.h_pstart.resize((size_t)nd1);
    MF_HIP(hipMemcpyAsync(&ovf, &S->scal.p[5], 8, hipMemcpyDeviceToHost, st));
    MF_HIP(hipMemcpyAsync(pre.h_pstart.data(), pre.pstart.p, (size_t)nd1 * 8, hipMemcpyDeviceToHost, st));
    MF_HIP(hipStreamSynchronize(st));
    if (ovf) {
        if (ctx->opt_verbose) fprintf(stderr, "[mf] skm (streamed): a digit region sized from the sample was too small\n");
        return MF_SKM_FALLBACK;
    }
    pre.buf.swap(S->buf); pre.cap = S->cap; pre.ready = true;
    S->gcur.reset(); S->save.reset(); S->rstart.reset();
    MF_HIP(hipMemsetAsync(S->scal.p, 0, S->scal.bytes(), st));
    const int rc = skm_run<K>(ctx, nullptr, n_bases, nullptr, (n_bases + 31) / 32, n_occ, S->lv, S->adaptive, S->table_bits, S->scal.p, thr, n_all, out, &pre);
    return skm_cut_above(rc, thr, out);
}
template <int K, typename... A> static int skm_stream_begin_if(A &&... a) {
    if constexpr (K - mf_skm_m(K) + 1 >= 3 && K - mf_skm_m(K) + 1 <= 17) return skm_stream_begin<K>(std::forward<A>(a)...); else return MF_SKM_FALLBACK;
}
template <int K, typename... A> static int skm_stream_piece_if(A &&... a) {
    if constexpr (K - mf_skm_m(K) + 1 >= 3 && K - mf_skm_m(K) + 1 <= 17) return skm_stream_piece<K>(std::forward<A>(a)...); else return MF_SKM_FALLBACK;
}
template <int K, typename... A> static int skm_stream_finish_if(A &&... a) {
    if constexpr (K - mf_skm_m(K) + 1 >= 3 && K - mf_skm_m(K) + 1 <= 17) return skm_stream_finish<K>(std::forward<A>(a)...); else return MF_SKM_FALLBACK;
}
#ifndef MF_SKM_ONLY_K31
#define SKM_STREAM_SWITCH(FN, ...) switch (S->k) { \
        case 20: return FN<20>(__VA_ARGS__); case 21: return FN<21>(__VA_ARGS__); case 22: return FN<22>(__VA_ARGS__); case 23: return FN<23>(__VA_ARGS__); \
        case 24: return FN<24>(__VA_ARGS__); case 25: return FN<25>(__VA_ARGS__); case 26: return FN<26>(__VA_ARGS__); case 27: return FN<27>(__VA_ARGS__); \
        case 28: return FN<28>(__VA_ARGS__); case 29: return FN<29>(__VA_ARGS__); case 30: return FN<30>(__VA_ARGS__); case 31: return FN<31>(__VA_ARGS__); \
        default: return MF_SKM_FALLBACK; }
#else
#define SKM_STREAM_SWITCH(FN, ...) switch (S->k) { case 31: return FN<31>(__VA_ARGS__); default: return MF_SKM_FALLBACK; }
#endif
// lv / adaptive / table_bits: the counting plan (mf_count_plan) made from an ESTIMATE of the occurrences; MF_SKM_FALLBACK: not this way
int mf_skm_stream_begin(mf_ctx *ctx, int k, const std::vector<int> &lv, bool adaptive, int table_bits, const uint8_t *s_bases, uint64_t s_nbases,
                        const uint32_t *s_vmask, uint64_t s_nwords, double scale, mf_skm_stream **out) {
    *out = nullptr;
    if (!ctx->opt_skm || k < MF_SKM_MIN_K || k > 31 || ctx->own_world > 1) return MF_SKM_FALLBACK;
    std::unique_ptr<mf_skm_stream> U(new mf_skm_stream());
    mf_skm_stream *S = U.get();
    S->ctx = ctx; S->k = k; S->lv = lv; S->adaptive = adaptive; S->table_bits = table_bits;
    auto go = [&]() -> int { SKM_STREAM_SWITCH(skm_stream_begin_if, S, s_bases, s_nbases, s_vmask, s_nwords, scale) };
    const int rc = go();
    if (rc == MF_OK) *out = U.release();
    return rc;
}
int mf_skm_stream_piece(mf_skm_stream *S, const uint8_t *d_bases, uint64_t n_bases, const uint32_t *vmask, uint64_t n_words) {
    SKM_STREAM_SWITCH(skm_stream_piece_if, S, d_bases, n_bases, vmask, n_words)
}
int mf_skm_stream_finish(mf_skm_stream *S, uint64_t n_occ, uint64_t n_bases, int thr, uint64_t *n_all, mf_table **out) {
    SKM_STREAM_SWITCH(skm_stream_finish_if, S, n_occ, n_bases, thr, n_all, out)
}
void mf_skm_stream_free(mf_skm_stream *S) { delete S; }

// adaptive: the plan was made from the number of occurrences alone -- the levels after the first may be re-planned from a pilot
// table_bits: partition bits the caller wants the TABLE to have (0: one more than the counting plan's)
int mf_count_skm(mf_ctx *ctx, const uint8_t *d_bases, uint64_t n_bases, const uint32_t *vmask, uint64_t n_words, uint64_t n_occ,
                 int k, const std::vector<int> &lv, bool adaptive, int table_bits, unsigned long long *scal, int thr, uint64_t *n_all, mf_table **out) {
    int rc = MF_SKM_FALLBACK;
    switch (k) {
#define SKM_CASE(KK) case KK: rc = skm_run_if<KK>(ctx, d_bases, n_bases, vmask, n_words, n_occ, lv, adaptive, table_bits, scal, thr, n_all, out); break;
#ifndef MF_SKM_ONLY_K31              /* (a quick look at one instantiation's code: hipcc -DMF_SKM_ONLY_K31 -S) */
        SKM_CASE(20) SKM_CASE(21) SKM_CASE(22) SKM_CASE(23) SKM_CASE(24) SKM_CASE(25)
        SKM_CASE(26) SKM_CASE(27) SKM_CASE(28) SKM_CASE(29) SKM_CASE(30)
#endif
        SKM_CASE(31)
#undef SKM_CASE
        default: return MF_SKM_FALLBACK;
    }
    return skm_cut_above(rc, thr, out);
}

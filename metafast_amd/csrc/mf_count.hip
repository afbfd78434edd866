// mf_count.hip -- reads (ASCII in HBM) -> canonical k-mer counts.
//
// Replaces the hot loop of IOUtils.loadReads (src/io/IOUtils.java:756-768: for every read, for every
// k-mer: BigLong2ShortHashMap.addAndBound(canonical, 1)) with an HBM-streaming design for MI355X:
//
//   K0  k_mask_*      valid-start bitmap (1 bit / base): a position starts a k-mer iff the whole
//                     k-mer lies inside one read of length >= max(k, min_len)
//   K1  k_l1_hist     ASCII -> 2-bit -> rolling canonical k-mer -> hash -> per-block digit histogram
//       k_scan        exclusive scan of the (digit x block) matrix -> exact output ranges
//       k_l1_scatter  same k-mer stream again, LDS-staged radix partitioning: one 64-byte staging line
//                     per digit in LDS, flushed with full-line stores; ranges padded with sentinels
//   K2  k_split       one workgroup per partition: LDS histogram + scan + LDS-staged scatter into
//                     2^bits sub-partitions (exact, no global atomics)
//   K3  k_count       one partition at a time per workgroup: open-addressed count table in LDS
//                     (64-bit CAS on keys, 32-bit add on counts), compacted in place
//       k_gather      dense (key,count) arrays
//
// Random accesses never leave LDS; HBM only sees streaming reads and full 64-byte line writes.
#include "mf_common.h"
#include "mf_count_dev.h"
#include <algorithm>
#include <vector>

int mf_count_skm(mf_ctx *ctx, const uint8_t *d_bases, uint64_t n_bases, const uint32_t *vmask, uint64_t n_words, uint64_t n_occ,
                 int k, const std::vector<int> &lv, bool adaptive, int table_bits, unsigned long long *scal, int thr, uint64_t *n_all, mf_table **out);

// =============================================================================================
// K0: valid-start bitmap
// =============================================================================================
__global__ void k_mask_init(uint32_t *__restrict__ vmask, uint64_t n_words, uint64_t n_bases) {
    uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    uint64_t lo = w * 32;
    uint32_t m = 0xFFFFFFFFu;
    if (lo + 32 > n_bases) m = (n_bases > lo) ? ((1u << (n_bases - lo)) - 1u) : 0u;
    vmask[w] = m;
}

template <bool ATOMIC>
__device__ __forceinline__ void mf_clear_bits(uint32_t *vmask, uint64_t lo, uint64_t hi) {  // clear [lo,hi)
    while (lo < hi) {
        uint64_t w = lo >> 5;
        uint32_t b0 = (uint32_t)(lo & 31);
        uint64_t wend = (w + 1) << 5;
        uint32_t b1 = (uint32_t)((hi < wend ? hi : wend) - (w << 5));  // 1..32
        uint32_t m = (b1 == 32 ? 0xFFFFFFFFu : ((1u << b1) - 1u)) & ~((1u << b0) - 1u);
        if (ATOMIC) atomicAnd(&vmask[w], ~m);
        else vmask[w] &= ~m;
        lo = wend;
    }
}

// one thread per read: clear the last k-1 start positions (or the whole read if too short).
// The words of a long read's window [e-k+1, e) can only be shared with the NEXT read's cleared range, and only if that
// read is short; so when this read and the next are both >= 64 bases the update is a plain read-modify-write
// (1e8 reads x 2 atomics on the bitmap were 20 ms).
__global__ void k_mask_reads(const uint64_t *__restrict__ off, uint64_t n_reads, int k, int min_len,
                             uint32_t *__restrict__ vmask, unsigned long long *__restrict__ n_occ) {
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t occ = 0;
    if (r < n_reads) {
        uint64_t s = off[r], e = off[r + 1];
        uint64_t len = e - s;
        uint64_t len_next = r + 1 < n_reads ? off[r + 2] - e : 64;
        if (len < (uint64_t)k || (int64_t)len < (int64_t)min_len) mf_clear_bits<true>(vmask, s, e);
        else {
            if (len >= 64 && len_next >= 64 && (int64_t)len_next >= (int64_t)min_len) mf_clear_bits<false>(vmask, e - (uint64_t)k + 1, e);   // (a next read below min_len is cleared whole, with atomics, starting in our last word)
            else mf_clear_bits<true>(vmask, e - (uint64_t)k + 1, e);
            occ = len - (uint64_t)k + 1;
        }
    }
    // block reduce, ONE atomic per 1024-thread workgroup (an atomic per wave on this single counter is 1.6e6 serialised
    // atomics at 100 M reads = the whole 19 ms of this kernel)
    __shared__ unsigned long long part[16];
    for (int d = 32; d >= 1; d >>= 1) occ += __shfl_down(occ, d, 64);
    if (mf_lane() == 0) part[threadIdx.x >> 6] = occ;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); w++) t += part[w];
        if (t) atomicAdd(n_occ, t);
    }
}

// =============================================================================================
// k-mer producer: one lane = one 32-position word of the base stream
// =============================================================================================
// Calls f(j, canonical_kmer, valid) for j = 0..31 (every lane makes all 32 calls so that wave-level
// protocols inside f stay convergent).  Word w covers base positions [32w, 32w+32).
template <typename F>
__device__ __forceinline__ void mf_word_kmers(const uint8_t *__restrict__ bases, uint64_t n_bases, uint64_t w,
                                              uint32_t m, int k, F &&f) {
    const uint4 *p = reinterpret_cast<const uint4 *>(bases + w * 32);
    uint64_t b0 = w * 32;
    uint4 z = make_uint4(0, 0, 0, 0);
    uint4 c0 = (b0 < n_bases) ? p[0] : z;
    uint4 c1 = (b0 + 16 < n_bases) ? p[1] : z;
    uint4 c2 = (b0 + 32 < n_bases) ? p[2] : z;
    uint4 c3 = (b0 + 48 < n_bases) ? p[3] : z;
    uint64_t W0 = ((uint64_t)mf_dec16(c0) << 32) | mf_dec16(c1);
    uint64_t W1 = ((uint64_t)mf_dec16(c2) << 32) | mf_dec16(c3);
    const int sh = 64 - 2 * k;
    const int top = 2 * k - 2;
    uint64_t fw = W0 >> sh;
    uint64_t rc = mf_revcomp(fw, k);
#pragma unroll 4
    for (int j = 0; j < 32; j++) {
        uint64_t cn = fw < rc ? fw : rc;
        f(j, cn, (bool)((m >> j) & 1u));
        W0 = (W0 << 2) | (W1 >> 62);
        W1 <<= 2;
        fw = W0 >> sh;
        rc = (rc >> 2) | ((uint64_t)(3u - (uint32_t)(fw & 3u)) << top);
    }
}

// Same walk, handing the k-mers to f in groups of 4: f(keys[4], valid[4]) is called 8 times.
template <typename F>
__device__ __forceinline__ void mf_word_kmers4(const uint8_t *__restrict__ bases, uint64_t n_bases, uint64_t w,
                                               uint32_t m, int k, F &&f) {
    const uint4 *p = reinterpret_cast<const uint4 *>(bases + w * 32);
    uint64_t b0 = w * 32;
    uint4 z = make_uint4(0, 0, 0, 0);
    uint4 c0 = (b0 < n_bases) ? p[0] : z;
    uint4 c1 = (b0 + 16 < n_bases) ? p[1] : z;
    uint4 c2 = (b0 + 32 < n_bases) ? p[2] : z;
    uint4 c3 = (b0 + 48 < n_bases) ? p[3] : z;
    uint64_t W0 = ((uint64_t)mf_dec16(c0) << 32) | mf_dec16(c1);
    uint64_t W1 = ((uint64_t)mf_dec16(c2) << 32) | mf_dec16(c3);
    const int sh = 64 - 2 * k;
    const int top = 2 * k - 2;
    uint64_t fw = W0 >> sh;
    uint64_t rc = mf_revcomp(fw, k);
#pragma unroll 2
    for (int j = 0; j < 32; j += 4) {
        uint64_t keys[4]; bool valid[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            keys[u] = fw < rc ? fw : rc;
            valid[u] = (bool)((m >> (j + u)) & 1u);
            W0 = (W0 << 2) | (W1 >> 62);
            W1 <<= 2;
            fw = W0 >> sh;
            rc = (rc >> 2) | ((uint64_t)(3u - (uint32_t)(fw & 3u)) << top);
        }
        f(keys, valid);
    }
}

__device__ __forceinline__ uint32_t mf_digit(uint64_t h, int shift_hi, int bits) {
    // digit = bits [64-shift_hi-bits, 64-shift_hi) of h, i.e. skip the shift_hi top bits already used
    return bits ? (uint32_t)((h << shift_hi) >> (64 - bits)) : 0u;
}

// =============================================================================================
// K1a: per-block digit histogram
// =============================================================================================
__global__ __launch_bounds__(1024) void k_l1_hist(const uint8_t *__restrict__ bases, uint64_t n_bases,
                                                  const uint32_t *__restrict__ vmask, uint64_t n_words,
                                                  uint64_t words_per_block, int k, int bits,
                                                  uint32_t *__restrict__ blockhist, int G) {
    __shared__ uint32_t hist[1 << MF_MAX_DIGIT_BITS];
    const int nd = 1 << bits;
    for (int i = threadIdx.x; i < nd; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    uint64_t wlo = (uint64_t)blockIdx.x * words_per_block;
    uint64_t whi = wlo + words_per_block < n_words ? wlo + words_per_block : n_words;
    for (uint64_t w = wlo + threadIdx.x; w < whi; w += blockDim.x) {
        uint32_t m = vmask[w];
        if (!m) continue;
        mf_word_kmers(bases, n_bases, w, m, k, [&](int, uint64_t key, bool valid) {
            if (valid) atomicAdd(&hist[mf_digit(mf_phash(key), 0, bits)], 1u);
        });
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nd; i += blockDim.x) blockhist[(size_t)i * G + blockIdx.x] = hist[i];
}

// =============================================================================================
// LDS-staged scatter: one 64-byte line (8 k-mers) per digit
// =============================================================================================
struct mf_stage {
    uint64_t *line;   // [nd][8] staging lines + 64 dummy slots (one per lane)
    uint64_t *cur;    // [nd] next global element index (multiple of 8) for this workgroup's range of digit d
    uint32_t *ctr;    // [nd] low 16 = slots reserved, high 16 = slots committed; + 64 dummy counters (always 0)
    uint64_t *q_pos;  // [16 waves][MF_QCAP] flush queue: output position
    uint32_t *q_d;    // [16 waves][MF_QCAP] flush queue: digit
    int nd;
};
#define MF_QCAP 16
#define MF_MLP 4          // 16-byte loads in flight per thread in the streaming loops

// Lock-free within the workgroup: reserve a slot, write it, commit; the 8th committer flushes the line to HBM with
// four 16-byte stores and reopens it.  Lanes that find the line full retry.
//
// Two things in this function were learnt the hard way on gfx950:
//  * The retry loop is WAVE-UNIFORM (ballot): every lane stays in the loop until the whole wave has placed its
//    elements, and the write + commit + flush of a successful lane happen INSIDE the iteration.  With a per-lane
//    `while (!done)` loop hipcc sinks the success path below the loop, and the lane holding slot 7 then waits at the
//    loop exit for a same-wave lane that spins on the full line forever (SIMT deadlock).
//  * MF_B elements per lane go through the protocol together and every step is BRANCH-FREE: lanes with nothing to do
//    in a step still execute the LDS instruction, on a private dummy counter / dummy slot, and "add" 0.  Written
//    with `if (pending[b])` around each step, hipcc emits one exec-mask branch + s_waitcnt per element and the 4 x 3
//    LDS round trips are fully serialised (the kernel was LDS-latency bound at 12 round trips per 4 k-mers).
#define MF_B 4
// four 8-byte stores, then four returning adds (the commits); the stores need no wait of their own (in-order LDS)
__device__ __forceinline__ void mf_lds_write4_add_rtn4(const uint32_t (&wa)[4], const uint64_t (&wv)[4], const uint32_t (&a)[4],
                                                       const uint32_t (&inc)[4], uint32_t (&old)[4]) {
    asm volatile("ds_write_b64 %4, %8\n\tds_write_b64 %5, %9\n\tds_write_b64 %6, %10\n\tds_write_b64 %7, %11\n\t"
                 "ds_add_rtn_u32 %0, %12, %16\n\tds_add_rtn_u32 %1, %13, %17\n\tds_add_rtn_u32 %2, %14, %18\n\t"
                 "ds_add_rtn_u32 %3, %15, %19\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(old[0]), "=&v"(old[1]), "=&v"(old[2]), "=&v"(old[3])
                 : "v"(wa[0]), "v"(wa[1]), "v"(wa[2]), "v"(wa[3]), "v"(wv[0]), "v"(wv[1]), "v"(wv[2]), "v"(wv[3]),
                   "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(inc[0]), "v"(inc[1]), "v"(inc[2]), "v"(inc[3])
                 : "memory");
}
__device__ __forceinline__ void mf_stage_insert_batch(const mf_stage &L, uint64_t *__restrict__ out, const uint32_t (&d)[MF_B],
                                                      const uint64_t (&key)[MF_B], bool (&pending)[MF_B], int ablate = 0) {
    if (ablate == 1) {            // diagnostic: k-mer generation + hashing only (keeps the values alive)
        uint64_t acc = 0;
#pragma unroll
        for (int b = 0; b < MF_B; b++) acc += pending[b] ? key[b] + d[b] : 0;
        if (acc == 0x123456789ull) out[0] = acc;
        return;
    }
    const uint32_t ctr0 = mf_lds_addr(L.ctr), line0 = mf_lds_addr(L.line);
    const uint32_t dummy_ctr = ctr0 + 4u * ((uint32_t)L.nd + (uint32_t)mf_lane());               // ctr[nd .. nd+64): one per lane
    const uint32_t dummy_slot = line0 + 8u * ((uint32_t)L.nd * MF_LINE + (uint32_t)mf_lane());   // line[nd*8 .. nd*8+64)
    for (;;) {
        bool any = false;
#pragma unroll
        for (int b = 0; b < MF_B; b++) any |= pending[b];
        if (__ballot(any) == 0ull) break;
        uint32_t ca[MF_B], w[MF_B], inc[MF_B], old[MF_B], old2[MF_B], wa[MF_B];
        bool got[MF_B];
#pragma unroll
        for (int b = 0; b < MF_B; b++) ca[b] = pending[b] ? ctr0 + 4u * d[b] : dummy_ctr;
        mf_lds_read4(ca, w);                                                     // peek: is the line open?
#pragma unroll
        for (int b = 0; b < MF_B; b++) inc[b] = (pending[b] && (w[b] & 0xFFFFu) < (uint32_t)MF_LINE) ? 1u : 0u;
        mf_lds_add_rtn4(ca, inc, old);                                           // reserve a slot
#pragma unroll
        for (int b = 0; b < MF_B; b++) {
            got[b] = inc[b] && (old[b] & 0xFFFFu) < (uint32_t)MF_LINE;
            wa[b] = got[b] ? line0 + 8u * (d[b] * MF_LINE + (old[b] & 0xFFFFu)) : dummy_slot;
            ca[b] = got[b] ? ctr0 + 4u * d[b] : dummy_ctr;
            inc[b] = got[b] ? 0x10000u : 0u;
        }
        mf_lds_write4_add_rtn4(wa, key, ca, inc, old2);                          // write the slot, commit
        // Flush of the completed lines, wave-cooperative: the completing lanes queue (digit, position) in a small per-wave
        // LDS queue and then FOUR lanes write one 64-byte line with ONE store instruction (16 lines per instruction).
        // One lane writing its line with four 16-byte stores costs 4x the L2 write requests and made the HBM flush half
        // of the kernel time (ablation: 18.4 ms -> 8.8 ms without the stores at 2.4e9 k-mers).
        uint64_t *qpos = L.q_pos + (threadIdx.x >> 6) * MF_QCAP;
        uint32_t *qd = L.q_d + (threadIdx.x >> 6) * MF_QCAP;
        const uint64_t lt_mask = (1ull << mf_lane()) - 1ull;
#pragma unroll
        for (int b = 0; b < MF_B; b++) {
            const bool fl = got[b] && (old2[b] >> 16) == (uint32_t)(MF_LINE - 1);     // 8th committer of its line
            const unsigned long long F = __ballot(fl);
            if (got[b]) pending[b] = false;
            if (F == 0ull) continue;
            const uint32_t n = (uint32_t)__popcll(F), qi = (uint32_t)__popcll(F & lt_mask);
            uint64_t mypos = 0;
            if (fl) { mypos = L.cur[d[b]]; L.cur[d[b]] = mypos + MF_LINE; }
            for (uint32_t e0 = 0; e0 < n; e0 += MF_QCAP) {
                if (fl && qi >= e0 && qi < e0 + MF_QCAP) { qpos[qi - e0] = mypos; qd[qi - e0] = d[b]; }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // queue visible to the wave (in-order LDS)
                const uint32_t e = e0 + ((uint32_t)mf_lane() >> 2), c = (uint32_t)mf_lane() & 3u;
                if (e < n) {
                    const uint32_t dd = qd[e - e0];
                    const uint64_t pp = qpos[e - e0];
                    const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(&L.line[dd * MF_LINE + 2 * c]);
                    if (ablate != 2) *reinterpret_cast<ulonglong2 *>(out + pp + 2 * c) = v;
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
// after a barrier: write every partly filled line, padding with the sentinel
__device__ __forceinline__ void mf_stage_flush_all(const mf_stage &L, uint64_t *__restrict__ out, int nd) {
    for (int d = threadIdx.x; d < nd; d += blockDim.x) {
        uint32_t c = L.ctr[d] >> 16;
        if (c) {
            uint64_t pos = L.cur[d];
            for (uint32_t s = 0; s < (uint32_t)MF_LINE; s++) out[pos + s] = s < c ? L.line[d * MF_LINE + s] : MF_EMPTY;
            L.cur[d] = pos + MF_LINE;
            L.ctr[d] = 0;
        }
    }
}
__device__ __forceinline__ mf_stage mf_stage_carve(unsigned char *smem, int nd) {
    mf_stage L;
    L.nd = nd;
    L.line = reinterpret_cast<uint64_t *>(smem);
    L.cur = L.line + (size_t)nd * MF_LINE + 64;
    L.q_pos = L.cur + nd;
    L.ctr = reinterpret_cast<uint32_t *>(L.q_pos + 16 * MF_QCAP);
    L.q_d = L.ctr + nd + 64;
    return L;
}
static inline size_t mf_stage_bytes(int nd) { return (size_t)nd * (MF_LINE * 8 + 8 + 4) + 64 * 8 + 64 * 4 + 16 * MF_QCAP * 12; }

// =============================================================================================
// K1b: scatter reads' k-mers into 2^bits partitions (ranges from k_l1_hist + k_scan)
// =============================================================================================
template <bool STAGED>
__global__ __launch_bounds__(1024) void k_l1_scatter(const uint8_t *__restrict__ bases, uint64_t n_bases,
                                                     const uint32_t *__restrict__ vmask, uint64_t n_words,
                                                     uint64_t words_per_block, int k, int bits,
                                                     const uint64_t *__restrict__ blockstart, int G,
                                                     uint64_t *__restrict__ out, int ablate) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nd = 1 << bits;
    mf_stage L = mf_stage_carve(smem, nd);
    for (int i = threadIdx.x; i < nd; i += blockDim.x) { L.cur[i] = blockstart[(size_t)i * G + blockIdx.x]; L.ctr[i] = 0; }
    if (threadIdx.x < 64) L.ctr[nd + threadIdx.x] = 0;
    __syncthreads();
    uint64_t wlo = (uint64_t)blockIdx.x * words_per_block;
    uint64_t whi = wlo + words_per_block < n_words ? wlo + words_per_block : n_words;
    // ALL 64 lanes of a wave must reach mf_stage_insert_batch together (its flush is wave-cooperative: four lanes per
    // line, lane 4e reopens line e), so the loop bounds and the skip of empty words are wave-uniform
    for (uint64_t wb = wlo; wb < whi; wb += blockDim.x) {
        const uint64_t w = wb + threadIdx.x;
        const uint32_t m = w < whi ? vmask[w] : 0u;
        if (__ballot(m != 0u) == 0ull) continue;
        mf_word_kmers4(bases, n_bases, w, m, k, [&](const uint64_t (&keys)[4], bool (&valid)[4]) {
            uint32_t d[4];
#pragma unroll
            for (int u = 0; u < 4; u++) d[u] = mf_digit(mf_phash(keys[u]), 0, bits);
            if (STAGED) mf_stage_insert_batch(L, out, d, keys, valid, ablate);
            else {
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (valid[u]) {
                        uint64_t pos = atomicAdd(reinterpret_cast<unsigned long long *>(&L.cur[d[u]]), 1ull);
                        out[pos] = keys[u];
                    }
            }
        });
    }
    __syncthreads();
    if (STAGED) mf_stage_flush_all(L, out, nd);
    else {
        for (int d = threadIdx.x; d < nd; d += blockDim.x) {
            uint64_t pos = L.cur[d];
            while (pos & 7) out[pos++] = MF_EMPTY;
        }
    }
}

// partition directory after level 1: start / padded length per digit
__global__ void k_l1_dir(const uint64_t *__restrict__ blockstart, int G, int nd, uint64_t *__restrict__ pstart,
                         uint32_t *__restrict__ plen) {
    int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= nd) return;
    uint64_t s = blockstart[(size_t)d * G], e = blockstart[(size_t)(d + 1) * G];
    pstart[d] = s;
    plen[d] = (uint32_t)(e - s);
}

// =============================================================================================
// K2: split every partition into 2^bits sub-partitions (one workgroup per partition at a time)
// =============================================================================================
template <bool STAGED>
__global__ __launch_bounds__(1024) void k_split(const uint64_t *__restrict__ in, const uint64_t *__restrict__ pstart,
                                                const uint32_t *__restrict__ plen, uint32_t np, int bits_used, int bits,
                                                uint64_t *__restrict__ out, uint64_t *__restrict__ ostart,
                                                uint32_t *__restrict__ olen) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ uint32_t scratch[17];
    const int nd = 1 << bits;
    mf_stage L = mf_stage_carve(smem, nd);
    const int ipt = (nd + (int)blockDim.x - 1) / (int)blockDim.x;   // bins per thread for the scan
    for (uint32_t p = blockIdx.x; p < np; p += gridDim.x) {
        const uint64_t start = pstart[p];
        const uint32_t len = plen[p];
        const uint64_t obase = start + (uint64_t)p * (uint64_t)(MF_LINE * nd);   // room for per-bin padding
        for (int i = threadIdx.x; i < nd; i += blockDim.x) L.ctr[i] = 0;
        __syncthreads();
        // 16-byte loads, MF_MLP of them in flight per thread (start is 64-byte aligned, len a multiple of 8)
        const ulonglong2 *in2 = reinterpret_cast<const ulonglong2 *>(in + start);
        const uint32_t npairs = len >> 1;
        for (uint32_t jb = 0; jb < npairs; jb += MF_MLP * blockDim.x) {
            ulonglong2 v[MF_MLP];
#pragma unroll
            for (int u = 0; u < MF_MLP; u++) {
                uint32_t j = jb + u * blockDim.x + threadIdx.x;
                v[u] = j < npairs ? in2[j] : make_ulonglong2(MF_EMPTY, MF_EMPTY);
            }
#pragma unroll
            for (int u = 0; u < MF_MLP; u++) {
                if (v[u].x != MF_EMPTY) atomicAdd(&L.ctr[mf_digit(mf_phash(v[u].x), bits_used, bits)], 1u);
                if (v[u].y != MF_EMPTY) atomicAdd(&L.ctr[mf_digit(mf_phash(v[u].y), bits_used, bits)], 1u);
            }
        }
        __syncthreads();
        // exclusive scan of padded bin sizes
        uint32_t mine = 0;
        int b0 = threadIdx.x * ipt;
        for (int j = 0; j < ipt; j++) { int b = b0 + j; if (b < nd) mine += (L.ctr[b] + 7u) & ~7u; }
        uint32_t tot;
        uint32_t ex = mf_block_excl_scan(mine, scratch, &tot);
        for (int j = 0; j < ipt; j++) {
            int b = b0 + j;
            if (b < nd) {
                uint32_t c = (L.ctr[b] + 7u) & ~7u;
                uint64_t gs = obase + ex;
                L.cur[b] = gs;
                ostart[(size_t)p * nd + b] = gs;
                olen[(size_t)p * nd + b] = c;
                ex += c;
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nd + 64; i += blockDim.x) L.ctr[i] = 0;
        __syncthreads();
        // uniform trip count for every wave (the insert's retry loop is wave-uniform)
        for (uint32_t jb = 0; jb < npairs; jb += MF_MLP * blockDim.x) {
            ulonglong2 v[MF_MLP];
#pragma unroll
            for (int u = 0; u < MF_MLP; u++) {
                uint32_t j = jb + u * blockDim.x + threadIdx.x;
                v[u] = j < npairs ? in2[j] : make_ulonglong2(MF_EMPTY, MF_EMPTY);
            }
#pragma unroll
            for (int u = 0; u < MF_MLP; u += 2) {
                uint64_t keys[4] = {v[u].x, v[u].y, v[u + 1].x, v[u + 1].y};
                uint32_t d[4]; bool valid[4];
#pragma unroll
                for (int q = 0; q < 4; q++) { valid[q] = keys[q] != MF_EMPTY; d[q] = mf_digit(mf_phash(keys[q]), bits_used, bits); }
                if (STAGED) mf_stage_insert_batch(L, out, d, keys, valid);
                else {
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        if (valid[q]) {
                            uint64_t pos = atomicAdd(reinterpret_cast<unsigned long long *>(&L.cur[d[q]]), 1ull);
                            out[pos] = keys[q];
                        }
                }
            }
        }
        __syncthreads();
        if (STAGED) mf_stage_flush_all(L, out, nd);
        else {
            for (int d = threadIdx.x; d < nd; d += blockDim.x) {
                uint64_t pos = L.cur[d];
                while (pos & 7) out[pos++] = MF_EMPTY;
            }
        }
        __syncthreads();
    }
}

// =============================================================================================
// K3: hash-count one partition at a time in LDS, compact in place
// =============================================================================================
// keys[start .. start+len) (with sentinels) -> keys[start .. start+d) distinct, cnt[start .. start+d) counts.
// 256-thread workgroups with a 48 KiB table so that three of them share a CU.  The first MF_PF*2*256 keys of the
// NEXT partition are loaded into registers while the current one is being counted, so the HBM latency of a
// partition (one per ~3 k k-mers) is hidden behind the LDS work of the previous one.
#define MF_PF 6            // 16-byte prefetch loads per thread: 12 keys x 256 threads = 3072 keys
__global__ __launch_bounds__(256) void k_count(uint64_t *__restrict__ keys, uint16_t *__restrict__ cnt,
                                               const uint64_t *__restrict__ pstart, const uint32_t *__restrict__ plen,
                                               uint32_t np, uint32_t *__restrict__ dcount,
                                               unsigned int *__restrict__ overflow, int ablate) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ uint32_t out_cursor;
    uint64_t *tk = reinterpret_cast<uint64_t *>(smem);                    // [MF_COUNT_SLOTS] + 64 dummy slots
    uint32_t *tc = reinterpret_cast<uint32_t *>(tk + MF_COUNT_SLOTS + 64);   // [MF_COUNT_SLOTS] + 64 dummy counters
    const uint32_t tk0 = mf_lds_addr(tk), tc0 = mf_lds_addr(tc);
    const uint32_t dummy_k = tk0 + 8u * ((uint32_t)MF_COUNT_SLOTS + (uint32_t)mf_lane());
    const uint32_t dummy_c = tc0 + 4u * ((uint32_t)MF_COUNT_SLOTS + (uint32_t)mf_lane());
    if (threadIdx.x < 64) { tk[MF_COUNT_SLOTS + threadIdx.x] = 0; tc[MF_COUNT_SLOTS + threadIdx.x] = 0; }   // dummies: never EMPTY
    const ulonglong2 EE = make_ulonglong2(MF_EMPTY, MF_EMPTY);
    uint32_t p = blockIdx.x;
    if (p >= np) return;
    uint64_t start = pstart[p];
    uint32_t len = plen[p];
    ulonglong2 R[MF_PF];
    {
        const ulonglong2 *in2 = reinterpret_cast<const ulonglong2 *>(keys + start);
        const uint32_t npairs = len >> 1;
#pragma unroll
        for (int u = 0; u < MF_PF; u++) { uint32_t j = u * blockDim.x + threadIdx.x; R[u] = j < npairs ? in2[j] : EE; }
    }
    for (;;) {
        const uint32_t pn = p + gridDim.x;
        uint64_t start_n = 0; uint32_t len_n = 0;
        if (pn < np) { start_n = pstart[pn]; len_n = plen[pn]; }
        // table size: power of two >= 9/8 * len, in [blockDim, MF_COUNT_SLOTS]
        uint32_t want = len + len / 8 + 1;
        uint32_t slots = blockDim.x;
        while (slots < want && slots < (uint32_t)MF_COUNT_SLOTS) slots <<= 1;
        const uint32_t mask = slots - 1;
        for (uint32_t i = threadIdx.x; i < slots; i += blockDim.x) { tk[i] = MF_EMPTY; tc[i] = 0; }
        if (threadIdx.x == 0) out_cursor = 0;
        __syncthreads();
        // take over the prefetched keys, then start fetching the next partition
        ulonglong2 K[MF_PF];
#pragma unroll
        for (int u = 0; u < MF_PF; u++) K[u] = R[u];
        if (pn < np) {
            const ulonglong2 *nx2 = reinterpret_cast<const ulonglong2 *>(keys + start_n);
            const uint32_t npairs_n = len_n >> 1;
#pragma unroll
            for (int u = 0; u < MF_PF; u++) { uint32_t j = u * blockDim.x + threadIdx.x; R[u] = j < npairs_n ? nx2[j] : EE; }
        }
        if (ablate != 3) {
#pragma unroll
        for (int u = 0; u < MF_PF; u += 2) {
            uint64_t k4[4] = {K[u].x, K[u].y, K[u + 1].x, K[u + 1].y};
            mf_count_insert4(tk0, tc0, dummy_k, dummy_c, mask, slots, k4, overflow);
        }
        } else { uint64_t acc = 0;
#pragma unroll
            for (int u = 0; u < MF_PF; u++) acc += K[u].x ^ K[u].y;
            if (acc == 0x1234567ull) tk[0] = acc; }
        {   // partitions longer than the prefetch window (heavy hitters): the rest straight from HBM
            const ulonglong2 *in2 = reinterpret_cast<const ulonglong2 *>(keys + start);
            const uint32_t npairs = len >> 1;
            for (uint32_t jb = MF_PF * blockDim.x; jb < npairs; jb += 2 * blockDim.x) {
                uint32_t j0 = jb + threadIdx.x, j1 = j0 + blockDim.x;
                ulonglong2 a = j0 < npairs ? in2[j0] : EE, b = j1 < npairs ? in2[j1] : EE;
                uint64_t k4[4] = {a.x, a.y, b.x, b.y};
                mf_count_insert4(tk0, tc0, dummy_k, dummy_c, mask, slots, k4, overflow);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the ds_add_u32 of the asm blocks are invisible to hipcc's waitcnt pass
        __syncthreads();
        // compaction: each wave walks 64-slot chunks (lane = slot: conflict-free LDS reads; a per-thread run of 16
        // consecutive slots is a 128-byte lane stride = 32-way bank conflict and made this phase dominate the kernel),
        // keeps them in registers, reserves its output range with ONE LDS atomic and writes; the order inside a
        // partition does not matter
        {
            constexpr int NCH = MF_COUNT_SLOTS / 256;
            const int nch = (int)(slots >> 8);
            uint64_t ck[NCH]; uint32_t cv[NCH], pre[NCH]; uint32_t total = 0;
            const uint64_t lt_mask = (1ull << mf_lane()) - 1ull;
#pragma unroll
            for (int i = 0; i < NCH; i++) {
                ck[i] = MF_EMPTY; cv[i] = 0; pre[i] = 0;
                if (i < nch) {
                    const uint32_t sl = ((threadIdx.x >> 6) << 6) + (uint32_t)i * blockDim.x + (uint32_t)mf_lane();
                    ck[i] = tk[sl]; cv[i] = tc[sl];
                    const unsigned long long bal = __ballot(ck[i] != MF_EMPTY);
                    pre[i] = total + (uint32_t)__popcll(bal & lt_mask);
                    total += (uint32_t)__popcll(bal);
                }
            }
            uint32_t wb = 0;
            if (mf_lane() == 0 && total) wb = atomicAdd(&out_cursor, total);
            wb = __shfl(wb, 0, 64);
#pragma unroll
            for (int i = 0; i < NCH; i++) {
                if (i < nch && ck[i] != MF_EMPTY && ablate != 5) {
                    const uint32_t pos = wb + pre[i];
                    keys[start + pos] = ck[i];
                    cnt[start + pos] = (uint16_t)(cv[i] > (uint32_t)MF_MAX_COUNT ? (uint32_t)MF_MAX_COUNT : cv[i]);
                }
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) dcount[p] = out_cursor;
        if (pn >= np) break;
        p = pn; start = start_n; len = len_n;
    }
}

__global__ void k_sum_counts(const uint16_t *__restrict__ c, uint64_t n, unsigned long long *__restrict__ total) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t s = 0;
    for (; i < n; i += stride) s += c[i];
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_down(s, d, 64);
    if (mf_lane() == 0 && s) atomicAdd(total, (unsigned long long)s);
}

// =============================================================================================
// host orchestration
// =============================================================================================
static int ceil_log2_u64(uint64_t x) { int b = 0; while ((1ull << b) < x) b++; return b; }

template <typename K> static int set_lds(K kern, size_t bytes) {
    MF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return MF_OK;
}

// K0 as a call (the streamed count masks piece by piece): vmask[ceil(n_bases / 32)], *d_nocc += the piece's k-mer occurrences; asynchronous
int mf_mask_reads(mf_ctx *ctx, const uint64_t *d_offsets, uint64_t n_reads, uint64_t n_bases, int k, int min_len, uint32_t *vmask, unsigned long long *d_nocc) {
    const uint64_t n_words = (n_bases + 31) / 32;
    if (!n_words || !n_reads) return MF_OK;
    mf_ktimer t(ctx, "k_mask");
    k_mask_init<<<(unsigned)((n_words + 255) / 256), 256, 0, ctx->stream>>>(vmask, n_words, n_bases);
    k_mask_reads<<<(unsigned)((n_reads + 1023) / 1024), 1024, 0, ctx->stream>>>(d_offsets, n_reads, k, min_len, vmask, d_nocc);
    MF_HIP(hipGetLastError());
    return MF_OK;
}
// the partition plan of a count: lv = the k-mer path's levels, slv = the super-k-mer path's (counting units), B = partition bits of the table,
// assembled = the input is taken for assembled sequences (no pilot).  Shared by mf_count_core and the streamed count (mf_stream.hip), which plans
// from an estimate of n_occ.
int mf_count_plan(mf_ctx *ctx, uint64_t n_occ, uint64_t n_reads, uint64_t n_bases, int k, int min_len, std::vector<int> &lv, std::vector<int> &slv, bool &assembled, int &B) {
    // ---- partition plan ----
    // Partitions are sized by k-mer OCCURRENCES (the LDS table must hold a partition's distinct k-mers, and reads repeat
    // theirs: several occurrences per distinct k-mer).  Long sequences (mean length >= 8k) are assembled ones -- the
    // cutter's input, ComponentCutterMain.java:81 -- whose k-mers are nearly all distinct: size those by a sixth, so that
    // a partition's index region stays a few KB and the 8 neighbour probes of a k-mer stay cache-local.
    uint64_t target = (uint64_t)ctx->opt_part_target;
    // (assembled: long sequences, or a caller that filters by length -- ComponentCutterMain.java:81 is the one that does --, or
    // the pipeline's hint; at low coverage unitigs are short, and partitions planned for reads then hold 4000 distinct k-mers)
    // (round 5: length alone says "assembled" only where no pilot will look at the input: 250-base reads at k = 25 .. 31 are "long sequences"
    // by the 8 k rule, were planned as unitigs -- 2^23 units of 200 records, three radix levels -- and took 387 ms where 150-base reads of the
    // same volume take 130 (profiles/r05az_probe_shapes.txt); the pilot measures what the rule guesses: distinct k-mers per occurrence)
    const bool pilot_looks = ctx->opt_skm && k >= MF_SKM_MIN_K && ctx->opt_skm_pilot != 0 && ctx->opt_l1_bits < 0;
    assembled = (n_bases / n_reads >= (uint64_t)(8 * k) && !pilot_looks) || min_len > 0 || ctx->opt_union_samples > 0 || ctx->own_world > 1;
    if (assembled && target > (uint64_t)ctx->opt_part_target_long) target = (uint64_t)ctx->opt_part_target_long;
    // (a shard of the union of many samples' unitigs: the samples share most of their k-mers -- 0.36 distinct per occurrence at 8
    // samples --, and it is the DISTINCT k-mers of a partition that must fit the LDS tables)
    if ((ctx->own_world >= 4 || ctx->opt_union_samples >= 4) && target == (uint64_t)ctx->opt_part_target_long) target *= 2;
    B = ceil_log2_u64((n_occ + target - 1) / target);
    if (ctx->own_world > 1) {                       // (a shard: every rank must own at least one level-1 digit)
        int lw = 0; while ((1 << lw) < ctx->own_world) lw++;
        if (B < lw) B = lw;
        if (!ctx->opt_skm || k < MF_SKM_MIN_K) return mf_set_error("mf_count_device_shard: k >= %d needed (minimizer partitions decide the owner)", MF_SKM_MIN_K);
    }
    lv.clear();
    if (ctx->opt_l1_bits >= 0) {
        lv.push_back((int)ctx->opt_l1_bits);
        int rest = ctx->opt_l2_bits >= 0 ? (int)ctx->opt_l2_bits : std::max(0, B - lv[0]);
        while (rest > 0) { int b = std::min(rest, MF_MAX_DIGIT_BITS); lv.push_back(b); rest -= b; }
    } else {
        int levels = std::max(1, (B + MF_MAX_DIGIT_BITS - 1) / MF_MAX_DIGIT_BITS);
        int rest = B;
        for (int i = 0; i < levels; i++) { int b = (rest + (levels - i) - 1) / (levels - i); lv.push_back(b); rest -= b; }
    }
    int total_bits = 0; for (int b : lv) total_bits += b;
    if (total_bits > 40) return mf_set_error("partition plan needs %d bits", total_bits);

    // ---- the super-k-mer path's plan (mf_skm.hip) ----
    {
        // A third radix level costs a whole extra pass over the records (k = 21, 100 M reads: 23 bits, +96 ms).  The LDS table
        // limits a partition's DISTINCT k-mers, the plan sizes it by occurrences: up to twice the target is tried with two
        // full levels first (if a partition turns out too rich the call ends up on the k-mer path, with its own plan).
        // The counting pass runs on partitions TWICE the planned size (one bit fewer): its per-partition costs (directory,
        // barriers, the sweep of the LDS table) halve, and the gather cuts every counting partition in two by the next bit of
        // the partition hash, so the table still has the B bits of partitions the graph kernels are planned for (k_gather_split).
        // Forced plans (l1_bits / l2_bits: tests) are taken as the counting plan.
        slv = lv;
        if (ctx->opt_l1_bits < 0) {
            int Bc = B > 0 ? B - 1 : 0;
            // (assembled sequences: nearly every k-mer distinct, and the table's partitions are planned small for the graph kernels --
            // 256 occurrences -- where the counting kernel pays four workgroup barriers and a sweep of its LDS table per unit: it
            // counts EIGHT table partitions as one unit (~2000 distinct k-mers, k_gather_split_n cuts them apart); the cutter-table
            // launch of the benchmark cost 13 x the sample's time per occurrence with units of two)
            if (assembled) Bc = std::max(0, B - (int)ctx->opt_unit_parts_long);
            if (ctx->own_world > 1) { int lw = 0; while ((1 << lw) < ctx->own_world) lw++; if (Bc < lw) Bc = lw; }
            slv.clear();
            const int levels = std::max(1, (Bc + MF_MAX_DIGIT_BITS - 1) / MF_MAX_DIGIT_BITS);
            int rest = Bc;
            for (int i = 0; i < levels; i++) { int b = (rest + (levels - i) - 1) / (levels - i); slv.push_back(b); rest -= b; }
            if (slv.size() == 3 && Bc <= 2 * MF_MAX_DIGIT_BITS + 1 && (n_occ >> (2 * MF_MAX_DIGIT_BITS)) <= 4 * target)
                slv = {MF_MAX_DIGIT_BITS, MF_MAX_DIGIT_BITS};
        }
    }
    return MF_OK;
}

// thr >= 0: keep only the k-mers with count > thr (*n_all = distinct k-mers before the cut); thr < 0: keep everything
int mf_count_core(mf_ctx *ctx, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n_reads, uint64_t n_bases,
                  int k, int min_len, mf_table **out, int thr, uint64_t *n_all) {
    if (n_all) *n_all = 0;
    if (k < 1) return mf_set_error("The size of k-mer must be at least 1.");           // KmersCounterMain.java:66-69
    if (k > 31) return mf_set_error("The size of k-mer must be no more than 31.");     // KmersCounterMain.java:70-73
    if (((uintptr_t)d_bases & 15) != 0) return mf_set_error("mf_count_device: d_bases must be 16-byte aligned");
    MF_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;

    if (n_reads == 0 || n_bases == 0) return mf_table_adopt(ctx, k, 0, 0, nullptr, 0, nullptr, 0, out);

    // ---- K0 ----
    const uint64_t n_words = (n_bases + 31) / 32;
    mf_buf<uint32_t> vmask; MF_TRY(vmask.alloc(ctx, n_words));
    mf_buf<unsigned long long> scal; MF_TRY(scal.alloc(ctx, 12));  // [0]=n_occ [1]=scan total [2]=overflow [3]=scan total 2
    MF_HIP(hipMemsetAsync(scal.p, 0, scal.bytes(), st));
    {
        mf_ktimer t(ctx, "k_mask");
        k_mask_init<<<(unsigned)((n_words + 255) / 256), 256, 0, st>>>(vmask.p, n_words, n_bases);
        k_mask_reads<<<(unsigned)((n_reads + 1023) / 1024), 1024, 0, st>>>(d_offsets, n_reads, k, min_len, vmask.p, &scal.p[0]);
    }
    MF_DBG(ctx, "k_mask");
    MF_HIP(hipGetLastError());
    unsigned long long n_occ = 0;
    MF_HIP(hipMemcpyAsync(&n_occ, &scal.p[0], 8, hipMemcpyDeviceToHost, st));
    MF_HIP(hipStreamSynchronize(st));
    if (n_occ == 0) return mf_table_adopt(ctx, k, 0, 0, nullptr, 0, nullptr, 0, out);

    std::vector<int> lv, slv; bool assembled = false; int B = 0;
    MF_TRY(mf_count_plan(ctx, n_occ, n_reads, n_bases, k, min_len, lv, slv, assembled, B));
    int total_bits = 0; for (int b : lv) total_bits += b;

    // ---- super-k-mer path (mf_skm.hip); falls through to the k-mer path below if the input does not suit it ----
    if (ctx->opt_skm && k >= MF_SKM_MIN_K) {
        // (reads: the plan is provisional -- a pilot measures the distinct k-mers per occurrence and sets the later levels, mf_skm.hip)
        int rc = mf_count_skm(ctx, d_bases, n_bases, vmask.p, n_words, n_occ, k, slv, ctx->opt_l1_bits < 0 && !assembled,
                              ctx->opt_l1_bits < 0 && assembled ? B : 0, scal.p, thr, n_all, out);
        if (rc != MF_SKM_FALLBACK) return rc;
    }
    if (ctx->own_world > 1) return mf_set_error("mf_count_device_shard: the input does not suit the minimizer-partition path");

    // ---- K1 ----
    const int bits1 = lv[0], nd1 = 1 << bits1;
    int G = ctx->opt_l1_blocks > 0 ? (int)ctx->opt_l1_blocks : ctx->n_cu;
    {
        uint64_t maxG = (n_words + 1023) / 1024;   // at least one word per thread
        if ((uint64_t)G > maxG) G = (int)maxG;
        if (G < 1) G = 1;
    }
    const uint64_t wpb = (n_words + G - 1) / G;
    mf_buf<uint32_t> blockhist; MF_TRY(blockhist.alloc(ctx, (size_t)nd1 * G));
    mf_buf<uint64_t> blockstart; MF_TRY(blockstart.alloc(ctx, (size_t)nd1 * G + 1));
    {
        mf_ktimer t(ctx, "k_l1_hist");
        k_l1_hist<<<G, 1024, 0, st>>>(d_bases, n_bases, vmask.p, n_words, wpb, k, bits1, blockhist.p, G);
    }
    MF_DBG(ctx, "k_l1_hist");
    {
        mf_ktimer t(ctx, "k_scan");
        MF_TRY(mf_scan<8>(ctx, blockhist.p, blockstart.p, (uint64_t)nd1 * G, (uint64_t *)&scal.p[1]));
    }
    MF_DBG(ctx, "k_scan");
    uint64_t cap = n_occ + (uint64_t)MF_LINE * nd1 * G;    // upper bound of the padded total
    mf_buf<uint64_t> bufA; MF_TRY(bufA.alloc(ctx, cap));
    const bool staged = ctx->opt_scatter_staged != 0;
    {
        size_t lds = mf_stage_bytes(nd1);
        mf_ktimer t(ctx, "k_l1_scatter");
        if (staged) {
            MF_TRY(set_lds(k_l1_scatter<true>, lds));
            k_l1_scatter<true><<<G, 1024, lds, st>>>(d_bases, n_bases, vmask.p, n_words, wpb, k, bits1, blockstart.p, G, bufA.p, (int)ctx->opt_ablate);
        } else {
            MF_TRY(set_lds(k_l1_scatter<false>, lds));
            k_l1_scatter<false><<<G, 1024, lds, st>>>(d_bases, n_bases, vmask.p, n_words, wpb, k, bits1, blockstart.p, G, bufA.p, 0);
        }
    }
    MF_DBG(ctx, "k_l1_scatter");
    uint32_t np = (uint32_t)nd1;
    mf_buf<uint64_t> pstart; MF_TRY(pstart.alloc(ctx, np));
    mf_buf<uint32_t> plen; MF_TRY(plen.alloc(ctx, np));
    k_l1_dir<<<(nd1 + 255) / 256, 256, 0, st>>>(blockstart.p, G, nd1, pstart.p, plen.p);
    vmask.reset(); blockhist.reset(); blockstart.reset();

    // ---- K2 levels ----
    int bits_used = bits1;
    for (size_t li = 1; li < lv.size(); li++) {
        const int bits = lv[li], nd = 1 << bits;
        uint64_t cap2 = cap + (uint64_t)np * MF_LINE * nd;
        uint64_t np2 = (uint64_t)np * nd;
        if (np2 > 0xFFFFFFF0ull) return mf_set_error("too many partitions");
        mf_buf<uint64_t> bufB; MF_TRY(bufB.alloc(ctx, cap2));
        mf_buf<uint64_t> ostart; MF_TRY(ostart.alloc(ctx, np2));
        mf_buf<uint32_t> olen; MF_TRY(olen.alloc(ctx, np2));
        size_t lds = mf_stage_bytes(nd);
        unsigned grid = (unsigned)std::min<uint64_t>(np, (uint64_t)ctx->n_cu * (lds > 72 * 1024 ? 1 : 2));
        {
            mf_ktimer t(ctx, "k_split");
            if (staged) {
                MF_TRY(set_lds(k_split<true>, lds));
                k_split<true><<<grid, 1024, lds, st>>>(bufA.p, pstart.p, plen.p, np, bits_used, bits, bufB.p, ostart.p, olen.p);
            } else {
                MF_TRY(set_lds(k_split<false>, lds));
                k_split<false><<<grid, 1024, lds, st>>>(bufA.p, pstart.p, plen.p, np, bits_used, bits, bufB.p, ostart.p, olen.p);
            }
        }
        MF_DBG(ctx, "k_split");
        // swap
        std::swap(bufA.p, bufB.p); std::swap(bufA.n, bufB.n);
        std::swap(pstart.p, ostart.p); std::swap(pstart.n, ostart.n);
        std::swap(plen.p, olen.p); std::swap(plen.n, olen.n);
        cap = cap2; np = (uint32_t)np2; bits_used += bits;
    }

    // ---- K3 ----
    mf_buf<uint16_t> cnt; MF_TRY(cnt.alloc(ctx, cap));
    mf_buf<uint32_t> dcount; MF_TRY(dcount.alloc(ctx, np));
    {
        size_t lds = (size_t)(MF_COUNT_SLOTS + 64) * 12;
        MF_TRY(set_lds(k_count, lds));
        unsigned grid = (unsigned)std::min<uint64_t>(np, (uint64_t)ctx->n_cu * 3);
        mf_ktimer t(ctx, "k_count");
        k_count<<<grid, 256, lds, st>>>(bufA.p, cnt.p, pstart.p, plen.p, np, dcount.p, (unsigned int *)&scal.p[2], (int)ctx->opt_ablate);
    }
    MF_DBG(ctx, "k_count");
    mf_buf<uint64_t> doff; MF_TRY(doff.alloc(ctx, (size_t)np + 1));
    {
        mf_ktimer t(ctx, "k_scan");
        MF_TRY(mf_scan<1>(ctx, dcount.p, doff.p, np, (uint64_t *)&scal.p[3]));
    }
    MF_DBG(ctx, "k_scan");
    MF_HIP(hipGetLastError());
    unsigned long long res[4];
    MF_HIP(hipMemcpyAsync(res, scal.p, 32, hipMemcpyDeviceToHost, st));
    MF_HIP(hipStreamSynchronize(st));
    if (res[2]) return mf_set_error("k_count: LDS table overflow (about %d or more distinct k-mers in one partition); "
                                    "lower option part_target", MF_COUNT_SLOTS);
    const uint64_t n_dist = res[3];
    mf_buf<uint64_t> dk; MF_TRY(dk.alloc(ctx, n_dist));
    mf_buf<uint16_t> dc; MF_TRY(dc.alloc(ctx, n_dist));
    {
        unsigned grid = (unsigned)std::min<uint64_t>(np, (uint64_t)ctx->n_cu * 32);
        mf_ktimer t(ctx, "k_gather");
        k_gather<<<grid, 256, 0, st>>>(bufA.p, cnt.p, pstart.p, dcount.p, doff.p, np, dk.p, dc.p);
    }
    MF_DBG(ctx, "k_gather");
    if (ctx->opt_verbose)
        fprintf(stderr, "[mf] count: n_occ=%llu levels=%zu bits=%d np=%u distinct=%llu\n", (unsigned long long)n_occ,
                lv.size(), total_bits, np, (unsigned long long)n_dist);
    MF_HIP(hipGetLastError());
    size_t kb = dk.bytes(), cb = dc.bytes();
    MF_TRY(mf_table_adopt(ctx, k, n_dist, n_occ, dk.take(), kb, dc.take(), cb, out));
    (*out)->n_records = cap; (*out)->record_bytes = 8;
    if (total_bits > 0 && total_bits <= 30) {          // keep the partition structure (see mf_table_ensure_index)
        (*out)->part_bits = total_bits;
        (*out)->part_off_bytes = doff.bytes();
        (*out)->d_part_off = doff.take();
    }
    if (n_all) *n_all = n_dist;
    if (thr >= 0) {                                     // (this path keeps the cut as a separate pass)
        mf_table *all = *out, *good = nullptr;
        int rc = mf_table_filter(all, thr, &good);
        if (rc == MF_OK) { good->n_occ = all->n_occ; good->n_records = all->n_records; good->record_bytes = all->record_bytes; good->cut_thr = thr; }
        mf_table_destroy(all);
        *out = good;
        return rc;
    }
    return MF_OK;
}

int mf_sum_counts(mf_ctx *ctx, const uint16_t *d_counts, uint64_t n, uint64_t *total) {
    *total = 0;
    if (!n) return MF_OK;
    mf_buf<unsigned long long> acc; MF_TRY(acc.alloc(ctx, 1));
    MF_HIP(hipMemsetAsync(acc.p, 0, 8, ctx->stream));
    unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, 4096);
    k_sum_counts<<<grid, 256, 0, ctx->stream>>>(d_counts, n, acc.p);
    unsigned long long t = 0;
    MF_HIP(hipMemcpyAsync(&t, acc.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    MF_HIP(hipStreamSynchronize(ctx->stream));
    *total = t;
    return MF_OK;
}

// ---- occurrences of the k-mers of an index in the reads (ReadsPresenceWorker.process, src/io/IOUtils.java:816-825):
// one lane = one 32-position word, every valid k-mer is looked up and, if present, its entry's counter goes up by one ----
__global__ __launch_bounds__(256) void k_presence(const uint8_t *__restrict__ bases, uint64_t n_bases, const uint32_t *__restrict__ vmask,
                                                  uint64_t n_words, int k, mf_index_view ix, unsigned long long *__restrict__ occ) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    const uint32_t m = vmask[w];
    if (!m) return;
    mf_word_kmers(bases, n_bases, w, m, k, [&](int, uint64_t key, bool valid) {
        uint32_t idx, val;
        if (valid && mf_index_find(ix, key, &idx, &val)) atomicAdd(&occ[idx], 1ull);
    });
}
// occ[i] += occurrences in the reads of the key at entry i of the index (reads shorter than k contribute nothing)
int mf_presence_core(mf_ctx *ctx, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n_reads, uint64_t n_bases, int k,
                     const mf_index &index, unsigned long long *d_occ) {
    if (k < 1 || k > 31) return mf_set_error("k must be in [1,31]");
    if (((uintptr_t)d_bases & 15) != 0) return mf_set_error("reads: d_bases must be 16-byte aligned");
    if (n_reads == 0 || n_bases == 0) return MF_OK;
    hipStream_t st = ctx->stream;
    const uint64_t n_words = (n_bases + 31) / 32;
    mf_buf<uint32_t> vmask; MF_TRY(vmask.alloc(ctx, n_words));
    mf_buf<unsigned long long> scal; MF_TRY(scal.alloc(ctx, 1));
    MF_HIP(hipMemsetAsync(scal.p, 0, 8, st));
    k_mask_init<<<(unsigned)((n_words + 255) / 256), 256, 0, st>>>(vmask.p, n_words, n_bases);
    k_mask_reads<<<(unsigned)((n_reads + 1023) / 1024), 1024, 0, st>>>(d_offsets, n_reads, k, 0, vmask.p, scal.p);
    {
        mf_ktimer t(ctx, "k_presence");
        k_presence<<<(unsigned)((n_words + 255) / 256), 256, 0, st>>>(d_bases, n_bases, vmask.p, n_words, k, mf_view(index), d_occ);
    }
    MF_HIP(hipGetLastError());
    MF_HIP(hipStreamSynchronize(st));
    return MF_OK;
}

extern "C" int mf_count_device(mf_ctx *ctx, const void *d_bases, const void *d_offsets, uint64_t n_reads,
                               uint64_t n_bases, int k, int min_read_len, mf_table **out) {
    mf_range rng_("mf:count");
    if (!ctx || !out) return mf_set_error("mf_count_device: NULL argument");
    *out = nullptr;
    return mf_count_core(ctx, (const uint8_t *)d_bases, (const uint64_t *)d_offsets, n_reads, n_bases, k, min_read_len, out, -1, nullptr);
}
// rank's shard of the table of ALL the given sequences: the k-mers whose minimizer-partition hash starts with `rank` (its top
// log2(world) bits).  The table has the partitions of the whole table; the other ranks' partitions are empty.
extern "C" int mf_count_device_shard(mf_ctx *ctx, const void *d_bases, const void *d_offsets, uint64_t n_reads, uint64_t n_bases, int k,
                                     int min_read_len, int rank, int world, mf_table **out) {
    mf_range rng_("mf:count_shard");
    if (!ctx || !out) return mf_set_error("mf_count_device_shard: NULL argument");
    *out = nullptr;
    if (world < 1 || world > 64 || (world & (world - 1)) || rank < 0 || rank >= world)
        return mf_set_error("mf_count_device_shard: the world size must be a power of two <= 64 and 0 <= rank < world");
    ctx->own_rank = rank; ctx->own_world = world;
    const int rc = mf_count_core(ctx, (const uint8_t *)d_bases, (const uint64_t *)d_offsets, n_reads, n_bases, k, min_read_len, out, -1, nullptr);
    ctx->own_rank = 0; ctx->own_world = 1;
    return rc;
}
extern "C" int mf_count_device_above(mf_ctx *ctx, const void *d_bases, const void *d_offsets, uint64_t n_reads,
                                     uint64_t n_bases, int k, int min_read_len, int threshold, mf_table **out,
                                     uint64_t *n_distinct_all) {
    mf_range rng_("mf:count");
    if (!ctx || !out) return mf_set_error("mf_count_device_above: NULL argument");
    *out = nullptr;
    return mf_count_core(ctx, (const uint8_t *)d_bases, (const uint64_t *)d_offsets, n_reads, n_bases, k, min_read_len, out,
                         threshold < 0 ? -1 : threshold, n_distinct_all);
}

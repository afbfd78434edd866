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
#include <algorithm>

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

__device__ __forceinline__ void mf_clear_bits(uint32_t *vmask, uint64_t lo, uint64_t hi) {  // clear [lo,hi)
    while (lo < hi) {
        uint64_t w = lo >> 5;
        uint32_t b0 = (uint32_t)(lo & 31);
        uint64_t wend = (w + 1) << 5;
        uint32_t b1 = (uint32_t)((hi < wend ? hi : wend) - (w << 5));  // 1..32
        uint32_t m = (b1 == 32 ? 0xFFFFFFFFu : ((1u << b1) - 1u)) & ~((1u << b0) - 1u);
        atomicAnd(&vmask[w], ~m);
        lo = wend;
    }
}

// one thread per read: clear the last k-1 start positions (or the whole read if too short)
__global__ void k_mask_reads(const uint64_t *__restrict__ off, uint64_t n_reads, int k, int min_len,
                             uint32_t *__restrict__ vmask, unsigned long long *__restrict__ n_occ) {
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t occ = 0;
    if (r < n_reads) {
        uint64_t s = off[r], e = off[r + 1];
        uint64_t len = e - s;
        if (len < (uint64_t)k || (int64_t)len < (int64_t)min_len) mf_clear_bits(vmask, s, e);
        else { mf_clear_bits(vmask, e - (uint64_t)k + 1, e); occ = len - (uint64_t)k + 1; }
    }
    // wave reduce, one atomic per wave
    for (int d = 32; d >= 1; d >>= 1) occ += __shfl_down(occ, d, 64);
    if (mf_lane() == 0 && occ) atomicAdd(n_occ, (unsigned long long)occ);
}

// =============================================================================================
// k-mer producer: one lane = one 32-position word of the base stream
// =============================================================================================
// 4 ASCII bases (byte 0 = first base) -> 8 bits, first base most significant, code A0 G1 C2 T3.
// (c>>1)&3 maps A,C,T,G (either case) to 0,1,2,3; the reference order needs f(x)=((x0^x1)<<1)|x1.
__device__ __forceinline__ uint32_t mf_dec4(uint32_t w) {
    uint32_t t = (w >> 1) & 0x03030303u;
    uint32_t x1 = (t >> 1) & 0x01010101u;
    uint32_t x0 = t & 0x01010101u;
    uint32_t c = ((x0 ^ x1) << 1) | x1;
    return (c * 0x40100401u) >> 24;   // gathers the four 2-bit fields into one byte
}
__device__ __forceinline__ uint32_t mf_dec16(uint4 v) {
    return (mf_dec4(v.x) << 24) | (mf_dec4(v.y) << 16) | (mf_dec4(v.z) << 8) | mf_dec4(v.w);
}

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
            if (valid) atomicAdd(&hist[mf_digit(mf_hash64(key), 0, bits)], 1u);
        });
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nd; i += blockDim.x) blockhist[(size_t)i * G + blockIdx.x] = hist[i];
}

// =============================================================================================
// LDS-staged scatter: one 64-byte line (8 k-mers) per digit
// =============================================================================================
struct mf_stage {
    uint64_t *line;   // [nd][8]
    uint64_t *cur;    // [nd] next global element index (multiple of 8) for this workgroup's range of digit d
    uint32_t *ctr;    // [nd] low 16 = slots reserved, high 16 = slots committed
};
#define MF_WG __HIP_MEMORY_SCOPE_WORKGROUP
#define MF_MLP 4          // 16-byte loads in flight per thread in the streaming loops

// Lock-free within the workgroup: reserve a slot, write it, commit; the 8th committer flushes the
// line to HBM with four 16-byte stores and reopens it.  Lanes that find the line full retry.
//
// The retry loop is WAVE-UNIFORM (ballot): every lane of the wave stays in the loop until the whole
// wave has placed its element, and the write + commit + flush of a successful lane happen INSIDE the
// iteration.  A per-lane `while (!done)` loop is wrong here: hipcc sinks the success path below the
// loop, so the lane holding slot 7 would wait at the loop exit for a same-wave lane that spins on the
// full line forever (SIMT deadlock, observed on gfx950).
__device__ __forceinline__ void mf_stage_insert(const mf_stage &L, uint64_t *__restrict__ out, uint32_t d, uint64_t key,
                                                bool active) {
    bool pending = active;
    while (__ballot(pending) != 0ull) {
        if (pending) {
            uint32_t w = __hip_atomic_load(&L.ctr[d], __ATOMIC_RELAXED, MF_WG);
            if ((w & 0xFFFFu) < (uint32_t)MF_LINE) {
                uint32_t old = __hip_atomic_fetch_add(&L.ctr[d], 1u, __ATOMIC_RELAXED, MF_WG);
                uint32_t r = old & 0xFFFFu;
                if (r < (uint32_t)MF_LINE) {
                    L.line[d * MF_LINE + r] = key;
                    uint32_t old2 = __hip_atomic_fetch_add(&L.ctr[d], 0x10000u, __ATOMIC_ACQ_REL, MF_WG);
                    if ((old2 >> 16) == (uint32_t)(MF_LINE - 1)) {
                        uint64_t pos = L.cur[d];
                        L.cur[d] = pos + MF_LINE;
                        const ulonglong2 *s = reinterpret_cast<const ulonglong2 *>(&L.line[d * MF_LINE]);
                        ulonglong2 a = s[0], b = s[1], c = s[2], e = s[3];
                        ulonglong2 *o = reinterpret_cast<ulonglong2 *>(out + pos);
                        o[0] = a; o[1] = b; o[2] = c; o[3] = e;
                        __hip_atomic_store(&L.ctr[d], 0u, __ATOMIC_RELEASE, MF_WG);
                    }
                    pending = false;
                }
            }
        }
        // keep every memory operation of this iteration inside it
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
}
// Same protocol for MF_B elements per lane at once: the peek / reserve / write / commit steps of the MF_B
// independent elements are issued back to back, so their LDS round trips overlap instead of adding up
// (with one element per call the kernel is bound by three dependent LDS latencies per k-mer).
#define MF_B 4
__device__ __forceinline__ void mf_stage_insert_batch(const mf_stage &L, uint64_t *__restrict__ out, const uint32_t (&d)[MF_B],
                                                      const uint64_t (&key)[MF_B], bool (&pending)[MF_B]) {
    for (;;) {
        bool any = false;
#pragma unroll
        for (int b = 0; b < MF_B; b++) any |= pending[b];
        if (__ballot(any) == 0ull) break;
        uint32_t w[MF_B], r[MF_B];
        bool got[MF_B];
#pragma unroll
        for (int b = 0; b < MF_B; b++) w[b] = pending[b] ? __hip_atomic_load(&L.ctr[d[b]], __ATOMIC_RELAXED, MF_WG) : 0xFFFFu;
#pragma unroll
        for (int b = 0; b < MF_B; b++) {
            got[b] = false; r[b] = 0;
            if (pending[b] && (w[b] & 0xFFFFu) < (uint32_t)MF_LINE) {
                uint32_t old = __hip_atomic_fetch_add(&L.ctr[d[b]], 1u, __ATOMIC_RELAXED, MF_WG);
                r[b] = old & 0xFFFFu;
                got[b] = r[b] < (uint32_t)MF_LINE;
            }
        }
#pragma unroll
        for (int b = 0; b < MF_B; b++)
            if (got[b]) L.line[d[b] * MF_LINE + r[b]] = key[b];
#pragma unroll
        for (int b = 0; b < MF_B; b++) {
            if (got[b]) {
                uint32_t old2 = __hip_atomic_fetch_add(&L.ctr[d[b]], 0x10000u, __ATOMIC_ACQ_REL, MF_WG);
                if ((old2 >> 16) == (uint32_t)(MF_LINE - 1)) {
                    uint64_t pos = L.cur[d[b]];
                    L.cur[d[b]] = pos + MF_LINE;
                    const ulonglong2 *s = reinterpret_cast<const ulonglong2 *>(&L.line[d[b] * MF_LINE]);
                    ulonglong2 a = s[0], bb = s[1], c = s[2], e = s[3];
                    ulonglong2 *o = reinterpret_cast<ulonglong2 *>(out + pos);
                    o[0] = a; o[1] = bb; o[2] = c; o[3] = e;
                    __hip_atomic_store(&L.ctr[d[b]], 0u, __ATOMIC_RELEASE, MF_WG);
                }
                pending[b] = false;
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
    L.line = reinterpret_cast<uint64_t *>(smem);
    L.cur = L.line + (size_t)nd * MF_LINE;
    L.ctr = reinterpret_cast<uint32_t *>(L.cur + nd);
    return L;
}
static inline size_t mf_stage_bytes(int nd) { return (size_t)nd * (MF_LINE * 8 + 8 + 4); }

// =============================================================================================
// K1b: scatter reads' k-mers into 2^bits partitions (ranges from k_l1_hist + k_scan)
// =============================================================================================
template <bool STAGED>
__global__ __launch_bounds__(1024) void k_l1_scatter(const uint8_t *__restrict__ bases, uint64_t n_bases,
                                                     const uint32_t *__restrict__ vmask, uint64_t n_words,
                                                     uint64_t words_per_block, int k, int bits,
                                                     const uint64_t *__restrict__ blockstart, int G,
                                                     uint64_t *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nd = 1 << bits;
    mf_stage L = mf_stage_carve(smem, nd);
    for (int i = threadIdx.x; i < nd; i += blockDim.x) { L.cur[i] = blockstart[(size_t)i * G + blockIdx.x]; L.ctr[i] = 0; }
    __syncthreads();
    uint64_t wlo = (uint64_t)blockIdx.x * words_per_block;
    uint64_t whi = wlo + words_per_block < n_words ? wlo + words_per_block : n_words;
    // every wave runs the same number of iterations so that no wave exits while others still spin
    for (uint64_t w = wlo + threadIdx.x; w < whi; w += blockDim.x) {
        uint32_t m = vmask[w];
        if (!m) continue;
        mf_word_kmers4(bases, n_bases, w, m, k, [&](const uint64_t (&keys)[4], bool (&valid)[4]) {
            uint32_t d[4];
#pragma unroll
            for (int u = 0; u < 4; u++) d[u] = mf_digit(mf_hash64(keys[u]), 0, bits);
            if (STAGED) mf_stage_insert_batch(L, out, d, keys, valid);
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
                if (v[u].x != MF_EMPTY) atomicAdd(&L.ctr[mf_digit(mf_hash64(v[u].x), bits_used, bits)], 1u);
                if (v[u].y != MF_EMPTY) atomicAdd(&L.ctr[mf_digit(mf_hash64(v[u].y), bits_used, bits)], 1u);
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
        for (int i = threadIdx.x; i < nd; i += blockDim.x) L.ctr[i] = 0;
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
                for (int q = 0; q < 4; q++) { valid[q] = keys[q] != MF_EMPTY; d[q] = mf_digit(mf_hash64(keys[q]), bits_used, bits); }
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
__device__ __forceinline__ void mf_count_insert(uint64_t *tk, uint32_t *tc, uint32_t mask, uint32_t slots, uint64_t key, uint32_t s,
                                                unsigned int *overflow) {
    uint32_t probes = 0;
    for (;;) {
        uint64_t cur = *reinterpret_cast<volatile uint64_t *>(&tk[s]);
        if (cur == MF_EMPTY) {
            cur = atomicCAS(reinterpret_cast<unsigned long long *>(&tk[s]), (unsigned long long)MF_EMPTY, (unsigned long long)key);
            if (cur == MF_EMPTY) cur = key;
        }
        if (cur == key) { atomicAdd(&tc[s], 1u); return; }
        s = (s + 1) & mask;
        if (++probes > slots) { atomicExch(overflow, 1u); return; }
    }
}
// four keys at a time: the four first-probe reads are in flight together; a hit is one fire-and-forget LDS add
__device__ __forceinline__ void mf_count_insert4(uint64_t *tk, uint32_t *tc, uint32_t mask, uint32_t slots, const uint64_t (&key)[4],
                                                 unsigned int *overflow) {
    uint32_t s[4]; uint64_t cur[4];
#pragma unroll
    for (int b = 0; b < 4; b++) s[b] = (uint32_t)mf_hash64(key[b]) & mask;
#pragma unroll
    for (int b = 0; b < 4; b++) cur[b] = *reinterpret_cast<volatile uint64_t *>(&tk[s[b]]);
#pragma unroll
    for (int b = 0; b < 4; b++) {
        if (key[b] == MF_EMPTY) continue;
        if (cur[b] == key[b]) atomicAdd(&tc[s[b]], 1u);
        else mf_count_insert(tk, tc, mask, slots, key[b], s[b], overflow);
    }
}
__global__ __launch_bounds__(256) void k_count(uint64_t *__restrict__ keys, uint16_t *__restrict__ cnt,
                                               const uint64_t *__restrict__ pstart, const uint32_t *__restrict__ plen,
                                               uint32_t np, uint32_t *__restrict__ dcount,
                                               unsigned int *__restrict__ overflow) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ uint32_t out_cursor;
    uint64_t *tk = reinterpret_cast<uint64_t *>(smem);                    // [MF_COUNT_SLOTS]
    uint32_t *tc = reinterpret_cast<uint32_t *>(tk + MF_COUNT_SLOTS);     // [MF_COUNT_SLOTS]
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
#pragma unroll
        for (int u = 0; u < MF_PF; u += 2) {
            uint64_t k4[4] = {K[u].x, K[u].y, K[u + 1].x, K[u + 1].y};
            mf_count_insert4(tk, tc, mask, slots, k4, overflow);
        }
        {   // partitions longer than the prefetch window (heavy hitters): the rest straight from HBM
            const ulonglong2 *in2 = reinterpret_cast<const ulonglong2 *>(keys + start);
            const uint32_t npairs = len >> 1;
            for (uint32_t jb = MF_PF * blockDim.x; jb < npairs; jb += 2 * blockDim.x) {
                uint32_t j0 = jb + threadIdx.x, j1 = j0 + blockDim.x;
                ulonglong2 a = j0 < npairs ? in2[j0] : EE, b = j1 < npairs ? in2[j1] : EE;
                uint64_t k4[4] = {a.x, a.y, b.x, b.y};
                mf_count_insert4(tk, tc, mask, slots, k4, overflow);
            }
        }
        __syncthreads();
        // compaction: each wave walks 64-slot chunks (lane = slot: conflict-free LDS reads; a per-thread run of 16
        // consecutive slots is a 128-byte lane stride = 32-way bank conflict and made this phase dominate the kernel)
        // and appends the occupied ones behind a cursor in LDS; the order inside a partition does not matter
        for (uint32_t base = (threadIdx.x >> 6) << 6; base < slots; base += blockDim.x) {
            const uint32_t sl = base + (uint32_t)mf_lane();
            const uint64_t key = tk[sl];
            const bool has = key != MF_EMPTY;
            const unsigned long long b = __ballot(has);
            if (b) {
                uint32_t wb = 0;
                if (mf_lane() == 0) wb = atomicAdd(&out_cursor, (uint32_t)__popcll(b));
                wb = __shfl(wb, 0, 64);
                if (has) {
                    const uint32_t pos = wb + (uint32_t)__popcll(b & ((1ull << mf_lane()) - 1ull));
                    const uint32_t v = tc[sl];
                    keys[start + pos] = key;
                    cnt[start + pos] = (uint16_t)(v > (uint32_t)MF_MAX_COUNT ? (uint32_t)MF_MAX_COUNT : v);
                }
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) dcount[p] = out_cursor;
        if (pn >= np) break;
        p = pn; start = start_n; len = len_n;
    }
}

// dense output: partition p's d distinct entries go to [doff[p], doff[p]+d)
__global__ __launch_bounds__(256) void k_gather(const uint64_t *__restrict__ keys, const uint16_t *__restrict__ cnt,
                                                const uint64_t *__restrict__ pstart, const uint32_t *__restrict__ dcount,
                                                const uint64_t *__restrict__ doff, uint32_t np,
                                                uint64_t *__restrict__ dk, uint16_t *__restrict__ dc) {
    for (uint32_t p = blockIdx.x; p < np; p += gridDim.x) {
        uint64_t s = pstart[p], o = doff[p];
        uint32_t d = dcount[p];
        for (uint32_t j = threadIdx.x; j < d; j += blockDim.x) { dk[o + j] = keys[s + j]; dc[o + j] = cnt[s + j]; }
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

int mf_count_core(mf_ctx *ctx, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n_reads, uint64_t n_bases,
                  int k, int min_len, mf_table **out) {
    if (k < 1) return mf_set_error("The size of k-mer must be at least 1.");           // KmersCounterMain.java:66-69
    if (k > 31) return mf_set_error("The size of k-mer must be no more than 31.");     // KmersCounterMain.java:70-73
    if (((uintptr_t)d_bases & 15) != 0) return mf_set_error("mf_count_device: d_bases must be 16-byte aligned");
    MF_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;

    if (n_reads == 0 || n_bases == 0) return mf_table_adopt(ctx, k, 0, 0, nullptr, 0, nullptr, 0, out);

    // ---- K0 ----
    const uint64_t n_words = (n_bases + 31) / 32;
    mf_buf<uint32_t> vmask; MF_TRY(vmask.alloc(ctx, n_words));
    mf_buf<unsigned long long> scal; MF_TRY(scal.alloc(ctx, 8));   // [0]=n_occ [1]=scan total [2]=overflow [3]=scan total 2
    MF_HIP(hipMemsetAsync(scal.p, 0, scal.bytes(), st));
    {
        mf_ktimer t(ctx, "k_mask");
        k_mask_init<<<(unsigned)((n_words + 255) / 256), 256, 0, st>>>(vmask.p, n_words, n_bases);
        k_mask_reads<<<(unsigned)((n_reads + 255) / 256), 256, 0, st>>>(d_offsets, n_reads, k, min_len, vmask.p, &scal.p[0]);
    }
    MF_DBG(ctx, "k_mask");
    MF_HIP(hipGetLastError());
    unsigned long long n_occ = 0;
    MF_HIP(hipMemcpyAsync(&n_occ, &scal.p[0], 8, hipMemcpyDeviceToHost, st));
    MF_HIP(hipStreamSynchronize(st));
    if (n_occ == 0) return mf_table_adopt(ctx, k, 0, 0, nullptr, 0, nullptr, 0, out);

    // ---- partition plan ----
    const uint64_t target = (uint64_t)ctx->opt_part_target;
    int B = ceil_log2_u64((n_occ + target - 1) / target);
    std::vector<int> lv;
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
        k_scan<true><<<1, 1024, 0, st>>>(blockhist.p, blockstart.p, (uint64_t)nd1 * G, (uint64_t *)&scal.p[1]);
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
            k_l1_scatter<true><<<G, 1024, lds, st>>>(d_bases, n_bases, vmask.p, n_words, wpb, k, bits1, blockstart.p, G, bufA.p);
        } else {
            MF_TRY(set_lds(k_l1_scatter<false>, lds));
            k_l1_scatter<false><<<G, 1024, lds, st>>>(d_bases, n_bases, vmask.p, n_words, wpb, k, bits1, blockstart.p, G, bufA.p);
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
        size_t lds = (size_t)MF_COUNT_SLOTS * 12;
        MF_TRY(set_lds(k_count, lds));
        unsigned grid = (unsigned)std::min<uint64_t>(np, (uint64_t)ctx->n_cu * 3);
        mf_ktimer t(ctx, "k_count");
        k_count<<<grid, 256, lds, st>>>(bufA.p, cnt.p, pstart.p, plen.p, np, dcount.p, (unsigned int *)&scal.p[2]);
    }
    MF_DBG(ctx, "k_count");
    mf_buf<uint64_t> doff; MF_TRY(doff.alloc(ctx, (size_t)np + 1));
    {
        mf_ktimer t(ctx, "k_scan");
        k_scan<false><<<1, 1024, 0, st>>>(dcount.p, doff.p, np, (uint64_t *)&scal.p[3]);
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
    return mf_table_adopt(ctx, k, n_dist, n_occ, dk.take(), kb, dc.take(), cb, out);
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

extern "C" int mf_count_device(mf_ctx *ctx, const void *d_bases, const void *d_offsets, uint64_t n_reads,
                               uint64_t n_bases, int k, int min_read_len, mf_table **out) {
    if (!ctx || !out) return mf_set_error("mf_count_device: NULL argument");
    *out = nullptr;
    return mf_count_core(ctx, (const uint8_t *)d_bases, (const uint64_t *)d_offsets, n_reads, n_bases, k, min_read_len, out);
}

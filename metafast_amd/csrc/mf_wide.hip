// mf_wide.hip -- NO-REFERENCE EXTENSION: canonical k-mer counts for 32 <= k <= 63 (128-bit k-mers).
//
// The reference stops at k = 31 (one Java long per k-mer; src/tools/KmersCounterMain.java:66-73 rejects k > 31), so nothing here
// replaces a reference function and nothing here takes part in any parity claim: BASELINE.json's config 4 asks for a k = 63 leg,
// SURVEY.md 8 asks for it as a labelled extension with a checker of its own (the test suite's 128-bit CPU restatement, or_count_wide).
// Same definitions as for k <= 31, on 2k-bit numbers: first base most significant (A0 G1 C2 T3), canonical = min(forward,
// reverse complement), counts saturate at 32767, reads shorter than max(k, min_len) give nothing.
//
// Not the hot path.  Round 5 (200 M reads at k = 63: 9.1 s -> 1.98 s):
//   k_wide_kmers<true>   the k-mers per class (top 10 bits of the canonical value): the host cuts the class range into passes of about equal size
//                        that fit the device (36 bytes per occurrence) and the sort (< 2^32 entries);
//   k_wide_kmers<false>  the canonical k-mers of a pass's classes, two 64-bit words each, through an LDS stage (bases packed 2 bits in LDS, no
//                        warm-up of k - 1 bases, no byte loads in the loop);
//   mf_sort.hip          radix passes over the LEADING 32 bits only (4 of the 16 passes a full sort of 2k = 126 bits takes);
//   k_wide_finish, k_wide_big  the order inside the buckets of equal leading bits, in LDS;
//   k_wide_flags/heads/counts  run lengths = counts.
#include <cstring>
#include <memory>
#include <vector>
#include "mf_common.h"
#include "mf_count_dev.h"
#include "mf_wide.h"

int mf_sort_u64_u32(mf_ctx *ctx, const uint64_t *d_keys_in, const uint32_t *d_vals_in, uint64_t n, int bits, uint64_t *d_keys_out, uint32_t *d_vals_out);

__global__ void k_wide_mask_init(uint32_t *__restrict__ vmask, uint64_t n_words) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w < n_words) vmask[w] = 0u;
}
// bit i of the bitmap: a k-mer starts at base i (inside one read of length >= max(k, min_len))
__global__ void k_wide_mask_reads(const uint64_t *__restrict__ off, uint64_t n_reads, int k, int min_len, uint32_t *__restrict__ vmask) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const uint64_t s = off[r], e = off[r + 1], len = e - s;
    if (len < (uint64_t)k || (int64_t)len < (int64_t)min_len) return;
    uint64_t lo = s;
    const uint64_t hi = e - (uint64_t)k + 1;                                     // starts [s, hi)
    while (lo < hi) {
        const uint64_t w = lo >> 5, wend = (w + 1) << 5;
        const uint32_t b0 = (uint32_t)(lo & 31), b1 = (uint32_t)((hi < wend ? hi : wend) - (w << 5));
        const uint32_t m = (b1 == 32 ? 0xFFFFFFFFu : ((1u << b1) - 1u)) & ~((1u << b0) - 1u);
        atomicOr(&vmask[w], m);
        lo = wend;
    }
}
struct wide128 { uint64_t hi, lo; };
__device__ __forceinline__ bool wide_less(const wide128 &a, const wide128 &b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }

// ---- the canonical k-mers of the reads (round 5) ----
// A workgroup takes 8192 positions at a time: their bases (+ 64 that follow) come in 16 at a time, coalesced, and are kept in LDS two bits each,
// first base on top.  A thread owns 32 positions = the 64 + 128 bits it holds in three registers: the k-mer at its first position is cut out of
// them, the reverse complement is that number reversed, and 31 steps roll both on -- no warm-up of k - 1 bases and no byte loads in the loop (the
// first version: one thread per 32 positions rolling 94 bytes from HBM, twice per pass, 96 ms per launch at 3e10 bases; this one: see DESIGN.md 4.4).
// HIST: the k-mers per class (the top WG_CB bits of the canonical value) -- the host cuts the class range into passes that fit.
// otherwise: the k-mers of classes [c0, c1) go out through an LDS stage, a run per flush at a place taken from *cursor (the order inside a pass
// does not matter: it is sorted next).
#define WG_T 256
#define WG_TILE (WG_T * 32)
#define WG_PK ((WG_TILE + 64) / 16 + 2)
#define WG_CB 10
#define WG_CAP 2048
__device__ __forceinline__ uint32_t wide_pack16(const uint8_t *__restrict__ bases, uint64_t q, uint64_t n_bases, bool aligned) {
    uint32_t w[4];
    if (aligned && q + 16 <= n_bases) { const uint4 v = *reinterpret_cast<const uint4 *>(bases + q); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
    else {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            w[j] = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) { const uint64_t a = q + 4 * j + b; if (a < n_bases) w[j] |= (uint32_t)bases[a] << (8 * b); }
        }
    }
    uint32_t out = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {                                                // A0 G1 C2 T3 of four bytes at once; the first base ends up on top
        const uint32_t x = (w[j] >> 1) & 0x03030303u, x0 = x & 0x01010101u, x1 = (x >> 1) & 0x01010101u;
        const uint32_t c = ((x0 ^ x1) << 1) | x1;
        out = (out << 8) | ((c * 0x40100401u) >> 24);
    }
    return out;
}
// the order of the 2-bit groups of a word reversed
__device__ __forceinline__ uint64_t wide_rev2(uint64_t x) {
    const uint64_t y = __brevll(x);
    return ((y & 0x5555555555555555ull) << 1) | ((y >> 1) & 0x5555555555555555ull);
}
__device__ __forceinline__ uint32_t wide_class(const wide128 &cn, int hb, int bits) {   // the top `bits` (<= 32) bits of the 2k-bit number
    return hb >= bits ? (uint32_t)(cn.hi >> (hb - bits)) : (uint32_t)((hb ? cn.hi << (bits - hb) : 0ull) | (cn.lo >> (64 - (bits - hb))));
}
template <bool HIST>
__global__ __launch_bounds__(WG_T) void k_wide_kmers(const uint8_t *__restrict__ bases, uint64_t n_bases, const uint32_t *__restrict__ vmask, uint64_t n_words, int k,
                                                     uint32_t c0, uint32_t c1, unsigned long long *__restrict__ class_hist, uint64_t *__restrict__ out_hi,
                                                     uint64_t *__restrict__ out_lo, unsigned long long *__restrict__ cursor) {
    __shared__ uint32_t pk[WG_PK];
    __shared__ uint64_t st_hi[HIST ? 1 : WG_CAP], st_lo[HIST ? 1 : WG_CAP];
    __shared__ uint32_t hist[HIST ? (1 << WG_CB) : 1];
    __shared__ uint32_t s_n;
    __shared__ unsigned long long s_base;
    const uint32_t tid = threadIdx.x;
    const bool aligned = (reinterpret_cast<uintptr_t>(bases) & 15u) == 0;
    const int hb = 2 * k - 64, sh = 128 - 2 * k;                                  // bits of the k-mer in the high word (0 .. 62); 2 <= sh <= 64
    const uint64_t hmask = (1ull << hb) - 1ull;
    if (HIST) for (uint32_t i = tid; i < (1u << WG_CB); i += WG_T) hist[i] = 0;
    if (tid == 0) s_n = 0;
    const uint64_t n_tiles = (n_words + WG_T - 1) / WG_T;
    auto flush = [&]() {                                                         // (every thread, after a barrier)
        const uint32_t n = s_n;
        if (tid == 0) s_base = atomicAdd(cursor, (unsigned long long)n);
        __syncthreads();
        const unsigned long long at = s_base;
        for (uint32_t s = tid; s < n; s += WG_T) { out_hi[at + s] = st_hi[s]; out_lo[at + s] = st_lo[s]; }
        __syncthreads();
        if (tid == 0) s_n = 0;
        __syncthreads();
    };
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        __syncthreads();
        const uint64_t q0 = tile * WG_TILE;
        for (uint32_t i = tid; i < WG_PK; i += WG_T) pk[i] = wide_pack16(bases, q0 + (uint64_t)i * 16, n_bases, aligned);
        __syncthreads();
        const uint64_t w = tile * WG_T + tid;
        const uint32_t m = w < n_words ? vmask[w] : 0u;
        wide128 fw = {0, 0}, rc = {0, 0};
        uint64_t W1 = 0, W2 = 0;
        if (m) {
            const uint32_t *q = pk + 2 * tid;
            const uint64_t W0 = ((uint64_t)q[0] << 32) | q[1];
            W1 = ((uint64_t)q[2] << 32) | q[3]; W2 = ((uint64_t)q[4] << 32) | q[5];
            if (sh == 64) { fw.hi = 0; fw.lo = W0; } else { fw.hi = W0 >> sh; fw.lo = (W0 << (64 - sh)) | (W1 >> sh); }
            const uint64_t rh = ~wide_rev2(fw.lo), rl = ~wide_rev2(fw.hi);         // the 128-bit number reversed and complemented: the k-mer's 2k bits are on top
            if (sh == 64) { rc.hi = 0; rc.lo = rh; } else { rc.hi = rh >> sh; rc.lo = (rh << (64 - sh)) | (rl >> sh); }
        }
        for (int i = 0; i < 32; i++) {
            if (m) {
                if (i) {                                                         // the base at window position i + k - 1 comes in
                    const int p = i + k - 1;
                    const uint64_t c = ((p < 64 ? W1 : W2) >> (62 - 2 * (p & 31))) & 3ull;
                    fw.hi = ((fw.hi << 2) | (fw.lo >> 62)) & hmask; fw.lo = (fw.lo << 2) | c;
                    if (hb >= 2) { rc.lo = (rc.lo >> 2) | (rc.hi << 62); rc.hi = (rc.hi >> 2) | ((3ull - c) << (hb - 2)); }
                    else rc.lo = (rc.lo >> 2) | ((3ull - c) << 62);              // (k = 32: the k-mer is the low word)
                }
                if ((m >> i) & 1u) {
                    const wide128 &cn = wide_less(rc, fw) ? rc : fw;
                    const uint32_t cls = wide_class(cn, hb, WG_CB);
                    if (HIST) atomicAdd(&hist[cls], 1u);
                    else if (cls >= c0 && cls < c1) { const uint32_t s = atomicAdd(&s_n, 1u); st_hi[s] = cn.hi; st_lo[s] = cn.lo; }
                }
            }
            if (!HIST && (i & 3) == 3) {                                         // at most 4 x 256 more before the next look: the stage never overflows
                __syncthreads();
                const uint32_t staged = s_n;
                __syncthreads();                                                 // (nobody adds before everybody has looked)
                if (staged > WG_CAP - 4 * WG_T) flush();
            }
        }
    }
    __syncthreads();
    if (HIST) { for (uint32_t i = tid; i < (1u << WG_CB); i += WG_T) if (hist[i]) atomicAdd(&class_hist[i], (unsigned long long)hist[i]); }
    else if (s_n) flush();
}

// ---- order inside the leading-bits buckets (round 5) ----
// After the radix passes over the leading WF_BITS bits of the 2k-bit numbers, equal leading bits sit together ("buckets": a few distinct k-mers and
// their repeats) -- the remaining 2k - WF_BITS bits never go through HBM again (12 of 16 radix passes saved at k = 63):
//   k_wide_finish      a workgroup takes WF_TILE entries + the WF_BIG that follow into LDS, numbers the buckets (a scan over the bucket heads) and
//                      orders the buckets of <= WF_BIG entries that START in its tile: every entry looks its k-mer up in a hash table of the tile (a
//                      slot holds the index of the first entry that came with the k-mer, its "representative"; the keys stay where they are) and takes
//                      a number among its equals; only the representatives look at their bucket, and only at its other representatives (a bitmap),
//                      adding up the counts of the smaller ones; place = bucket start + entries below + number among equals.  (Walking the bucket
//                      with every entry: 552 ms per 4.4e9 entries, 66 ms of it not the walk, profiles/r05ah_finish_ablation.txt.)  Larger buckets
//                      (abundant k-mers and their error variants: 40 % of the entries at 200 M reads, 616 entries and 90 distinct k-mers on
//                      average) go on a list;
//   k_wide_big<64>     ONE WAVE per listed bucket (no workgroup barriers, 12 buckets in flight per CU): streams the bucket -- it may run over many
//                      tiles -- through a hash table of 512 slots in LDS (distinct k-mers and counts), ranks the distinct k-mers, writes the runs;
//                      more than 320 distinct k-mers: on to the next list;
//   k_wide_big<256>    the same with a workgroup and 2048 slots; more than `dlimit` (<= 1280) distinct k-mers (low-complexity reads): the last list --
//                      the host gathers those buckets, sorts them with the full radix sort and puts them back (or sorts the whole pass, if they
//                      hold more than a quarter of it).
#define WF_BITS 32
#define WF_T 256
#define WF_PER 3
#define WF_N (WF_T * WF_PER)          // 768 entries in LDS
#define WF_BIG 256
#define WF_TILE (WF_N - WF_BIG)       // 512
#define WF_TAB 1024                   // slots of the tile's table (<= WF_N distinct k-mers)
#define WF_LIST 256                   // large buckets a tile can own (512 / 2)
#define WF_LOCK 0xFFFFFFFFu
#define WF_EMPTY 0xFFFFFFFFu
__device__ __forceinline__ uint32_t wide_hash(const wide128 &x) { return (uint32_t)((x.lo * 0x9E3779B97F4A7C15ull ^ x.hi * 0xC2B2AE3D27D4EB4Full) >> 40); }
__global__ __launch_bounds__(WF_T) void k_wide_finish(const uint64_t *__restrict__ hi, const uint64_t *__restrict__ lo, uint64_t n, int hb_tb, uint32_t big, int dbg,
                                                      uint64_t *__restrict__ ohi, uint64_t *__restrict__ olo, unsigned long long *__restrict__ tile_big, uint32_t tile_cap,
                                                      uint32_t *__restrict__ tile_cnt) {
    __shared__ uint64_t l_hi[WF_N], l_lo[WF_N];
    __shared__ uint32_t bstart[WF_N + 1], tab[WF_TAB], ecnt[WF_N], reps[WF_N], repbits[WF_N / 32];
    __shared__ uint16_t mybig[WF_LIST];
    __shared__ uint32_t scratch[17];
    __shared__ uint32_t s_cont0, s_open, s_nbig, s_nreps;
    const int hb = hb_tb & 255, tb = hb_tb >> 8;                                  // bits of the k-mer in the high word; leading bits the array is ascending in
    const uint32_t tid = threadIdx.x;
    const uint64_t base = (uint64_t)blockIdx.x * WF_TILE;
    const uint32_t cnt = (uint32_t)(n - base < WF_TILE ? n - base : WF_TILE), N = (uint32_t)(n - base < WF_N ? n - base : WF_N);
    for (uint32_t i = tid; i < N; i += WF_T) { l_hi[i] = hi[base + i]; l_lo[i] = lo[base + i]; }
    for (uint32_t i = tid; i < WF_TAB; i += WF_T) tab[i] = WF_EMPTY;
    for (uint32_t i = tid; i < WF_N; i += WF_T) ecnt[i] = 0;
    if (tid < WF_N / 32) repbits[tid] = 0;
    if (tid == 0) { s_nbig = 0; s_nreps = 0; }
    __syncthreads();
    auto pre = [&](uint32_t i) { const wide128 x = {l_hi[i], l_lo[i]}; return wide_class(x, hb, tb); };
    if (tid == 0) {
        const wide128 p = {base ? hi[base - 1] : 0ull, base ? lo[base - 1] : 0ull};
        s_cont0 = base && wide_class(p, hb, tb) == pre(0);
        const wide128 q = {base + N < n ? hi[base + N] : 0ull, base + N < n ? lo[base + N] : 0ull};
        s_open = base + N < n && wide_class(q, hb, tb) == pre(N - 1);
    }
    // bucket numbers: entry i of thread t's run [t * WF_PER, ..) is a head when its leading bits differ from the entry before (entry 0 always)
    uint32_t heads = 0, hm = 0;
    const uint32_t i0 = tid * WF_PER;
#pragma unroll
    for (uint32_t j = 0; j < WF_PER; j++) {
        const uint32_t i = i0 + j;
        if (i < N && (i == 0 || pre(i) != pre(i - 1))) { hm |= 1u << j; heads++; }
    }
    uint32_t n_buckets;
    uint32_t id = mf_block_excl_scan(heads, scratch, &n_buckets);                // the number of the first head of this thread
#pragma unroll
    for (uint32_t j = 0; j < WF_PER; j++) if ((hm >> j) & 1u) bstart[id++] = i0 + j;
    if (tid == 0) bstart[n_buckets] = N;
    __syncthreads();
    uint32_t b = id - heads;                                                     // heads before this thread's run
    uint32_t my_rep[WF_PER], my_at[WF_PER];                                      // per entry: its representative (WF_EMPTY: not placed here); bucket start + number among equals
#pragma unroll
    for (uint32_t j = 0; j < WF_PER; j++) {
        const uint32_t i = i0 + j;
        my_rep[j] = WF_EMPTY; my_at[j] = 0;
        if (i >= N) continue;
        if ((hm >> j) & 1u) b++;                                                 // entry i is in bucket b - 1
        const uint32_t me = b - 1, s = bstart[me], e = bstart[me + 1];
        if (s >= cnt || (me == 0 && s_cont0)) continue;                          // another tile's bucket
        if (e - s > big || (me == n_buckets - 1 && s_open)) {                    // (open: it runs on past what is in LDS, i.e. more than WF_BIG entries)
            if (i == s) mybig[atomicAdd(&s_nbig, 1u)] = (uint16_t)s;             // (at most WF_TILE / 2 of them: big >= 1)
            continue;
        }
        const wide128 x = {l_hi[i], l_lo[i]};
        uint32_t h = wide_hash(x) & (WF_TAB - 1), rep;
        for (;;) {
            uint32_t v = __hip_atomic_load(&tab[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (v == WF_EMPTY) {
                v = atomicCAS(&tab[h], WF_EMPTY, i);
                if (v == WF_EMPTY) { rep = i; reps[atomicAdd(&s_nreps, 1u)] = (me << 16) | i; atomicOr(&repbits[i >> 5], 1u << (i & 31u)); break; }
            }
            if (l_hi[v] == x.hi && l_lo[v] == x.lo) { rep = v; break; }
            h = (h + 1) & (WF_TAB - 1);
        }
        my_rep[j] = rep;
        my_at[j] = s + atomicAdd(&ecnt[rep], 1u);
    }
    __syncthreads();
    // the representatives: the entries of the bucket below them = the counts of the bucket's smaller representatives (a bucket of 200 repeats of
    // 3 k-mers costs 3 x 3 comparisons, not 3 x 200).  ecnt[]: count in the low half, that number in the high half
    const uint32_t nreps = s_nreps, nbig = (dbg & 1) ? 0u : s_nbig;
    if (tid == 0) tile_cnt[blockIdx.x] = nbig;                                     // (no list with one cursor: 2.3e6 atomics on one address per pass)
    if (!(dbg & 2)) for (uint32_t r = tid; r < nreps; r += WF_T) {
        const uint32_t i = reps[r] & 0xFFFFu, me = reps[r] >> 16, s = bstart[me], e = bstart[me + 1];
        const wide128 x = {l_hi[i], l_lo[i]};
        uint32_t below = 0;
        const uint32_t w0 = s >> 5, w1 = (e - 1) >> 5;
        for (uint32_t w = w0; w <= w1; w++) {
            uint32_t m = repbits[w];
            if (w == w0) m &= ~0u << (s & 31u);
            if (w == w1 && (e & 31u)) m &= (1u << (e & 31u)) - 1u;
            while (m) {
                const uint32_t o = w * 32 + (uint32_t)__builtin_ctz(m);
                m &= m - 1;
                const wide128 y = {l_hi[o], l_lo[o]};
                if (wide_less(y, x)) below += ecnt[o] & 0xFFFFu;
            }
        }
        ecnt[i] = (below << 16) | (ecnt[i] & 0xFFFFu);
    }
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < WF_PER; j++) {
        if (my_rep[j] == WF_EMPTY) continue;
        const uint32_t i = i0 + j;
        const uint64_t at = base + my_at[j] + (ecnt[my_rep[j]] >> 16);
        ohi[at] = l_hi[i]; olo[at] = l_lo[i];
    }
    for (uint32_t q = tid; q < nbig; q += WF_T) tile_big[(uint64_t)blockIdx.x * tile_cap + q] = base + mybig[q];       // (nbig <= tile_cap = (WF_TILE - 1) / (big + 1) + 1)
}
// the tiles' large buckets as one list
__global__ void k_wide_big_list(const unsigned long long *__restrict__ tile_big, uint32_t tile_cap, const uint32_t *__restrict__ tile_cnt, const uint64_t *__restrict__ toff,
                                uint64_t n_tiles, unsigned long long *__restrict__ list) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    const uint32_t c = tile_cnt[t];
    for (uint32_t q = 0; q < c; q++) list[toff[t] + q] = tile_big[t * tile_cap + q];
}

// One workgroup of T threads (T = 64: one wave) per listed bucket: tcnt[] 0 = free, WF_LOCK = being written, else the count of the k-mer in the slot.
// stats: 128 stripes of 16 words (a stripe per cache line: 1.1e7 buckets adding to ONE line cost more than the kernel), [0] entries, [1] buckets,
// [2] the largest, [3] distinct k-mers (summed)
template <int T, int SLOTS>
__global__ __launch_bounds__(T) void k_wide_big(const uint64_t *__restrict__ hi, const uint64_t *__restrict__ lo, uint64_t n, int hb_tb, const unsigned long long *__restrict__ list,
                                                uint32_t dlimit, uint64_t *__restrict__ ohi, uint64_t *__restrict__ olo, unsigned long long *__restrict__ over_list,
                                                unsigned int *__restrict__ n_over, unsigned long long *__restrict__ stats, int dbg) {
    constexpr int DCAP = SLOTS * 7 / 8;                                              // dlimit + 2 T <= DCAP: the table never fills
    __shared__ uint64_t t_hi[SLOTS], t_lo[SLOTS];
    __shared__ uint32_t tcnt[SLOTS];
    __shared__ uint64_t d_hi[DCAP], d_lo[DCAP];                                      // the distinct k-mers side by side (the ranking loop reads them in step, no slot numbers in between) ...
    __shared__ uint32_t d_cnt[DCAP];                                                 // ... and their counts; later: where the run of the k-mer of rank r starts
    __shared__ uint16_t order[DCAP];                                                 // rank -> place in d_hi / d_lo
    __shared__ uint32_t s_ndist, s_nocc, s_stop;
    __shared__ unsigned long long s_end;
    const int hb = hb_tb & 255, tb = hb_tb >> 8;
    const uint32_t tid = threadIdx.x;
    const uint64_t start = list[blockIdx.x];
    for (uint32_t i = tid; i < SLOTS; i += T) tcnt[i] = 0;
    if (tid == 0) { s_ndist = 0; s_nocc = 0; s_stop = 0; s_end = n; }
    const wide128 first = {hi[start], lo[start]};
    const uint32_t p = wide_class(first, hb, tb);
    wide128 cur[2], nxt[2];
    auto fetch = [&](wide128 *x, uint64_t pos) {
#pragma unroll
        for (int c = 0; c < 2; c++) { const uint64_t e = pos + (uint64_t)c * T + tid; x[c].hi = 0; x[c].lo = 0; if (e < n) { x[c].hi = hi[e]; x[c].lo = lo[e]; } }
    };
    fetch(cur, start);
    __syncthreads();
    for (uint64_t pos = start;; pos += 2 * T) {                                      // two entries per thread, and the next two on their way
        fetch(nxt, pos + 2 * T);
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const uint64_t e = pos + (uint64_t)c * T + tid;
            const bool same = e < n && wide_class(cur[c], hb, tb) == p;
            if (!same) { if (e < n) atomicMin(&s_end, (unsigned long long)e); s_stop = 1; }
            uint32_t slot = wide_hash(cur[c]) & (SLOTS - 1);
            bool done = !same || (dbg & 16);
            // the lanes of a wave that hold the k-mer of its first lane (twice: the first of the rest) send ONE of them with their number -- most of a
            // large bucket is one abundant k-mer, and 64 lanes adding to one LDS word take turns (50 of 108 ms were the inserts, r05ar)
            uint32_t weight = 1;
            bool led = false;
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const unsigned long long act = __ballot(!done && !led);
                if (!act) break;                                                     // (uniform)
                const int leader = __ffsll((long long)act) - 1;
                const uint64_t lh = (uint64_t)__shfl((unsigned long long)cur[c].hi, leader), ll = (uint64_t)__shfl((unsigned long long)cur[c].lo, leader);
                const bool eq = !done && !led && cur[c].hi == lh && cur[c].lo == ll;
                const unsigned long long m = __ballot(eq);
                if (eq) { if ((int)(tid & 63u) == leader) { weight = (uint32_t)__popcll(m); led = true; } else done = true; }
            }
            while (!done) {
                const uint32_t v = __hip_atomic_load(&tcnt[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (v == 0) {
                    if (atomicCAS(&tcnt[slot], 0u, WF_LOCK) == 0u) {
                        t_hi[slot] = cur[c].hi; t_lo[slot] = cur[c].lo;
                        __hip_atomic_store(&tcnt[slot], weight, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        atomicAdd(&s_ndist, 1u);
                        done = true;
                    }
                } else if (v != WF_LOCK) {
                    if (t_hi[slot] == cur[c].hi && t_lo[slot] == cur[c].lo) { atomicAdd(&tcnt[slot], weight); done = true; }
                    else slot = (slot + 1) & (SLOTS - 1);
                }
            }
        }
        __syncthreads();
        const uint32_t nd = s_ndist, stop = s_stop;
        __syncthreads();
        if (nd > dlimit) { if (tid == 0) over_list[atomicAdd(n_over, 1u)] = start; return; }     // (uniform)
        if (stop) break;
        cur[0] = nxt[0]; cur[1] = nxt[1];
    }
    // (a bucket that ends inside the first of the two chunks: the second chunk's entries are another bucket's and were not taken -- the array is
    // ascending in the leading bits)
    const uint64_t len = s_end - start;
    for (uint32_t i = tid; i < SLOTS; i += T) {
        const uint32_t c = tcnt[i];
        if (c) { const uint32_t a = atomicAdd(&s_nocc, 1u); d_hi[a] = t_hi[i]; d_lo[a] = t_lo[i]; d_cnt[a] = c; }
    }
    __syncthreads();
    const uint32_t D = s_nocc;
    if (tid == 0) { unsigned long long *sp = stats + 16 * (blockIdx.x & 127u); atomicAdd(sp, (unsigned long long)len); atomicAdd(sp + 1, 1ull); atomicMax(sp + 2, (unsigned long long)len); atomicAdd(sp + 3, (unsigned long long)D); }
    // every distinct k-mer: how many distinct k-mers and how many entries are below it
    uint32_t my_rank[DCAP / T], my_before[DCAP / T];
#pragma unroll
    for (uint32_t c = 0; c < DCAP / T; c++) {
        const uint32_t a = tid + c * T;
        my_rank[c] = 0; my_before[c] = 0;
        if (a >= D || (dbg & 8)) continue;
        const wide128 xk = {d_hi[a], d_lo[a]};
        uint32_t rk = 0, bf = 0;
#pragma unroll 4
        for (uint32_t o = 0; o < D; o++) { const wide128 y = {d_hi[o], d_lo[o]}; const uint32_t cy = d_cnt[o]; if (wide_less(y, xk)) { rk++; bf += cy; } }
        my_rank[c] = rk; my_before[c] = bf;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t c = 0; c < DCAP / T; c++) if (tid + c * T < D) { order[my_rank[c]] = (uint16_t)(tid + c * T); d_cnt[my_rank[c]] = my_before[c]; }
    __syncthreads();
    // the bucket in order: entry j is the k-mer of the last rank whose run starts at or before j
    if (!(dbg & 4)) for (uint64_t j = tid; j < len; j += T) {
        uint32_t lo_r = 0, hi_r = D;                                                    // the answer is in [lo_r, hi_r)
        while (hi_r - lo_r > 1) { const uint32_t mid = (lo_r + hi_r) >> 1; if ((uint64_t)d_cnt[mid] <= j) lo_r = mid; else hi_r = mid; }
        const uint32_t a = order[lo_r];
        ohi[start + j] = d_hi[a]; olo[start + j] = d_lo[a];
    }
}
// the extent of a listed bucket: the first entry after `start` with other leading bits (the array is ascending in them)
__global__ void k_wide_big_extent(const uint64_t *__restrict__ hi, const uint64_t *__restrict__ lo, uint64_t n, int hb_tb, const uint64_t *__restrict__ start, uint32_t n_big,
                                  uint32_t *__restrict__ len) {
    const int hb = hb_tb & 255, tb = hb_tb >> 8;
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_big) return;
    const uint64_t s = start[j];
    const wide128 x = {hi[s], lo[s]};
    const uint32_t p = wide_class(x, hb, tb);
    uint64_t a = s + 1, b = n;                                                    // the answer is in [a, b]
    while (a < b) {
        const uint64_t mid = a + ((b - a) >> 1);
        const wide128 y = {hi[mid], lo[mid]};
        if (wide_class(y, hb, tb) == p) a = mid + 1; else b = mid;
    }
    len[j] = (uint32_t)(a - s);
}
// entry e of the gathered buckets <-> its place in the pass (BACK: sorted entries return)
template <bool BACK>
__global__ void k_wide_big_move(const uint64_t *__restrict__ start, const uint64_t *__restrict__ toff, uint32_t n_big, uint64_t total, const uint64_t *__restrict__ src_hi,
                                const uint64_t *__restrict__ src_lo, uint64_t *__restrict__ dst_hi, uint64_t *__restrict__ dst_lo) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    uint32_t a = 0, b = n_big;                                                    // the last bucket with toff <= e
    while (b - a > 1) { const uint32_t mid = (a + b) >> 1; if (toff[mid] <= e) a = mid; else b = mid; }
    const uint64_t at = start[a] + (e - toff[a]);
    if (BACK) { dst_hi[at] = src_hi[e]; dst_lo[at] = src_lo[e]; } else { dst_hi[e] = src_hi[at]; dst_lo[e] = src_lo[at]; }
}
__global__ void k_wide_flags(const uint64_t *__restrict__ hi, const uint64_t *__restrict__ lo, uint64_t n, uint32_t *__restrict__ flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (i == 0 || hi[i] != hi[i - 1] || lo[i] != lo[i - 1]) ? 1u : 0u;
}
__global__ void k_wide_heads(const uint64_t *__restrict__ hi, const uint64_t *__restrict__ lo, const uint32_t *__restrict__ flag,
                             const uint64_t *__restrict__ idx, uint64_t n, uint64_t *__restrict__ ohi, uint64_t *__restrict__ olo,
                             uint64_t *__restrict__ start) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flag[i]) { const uint64_t j = idx[i]; ohi[j] = hi[i]; olo[j] = lo[i]; start[j] = i; }
}
__global__ void k_wide_counts(const uint64_t *__restrict__ start, uint64_t nd, uint64_t n, uint16_t *__restrict__ cnt) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nd) return;
    const uint64_t c = (j + 1 < nd ? start[j + 1] : n) - start[j];
    cnt[j] = (uint16_t)(c > (uint64_t)MF_MAX_COUNT ? (uint64_t)MF_MAX_COUNT : c);
}

static unsigned wgrid(uint64_t n, unsigned bs = 256) { return (unsigned)std::min<uint64_t>((n + bs - 1) / bs, 0x7FFFFFFFull); }

// ---- the entries with count > thr of ascending arrays, order kept (the cut of IOUtils.printKmers, src/io/IOUtils.java:52-60, carried over) ----
__global__ void k_wide_keep_flags(const uint16_t *__restrict__ cnt, uint64_t n, int thr, uint32_t *__restrict__ flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (int)cnt[i] > thr ? 1u : 0u;
}
__global__ void k_wide_keep_move(const uint64_t *__restrict__ hi, const uint64_t *__restrict__ lo, const uint16_t *__restrict__ cnt, const uint32_t *__restrict__ flag,
                                 const uint64_t *__restrict__ idx, uint64_t n, uint64_t *__restrict__ ohi, uint64_t *__restrict__ olo, uint16_t *__restrict__ ocnt) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flag[i]) return;
    const uint64_t j = idx[i];
    ohi[j] = hi[i]; olo[j] = lo[i]; ocnt[j] = cnt[i];
}
int mf_wide_compact(mf_ctx *ctx, const uint64_t *hi, const uint64_t *lo, const uint16_t *cnt, uint64_t n, int thr, mf_wtable::piece &out) {
    hipStream_t st = ctx->stream;
    mf_buf<uint32_t> flag; mf_buf<uint64_t> idx, tot;
    MF_TRY(flag.alloc(ctx, n)); MF_TRY(idx.alloc(ctx, n + 1)); MF_TRY(tot.alloc(ctx, 1));
    uint64_t keep = 0;
    if (n) {
        k_wide_keep_flags<<<wgrid(n), 256, 0, st>>>(cnt, n, thr, flag.p);
        MF_TRY(mf_scan<1>(ctx, flag.p, idx.p, n, tot.p));
        MF_HIP(hipMemcpyAsync(&keep, tot.p, 8, hipMemcpyDeviceToHost, st));
        MF_HIP(hipStreamSynchronize(st));
    }
    MF_TRY(out.hi.alloc(ctx, keep)); MF_TRY(out.lo.alloc(ctx, keep)); MF_TRY(out.cnt.alloc(ctx, keep));
    if (keep) k_wide_keep_move<<<wgrid(n), 256, 0, st>>>(hi, lo, cnt, flag.p, idx.p, n, out.hi.p, out.lo.p, out.cnt.p);
    MF_HIP(hipStreamSynchronize(st));
    out.n = keep;
    return MF_OK;
}

static int count_wide_impl(mf_ctx *ctx, const void *d_bases, const void *d_offsets, uint64_t n_reads, uint64_t n_bases, int k, int min_read_len, int threshold,
                           mf_wtable **out) {
    mf_range rng_("mf:count_wide");
    if (!ctx || !out) return mf_set_error("mf_count_wide_device: NULL argument");
    *out = nullptr;
    if (k < 32 || k > 63) return mf_set_error("mf_count_wide_device: 32 <= k <= 63 (k <= 31: mf_count_device)");
    if (n_reads && (!d_bases || !d_offsets)) return mf_set_error("mf_count_wide_device: NULL argument");
    MF_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    mf_wtable *t = new mf_wtable();
    t->ctx = ctx; t->k = k;
    *out = t;
    if (!n_reads || !n_bases) return MF_OK;
    auto fail = [&](int rc) { delete t; *out = nullptr; return rc; };
    const uint64_t n_words = (n_bases + 31) / 32;
    const int hb = 2 * k - 64;
    mf_buf<uint32_t> vmask; mf_buf<unsigned long long> chist, cursor; mf_buf<uint64_t> tot;
    if (vmask.alloc(ctx, n_words) < 0 || chist.alloc(ctx, (1u << WG_CB) + 2) < 0 || cursor.alloc(ctx, 8) < 0 || tot.alloc(ctx, 2) < 0) return fail(MF_ERR);
    const unsigned gen_grid = (unsigned)std::min<uint64_t>((n_words + WG_T - 1) / WG_T, (uint64_t)ctx->n_cu * 8);
    std::vector<unsigned long long> h_class(1u << WG_CB);
    {
        mf_ktimer tm(ctx, "k_wide_mask");
        k_wide_mask_init<<<wgrid(n_words), 256, 0, st>>>(vmask.p, n_words);
        k_wide_mask_reads<<<wgrid(n_reads), 256, 0, st>>>((const uint64_t *)d_offsets, n_reads, k, min_read_len, vmask.p);
    }
    // (assembled sequences -- the cutter table's input, filtered by length: ComponentCutterMain.java:81 -- hold every k-mer once: LDS tables merge nothing,
    // and the sort path counts the 3.9e8 k-mers of the 200 M-read sample's unitigs in 53 ms where the record path takes 144; wide_skm = 2: always)
    if (ctx->opt_wide_skm == 2 || (ctx->opt_wide_skm && min_read_len <= 0)) {
        // the record path (mf_wskm.hip): super-k-mer records + LDS tables; 1 = not an input for it (tiny, no room): the sort path below
        const int rc = mf_count_wide_skm(ctx, (const uint8_t *)d_bases, (const uint64_t *)d_offsets, n_reads, n_bases, k, min_read_len, threshold, vmask.p, n_words, t);
        if (rc < 0) return fail(rc);
        if (rc == 0) return MF_OK;
    }
    {
        mf_ktimer tm(ctx, "k_wide_kmers");
        if (hipMemsetAsync(chist.p, 0, sizeof(unsigned long long) << WG_CB, st) != hipSuccess) return fail(mf_set_error("mf_count_wide_device: memset failed"));
        k_wide_kmers<true><<<gen_grid, WG_T, 0, st>>>((const uint8_t *)d_bases, n_bases, vmask.p, n_words, k, 0u, 0u, chist.p, nullptr, nullptr, nullptr);
    }
    if (hipMemcpyAsync(h_class.data(), chist.p, sizeof(unsigned long long) << WG_CB, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return fail(mf_set_error("mf_count_wide_device: %s", hipGetErrorString(hipGetLastError())));
    uint64_t n_occ = 0;
    for (unsigned long long c : h_class) n_occ += c;
    t->n_occ = n_occ;
    if (!n_occ) return MF_OK;
    // passes: each takes the k-mers of a range of classes (the top WG_CB bits of the canonical value), as many classes as fit: under 4e9
    // occurrences (the sort) and 36 bytes of device memory per occurrence (two pairs of 8-byte arrays that are each other's ping-pong in the
    // sort; the run-length arrays take the place of one pair).  Option wide_passes: about that many passes (tests).
    uint64_t per_pass;
    if (ctx->opt_wide_passes > 0) per_pass = std::max<uint64_t>(1, (n_occ + (uint64_t)ctx->opt_wide_passes - 1) / (uint64_t)ctx->opt_wide_passes);
    else {
        size_t fr = 0, total = 0;
        if (hipMemGetInfo(&fr, &total) != hipSuccess) return fail(mf_set_error("mf_count_wide_device: hipMemGetInfo failed"));
        per_pass = std::min<uint64_t>(4000000000ull, std::max<uint64_t>(1ull << 20, (uint64_t)(((double)fr + (double)mf_arena_idle(ctx)) * 0.6 / 36.0)));
    }
    const uint64_t target = (n_occ + (n_occ + per_pass - 1) / per_pass - 1) / ((n_occ + per_pass - 1) / per_pass);   // passes of about equal size
    typedef mf_wtable::piece piece;
    std::vector<std::unique_ptr<piece>> &pieces = t->pieces;
    uint64_t nd_total = 0, occ_seen = 0, n_kept = 0;
    // the leading bits the radix passes order: 32 -- or the high word's 24 .. 31 bits alone (k = 44 .. 47: a pass over a few bits of the low word saved)
    const int tb = (hb >= 24 && hb < WF_BITS) ? hb : WF_BITS, hb_tb = hb | (tb << 8);
    const uint32_t big = (uint32_t)std::min<int64_t>(WF_BIG, std::max<int64_t>(1, ctx->opt_wide_big_bucket));
    const uint32_t dlimit = (uint32_t)std::min<int64_t>(1280, std::max<int64_t>(1, ctx->opt_wide_distinct));
    for (uint32_t c0 = 0; c0 < (1u << WG_CB);) {
        uint32_t c1 = c0;
        uint64_t n_p = 0;
        while (c1 < (1u << WG_CB) && (c1 == c0 || (n_p + h_class[c1] <= per_pass && n_p < target))) n_p += h_class[c1++];
        const uint32_t cfirst = c0;
        c0 = c1;
        occ_seen += n_p;
        if (!n_p) continue;
        if (n_p >= (1ull << 32)) return fail(mf_set_error("mf_count_wide_device: one class of canonical k-mers holds more than 2^32 occurrences"));
        mf_buf<uint64_t> h0, l0, h1, l1;
        if (h0.alloc(ctx, n_p) < 0 || l0.alloc(ctx, n_p) < 0 || h1.alloc(ctx, n_p) < 0 || l1.alloc(ctx, n_p) < 0) return fail(MF_ERR);
        {
            mf_ktimer tm(ctx, "k_wide_kmers");
            if (hipMemsetAsync(cursor.p, 0, 8, st) != hipSuccess) return fail(mf_set_error("mf_count_wide_device: memset failed"));
            k_wide_kmers<false><<<gen_grid, WG_T, 0, st>>>((const uint8_t *)d_bases, n_bases, vmask.p, n_words, k, cfirst, c1, nullptr, h0.p, l0.p, cursor.p);
        }
        // ascending (hi, lo).  a: the pair of arrays that holds the pass, b: the other pair (the sorts write both: no temporaries)
        uint64_t *ah = h0.p, *al = l0.p, *bh = h1.p, *bl = l1.p;
        auto swap_ab = [&]() { std::swap(ah, bh); std::swap(al, bl); };
        auto sort_hi = [&](int first, int bits) { int sec = 0; if (mf_sort_u64_u64_pingpong(ctx, ah, al, n_p, first, bits, bh, bl, &sec) < 0) return MF_ERR; if (sec) swap_ab(); return MF_OK; };
        auto sort_lo = [&](int first, int bits) { int sec = 0; if (mf_sort_u64_u64_pingpong(ctx, al, ah, n_p, first, bits, bl, bh, &sec) < 0) return MF_ERR; if (sec) swap_ab(); return MF_OK; };
        bool sorted = false;
        unsigned int nlist[3] = {0, 0, 0};
        unsigned long long hstat[4] = {0, 0, 0, 0};                                      // the large buckets: entries, buckets, the largest, distinct k-mers
        if (ctx->opt_wide_finish) {
            mf_ktimer tm(ctx, "k_wide_sort");
            // radix passes over the leading WF_BITS bits alone (stable LSD: the low word's share of them first), then the order inside the buckets in LDS
            const uint64_t n_tiles = (n_p + WF_TILE - 1) / WF_TILE;
            const uint32_t tile_cap = (WF_TILE - 1) / (big + 1) + 1;                  // (a listed bucket holds more than `big` entries and starts in the tile)
            mf_buf<unsigned long long> tile_big, bigs, wstats; mf_buf<unsigned int> n_big; mf_buf<uint32_t> tile_cnt; mf_buf<uint64_t> tile_off;
            if (tile_big.alloc(ctx, n_tiles * tile_cap) < 0 || tile_cnt.alloc(ctx, n_tiles) < 0 || tile_off.alloc(ctx, n_tiles + 1) < 0 || n_big.alloc(ctx, 1) < 0 || wstats.alloc(ctx, 128 * 16) < 0) return fail(MF_ERR);
            if (hipMemsetAsync(wstats.p, 0, 128 * 16 * 8, st) != hipSuccess) return fail(mf_set_error("mf_count_wide_device: memset failed"));
            if (hb >= tb) { if (sort_hi(hb - tb, tb) < 0) return fail(MF_ERR); }
            else {
                if (sort_lo(64 - (tb - hb), tb - hb) < 0) return fail(MF_ERR);
                if (hb && sort_hi(0, hb) < 0) return fail(MF_ERR);
            }
            // a: ascending in the leading bits -> b: ascending
            mf_buf<unsigned long long> list2, list3;
            unsigned int nb1 = 0, nb2 = 0, nb = 0;
            auto fetch = [&](unsigned int *v) { return hipMemcpyAsync(v, n_big.p, 4, hipMemcpyDeviceToHost, st) == hipSuccess && hipMemsetAsync(n_big.p, 0, 4, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess; };
            if (hipMemsetAsync(n_big.p, 0, 4, st) != hipSuccess) return fail(mf_set_error("mf_count_wide_device: memset failed"));
            { mf_ktimer tf(ctx, "k_wide_finish");
            k_wide_finish<<<(unsigned)n_tiles, WF_T, 0, st>>>(ah, al, n_p, hb_tb, big, (int)ctx->opt_wide_ablate, bh, bl, tile_big.p, tile_cap, tile_cnt.p); }
            if (mf_scan<1>(ctx, tile_cnt.p, tile_off.p, n_tiles, tot.p) < 0) return fail(MF_ERR);
            { uint64_t v = 0;
              if (hipMemcpyAsync(&v, tot.p, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail(mf_set_error("mf_count_wide_device: %s", hipGetErrorString(hipGetLastError())));
              nb1 = (unsigned int)v; }
            if (nb1) {
                if (bigs.alloc(ctx, nb1) < 0 || list2.alloc(ctx, nb1) < 0) return fail(MF_ERR);
                { mf_ktimer tf(ctx, "k_wide_big");
                k_wide_big_list<<<wgrid(n_tiles), 256, 0, st>>>(tile_big.p, tile_cap, tile_cnt.p, tile_off.p, n_tiles, bigs.p);
                k_wide_big<64, 512><<<nb1, 64, 0, st>>>(ah, al, n_p, hb_tb, bigs.p, std::min<uint32_t>(dlimit, 320u), bh, bl, list2.p, n_big.p, wstats.p, (int)ctx->opt_wide_ablate); }
                if (!fetch(&nb2)) return fail(mf_set_error("mf_count_wide_device: %s", hipGetErrorString(hipGetLastError())));
            }
            if (nb2) {
                if (list3.alloc(ctx, nb2) < 0) return fail(MF_ERR);
                { mf_ktimer tf(ctx, "k_wide_big2");
                k_wide_big<256, 2048><<<nb2, 256, 0, st>>>(ah, al, n_p, hb_tb, list2.p, dlimit, bh, bl, list3.p, n_big.p, wstats.p, (int)ctx->opt_wide_ablate); }
                if (!fetch(&nb)) return fail(mf_set_error("mf_count_wide_device: %s", hipGetErrorString(hipGetLastError())));
            }
            sorted = true;
            if (nb) {
                mf_buf<uint64_t> bs, toff, th0, tl0, th1, tl1; mf_buf<uint32_t> blen, dummy, dummy2;
                if (bs.alloc(ctx, nb) < 0 || toff.alloc(ctx, (uint64_t)nb + 1) < 0 || blen.alloc(ctx, nb) < 0 || dummy.alloc(ctx, nb) < 0 || dummy2.alloc(ctx, nb) < 0) return fail(MF_ERR);
                if (hipMemsetAsync(dummy.p, 0, (size_t)nb * 4, st) != hipSuccess) return fail(mf_set_error("mf_count_wide_device: memset failed"));
                if (mf_sort_u64_u32(ctx, (const uint64_t *)list3.p, dummy.p, nb, 33, bs.p, dummy2.p) < 0) return fail(MF_ERR);
                k_wide_big_extent<<<wgrid(nb), 256, 0, st>>>(ah, al, n_p, hb_tb, bs.p, nb, blen.p);
                if (mf_scan<1>(ctx, blen.p, toff.p, nb, tot.p) < 0) return fail(MF_ERR);
                uint64_t big_total = 0;
                if (hipMemcpyAsync(&big_total, tot.p, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail(mf_set_error("mf_count_wide_device: %s", hipGetErrorString(hipGetLastError())));
                ctx->n_wide_big += big_total;
                if (big_total > n_p / 4) sorted = false;                              // too many for a side sort: all of the pass below (a still holds it)
                else {
                    if (th0.alloc(ctx, big_total) < 0 || tl0.alloc(ctx, big_total) < 0 || th1.alloc(ctx, big_total) < 0 || tl1.alloc(ctx, big_total) < 0) return fail(MF_ERR);
                    k_wide_big_move<false><<<wgrid(big_total), 256, 0, st>>>(bs.p, toff.p, nb, big_total, ah, al, th0.p, tl0.p);
                    if (mf_sort_u64_u64(ctx, tl0.p, th0.p, big_total, 64, tl1.p, th1.p) < 0) return fail(MF_ERR);
                    if (hb) { if (mf_sort_u64_u64(ctx, th1.p, tl1.p, big_total, hb, th0.p, tl0.p) < 0) return fail(MF_ERR); } else { th0.swap(th1); tl0.swap(tl1); }
                    k_wide_big_move<true><<<wgrid(big_total), 256, 0, st>>>(bs.p, toff.p, nb, big_total, th0.p, tl0.p, bh, bl);
                    if (hipStreamSynchronize(st) != hipSuccess) return fail(mf_set_error("mf_count_wide_device: %s", hipGetErrorString(hipGetLastError())));
                }
            }
            {
                std::vector<unsigned long long> hs(128 * 16);
                if (hipMemcpyAsync(hs.data(), wstats.p, hs.size() * 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail(mf_set_error("mf_count_wide_device: %s", hipGetErrorString(hipGetLastError())));
                for (int i = 0; i < 128; i++) { hstat[0] += hs[16 * i]; hstat[1] += hs[16 * i + 1]; hstat[2] = std::max(hstat[2], hs[16 * i + 2]); hstat[3] += hs[16 * i + 3]; }
            }
            nlist[0] = nb1; nlist[1] = nb2; nlist[2] = nb;
            if (sorted) swap_ab();
        }
        if (!sorted) {   // LSD over all 2k bits -- by the low word, then (stable) by the high word's 2k - 64 bits
            mf_ktimer tm(ctx, "k_wide_sort");
            if (sort_lo(0, 64) < 0 || (hb && sort_hi(0, hb) < 0)) return fail(MF_ERR);
        }
        if (ah != h0.p) { h0.swap(h1); l0.swap(l1); }                                  // (h0, l0): the ascending pass
        h1.reset(); l1.reset();
        // run lengths
        mf_buf<uint32_t> flag; mf_buf<uint64_t> idx;
        if (flag.alloc(ctx, n_p) < 0 || idx.alloc(ctx, n_p + 1) < 0) return fail(MF_ERR);
        k_wide_flags<<<wgrid(n_p), 256, 0, st>>>(h0.p, l0.p, n_p, flag.p);
        if (mf_scan<1>(ctx, flag.p, idx.p, n_p, tot.p) < 0) return fail(MF_ERR);
        uint64_t nd = 0; unsigned long long emitted = 0;
        if (hipMemcpyAsync(&nd, tot.p, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipMemcpyAsync(&emitted, cursor.p, 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) return fail(mf_set_error("mf_count_wide_device: %s", hipGetErrorString(hipGetLastError())));
        ctx->n_wide_hashed += hstat[0];
        if (ctx->opt_verbose) fprintf(stderr, "[mf] count_wide: classes [%u, %u): %llu k-mers, %llu distinct; %llu in %llu buckets of more than %u entries (hash table; the largest %llu entries, %.1f distinct k-mers on average; lists: %u -> %u -> %u)\n", cfirst, c1,
                                      (unsigned long long)n_p, (unsigned long long)nd, hstat[0], hstat[1], big, hstat[2], hstat[1] ? (double)hstat[3] / (double)hstat[1] : 0.0, nlist[0], nlist[1], nlist[2]);
        if (emitted != n_p) return fail(mf_set_error("mf_count_wide_device: internal error, a pass wrote %llu of %llu k-mers", emitted, (unsigned long long)n_p));
        auto pc = std::make_unique<piece>();
        mf_buf<uint64_t> start;
        if (pc->hi.alloc(ctx, nd) < 0 || pc->lo.alloc(ctx, nd) < 0 || pc->cnt.alloc(ctx, nd) < 0 || start.alloc(ctx, nd) < 0) return fail(MF_ERR);
        {
            mf_ktimer tm(ctx, "k_wide_runs");
            k_wide_heads<<<wgrid(n_p), 256, 0, st>>>(h0.p, l0.p, flag.p, idx.p, n_p, pc->hi.p, pc->lo.p, start.p);
            k_wide_counts<<<wgrid(nd), 256, 0, st>>>(start.p, nd, n_p, pc->cnt.p);
        }
        if (hipStreamSynchronize(st) != hipSuccess) return fail(mf_set_error("mf_count_wide_device: %s", hipGetErrorString(hipGetLastError())));
        pc->n = nd; nd_total += nd;
        if (threshold >= 1) {                                                            // the cut inside the pass: the uncut piece goes back to the arena
            auto kept = std::make_unique<piece>();
            if (mf_wide_compact(ctx, pc->hi.p, pc->lo.p, pc->cnt.p, nd, threshold, *kept) < 0) return fail(MF_ERR);
            pc = std::move(kept);
        }
        n_kept += pc->n;
        pieces.push_back(std::move(pc));
    }
    if (occ_seen != n_occ) return fail(mf_set_error("mf_count_wide_device: internal error, the passes saw %llu of %llu k-mers", (unsigned long long)occ_seen, (unsigned long long)n_occ));
    t->n = n_kept; t->n_all = nd_total;
    t->cut_thr = threshold >= 1 ? threshold : 0;                                  // (a counted k-mer has count >= 1 > 0)
    return MF_OK;
}
extern "C" int mf_count_wide_device(mf_ctx *ctx, const void *d_bases, const void *d_offsets, uint64_t n_reads, uint64_t n_bases, int k, int min_read_len,
                                    mf_wtable **out) {
    return count_wide_impl(ctx, d_bases, d_offsets, n_reads, n_bases, k, min_read_len, 0, out);
}
extern "C" int mf_count_wide_device_above(mf_ctx *ctx, const void *d_bases, const void *d_offsets, uint64_t n_reads, uint64_t n_bases, int k, int min_read_len,
                                          int threshold, mf_wtable **out, uint64_t *n_distinct_all) {
    MF_TRY(count_wide_impl(ctx, d_bases, d_offsets, n_reads, n_bases, k, min_read_len, threshold, out));
    if (n_distinct_all) *n_distinct_all = (*out)->n_all;
    return MF_OK;
}
// the entries with count > threshold as a table of its own (one piece)
extern "C" int mf_wtable_filter(const mf_wtable *t, int threshold, mf_wtable **out) {
    if (!t || !out) return mf_set_error("mf_wtable_filter: NULL argument");
    *out = nullptr;
    mf_ctx *ctx = t->ctx;
    MF_HIP(hipSetDevice(ctx->device));
    auto nt = std::make_unique<mf_wtable>();
    nt->ctx = ctx; nt->k = t->k; nt->n_occ = t->n_occ; nt->n_all = t->n_all; nt->cut_thr = std::max(t->cut_thr, threshold);
    uint64_t tot = 0;
    for (auto &pc : t->pieces) {
        auto kept = std::make_unique<mf_wtable::piece>();
        MF_TRY(mf_wide_compact(ctx, pc->hi.p, pc->lo.p, pc->cnt.p, pc->n, threshold, *kept));
        tot += kept->n;
        if (kept->n) nt->pieces.push_back(std::move(kept));
    }
    nt->n = tot;
    nt->ascending = t->ascending;                         // (the cut keeps the order it finds)
    MF_TRY(mf_wtable_flatten(nt.get()));
    *out = nt.release();
    return MF_OK;
}
// a table fresh from the record path: its one piece in ascending order (the index, if any, was built over the other order)
int mf_wtable_ensure_ascending(mf_wtable *t) {
    if (t->ascending) return MF_OK;
    if (t->pieces.size() > 1) return mf_set_error("wide table: %zu pieces out of order", t->pieces.size());
    if (t->pieces.size() == 1 && t->pieces[0]->n) {
        MF_HIP(hipSetDevice(t->ctx->device));
        auto pc = std::make_unique<mf_wtable::piece>();
        mf_wtable::piece &old = *t->pieces[0];
        MF_TRY(mf_wide_order(t->ctx, t->k, old.hi.p, old.lo.p, old.cnt.p, old.n, pc.get()));
        t->pieces[0] = std::move(pc);
        t->index.reset(); t->index_mask = 0;
    }
    t->ascending = true;
    return MF_OK;
}
int mf_wtable_flatten(mf_wtable *t) {
    if (t->pieces.size() <= 1) return MF_OK;
    mf_ctx *ctx = t->ctx;
    hipStream_t st = ctx->stream;
    auto one = std::make_unique<mf_wtable::piece>();
    MF_TRY(one->hi.alloc(ctx, t->n)); MF_TRY(one->lo.alloc(ctx, t->n)); MF_TRY(one->cnt.alloc(ctx, t->n));
    uint64_t at = 0;
    for (auto &pc : t->pieces) {
        if (!pc->n) continue;
        MF_HIP(hipMemcpyAsync(one->hi.p + at, pc->hi.p, pc->n * 8, hipMemcpyDeviceToDevice, st));
        MF_HIP(hipMemcpyAsync(one->lo.p + at, pc->lo.p, pc->n * 8, hipMemcpyDeviceToDevice, st));
        MF_HIP(hipMemcpyAsync(one->cnt.p + at, pc->cnt.p, pc->n * 2, hipMemcpyDeviceToDevice, st));
        at += pc->n;
    }
    MF_HIP(hipStreamSynchronize(st));
    one->n = at;
    t->pieces.clear();
    t->pieces.push_back(std::move(one));
    return MF_OK;
}
extern "C" void mf_wtable_destroy(mf_wtable *t) { delete t; }
extern "C" int mf_wtable_stats(const mf_wtable *t, uint64_t *n_distinct, uint64_t *n_occ, int *k) {
    if (!t) return mf_set_error("wide table is NULL");
    if (n_distinct) *n_distinct = t->n;
    if (n_occ) *n_occ = t->n_occ;
    if (k) *k = t->k;
    return MF_OK;
}
extern "C" int mf_wtable_pieces(const mf_wtable *t, uint32_t *n_pieces) {
    if (!t || !n_pieces) return mf_set_error("mf_wtable_pieces: NULL argument");
    *n_pieces = (uint32_t)t->pieces.size();
    return MF_OK;
}
extern "C" int mf_wtable_piece_view(const mf_wtable *t, uint32_t i, const void **d_keys_hi, const void **d_keys_lo, const void **d_counts, uint64_t *n) {
    if (!t || !n) return mf_set_error("mf_wtable_piece_view: NULL argument");
    MF_TRY(mf_wtable_ensure_ascending(const_cast<mf_wtable *>(t)));     // (what the caller sees is ascending)
    if (i >= t->pieces.size()) return mf_set_error("mf_wtable_piece_view: piece %u of %zu", i, t->pieces.size());
    const auto &pc = *t->pieces[i];
    if (d_keys_hi) *d_keys_hi = pc.hi.p;
    if (d_keys_lo) *d_keys_lo = pc.lo.p;
    if (d_counts) *d_counts = pc.cnt.p;
    *n = pc.n;
    return MF_OK;
}
// ascending 2k-bit k-mers as (high word, low word), counts; capacity in entries (NULL arrays: only *n)
extern "C" int mf_wtable_export(const mf_wtable *t, uint64_t *keys_hi, uint64_t *keys_lo, uint16_t *counts, uint64_t capacity, uint64_t *n) {
    if (!t || !n) return mf_set_error("mf_wtable_export: NULL argument");
    *n = t->n;
    if (!keys_hi && !keys_lo && !counts) return MF_OK;
    if (capacity < t->n) return mf_set_error("mf_wtable_export: capacity %llu < %llu entries", (unsigned long long)capacity, (unsigned long long)t->n);
    if (!t->n) return MF_OK;
    MF_HIP(hipSetDevice(t->ctx->device));
    MF_TRY(mf_wtable_ensure_ascending(const_cast<mf_wtable *>(t)));
    uint64_t at = 0;
    for (auto &pc : t->pieces) {
        if (!pc->n) continue;
        if (keys_hi) MF_HIP(hipMemcpy(keys_hi + at, pc->hi.p, pc->n * 8, hipMemcpyDeviceToHost));
        if (keys_lo) MF_HIP(hipMemcpy(keys_lo + at, pc->lo.p, pc->n * 8, hipMemcpyDeviceToHost));
        if (counts) MF_HIP(hipMemcpy(counts + at, pc->cnt.p, pc->n * 2, hipMemcpyDeviceToHost));
        at += pc->n;
    }
    return MF_OK;
}

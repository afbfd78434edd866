// mf_wide.hip -- NO-REFERENCE EXTENSION: canonical k-mer counts for 32 <= k <= 63 (128-bit k-mers).
//
// The reference stops at k = 31 (one Java long per k-mer; src/tools/KmersCounterMain.java:66-73 rejects k > 31), so nothing here
// replaces a reference function and nothing here takes part in any parity claim: BASELINE.json's config 4 asks for a k = 63 leg,
// SURVEY.md 8 asks for it as a labelled extension with a checker of its own (the test suite's 128-bit CPU restatement, or_count_wide).
// Same definitions as for k <= 31, on 2k-bit numbers: first base most significant (A0 G1 C2 T3), canonical = min(forward,
// reverse complement), counts saturate at 32767, reads shorter than max(k, min_len) give nothing.
//
// Not the hot path.  Round 5 (200 M reads at k = 63: 9.1 s -> 2.8 s):
//   k_wide_kmers<true>   the k-mers per class (top 10 bits of the canonical value): the host cuts the class range into passes of about equal size
//                        that fit the device (48 bytes per occurrence) and the sort (< 2^31 entries);
//   k_wide_kmers<false>  the canonical k-mers of a pass's classes, two 64-bit words each, through an LDS stage (bases packed 2 bits in LDS, no
//                        warm-up of k - 1 bases, no byte loads in the loop);
//   mf_sort.hip          radix passes over the LEADING 32 bits only (4 of the 16 passes a full sort of 2k = 126 bits takes);
//   k_wide_finish        the order inside the buckets of equal leading bits, in LDS;
//   k_wide_flags/heads/counts  run lengths = counts.
#include <cstring>
#include <memory>
#include <vector>
#include "mf_common.h"
#include "mf_count_dev.h"

int mf_sort_u64_u32(mf_ctx *ctx, const uint64_t *d_keys_in, const uint32_t *d_vals_in, uint64_t n, int bits, uint64_t *d_keys_out, uint32_t *d_vals_out);

struct mf_wtable {
    mf_ctx *ctx = nullptr;
    int k = 0;
    uint64_t n = 0, n_occ = 0;
    mf_buf<uint64_t> hi, lo;          // ascending (hi, lo)
    mf_buf<uint16_t> cnt;
};

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
// their repeats).  A workgroup takes WF_TILE entries + the WF_BIG that follow into LDS, numbers the buckets (a scan over the bucket heads) and
// orders the buckets that START in its tile -- the remaining 2k - WF_BITS bits never go through HBM again (12 of 16 radix passes saved at k = 63):
//   <= WF_BIG entries: every entry looks its k-mer up in a hash table of the tile (a slot holds the index of the first entry that came with the
//     k-mer, its "representative"; the keys stay where they are) and takes a number among its equals; only the representatives walk their bucket
//     and count the entries below them (walking with every entry cost 8 x as much on abundant k-mers: 552 -> 66 ms per 4.4e9 entries without
//     the walk, profiles/r05ah_finish_ablation.txt); place = bucket start + entries below + number among equals;
//   more (abundant k-mers and their error variants; a bucket may run over many tiles): the workgroup streams the bucket through a second hash
//     table in LDS (the tile's own arrays, done with by then) -- distinct k-mers and their counts --, ranks the distinct k-mers and writes the runs;
//   more than `dlimit` distinct k-mers in one bucket (low-complexity reads): put on a list -- the host gathers those buckets, sorts them with the
//     full radix sort and puts them back (or sorts the whole pass, if they hold more than a quarter of it).
#define WF_BITS 32
#define WF_T 256
#define WF_PER 5
#define WF_N (WF_T * WF_PER)          // 1280 entries in LDS
#define WF_BIG 256
#define WF_TILE (WF_N - WF_BIG)       // 1024
#define WF_TAB 2048                   // slots of the tile's table (<= WF_N distinct k-mers)
#define WF_SLOTS 1024                 // slots of the large buckets' table (in l_hi / l_lo / bstart)
#define WF_DMAX 704                   // distinct k-mers it takes (+ 256 in flight < WF_SLOTS)
#define WF_LIST 512                   // large buckets a tile can own
#define WF_LOCK 0xFFFFFFFFu
#define WF_EMPTY 0xFFFFFFFFu
__device__ __forceinline__ uint32_t wide_hash(const wide128 &x) { return (uint32_t)((x.lo * 0x9E3779B97F4A7C15ull ^ x.hi * 0xC2B2AE3D27D4EB4Full) >> 40); }
__global__ __launch_bounds__(WF_T) void k_wide_finish(const uint64_t *__restrict__ hi, const uint64_t *__restrict__ lo, uint64_t n, int hb, uint32_t big, uint32_t dlimit, int dbg,
                                                      uint64_t *__restrict__ ohi, uint64_t *__restrict__ olo, unsigned long long *__restrict__ big_start,
                                                      unsigned int *__restrict__ n_big, unsigned long long *__restrict__ n_hashed) {
    __shared__ uint64_t l_hi[WF_N], l_lo[WF_N];
    __shared__ uint32_t bstart[WF_N + 1];
    __shared__ uint32_t tab[WF_TAB], ecnt[WF_N], reps[WF_N];
    __shared__ uint16_t occ[WF_DMAX + WF_T], mybig[WF_LIST];
    __shared__ uint32_t scratch[17];
    __shared__ uint32_t s_cont0, s_open, s_nbig, s_nreps, s_ndist, s_nocc, s_stop;
    __shared__ unsigned long long s_end;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t base = (uint64_t)blockIdx.x * WF_TILE;
    const uint32_t cnt = (uint32_t)(n - base < WF_TILE ? n - base : WF_TILE), N = (uint32_t)(n - base < WF_N ? n - base : WF_N);
    for (uint32_t i = tid; i < N; i += WF_T) { l_hi[i] = hi[base + i]; l_lo[i] = lo[base + i]; }
    for (uint32_t i = tid; i < WF_TAB; i += WF_T) tab[i] = WF_EMPTY;
    for (uint32_t i = tid; i < WF_N; i += WF_T) ecnt[i] = 0;
    if (tid == 0) { s_nbig = 0; s_nreps = 0; }
    __syncthreads();
    auto pre = [&](uint32_t i) { const wide128 x = {l_hi[i], l_lo[i]}; return wide_class(x, hb, WF_BITS); };
    if (tid == 0) {
        const wide128 p = {base ? hi[base - 1] : 0ull, base ? lo[base - 1] : 0ull};
        s_cont0 = base && wide_class(p, hb, WF_BITS) == pre(0);
        const wide128 q = {base + N < n ? hi[base + N] : 0ull, base + N < n ? lo[base + N] : 0ull};
        s_open = base + N < n && wide_class(q, hb, WF_BITS) == pre(N - 1);
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
        if (e - s > big || (me == n_buckets - 1 && s_open)) {
            if (i == s) {
                const uint32_t q = atomicAdd(&s_nbig, 1u);
                if (q < WF_LIST) mybig[q] = (uint16_t)s; else big_start[atomicAdd(n_big, 1u)] = base + s;
            }
            continue;
        }
        const wide128 x = {l_hi[i], l_lo[i]};
        uint32_t h = wide_hash(x) & (WF_TAB - 1), rep;
        for (;;) {
            uint32_t v = __hip_atomic_load(&tab[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (v == WF_EMPTY) {
                v = atomicCAS(&tab[h], WF_EMPTY, i);
                if (v == WF_EMPTY) { rep = i; reps[atomicAdd(&s_nreps, 1u)] = (me << 16) | i; break; }
            }
            if (l_hi[v] == x.hi && l_lo[v] == x.lo) { rep = v; break; }
            h = (h + 1) & (WF_TAB - 1);
        }
        my_rep[j] = rep;
        my_at[j] = s + atomicAdd(&ecnt[rep], 1u);
    }
    __syncthreads();
    // the representatives: entries of the bucket below them.  ecnt[] becomes that number
    const uint32_t nreps = s_nreps;
    if (!(dbg & 2)) for (uint32_t r = tid; r < nreps; r += WF_T) {
        const uint32_t i = reps[r] & 0xFFFFu, me = reps[r] >> 16, s = bstart[me], e = bstart[me + 1];
        const wide128 x = {l_hi[i], l_lo[i]};
        uint32_t below = 0;
        for (uint32_t o = s; o < e; o++) { const wide128 y = {l_hi[o], l_lo[o]}; below += wide_less(y, x) ? 1u : 0u; }
        ecnt[i] = below;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < WF_PER; j++) {
        if (my_rep[j] == WF_EMPTY) continue;
        const uint32_t i = i0 + j;
        const uint64_t at = base + my_at[j] + ecnt[my_rep[j]];
        ohi[at] = l_hi[i]; olo[at] = l_lo[i];
    }
    __syncthreads();
    // the large buckets this tile owns, one after the other: bstart[] becomes the table's counters (0: free, WF_LOCK: being written), l_hi / l_lo its keys
    const uint32_t nbig = (dbg & 1) ? 0u : (s_nbig < WF_LIST ? s_nbig : WF_LIST);
    uint32_t *tcnt = bstart;
    for (uint32_t q = 0; q < nbig; q++) {
        const uint64_t start = base + mybig[q];
        for (uint32_t i = tid; i < WF_SLOTS; i += WF_T) tcnt[i] = 0;
        if (tid == 0) { s_ndist = 0; s_nocc = 0; s_stop = 0; s_end = n; }
        const wide128 first = {hi[start], lo[start]};
        const uint32_t p = wide_class(first, hb, WF_BITS);
        __syncthreads();
        bool over = false;
        for (uint64_t pos = start;; pos += WF_T) {
            const uint64_t e = pos + tid;
            wide128 x = {0, 0};
            bool same = false;
            if (e < n) { x.hi = hi[e]; x.lo = lo[e]; same = wide_class(x, hb, WF_BITS) == p; }
            if (!same) { if (e < n) atomicMin(&s_end, (unsigned long long)e); s_stop = 1; }
            uint32_t slot = wide_hash(x) & (WF_SLOTS - 1);
            bool done = !same;
            while (!done) {
                const uint32_t c = __hip_atomic_load(&tcnt[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (c == 0) {
                    if (atomicCAS(&tcnt[slot], 0u, WF_LOCK) == 0u) {
                        l_hi[slot] = x.hi; l_lo[slot] = x.lo;
                        __hip_atomic_store(&tcnt[slot], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        atomicAdd(&s_ndist, 1u);
                        done = true;
                    }
                } else if (c != WF_LOCK) {
                    if (l_hi[slot] == x.hi && l_lo[slot] == x.lo) { atomicAdd(&tcnt[slot], 1u); done = true; }
                    else slot = (slot + 1) & (WF_SLOTS - 1);
                }
            }
            __syncthreads();
            const uint32_t nd = s_ndist, stop = s_stop;
            __syncthreads();
            if (nd > dlimit) { over = true; break; }                                 // (dlimit + 256 < WF_SLOTS: the table never fills)
            if (stop) break;
        }
        if (over) { if (tid == 0) big_start[atomicAdd(n_big, 1u)] = start; __syncthreads(); continue; }
        const uint64_t len = s_end - start;
        for (uint32_t i = tid; i < WF_SLOTS; i += WF_T) if (tcnt[i]) occ[atomicAdd(&s_nocc, 1u)] = (uint16_t)i;
        __syncthreads();
        const uint32_t D = s_nocc;
        if (tid == 0) atomicAdd(n_hashed, (unsigned long long)len);
        for (uint32_t a = wave; a < D; a += WF_T / 64) {                             // a wave per distinct k-mer: the entries before it, then its run
            const uint32_t sa = occ[a];
            const wide128 x = {l_hi[sa], l_lo[sa]};
            unsigned long long before = 0;
            for (uint32_t o = lane; o < D; o += 64) { const uint32_t so = occ[o]; const wide128 y = {l_hi[so], l_lo[so]}; if (wide_less(y, x)) before += tcnt[so]; }
            for (int d = 32; d; d >>= 1) before += __shfl_xor(before, d);
            const uint64_t c = tcnt[sa], at = start + before;
            for (uint64_t j = lane; j < c; j += 64) { ohi[at + j] = x.hi; olo[at + j] = x.lo; }
        }
        __syncthreads();
    }
}
// the extent of a listed bucket: the first entry after `start` with other leading bits (the array is ascending in them)
__global__ void k_wide_big_extent(const uint64_t *__restrict__ hi, const uint64_t *__restrict__ lo, uint64_t n, int hb, const uint64_t *__restrict__ start, uint32_t n_big,
                                  uint32_t *__restrict__ len) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_big) return;
    const uint64_t s = start[j];
    const wide128 x = {hi[s], lo[s]};
    const uint32_t p = wide_class(x, hb, WF_BITS);
    uint64_t a = s + 1, b = n;                                                    // the answer is in [a, b]
    while (a < b) {
        const uint64_t mid = a + ((b - a) >> 1);
        const wide128 y = {hi[mid], lo[mid]};
        if (wide_class(y, hb, WF_BITS) == p) a = mid + 1; else b = mid;
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

extern "C" int mf_count_wide_device(mf_ctx *ctx, const void *d_bases, const void *d_offsets, uint64_t n_reads, uint64_t n_bases, int k, int min_read_len,
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
    if (vmask.alloc(ctx, n_words) < 0 || chist.alloc(ctx, (1u << WG_CB) + 2) < 0 || cursor.alloc(ctx, 2) < 0 || tot.alloc(ctx, 2) < 0) return fail(MF_ERR);
    const unsigned gen_grid = (unsigned)std::min<uint64_t>((n_words + WG_T - 1) / WG_T, (uint64_t)ctx->n_cu * 8);
    std::vector<unsigned long long> h_class(1u << WG_CB);
    {
        mf_ktimer tm(ctx, "k_wide_mask");
        k_wide_mask_init<<<wgrid(n_words), 256, 0, st>>>(vmask.p, n_words);
        k_wide_mask_reads<<<wgrid(n_reads), 256, 0, st>>>((const uint64_t *)d_offsets, n_reads, k, min_read_len, vmask.p);
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
    // passes: each takes the k-mers of a range of classes (the top WG_CB bits of the canonical value), as many classes as fit: under 2^31
    // occurrences (the sort) and 48 bytes of device memory per occurrence (two pairs of 8-byte arrays + the sort's temporaries; the run-length
    // arrays reuse them).  Option wide_passes: about that many passes (tests).
    uint64_t per_pass;
    if (ctx->opt_wide_passes > 0) per_pass = std::max<uint64_t>(1, (n_occ + (uint64_t)ctx->opt_wide_passes - 1) / (uint64_t)ctx->opt_wide_passes);
    else {
        size_t fr = 0, total = 0;
        if (hipMemGetInfo(&fr, &total) != hipSuccess) return fail(mf_set_error("mf_count_wide_device: hipMemGetInfo failed"));
        per_pass = std::min<uint64_t>(1ull << 31, std::max<uint64_t>(1ull << 20, (uint64_t)(((double)fr + (double)mf_arena_idle(ctx)) * 0.6 / 48.0)));
    }
    const uint64_t target = (n_occ + (n_occ + per_pass - 1) / per_pass - 1) / ((n_occ + per_pass - 1) / per_pass);   // passes of about equal size
    struct piece { mf_buf<uint64_t> hi, lo; mf_buf<uint16_t> cnt; uint64_t n = 0; };
    std::vector<std::unique_ptr<piece>> pieces;
    uint64_t nd_total = 0, occ_seen = 0;
    const uint32_t big = (uint32_t)std::min<int64_t>(WF_BIG, std::max<int64_t>(1, ctx->opt_wide_big_bucket));
    const uint32_t dlimit = (uint32_t)std::min<int64_t>(WF_DMAX, std::max<int64_t>(1, ctx->opt_wide_distinct));
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
            if (hipMemsetAsync(cursor.p, 0, 16, st) != hipSuccess) return fail(mf_set_error("mf_count_wide_device: memset failed"));
            k_wide_kmers<false><<<gen_grid, WG_T, 0, st>>>((const uint8_t *)d_bases, n_bases, vmask.p, n_words, k, cfirst, c1, nullptr, h0.p, l0.p, cursor.p);
        }
        // ascending (hi, lo).  (h0, l0) -> (h1, l1)
        bool sorted = false;
        if (ctx->opt_wide_finish) {
            mf_ktimer tm(ctx, "k_wide_sort");
            // radix passes over the leading WF_BITS bits alone (stable LSD: the low word's share of them first), then the order inside the buckets in LDS
            mf_buf<unsigned long long> bigs; mf_buf<unsigned int> n_big;
            const uint64_t big_cap = n_p / big + 2;                      // (a listed bucket holds more than `big` entries)
            if (bigs.alloc(ctx, big_cap) < 0 || n_big.alloc(ctx, 1) < 0) return fail(MF_ERR);
            if (hb >= WF_BITS) { if (mf_sort_u64_u64_range(ctx, h0.p, l0.p, n_p, hb - WF_BITS, WF_BITS, h1.p, l1.p) < 0) return fail(MF_ERR); }
            else {
                if (mf_sort_u64_u64_range(ctx, l0.p, h0.p, n_p, 64 - (WF_BITS - hb), WF_BITS - hb, l1.p, h1.p) < 0) return fail(MF_ERR);
                if (hb) { if (mf_sort_u64_u64_range(ctx, h1.p, l1.p, n_p, 0, hb, h0.p, l0.p) < 0) return fail(MF_ERR); h0.swap(h1); l0.swap(l1); }
            }
            // (h1, l1): ascending in the leading bits -> (h0, l0): ascending
            if (hipMemsetAsync(n_big.p, 0, 4, st) != hipSuccess) return fail(mf_set_error("mf_count_wide_device: memset failed"));
            { mf_ktimer tf(ctx, "k_wide_finish");
            k_wide_finish<<<(unsigned)((n_p + WF_TILE - 1) / WF_TILE), WF_T, 0, st>>>(h1.p, l1.p, n_p, hb, big, dlimit, (int)ctx->opt_wide_ablate, h0.p, l0.p, bigs.p, n_big.p, cursor.p + 1); }
            unsigned int nb = 0;
            if (hipMemcpyAsync(&nb, n_big.p, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail(mf_set_error("mf_count_wide_device: %s", hipGetErrorString(hipGetLastError())));
            sorted = true;
            if (nb) {
                if ((uint64_t)nb > big_cap) return fail(mf_set_error("mf_count_wide_device: internal error, %u large buckets among %llu k-mers", nb, (unsigned long long)n_p));
                mf_buf<uint64_t> bs, bs2, toff, th0, tl0, th1, tl1; mf_buf<uint32_t> blen, dummy, dummy2;
                if (bs.alloc(ctx, nb) < 0 || bs2.alloc(ctx, nb) < 0 || toff.alloc(ctx, (uint64_t)nb + 1) < 0 || blen.alloc(ctx, nb) < 0 || dummy.alloc(ctx, nb) < 0 || dummy2.alloc(ctx, nb) < 0) return fail(MF_ERR);
                if (hipMemsetAsync(dummy.p, 0, (size_t)nb * 4, st) != hipSuccess) return fail(mf_set_error("mf_count_wide_device: memset failed"));
                if (mf_sort_u64_u32(ctx, (const uint64_t *)bigs.p, dummy.p, nb, 33, bs.p, dummy2.p) < 0) return fail(MF_ERR);
                k_wide_big_extent<<<wgrid(nb), 256, 0, st>>>(h1.p, l1.p, n_p, hb, bs.p, nb, blen.p);
                if (mf_scan<1>(ctx, blen.p, toff.p, nb, tot.p) < 0) return fail(MF_ERR);
                uint64_t big_total = 0;
                if (hipMemcpyAsync(&big_total, tot.p, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail(mf_set_error("mf_count_wide_device: %s", hipGetErrorString(hipGetLastError())));
                ctx->n_wide_big += big_total;
                if (big_total > n_p / 4) { sorted = false; h0.swap(h1); l0.swap(l1); }          // too many for a side sort: all of the pass below, (h0, l0) again the input
                else {
                    if (th0.alloc(ctx, big_total) < 0 || tl0.alloc(ctx, big_total) < 0 || th1.alloc(ctx, big_total) < 0 || tl1.alloc(ctx, big_total) < 0) return fail(MF_ERR);
                    k_wide_big_move<false><<<wgrid(big_total), 256, 0, st>>>(bs.p, toff.p, nb, big_total, h1.p, l1.p, th0.p, tl0.p);
                    if (mf_sort_u64_u64(ctx, tl0.p, th0.p, big_total, 64, tl1.p, th1.p) < 0) return fail(MF_ERR);
                    if (hb) { if (mf_sort_u64_u64(ctx, th1.p, tl1.p, big_total, hb, th0.p, tl0.p) < 0) return fail(MF_ERR); } else { th0.swap(th1); tl0.swap(tl1); }
                    k_wide_big_move<true><<<wgrid(big_total), 256, 0, st>>>(bs.p, toff.p, nb, big_total, th0.p, tl0.p, h0.p, l0.p);
                    if (hipStreamSynchronize(st) != hipSuccess) return fail(mf_set_error("mf_count_wide_device: %s", hipGetErrorString(hipGetLastError())));
                }
            }
        }
        if (!sorted) {   // LSD over all 2k bits -- by the low word, then (stable) by the high word's 2k - 64 bits
            mf_ktimer tm(ctx, "k_wide_sort");
            if (mf_sort_u64_u64(ctx, l0.p, h0.p, n_p, 64, l1.p, h1.p) < 0) return fail(MF_ERR);
            if (hb) { if (mf_sort_u64_u64(ctx, h1.p, l1.p, n_p, hb, h0.p, l0.p) < 0) return fail(MF_ERR); } else { h0.swap(h1); l0.swap(l1); }
        }
        h1.reset(); l1.reset();
        // run lengths
        mf_buf<uint32_t> flag; mf_buf<uint64_t> idx;
        if (flag.alloc(ctx, n_p) < 0 || idx.alloc(ctx, n_p + 1) < 0) return fail(MF_ERR);
        k_wide_flags<<<wgrid(n_p), 256, 0, st>>>(h0.p, l0.p, n_p, flag.p);
        if (mf_scan<1>(ctx, flag.p, idx.p, n_p, tot.p) < 0) return fail(MF_ERR);
        uint64_t nd = 0; unsigned long long emitted_hashed[2] = {0, 0};
        unsigned long long &emitted = emitted_hashed[0];
        if (hipMemcpyAsync(&nd, tot.p, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipMemcpyAsync(emitted_hashed, cursor.p, 16, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) return fail(mf_set_error("mf_count_wide_device: %s", hipGetErrorString(hipGetLastError())));
        ctx->n_wide_hashed += emitted_hashed[1];
        if (ctx->opt_verbose) fprintf(stderr, "[mf] count_wide: classes [%u, %u): %llu k-mers, %llu distinct; %llu in buckets of more than %u entries (hash table)\n", cfirst, c1,
                                      (unsigned long long)n_p, (unsigned long long)nd, emitted_hashed[1], big);
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
        pieces.push_back(std::move(pc));
    }
    if (occ_seen != n_occ) return fail(mf_set_error("mf_count_wide_device: internal error, the passes saw %llu of %llu k-mers", (unsigned long long)occ_seen, (unsigned long long)n_occ));
    if (pieces.size() == 1) { t->hi.swap(pieces[0]->hi); t->lo.swap(pieces[0]->lo); t->cnt.swap(pieces[0]->cnt); }
    else if (nd_total) {
        if (t->hi.alloc(ctx, nd_total) < 0 || t->lo.alloc(ctx, nd_total) < 0 || t->cnt.alloc(ctx, nd_total) < 0) return fail(MF_ERR);
        uint64_t at = 0;
        for (auto &pc : pieces) {
            if (!pc->n) continue;
            if (hipMemcpyAsync(t->hi.p + at, pc->hi.p, pc->n * 8, hipMemcpyDeviceToDevice, st) != hipSuccess || hipMemcpyAsync(t->lo.p + at, pc->lo.p, pc->n * 8, hipMemcpyDeviceToDevice, st) != hipSuccess ||
                hipMemcpyAsync(t->cnt.p + at, pc->cnt.p, pc->n * 2, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(mf_set_error("mf_count_wide_device: copy failed"));
            at += pc->n;
        }
        if (hipStreamSynchronize(st) != hipSuccess) return fail(mf_set_error("mf_count_wide_device: %s", hipGetErrorString(hipGetLastError())));
    }
    t->n = nd_total;
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
// ascending 2k-bit k-mers as (high word, low word), counts; capacity in entries (NULL arrays: only *n)
extern "C" int mf_wtable_export(const mf_wtable *t, uint64_t *keys_hi, uint64_t *keys_lo, uint16_t *counts, uint64_t capacity, uint64_t *n) {
    if (!t || !n) return mf_set_error("mf_wtable_export: NULL argument");
    *n = t->n;
    if (!keys_hi && !keys_lo && !counts) return MF_OK;
    if (capacity < t->n) return mf_set_error("mf_wtable_export: capacity %llu < %llu entries", (unsigned long long)capacity, (unsigned long long)t->n);
    if (!t->n) return MF_OK;
    MF_HIP(hipSetDevice(t->ctx->device));
    if (keys_hi) MF_HIP(hipMemcpy(keys_hi, t->hi.p, t->n * 8, hipMemcpyDeviceToHost));
    if (keys_lo) MF_HIP(hipMemcpy(keys_lo, t->lo.p, t->n * 8, hipMemcpyDeviceToHost));
    if (counts) MF_HIP(hipMemcpy(counts, t->cnt.p, t->n * 2, hipMemcpyDeviceToHost));
    return MF_OK;
}

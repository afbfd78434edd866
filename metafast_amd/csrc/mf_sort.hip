// mf_sort.hip -- stable LSD radix sorts for the file seams: ascending (k-mer, count) order of the writers and of mf_table_export, the members of
// a component in ascending order, the unitigs in output order, the partition order of a loaded table.
// The reference dumps its hash map in iteration order (src/io/IOUtils.java:45-71: not reproducible); this implementation
// writes ascending k-mers so that files can be compared byte for byte.  Not on the timed step.
//
// Round 5: hand-written for wave64 (rounds 1-4 called rocPRIM's device radix sort here).  One pass per 8 key bits:
//   k_rs_count    a workgroup owns RS_BLOCK consecutive elements; it counts their digits in LDS and writes its 256 counters digit-major:
//                 hist[digit][workgroup]
//   mf_scan       exclusive prefix over hist in that order = where each workgroup's elements of each digit go
//   k_rs_scatter  the workgroup orders its elements a tile of 4096 at a time in LDS (stable: waves own consecutive ranges, rounds go forward,
//                 lanes ascend; the lanes of a round that hold the same digit find each other with eight ballots) and writes the tile out in
//                 runs of equal digits: consecutive threads, consecutive places.
// 8 + 2 (K + V) bytes per element and pass.  Everything of a call is on ctx->stream.  Measured (profiles/r05ad_sort_rate.txt, 1e8 elements):
// (k-mer, count) 62 bits 8.9 ms = 11.2 G elements/s; (u64, u64) 64 bits 9.8 ms (the library's keys-only sort of the same: 8.8 ms).
#include <cstring>
#include "mf_common.h"

#define RS_T 256                       // threads per workgroup (4 waves)
#define RS_R 16                        // rounds of 64 elements a wave takes per tile
#define RS_TILE (RS_T * RS_R)          // 4096 elements: what a workgroup orders in LDS at a time
#define RS_TILES 4                     // tiles per workgroup
#define RS_BLOCK (RS_TILE * RS_TILES)  // 16384 consecutive elements a workgroup owns

template <typename K>
__global__ __launch_bounds__(RS_T) void k_rs_count(const K *__restrict__ keys, uint64_t n, int shift, uint32_t dmask, uint32_t *__restrict__ hist, uint64_t n_blocks) {
    __shared__ uint32_t cnt[256];
    cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t e0 = (uint64_t)blockIdx.x * RS_BLOCK;
    const uint64_t e1 = e0 + RS_BLOCK < n ? e0 + RS_BLOCK : n;
    for (uint64_t e = e0 + threadIdx.x; e < e1; e += RS_T) atomicAdd(&cnt[(uint32_t)(keys[e] >> shift) & dmask], 1u);
    __syncthreads();
    hist[(uint64_t)threadIdx.x * n_blocks + blockIdx.x] = cnt[threadIdx.x];
}

// A tile of 4096 elements in wave-major order (wave w holds elements w * 1024 + round * 64 + lane): every lane ranks its elements among the
// wave's earlier ones of the same digit (eight ballots find the lanes of a round that share it, a running counter per wave and digit carries
// the rounds before), the counters of the four waves are put in digit order, the tile is written digit-sorted into LDS and leaves it in runs:
// consecutive threads write consecutive places.
template <typename K, typename V>
__global__ __launch_bounds__(RS_T) void k_rs_scatter(const K *__restrict__ keys, const V *__restrict__ vals, uint64_t n, int shift, uint32_t dmask, const uint64_t *__restrict__ offs,
                                                     uint64_t n_blocks, K *__restrict__ keys_out, V *__restrict__ vals_out) {
    __shared__ K lk[RS_TILE];
    __shared__ V lv[RS_TILE];
    __shared__ uint32_t wcnt[RS_T / 64][256];           // per wave and digit: elements so far in this tile; then: where the wave's run of the digit starts in the tile
    __shared__ uint32_t toff[256], gcur[256], tot[256];
    __shared__ uint32_t scratch[17];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, tid = threadIdx.x;
    gcur[tid] = (uint32_t)offs[(uint64_t)tid * n_blocks + blockIdx.x];      // (places are < 2^32: the callers refuse more entries)
    for (int w = 0; w < RS_T / 64; w++) wcnt[w][tid] = 0;
    __syncthreads();
    const unsigned long long below = lane ? (~0ull >> (64u - lane)) : 0ull;
    const uint64_t b0 = (uint64_t)blockIdx.x * RS_BLOCK;
    for (int t = 0; t < RS_TILES; t++) {
        const uint64_t t0 = b0 + (uint64_t)t * RS_TILE;
        if (t0 >= n) break;                                                  // (uniform)
        K key[RS_R]; V val[RS_R]; uint32_t rank[RS_R];
#pragma unroll
        for (int r = 0; r < RS_R; r++) {
            const uint64_t e = t0 + (uint64_t)wave * (64 * RS_R) + (uint64_t)r * 64 + lane;
            const bool have = e < n;
            key[r] = 0; val[r] = V();
            if (have) { key[r] = keys[e]; val[r] = vals[e]; }
        }
#pragma unroll
        for (int r = 0; r < RS_R; r++) {
            const uint64_t e = t0 + (uint64_t)wave * (64 * RS_R) + (uint64_t)r * 64 + lane;
            const bool have = e < n;
            const uint32_t d = (uint32_t)(key[r] >> shift) & dmask;
            unsigned long long peers = __ballot(have);
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const unsigned long long m = __ballot((d >> b) & 1u);
                peers &= ((d >> b) & 1u) ? m : ~m;
            }
            rank[r] = 0xFFFFFFFFu;
            if (have) {
                const uint32_t lower = (uint32_t)__popcll(peers & below);
                // (every peer reads the counter before the lowest of them moves it: the LDS operations of a wave keep their order)
                const uint32_t c = __atomic_load_n(&wcnt[wave][d], __ATOMIC_RELAXED);
                rank[r] = c + lower;
                if (lower == 0u) __atomic_store_n(&wcnt[wave][d], c + (uint32_t)__popcll(peers), __ATOMIC_RELAXED);
            }
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        {   // thread d: the digit's elements in this tile, wave by wave -> where each wave's run starts
            uint32_t c[RS_T / 64], sum = 0;
            for (int w = 0; w < RS_T / 64; w++) { c[w] = wcnt[w][tid]; sum += c[w]; }
            uint32_t all;
            const uint32_t ex = mf_block_excl_scan(sum, scratch, &all);
            toff[tid] = ex; tot[tid] = sum;
            uint32_t at = ex;
            for (int w = 0; w < RS_T / 64; w++) { wcnt[w][tid] = at; at += c[w]; }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RS_R; r++) {
            if (rank[r] != 0xFFFFFFFFu) {
                const uint32_t d = (uint32_t)(key[r] >> shift) & dmask;
                const uint32_t s = wcnt[wave][d] + rank[r];
                lk[s] = key[r]; lv[s] = val[r];
            }
        }
        __syncthreads();
        const uint64_t left = n - t0;
        const uint32_t cnt = left < RS_TILE ? (uint32_t)left : (uint32_t)RS_TILE;
#pragma unroll 4
        for (uint32_t s = tid; s < cnt; s += RS_T) {
            const K k = lk[s];
            const uint32_t d = (uint32_t)(k >> shift) & dmask;
            const uint32_t at = gcur[d] + (s - toff[d]);
            keys_out[at] = k; vals_out[at] = lv[s];
        }
        __syncthreads();
        gcur[tid] += tot[tid];
        for (int w = 0; w < RS_T / 64; w++) wcnt[w][tid] = 0;
        __syncthreads();
    }
}

// in -> out, ascending by the `bits` key bits from bit `first_bit` up, stable.  tmp_k / tmp_v: n elements each (needed when more than one pass is made)
// pingpong != nullptr: no temporaries -- the input arrays are written too (d_keys_in / d_vals_in must be writable) and *pingpong says where the
// result is: 0 = in d_keys_out / d_vals_out, 1 = back in the input arrays (an even number of passes).
template <typename K, typename V>
static int rs_sort(mf_ctx *ctx, const K *d_keys_in, const V *d_vals_in, uint64_t n, int bits, K *d_keys_out, V *d_vals_out, int first_bit = 0, int *pingpong = nullptr) {
    if (pingpong) *pingpong = 0;
    if (!n) return MF_OK;
    if (n >= (1ull << 32)) return mf_set_error("sort: more than 2^32 entries is not supported");
    MF_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    bits = std::min<int>(bits, (int)sizeof(K) * 8 - first_bit);
    const int passes = std::max(1, (bits + 7) / 8);
    const uint64_t n_blocks = (n + RS_BLOCK - 1) / RS_BLOCK;
    const unsigned grid = (unsigned)n_blocks;
    mf_buf<uint32_t> hist; MF_TRY(hist.alloc(ctx, 256 * n_blocks));
    mf_buf<uint64_t> offs; MF_TRY(offs.alloc(ctx, 256 * n_blocks + 1));
    mf_buf<uint64_t> tot; MF_TRY(tot.alloc(ctx, 2));
    mf_buf<K> tk; mf_buf<V> tv;
    if (passes > 1 && !pingpong) { MF_TRY(tk.alloc(ctx, n)); MF_TRY(tv.alloc(ctx, n)); }
    // the last pass must land in the caller's buffers: with an even number of passes the first one goes to the temporaries
    const K *src_k = d_keys_in; const V *src_v = d_vals_in;
    mf_ktimer tm(ctx, "k_radix_sort");
    for (int p = 0; p < passes; p++) {
        const bool to_out = pingpong ? (p & 1) == 0 : ((passes - 1 - p) & 1) == 0;
        K *dst_k = to_out ? d_keys_out : (pingpong ? const_cast<K *>(d_keys_in) : tk.p); V *dst_v = to_out ? d_vals_out : (pingpong ? const_cast<V *>(d_vals_in) : tv.p);
        const int left = bits - 8 * p;                                                       // (bits above the range take no part in the order)
        const uint32_t dmask = left >= 8 ? 255u : (1u << left) - 1u;
        k_rs_count<K><<<grid, RS_T, 0, st>>>(src_k, n, first_bit + 8 * p, dmask, hist.p, n_blocks);
        MF_TRY(mf_scan<1>(ctx, hist.p, offs.p, 256 * n_blocks, tot.p));
        k_rs_scatter<K, V><<<grid, RS_T, 0, st>>>(src_k, src_v, n, first_bit + 8 * p, dmask, offs.p, n_blocks, dst_k, dst_v);
        src_k = dst_k; src_v = dst_v;
    }
    if (pingpong) *pingpong = (passes & 1) ? 0 : 1;
    MF_HIP(hipGetLastError());
    MF_HIP(hipStreamSynchronize(st));
    return MF_OK;
}

int mf_sort_pairs(mf_ctx *ctx, const uint64_t *d_keys_in, const uint16_t *d_vals_in, uint64_t n, int key_bits, uint64_t *d_keys_out,
                  uint16_t *d_vals_out) {
    return rs_sort<uint64_t, uint16_t>(ctx, d_keys_in, d_vals_in, n, std::min(64, std::max(1, key_bits)), d_keys_out, d_vals_out);
}

// k-mers grouped by component (ascending component id), ascending inside each component: two stable LSD sorts
int mf_sort_kmers_by_comp(mf_ctx *ctx, const uint32_t *d_comp, const uint64_t *d_kmers, uint64_t n, int key_bits, uint32_t n_comps,
                          uint64_t *d_out) {
    if (!n) return MF_OK;
    if (n >= (1ull << 32)) return mf_set_error("sort: more than 2^32 entries is not supported");
    mf_buf<uint64_t> k1; mf_buf<uint32_t> c1, c2;
    MF_TRY(k1.alloc(ctx, n)); MF_TRY(c1.alloc(ctx, n)); MF_TRY(c2.alloc(ctx, n));
    int cb = 1; while (cb < 32 && (1ull << cb) < (uint64_t)n_comps) cb++;
    MF_TRY((rs_sort<uint64_t, uint32_t>(ctx, d_kmers, d_comp, n, std::min(64, std::max(1, key_bits)), k1.p, c1.p)));
    return rs_sort<uint32_t, uint64_t>(ctx, c1.p, k1.p, n, cb, c2.p, d_out);
}

// (u32 key, u32 value) pairs, ascending keys of `bits` bits (stable)
int mf_sort_u32_pairs(mf_ctx *ctx, const uint32_t *d_keys_in, const uint32_t *d_vals_in, uint64_t n, int bits, uint32_t *d_keys_out,
                      uint32_t *d_vals_out) {
    return rs_sort<uint32_t, uint32_t>(ctx, d_keys_in, d_vals_in, n, std::min(32, std::max(1, bits)), d_keys_out, d_vals_out);
}

// (u32 key, u64 value) pairs, ascending keys of `bits` bits (stable): the members of the sharded cutter grouped by component
int mf_sort_u32_u64(mf_ctx *ctx, const uint32_t *d_keys_in, const uint64_t *d_vals_in, uint64_t n, int bits, uint32_t *d_keys_out,
                    uint64_t *d_vals_out) {
    return rs_sort<uint32_t, uint64_t>(ctx, d_keys_in, d_vals_in, n, std::min(32, std::max(1, bits)), d_keys_out, d_vals_out);
}

// (u64 key, u32 value) pairs, ascending keys of `bits` bits (stable)
int mf_sort_u64_u32(mf_ctx *ctx, const uint64_t *d_keys_in, const uint32_t *d_vals_in, uint64_t n, int bits, uint64_t *d_keys_out, uint32_t *d_vals_out) {
    return rs_sort<uint64_t, uint32_t>(ctx, d_keys_in, d_vals_in, n, std::min(64, std::max(1, bits)), d_keys_out, d_vals_out);
}

// (u64 key, u64 value) pairs (mf_wide.hip: the two words of a 2k-bit k-mer, sorted word by word)
int mf_sort_u64_u64(mf_ctx *ctx, const uint64_t *d_keys_in, const uint64_t *d_vals_in, uint64_t n, int bits, uint64_t *d_keys_out, uint64_t *d_vals_out) {
    return rs_sort<uint64_t, uint64_t>(ctx, d_keys_in, d_vals_in, n, std::min(64, std::max(1, bits)), d_keys_out, d_vals_out);
}

// ... by the key bits [first_bit, first_bit + bits) alone (mf_wide.hip: the leading bits of a 2k-bit k-mer; the rest is ordered in LDS)
int mf_sort_u64_u64_range(mf_ctx *ctx, const uint64_t *d_keys_in, const uint64_t *d_vals_in, uint64_t n, int first_bit, int bits, uint64_t *d_keys_out, uint64_t *d_vals_out) {
    if (first_bit < 0 || first_bit > 63 || bits < 1) return mf_set_error("sort: bits [%d, %d + %d) of a 64-bit key", first_bit, first_bit, bits);
    return rs_sort<uint64_t, uint64_t>(ctx, d_keys_in, d_vals_in, n, bits, d_keys_out, d_vals_out, first_bit);
}
// ... without temporaries: the two pairs of arrays are each other's ping-pong; *in_second = 1 when the result is in (k1, v1), 0 when it is back in (k0, v0)
int mf_sort_u64_u64_pingpong(mf_ctx *ctx, uint64_t *k0, uint64_t *v0, uint64_t n, int first_bit, int bits, uint64_t *k1, uint64_t *v1, int *in_second) {
    if (first_bit < 0 || first_bit > 63 || bits < 1 || !in_second) return mf_set_error("sort: bits [%d, %d + %d) of a 64-bit key", first_bit, first_bit, bits);
    int back = 0;
    const int rc = rs_sort<uint64_t, uint64_t>(ctx, k0, v0, n, bits, k1, v1, first_bit, &back);
    *in_second = back ? 0 : 1;
    return rc;
}

// test hook (not part of the ABI, tests/test_round5_gpu.py): sorts host arrays with the kernels above.  kind: 0 = (u64, u16), 1 = (u32, u32), 2 = (u32, u64),
// 3 = (u64, u32), 4 = (u64, u64)
template <typename K, typename V>
static int rs_debug(mf_ctx *ctx, const void *keys, const void *vals, uint64_t n, int bits, void *keys_out, void *vals_out) {
    mf_buf<K> ki, ko; mf_buf<V> vi, vo;
    MF_TRY(ki.alloc(ctx, n)); MF_TRY(ko.alloc(ctx, n)); MF_TRY(vi.alloc(ctx, n)); MF_TRY(vo.alloc(ctx, n));
    MF_HIP(hipMemcpyAsync(ki.p, keys, n * sizeof(K), hipMemcpyHostToDevice, ctx->stream));
    MF_HIP(hipMemcpyAsync(vi.p, vals, n * sizeof(V), hipMemcpyHostToDevice, ctx->stream));
    MF_TRY((rs_sort<K, V>(ctx, ki.p, vi.p, n, bits, ko.p, vo.p)));
    MF_HIP(hipMemcpyAsync(keys_out, ko.p, n * sizeof(K), hipMemcpyDeviceToHost, ctx->stream));
    MF_HIP(hipMemcpyAsync(vals_out, vo.p, n * sizeof(V), hipMemcpyDeviceToHost, ctx->stream));
    MF_HIP(hipStreamSynchronize(ctx->stream));
    return MF_OK;
}
extern "C" int mf_debug_sort(mf_ctx *ctx, int kind, const void *keys, const void *vals, uint64_t n, int bits, void *keys_out, void *vals_out) {
    if (!ctx || !n) return MF_OK;
    MF_HIP(hipSetDevice(ctx->device));
    switch (kind) {
    case 0: return rs_debug<uint64_t, uint16_t>(ctx, keys, vals, n, bits, keys_out, vals_out);
    case 1: return rs_debug<uint32_t, uint32_t>(ctx, keys, vals, n, bits, keys_out, vals_out);
    case 2: return rs_debug<uint32_t, uint64_t>(ctx, keys, vals, n, bits, keys_out, vals_out);
    case 3: return rs_debug<uint64_t, uint32_t>(ctx, keys, vals, n, bits, keys_out, vals_out);
    case 4: return rs_debug<uint64_t, uint64_t>(ctx, keys, vals, n, bits, keys_out, vals_out);
    default: return mf_set_error("mf_debug_sort: kind %d", kind);
    }
}

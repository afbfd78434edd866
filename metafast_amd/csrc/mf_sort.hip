// mf_sort.hip -- stable LSD radix sorts for the file seams: ascending (k-mer, count) order of the writers and of mf_table_export, the members of
// a component in ascending order, the unitigs in output order, the partition order of a loaded table.
// The reference dumps its hash map in iteration order (src/io/IOUtils.java:45-71: not reproducible); this implementation
// writes ascending k-mers so that files can be compared byte for byte.  Not on the timed step.
//
// Round 5: hand-written for wave64 (rounds 1-4 called rocPRIM's device radix sort here).  One pass per 8 key bits:
//   k_rs_count    a WAVE owns RS_WAVE consecutive elements; it counts their digits in LDS (one ds_add per element) and writes its 256
//                 counters digit-major: hist[digit][wave]
//   mf_scan       exclusive prefix over hist in that order = where each wave's elements of each digit go
//   k_rs_scatter  the wave walks its elements again, 64 at a time: the lanes that hold the same digit find each other with eight ballots
//                 (one per digit bit), a lane's place is its digit's running cursor + the number of such lanes below it; the lowest one
//                 moves the cursor on.  Stable by construction: waves own consecutive ranges, rounds go forward, lanes ascend.
// 8 + 2 (K + V) bytes per element and pass.  Everything of a call is on ctx->stream.  (About 2.5 x the time of rocPRIM's onesweep sort, whose
// workgroups sort tiles of 7 680 elements in LDS before they write: off the timed step that is +14 ms for components.bin's member order and
// a few ms per .kmers.bin; the no-reference k = 32..63 path, which sorts every k-mer OCCURRENCE twice per pass, keeps the library sort: 9.1 s
// against 13.5 s for 200 M reads at k = 63, profiles/r05aa_sort.txt.)
#include <cstring>
#include "mf_common.h"

#define RS_T 256                       // threads per workgroup
#define RS_WAVE 4096                   // elements a wave owns
#define RS_BLOCK (RS_WAVE * (RS_T / 64))

template <typename K>
__global__ __launch_bounds__(RS_T) void k_rs_count(const K *__restrict__ keys, uint64_t n, int shift, uint32_t dmask, uint32_t *__restrict__ hist, uint64_t n_waves) {
    __shared__ uint32_t cnt[RS_T / 64][256];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (uint32_t i = lane; i < 256u; i += 64u) cnt[wave][i] = 0;
    __builtin_amdgcn_wave_barrier();
    const uint64_t gw = (uint64_t)blockIdx.x * (RS_T / 64) + wave;
    const uint64_t e0 = gw * RS_WAVE;
    if (gw < n_waves) {
        const uint64_t e1 = e0 + RS_WAVE < n ? e0 + RS_WAVE : n;
        for (uint64_t e = e0 + lane; e < e1; e += 64) atomicAdd(&cnt[wave][(uint32_t)(keys[e] >> shift) & dmask], 1u);
        __builtin_amdgcn_wave_barrier();
        for (uint32_t d = lane; d < 256u; d += 64u) hist[(uint64_t)d * n_waves + gw] = cnt[wave][d];
    }
}

template <typename K, typename V>
__global__ __launch_bounds__(RS_T) void k_rs_scatter(const K *__restrict__ keys, const V *__restrict__ vals, uint64_t n, int shift, uint32_t dmask, const uint64_t *__restrict__ offs,
                                                     uint64_t n_waves, K *__restrict__ keys_out, V *__restrict__ vals_out) {
    __shared__ uint32_t cur[RS_T / 64][256];            // (places are < 2^32: the callers refuse more entries)
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint64_t gw = (uint64_t)blockIdx.x * (RS_T / 64) + wave;
    if (gw >= n_waves) return;
    for (uint32_t d = lane; d < 256u; d += 64u) cur[wave][d] = (uint32_t)offs[(uint64_t)d * n_waves + gw];
    __builtin_amdgcn_wave_barrier();
    const uint64_t e0 = gw * RS_WAVE;
    const uint64_t e1 = e0 + RS_WAVE < n ? e0 + RS_WAVE : n;
    const unsigned long long below = lane ? (~0ull >> (64u - lane)) : 0ull;
    for (uint64_t eb = e0; eb < e1; eb += 64) {
        const uint64_t e = eb + lane;
        const bool have = e < e1;
        K key = 0; V val = V();
        if (have) { key = keys[e]; val = vals[e]; }
        const uint32_t d = (uint32_t)(key >> shift) & dmask;
        unsigned long long peers = __ballot(have);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        if (have) {
            const uint32_t lower = (uint32_t)__popcll(peers & below);
            // (every peer reads the cursor before the lowest of them moves it: the LDS operations of a wave keep their order; atomics so that
            // the compiler keeps every one of them a real LDS access)
            const uint32_t at = __atomic_load_n(&cur[wave][d], __ATOMIC_RELAXED) + lower;
            keys_out[at] = key; vals_out[at] = val;
            if (lower == 0u) __atomic_store_n(&cur[wave][d], at + (uint32_t)__popcll(peers), __ATOMIC_RELAXED);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// in -> out, ascending by the low `bits` key bits, stable.  tmp_k / tmp_v: n elements each (needed when more than one pass is made)
template <typename K, typename V>
static int rs_sort(mf_ctx *ctx, const K *d_keys_in, const V *d_vals_in, uint64_t n, int bits, K *d_keys_out, V *d_vals_out) {
    if (!n) return MF_OK;
    if (n >= (1ull << 32)) return mf_set_error("sort: more than 2^32 entries is not supported");
    MF_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int passes = std::max(1, (std::min<int>(bits, (int)sizeof(K) * 8) + 7) / 8);
    const uint64_t n_waves = (n + RS_WAVE - 1) / RS_WAVE;
    const unsigned grid = (unsigned)((n_waves + (RS_T / 64) - 1) / (RS_T / 64));
    mf_buf<uint32_t> hist; MF_TRY(hist.alloc(ctx, 256 * n_waves));
    mf_buf<uint64_t> offs; MF_TRY(offs.alloc(ctx, 256 * n_waves + 1));
    mf_buf<uint64_t> tot; MF_TRY(tot.alloc(ctx, 2));
    mf_buf<K> tk; mf_buf<V> tv;
    if (passes > 1) { MF_TRY(tk.alloc(ctx, n)); MF_TRY(tv.alloc(ctx, n)); }
    // the last pass must land in the caller's buffers: with an even number of passes the first one goes to the temporaries
    const K *src_k = d_keys_in; const V *src_v = d_vals_in;
    for (int p = 0; p < passes; p++) {
        const bool to_out = ((passes - 1 - p) & 1) == 0;
        K *dst_k = to_out ? d_keys_out : tk.p; V *dst_v = to_out ? d_vals_out : tv.p;
        const int left = std::min<int>(bits, (int)sizeof(K) * 8) - 8 * p;                  // (bits above `bits` take no part in the order)
        const uint32_t dmask = left >= 8 ? 255u : (1u << left) - 1u;
        k_rs_count<K><<<grid, RS_T, 0, st>>>(src_k, n, 8 * p, dmask, hist.p, n_waves);
        MF_TRY(mf_scan<1>(ctx, hist.p, offs.p, 256 * n_waves, tot.p));
        k_rs_scatter<K, V><<<grid, RS_T, 0, st>>>(src_k, src_v, n, 8 * p, dmask, offs.p, n_waves, dst_k, dst_v);
        src_k = dst_k; src_v = dst_v;
    }
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

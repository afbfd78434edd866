// mf_table.hip -- table handle (dense (key,count) arrays in HBM) + the open-addressed index in HBM.
//
// The index stands in for BigLong2ShortHashMap's random-access role (get / contains / addAndBound,
// itmo!/structures/map/Long2ShortHashMap.java:119-175): 16-byte slots {key, idx, val}, linear probing,
// load <= 0.5, empty marker = all-ones key (key 0 = poly-A is a legal k-mer, the reference side-cars it).
#include "mf_common.h"
#include <algorithm>
#include <numeric>

int mf_sum_counts(mf_ctx *ctx, const uint16_t *d_counts, uint64_t n, uint64_t *total);

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
__global__ void k_index_init(mf_slot *__restrict__ slots, uint64_t cap) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < cap; i += stride) {
        ulonglong2 v; v.x = MF_EMPTY; v.y = 0;
        *reinterpret_cast<ulonglong2 *>(&slots[i]) = v;
    }
}

// distinct keys: claim a slot with one 64-bit CAS, then store {idx,val}
__global__ void k_index_insert(mf_slot *__restrict__ slots, uint64_t mask, const uint64_t *__restrict__ keys,
                               const uint16_t *__restrict__ vals, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint64_t key = keys[i];
        uint64_t s = mf_hash64(key) & mask;
        for (;;) {
            unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long *>(&slots[s].key),
                                               (unsigned long long)MF_EMPTY, (unsigned long long)key);
            if (old == MF_EMPTY) {
                uint64_t aux = (uint64_t)(uint32_t)i | ((uint64_t)(vals ? vals[i] : 0) << 32);
                *reinterpret_cast<uint64_t *>(&slots[s].idx) = aux;
                break;
            }
            s = (s + 1) & mask;
        }
    }
}

// insert-or-add (duplicates allowed): val accumulates in 32 bits, clamped to 32767 on extraction
__global__ void k_index_insert_add(mf_slot *__restrict__ slots, uint64_t mask, const uint64_t *__restrict__ keys,
                                   const uint16_t *__restrict__ vals, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint64_t key = keys[i];
        uint64_t s = mf_hash64(key) & mask;
        for (;;) {
            unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long *>(&slots[s].key),
                                               (unsigned long long)MF_EMPTY, (unsigned long long)key);
            if (old == MF_EMPTY || old == key) { atomicAdd(&slots[s].val, (uint32_t)vals[i]); break; }
            s = (s + 1) & mask;
        }
    }
}

__global__ void k_index_lookup(mf_index_view ix, const uint64_t *__restrict__ keys, uint64_t n, int32_t *__restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t idx, val;
    out[i] = mf_index_find(ix, keys[i], &idx, &val) ? (int32_t)val : -1;
}

// ---- partitioned index build: the dense table comes out of the counting pass grouped by partition, so every partition's
// region can be built in LDS (LDS CAS) and written out as one contiguous block: a streaming pass, no HBM atomics, no random
// HBM accesses (the generic build below costs one random CAS + one random store per key: 40 ms for 3.6e8 keys).
// Regions are sized per partition (minimizer partitions differ a lot in size): power of two > 2 x its keys. ----
#define MF_IDX_WAVE_SLOTS 1024      // regions up to this size are built by one wave, larger ones by a whole workgroup
__global__ void k_index_region_sizes(const uint64_t *__restrict__ part_off, uint32_t np, uint32_t *__restrict__ sz,
                                     uint32_t *__restrict__ biglist, unsigned int *__restrict__ n_big) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= np) return;
    uint64_t c = part_off[p + 1] - part_off[p];
    if (c >= (1ull << MF_CIDX_REL_BITS) - 1ull) { atomicAdd(n_big + 2, 1u); c = 1; }  // (a position inside the partition has 20 bits: the caller builds the generic index)
    const uint64_t want = 2 * c + 1;        // load <= 0.5: most probes are for absent neighbours, and a miss walks to the end of its cluster
    uint32_t S = 2;
    while (S < want) S <<= 1;
    sz[p] = S;
    // up to MF_IDX_WAVE_SLOTS: built in LDS by one wave; up to 8192: in LDS by a workgroup; larger (a partition that was
    // counted in several passes): in place in HBM.  biglist: the workgroup ones from the front, the HBM ones from the back.
    if (S > 8192u) biglist[np - 1u - atomicAdd(n_big + 4, 1u)] = p;
    else if (S > MF_IDX_WAVE_SLOTS) biglist[atomicAdd(n_big, 1u)] = p;
}
__global__ void k_index_dir_pack(const uint64_t *__restrict__ roff, const uint32_t *__restrict__ sz, const uint64_t *__restrict__ part_off, uint32_t np,
                                 uint64_t *__restrict__ dir) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < np) { dir[2 * (size_t)p] = (roff[p] << 6) | (uint64_t)(31 - __clz(sz[p])); dir[2 * (size_t)p + 1] = part_off[p]; }
}
// slot of entry i of partition [lo, ..): tag of the key's hash | position inside the partition (the keys of a table are distinct:
// the first free slot from the home slot on is taken)
__device__ __forceinline__ void mf_cidx_insert(uint32_t *reg, uint32_t rmask, uint64_t key, uint32_t rel, uint32_t skm_k) {
    const uint32_t hs = mf_pslot(mf_phash(mf_cidx_hkey(key, skm_k)));
    const uint32_t v = ((hs >> MF_CIDX_REL_BITS) << MF_CIDX_REL_BITS) | rel;
    uint32_t s = hs & rmask;
    for (;;) {
        if (atomicCAS(&reg[s], MF_CIDX_EMPTY, v) == MF_CIDX_EMPTY) break;
        s = (s + 1u) & rmask;
    }
}
// TEAM = 64: a wave per partition (4 regions per block), skips the large ones; TEAM = 256: the block per LISTED partition
template <int TEAM>
__global__ __launch_bounds__(256) void k_index_build_part(uint32_t *__restrict__ slots, const uint64_t *__restrict__ dir, const uint64_t *__restrict__ keys,
                                                          const uint64_t *__restrict__ part_off, uint32_t np, const uint32_t *__restrict__ biglist,
                                                          const unsigned int *__restrict__ n_big, uint32_t skm_k) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TEAMS = 256 / TEAM;
    const int team = threadIdx.x / TEAM, tl = threadIdx.x % TEAM;
    uint32_t *reg = reinterpret_cast<uint32_t *>(smem) + (size_t)team * MF_IDX_WAVE_SLOTS;     // (TEAM 256: team == 0, the whole buffer)
    auto sync = [&]() {
        if (TEAM == 64) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
        else __syncthreads();
    };
    const uint32_t nteams = gridDim.x * TEAMS;
    const uint32_t count = TEAM == 64 ? np : *n_big;
    for (uint32_t it = blockIdx.x * TEAMS + team; it < count; it += nteams) {   // team-uniform
        const uint32_t p = TEAM == 64 ? it : biglist[it];
        const uint64_t d = dir[2 * (size_t)p];
        const uint32_t S = 1u << (uint32_t)(d & 63ull), rmask = S - 1;
        if (TEAM == 64 && S > MF_IDX_WAVE_SLOTS) continue;
        for (uint32_t j = tl; j < S; j += TEAM) reg[j] = MF_CIDX_EMPTY;
        sync();
        const uint64_t lo = part_off[p], n_here = part_off[p + 1] - lo;
        for (uint64_t i = tl; i < n_here; i += TEAM) mf_cidx_insert(reg, rmask, keys[lo + i], (uint32_t)i, skm_k);
        sync();
        uint32_t *dst = slots + (d >> 6);
        for (uint32_t j = tl; j < S; j += TEAM) dst[j] = reg[j];
        sync();
    }
}
// regions too large for LDS: one workgroup per listed partition (biglist from the back) clears its region in HBM and
// inserts with global atomics
__global__ __launch_bounds__(256) void k_index_build_huge(uint32_t *__restrict__ slots, const uint64_t *__restrict__ dir, const uint64_t *__restrict__ keys,
                                                          const uint64_t *__restrict__ part_off, uint32_t np, const uint32_t *__restrict__ biglist, uint32_t n_huge, uint32_t skm_k) {
    for (uint32_t it = blockIdx.x; it < n_huge; it += gridDim.x) {
        const uint32_t p = biglist[np - 1u - it];
        const uint64_t d = dir[2 * (size_t)p];
        const uint32_t S = 1u << (uint32_t)(d & 63ull), rmask = S - 1u;
        uint32_t *reg = slots + (d >> 6);
        for (uint32_t j = threadIdx.x; j < S; j += blockDim.x) reg[j] = MF_CIDX_EMPTY;
        __threadfence();
        __syncthreads();
        const uint64_t lo = part_off[p], n_here = part_off[p + 1] - lo;
        for (uint64_t i = threadIdx.x; i < n_here; i += blockDim.x) mf_cidx_insert(reg, rmask, keys[lo + i], (uint32_t)i, skm_k);
        __syncthreads();
    }
}
// entries with count > thr per partition (one wave per partition)
__global__ __launch_bounds__(256) void k_part_selcount(const uint16_t *__restrict__ cnts, const uint64_t *__restrict__ part_off, uint32_t np,
                                                       int thr, uint32_t *__restrict__ pcount) {
    const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
    for (uint32_t p = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); p < np; p += nwaves) {
        uint32_t c = 0;
        for (uint64_t i = part_off[p] + mf_lane(); i < part_off[p + 1]; i += 64) c += (int)cnts[i] > thr;
        for (int d = 32; d >= 1; d >>= 1) c += __shfl_down(c, d, 64);
        if (mf_lane() == 0) pcount[p] = c;
    }
}

// ---- stable stream compaction in two passes: select entries with pred(i) ----
// MODE 0: table entries with count > thr.   MODE 1: non-empty index slots (val clamped to 32767).
template <int MODE>
__device__ __forceinline__ bool mf_sel_pred(const uint64_t *keys, const uint16_t *cnts, const mf_slot *slots, uint64_t i,
                                            int thr) {
    if (MODE == 0) return (int)cnts[i] > thr;
    return slots[i].key != MF_EMPTY;
}
template <int MODE>
__global__ __launch_bounds__(1024) void k_select_count(const uint64_t *__restrict__ keys, const uint16_t *__restrict__ cnts,
                                                        const mf_slot *__restrict__ slots, uint64_t n, uint64_t per_block,
                                                        int thr, uint32_t *__restrict__ bcount) {
    __shared__ uint32_t scratch[17];
    uint64_t lo = (uint64_t)blockIdx.x * per_block, hi = lo + per_block < n ? lo + per_block : n;
    uint32_t c = 0;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) c += mf_sel_pred<MODE>(keys, cnts, slots, i, thr);
    uint32_t tot;
    mf_block_excl_scan(c, scratch, &tot);
    if (threadIdx.x == 0) bcount[blockIdx.x] = tot;
}
template <int MODE>
__global__ __launch_bounds__(1024) void k_select_write(const uint64_t *__restrict__ keys, const uint16_t *__restrict__ cnts,
                                                        const mf_slot *__restrict__ slots, uint64_t n, uint64_t per_block,
                                                        int thr, const uint64_t *__restrict__ boff,
                                                        uint64_t *__restrict__ ok, uint16_t *__restrict__ oc) {
    __shared__ uint32_t scratch[17];
    uint64_t lo = (uint64_t)blockIdx.x * per_block, hi = lo + per_block < n ? lo + per_block : n;
    uint64_t base = boff[blockIdx.x];
    for (uint64_t i0 = lo; i0 < hi; i0 += blockDim.x) {   // uniform trip count (barriers inside)
        uint64_t i = i0 + threadIdx.x;
        bool sel = i < hi && mf_sel_pred<MODE>(keys, cnts, slots, i, thr);
        uint32_t tot;
        uint32_t ex = mf_block_excl_scan(sel ? 1u : 0u, scratch, &tot);
        if (sel) {
            if (MODE == 0) { ok[base + ex] = keys[i]; oc[base + ex] = cnts[i]; }
            else {
                uint32_t v = slots[i].val;
                ok[base + ex] = slots[i].key;
                oc[base + ex] = (uint16_t)(v > (uint32_t)MF_MAX_COUNT ? (uint32_t)MF_MAX_COUNT : v);
            }
        }
        base += tot;
    }
}

// histogram of counts (0..32767) -- LDS-privatised low bins, global atomics for the rest
__global__ __launch_bounds__(256) void k_count_hist(const uint16_t *__restrict__ c, uint64_t n,
                                                    unsigned long long *__restrict__ hist) {
    __shared__ uint32_t lo[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lo[i] = 0;
    __syncthreads();
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t v = c[i];
        if (v < 1024) atomicAdd(&lo[v], 1u);
        else atomicAdd(&hist[v], 1ull);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < 1024; j += blockDim.x)
        if (lo[j]) atomicAdd(&hist[j], (unsigned long long)lo[j]);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static uint64_t pow2_at_least(uint64_t x) { uint64_t p = 1; while (p < x) p <<= 1; return p; }

int mf_table_adopt(mf_ctx *ctx, int k, uint64_t n, uint64_t n_occ, uint64_t *d_keys, size_t kb, uint16_t *d_counts,
                   size_t cb, mf_table **out) {
    mf_table *t = new mf_table();
    t->ctx = ctx; t->k = k; t->n = n; t->n_occ = n_occ;
    t->d_keys = d_keys; t->keys_bytes = kb;
    t->d_counts = d_counts; t->counts_bytes = cb;
    *out = t;
    return MF_OK;
}

extern "C" void mf_table_destroy(mf_table *t) {
    if (!t) return;
    if (--t->refs > 0) return;                              // (another handle on the same table: the context's file cache, or a load that hit it)
    if (t->owns_arrays && t->d_keys) mf_release(t->ctx, t->d_keys, t->keys_bytes);
    if (t->owns_arrays && t->d_counts) mf_release(t->ctx, t->d_counts, t->counts_bytes);
    if (t->owns_arrays && t->index.slots) mf_release(t->ctx, t->index.slots, t->index_bytes);       // (an alias borrows the index too)
    if (t->owns_arrays && t->index.dir) mf_release(t->ctx, t->index.dir, t->index.dir_bytes);
    if (t->owns_arrays && t->d_part_off) mf_release(t->ctx, t->d_part_off, t->part_off_bytes);
    delete t;
}

// gives the lookup index back to the arena (it is rebuilt on the next lookup: 9 ms per 3.6e8 keys): with several samples per
// GPU (KmersCounterForManyFilesMain.java:80-108 loops over all libraries) a sample's index is needed while its unitigs are
// built and again for its feature vector, and a 4e8..1.4e9-key table's index is 3 .. 6 times the table
extern "C" int mf_table_drop_index(mf_table *t) {
    if (!t) return mf_set_error("table is NULL");
    if (!t->owns_arrays) return MF_OK;
    MF_HIP(hipSetDevice(t->ctx->device));
    MF_HIP(hipStreamSynchronize(t->ctx->stream));
    if (t->index.slots) mf_release(t->ctx, t->index.slots, t->index_bytes);
    if (t->index.dir) mf_release(t->ctx, t->index.dir, t->index.dir_bytes);
    t->index = mf_index();
    t->index_bytes = 0;
    return MF_OK;
}

int mf_index_build(mf_ctx *ctx, const uint64_t *d_keys, const uint16_t *d_vals, uint64_t n, mf_index *out, size_t *bytes) {
    uint64_t cap = pow2_at_least(std::max<uint64_t>(2 * n, 1024));
    void *p = nullptr;
    MF_TRY(mf_alloc(ctx, cap * sizeof(mf_slot), &p));
    unsigned grid = (unsigned)std::min<uint64_t>((cap + 255) / 256, 65536);
    {
        mf_ktimer t(ctx, "k_index_init");
        k_index_init<<<grid, 256, 0, ctx->stream>>>((mf_slot *)p, cap);
    }
    if (n) {
        unsigned g2 = (unsigned)std::min<uint64_t>((n + 255) / 256, 65536);
        mf_ktimer t(ctx, "k_index_insert");
        k_index_insert<<<g2, 256, 0, ctx->stream>>>((mf_slot *)p, cap - 1, d_keys, d_vals, n);
    }
    out->slots = p; out->cap = cap;
    *bytes = cap * sizeof(mf_slot);
    return MF_OK;
}

int mf_table_ensure_index(mf_table *t) {
    if (t->index.slots) return MF_OK;
    if (t->n >= 0xFFFFFFFFull) return mf_set_error("index supports < 2^32 entries");
    mf_ctx *ctx = t->ctx;
    if (t->part_bits > 0 && t->d_part_off && t->n) {
        const uint32_t np = 1u << t->part_bits;
        mf_buf<uint32_t> sz; MF_TRY(sz.alloc(ctx, np));
        mf_buf<uint32_t> biglist; MF_TRY(biglist.alloc(ctx, np));
        mf_buf<uint64_t> roff; MF_TRY(roff.alloc(ctx, (size_t)np + 1));
        mf_buf<uint64_t> scal; MF_TRY(scal.alloc(ctx, 4));            // [0] total slots, [1] number of large regions, [2] oversized partitions, [3] HBM-built regions
        mf_buf<uint64_t> dir; MF_TRY(dir.alloc(ctx, (size_t)2 * np));      // (region, first entry) per partition
        MF_HIP(hipMemsetAsync(scal.p, 0, 32, ctx->stream));
        k_index_region_sizes<<<(np + 255) / 256, 256, 0, ctx->stream>>>(t->d_part_off, np, sz.p, biglist.p, (unsigned int *)&scal.p[1]);
        MF_TRY(mf_scan<1>(ctx, sz.p, roff.p, np, &scal.p[0]));
        k_index_dir_pack<<<(np + 255) / 256, 256, 0, ctx->stream>>>(roff.p, sz.p, t->d_part_off, np, dir.p);
        uint64_t hs[4];
        MF_HIP(hipMemcpyAsync(hs, scal.p, 32, hipMemcpyDeviceToHost, ctx->stream));
        MF_HIP(hipStreamSynchronize(ctx->stream));
        if (hs[2]) return mf_index_build(ctx, t->d_keys, t->d_counts, t->n, &t->index, &t->index_bytes);
        const uint64_t cap = hs[0];
        const uint32_t n_big = (uint32_t)hs[1], n_huge = (uint32_t)hs[3];
        const uint32_t skm_k = t->part_skm ? (uint32_t)t->k : 0u;              // (minimizer partitions: home slot and tag from the key's interior, mf_cidx_hkey)
        void *p = nullptr;
        MF_TRY(mf_alloc(ctx, cap * sizeof(uint32_t), &p));
        {
            mf_ktimer tm(ctx, "k_index_build_part");
            const size_t lds = (size_t)4 * MF_IDX_WAVE_SLOTS * sizeof(uint32_t);
            const unsigned grid = (unsigned)std::min<uint64_t>((np + 3) / 4, (uint64_t)ctx->n_cu * 8);
            k_index_build_part<64><<<grid, 256, lds, ctx->stream>>>((uint32_t *)p, dir.p, t->d_keys, t->d_part_off, np, biglist.p, (const unsigned int *)&scal.p[1], skm_k);
            if (n_big) {      // regions of up to 8192 slots (the counting pass cannot produce partitions of more than 4096 keys)
                const size_t lds2 = (size_t)8192 * sizeof(uint32_t);
                k_index_build_part<256><<<(unsigned)std::min<uint32_t>(n_big, (uint32_t)ctx->n_cu * 2), 256, lds2, ctx->stream>>>((uint32_t *)p, dir.p, t->d_keys, t->d_part_off, np, biglist.p, (const unsigned int *)&scal.p[1], skm_k);
            }
            if (n_huge)
                k_index_build_huge<<<(unsigned)std::min<uint32_t>(n_huge, (uint32_t)ctx->n_cu * 4), 256, 0, ctx->stream>>>((uint32_t *)p, dir.p, t->d_keys, t->d_part_off, np, biglist.p, n_huge, skm_k);
        }
        MF_HIP(hipGetLastError());
        MF_HIP(hipStreamSynchronize(ctx->stream));          // (biglist / scal are released below)
        t->index.slots = p; t->index.cap = cap; t->index.part_bits = (uint32_t)t->part_bits;
        t->index.skm_k = skm_k;
        t->index.compact = 1; t->index.keys = t->d_keys; t->index.counts = t->d_counts;
        t->index.dir_bytes = dir.bytes(); t->index.dir = dir.take();
        t->index_bytes = cap * sizeof(uint32_t);
        if (ctx->opt_verbose) fprintf(stderr, "[mf] index: %u partitions, %llu slots for %llu keys (%u large regions), %.2f GB\n", np,
                                      (unsigned long long)cap, (unsigned long long)t->n, n_big, cap * 4 / 1e9);
        return MF_OK;
    }
    return mf_index_build(ctx, t->d_keys, t->d_counts, t->n, &t->index, &t->index_bytes);
}

extern "C" int mf_table_stats(const mf_table *t, uint64_t *n_distinct, uint64_t *n_total) {
    if (!t) return mf_set_error("table is NULL");
    if (n_distinct) *n_distinct = t->n;
    if (n_total) MF_TRY(mf_sum_counts(t->ctx, t->d_counts, t->n, n_total));
    return MF_OK;
}
extern "C" int mf_table_records(const mf_table *t, uint64_t *n_records, int *record_bytes) {
    if (!t) return mf_set_error("table is NULL");
    if (n_records) *n_records = t->n_records;
    if (record_bytes) *record_bytes = t->record_bytes;
    return MF_OK;
}
extern "C" int mf_table_occurrences(const mf_table *t, uint64_t *n_occ) {
    if (!t || !n_occ) return mf_set_error("NULL argument");
    *n_occ = t->n_occ;
    return MF_OK;
}
extern "C" int mf_table_device_view(const mf_table *t, const void **d_keys, const void **d_counts, uint64_t *n) {
    if (!t) return mf_set_error("table is NULL");
    if (d_keys) *d_keys = t->d_keys;
    if (d_counts) *d_counts = t->d_counts;
    if (n) *n = t->n;
    return MF_OK;
}

// device-side selection into fresh dense arrays
template <int MODE>
static int select_entries(mf_ctx *ctx, const uint64_t *keys, const uint16_t *cnts, const mf_slot *slots, uint64_t n, int thr,
                          mf_buf<uint64_t> &ok, mf_buf<uint16_t> &oc, uint64_t *n_out) {
    *n_out = 0;
    if (!n) { MF_TRY(ok.alloc(ctx, 0)); MF_TRY(oc.alloc(ctx, 0)); return MF_OK; }
    uint64_t nb = std::min<uint64_t>((n + 1023) / 1024, 2048);
    uint64_t per = ((n + nb - 1) / nb + 1023) / 1024 * 1024;
    nb = (n + per - 1) / per;
    mf_buf<uint32_t> bcount; MF_TRY(bcount.alloc(ctx, nb));
    mf_buf<uint64_t> boff; MF_TRY(boff.alloc(ctx, nb + 1));
    mf_buf<uint64_t> tot; MF_TRY(tot.alloc(ctx, 1));
    {
        mf_ktimer t(ctx, "k_select");
        k_select_count<MODE><<<(unsigned)nb, 1024, 0, ctx->stream>>>(keys, cnts, slots, n, per, thr, bcount.p);
        k_scan<1><<<1, 1024, 0, ctx->stream>>>(bcount.p, boff.p, nb, tot.p);
    }
    uint64_t m = 0;
    MF_HIP(hipMemcpyAsync(&m, tot.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    MF_HIP(hipStreamSynchronize(ctx->stream));
    MF_TRY(ok.alloc(ctx, m)); MF_TRY(oc.alloc(ctx, m));
    if (m) {
        mf_ktimer t(ctx, "k_select");
        k_select_write<MODE><<<(unsigned)nb, 1024, 0, ctx->stream>>>(keys, cnts, slots, n, per, thr, boff.p, ok.p, oc.p);
    }
    *n_out = m;
    return MF_OK;
}

extern "C" int mf_table_filter(const mf_table *t, int threshold, mf_table **out) {
    if (!t || !out) return mf_set_error("NULL argument");
    *out = nullptr;
    mf_ctx *ctx = t->ctx;
    MF_HIP(hipSetDevice(ctx->device));
    mf_buf<uint64_t> ok; mf_buf<uint16_t> oc; uint64_t m = 0;
    MF_TRY(select_entries<0>(ctx, t->d_keys, t->d_counts, nullptr, t->n, threshold, ok, oc, &m));
    size_t kb = ok.bytes(), cb = oc.bytes();
    MF_TRY(mf_table_adopt(ctx, t->k, m, 0, ok.take(), kb, oc.take(), cb, out));
    (*out)->cut_thr = std::max(t->cut_thr, threshold);
    if (threshold >= 0 && threshold > t->cut_thr) {
        // what this cut drops joins what earlier cuts dropped: the count histogram (.stat.txt) stays that of ALL k-mers
        std::vector<uint64_t> h;
        MF_TRY(mf_table_count_hist(t, h));
        (*out)->dropped_hist.assign(h.begin(), h.begin() + std::min<size_t>((size_t)threshold + 1, h.size()));
    } else (*out)->dropped_hist = t->dropped_hist;
    if (t->part_bits > 0 && t->d_part_off && m) {
        // the compaction is stable, so the selected entries of partition p are still contiguous: new offsets by a scan
        const uint32_t np = 1u << t->part_bits;
        mf_buf<uint32_t> pc; MF_TRY(pc.alloc(ctx, np));
        mf_buf<uint64_t> po; MF_TRY(po.alloc(ctx, (size_t)np + 1));
        mf_buf<uint64_t> tot; MF_TRY(tot.alloc(ctx, 1));
        unsigned grid = (unsigned)std::min<uint64_t>((np + 3) / 4, (uint64_t)ctx->n_cu * 32);
        k_part_selcount<<<grid, 256, 0, ctx->stream>>>(t->d_counts, t->d_part_off, np, threshold, pc.p);
        MF_TRY(mf_scan<1>(ctx, pc.p, po.p, np, tot.p));
        (*out)->part_bits = t->part_bits; (*out)->part_skm = t->part_skm;
        (*out)->part_off_bytes = po.bytes();
        (*out)->d_part_off = po.take();
    }
    return MF_OK;
}

int mf_table_filter_or_alias(const mf_table *t, int threshold, mf_table **out) {
    mf_ctx *ctx = t->ctx;
    bool all_pass = t->n && t->cut_thr >= threshold;          // the table went through this cut (or a stricter one) already
    if (t->n && !all_pass) {
        // count first; copy only if something is filtered out
        uint64_t n = t->n;
        uint64_t nb = std::min<uint64_t>((n + 1023) / 1024, 2048);
        uint64_t per = ((n + nb - 1) / nb + 1023) / 1024 * 1024;
        nb = (n + per - 1) / per;
        mf_buf<uint32_t> bcount; MF_TRY(bcount.alloc(ctx, nb));
        mf_buf<uint64_t> boff; MF_TRY(boff.alloc(ctx, nb + 1));
        mf_buf<uint64_t> tot; MF_TRY(tot.alloc(ctx, 1));
        k_select_count<0><<<(unsigned)nb, 1024, 0, ctx->stream>>>(t->d_keys, t->d_counts, nullptr, n, per, threshold, bcount.p);
        k_scan<1><<<1, 1024, 0, ctx->stream>>>(bcount.p, boff.p, nb, tot.p);
        uint64_t m = 0;
        MF_HIP(hipMemcpyAsync(&m, tot.p, 8, hipMemcpyDeviceToHost, ctx->stream));
        MF_HIP(hipStreamSynchronize(ctx->stream));
        all_pass = m == n;
    }
    if (all_pass) {
        // nothing to filter: an alias that borrows the arrays AND the index of t (built here on t, so that later users
        // of t -- the features step -- find it instead of building their own)
        MF_TRY(mf_table_ensure_index(const_cast<mf_table *>(t)));
        MF_TRY(mf_table_adopt(ctx, t->k, t->n, 0, t->d_keys, t->keys_bytes, t->d_counts, t->counts_bytes, out));
        (*out)->owns_arrays = false;
        (*out)->cut_thr = std::max(t->cut_thr, threshold);
        (*out)->dropped_hist = t->dropped_hist;
        (*out)->index = t->index; (*out)->index_bytes = t->index_bytes;
        (*out)->part_bits = t->part_bits; (*out)->part_skm = t->part_skm; (*out)->d_part_off = t->d_part_off; (*out)->part_off_bytes = t->part_off_bytes;
        return MF_OK;
    }
    return mf_table_filter(t, threshold, out);
}

// host arrays -> table (insert-or-add with saturation through the HBM index)
// ---- a table with MINIMIZER partitions from pairs whose keys are all different (a .kmers.bin file): partition hash per
// key, stable radix sort of (partition, position), gather, partition offsets by binary search.  The table then gets the
// per-partition index and its cache-local neighbour probes, like a table that comes out of the counting pass.
__global__ void k_pairs_part(const uint64_t *__restrict__ keys, uint64_t n, int k, int bits, uint32_t *__restrict__ part, uint32_t *__restrict__ pos,
                             unsigned int *__restrict__ unsorted) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t key = keys[i];
    part[i] = mf_skm_ph(key, k) >> (32 - bits);
    pos[i] = (uint32_t)i;
    if (i + 1 < n && keys[i + 1] <= key) atomicExch(unsorted, 1u);        // (not strictly ascending: duplicates are possible)
}
__global__ void k_pairs_gather(const uint64_t *__restrict__ keys, const uint16_t *__restrict__ vals, const uint32_t *__restrict__ pos, uint64_t n,
                               uint64_t *__restrict__ ok, uint16_t *__restrict__ ov) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const uint32_t s = pos[i]; ok[i] = keys[s]; ov[i] = vals[s]; }
}
__global__ void k_pairs_offsets(const uint32_t *__restrict__ part_sorted, uint64_t n, uint32_t np, uint64_t *__restrict__ off) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p > np) return;
    uint64_t lo = 0, hi = n;                      // first position with part >= p
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (part_sorted[mid] < p) lo = mid + 1; else hi = mid; }
    off[p] = lo;
}
// returns 1 if the pairs do not qualify (k too small for minimizers, keys not strictly ascending, too many)
static int table_from_unique_pairs(mf_ctx *ctx, const uint64_t *d_keys, const uint16_t *d_vals, uint64_t n, int k, mf_table **out) {
    if (k < MF_SKM_MIN_K || n < 4096 || n >= 0xFFFFFFFFull) return 1;
    int bits = 1; while (bits < 26 && (n >> bits) > 96) bits++;
    const uint32_t np = 1u << bits;
    mf_buf<uint32_t> part, pos, part2, pos2; mf_buf<unsigned int> flag;
    MF_TRY(part.alloc(ctx, n)); MF_TRY(pos.alloc(ctx, n)); MF_TRY(part2.alloc(ctx, n)); MF_TRY(pos2.alloc(ctx, n)); MF_TRY(flag.alloc(ctx, 1));
    MF_HIP(hipMemsetAsync(flag.p, 0, 4, ctx->stream));
    k_pairs_part<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(d_keys, n, k, bits, part.p, pos.p, flag.p);
    unsigned int uns = 0;
    MF_HIP(hipMemcpyAsync(&uns, flag.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    MF_HIP(hipStreamSynchronize(ctx->stream));
    if (uns) return 1;
    MF_TRY(mf_sort_u32_pairs(ctx, part.p, pos.p, n, bits, part2.p, pos2.p));
    mf_buf<uint64_t> ok; mf_buf<uint16_t> oc; mf_buf<uint64_t> off;
    MF_TRY(ok.alloc(ctx, n)); MF_TRY(oc.alloc(ctx, n)); MF_TRY(off.alloc(ctx, (size_t)np + 1));
    k_pairs_gather<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(d_keys, d_vals, pos2.p, n, ok.p, oc.p);
    k_pairs_offsets<<<(np + 1 + 255) / 256, 256, 0, ctx->stream>>>(part2.p, n, np, off.p);
    MF_HIP(hipStreamSynchronize(ctx->stream));
    size_t kb = ok.bytes(), cb = oc.bytes();
    MF_TRY(mf_table_adopt(ctx, k, n, 0, ok.take(), kb, oc.take(), cb, out));
    (*out)->part_bits = bits; (*out)->part_skm = 1;
    (*out)->part_off_bytes = off.bytes(); (*out)->d_part_off = off.take();
    return MF_OK;
}
int mf_table_from_device_pairs(mf_ctx *ctx, const uint64_t *d_keys, const uint16_t *d_vals, uint64_t n, int k, mf_table **out) {
    {
        const int rc = table_from_unique_pairs(ctx, d_keys, d_vals, n, k, out);
        if (rc <= 0) return rc;
    }
    uint64_t cap = pow2_at_least(std::max<uint64_t>(2 * n, 1024));
    mf_buf<mf_slot> slots; MF_TRY(slots.alloc(ctx, cap));
    unsigned grid = (unsigned)std::min<uint64_t>((cap + 255) / 256, 65536);
    k_index_init<<<grid, 256, 0, ctx->stream>>>(slots.p, cap);
    if (n) {
        unsigned g2 = (unsigned)std::min<uint64_t>((n + 255) / 256, 65536);
        mf_ktimer t(ctx, "k_index_insert_add");
        k_index_insert_add<<<g2, 256, 0, ctx->stream>>>(slots.p, cap - 1, d_keys, d_vals, n);
    }
    mf_buf<uint64_t> ok; mf_buf<uint16_t> oc; uint64_t m = 0;
    MF_TRY(select_entries<1>(ctx, nullptr, nullptr, slots.p, cap, 0, ok, oc, &m));
    size_t kb = ok.bytes(), cb = oc.bytes();
    return mf_table_adopt(ctx, k, m, 0, ok.take(), kb, oc.take(), cb, out);
}

extern "C" int mf_table_from_host(mf_ctx *ctx, const uint64_t *keys, const uint16_t *counts, uint64_t n, int k, mf_table **out) {
    if (!ctx || !out || (n && (!keys || !counts))) return mf_set_error("NULL argument");
    *out = nullptr;
    if (k < 1 || k > 31) return mf_set_error("k must be in [1,31]");
    MF_HIP(hipSetDevice(ctx->device));
    mf_buf<uint64_t> dk; MF_TRY(dk.alloc(ctx, n));
    mf_buf<uint16_t> dc; MF_TRY(dc.alloc(ctx, n));
    if (n) {
        MF_HIP(hipMemcpyAsync(dk.p, keys, n * 8, hipMemcpyHostToDevice, ctx->stream));
        MF_HIP(hipMemcpyAsync(dc.p, counts, n * 2, hipMemcpyHostToDevice, ctx->stream));
    }
    int r = mf_table_from_device_pairs(ctx, dk.p, dc.p, n, k, out);
    MF_HIP(hipStreamSynchronize(ctx->stream));
    return r;
}

// the entries with count > threshold in ascending k-mer order, in HBM (sorted there: a host sort of 3.6e8 entries takes
// a minute, the device-wide radix sort 30 ms)
int mf_table_select_sorted(const mf_table *t, int threshold, mf_buf<uint64_t> &sk, mf_buf<uint16_t> &sc, uint64_t *n) {
    mf_ctx *ctx = t->ctx;
    MF_HIP(hipSetDevice(ctx->device));
    mf_buf<uint64_t> ok; mf_buf<uint16_t> oc; uint64_t m = 0;
    MF_TRY(select_entries<0>(ctx, t->d_keys, t->d_counts, nullptr, t->n, threshold, ok, oc, &m));
    *n = m;
    MF_TRY(sk.alloc(ctx, m)); MF_TRY(sc.alloc(ctx, m));
    return mf_sort_pairs(ctx, ok.p, oc.p, m, 2 * t->k, sk.p, sc.p);
}
// counts of the entries that pass (count > thr and value in the filter's index > fthr, absent = 0), 0 for the others
__global__ void k_mask_by_filter(mf_index_view fx, const uint64_t *__restrict__ keys, const uint16_t *__restrict__ cnts, uint64_t n, int thr, int fthr,
                                 uint16_t *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t idx, val = 0;
    const int c = cnts[i];
    int fv = 0;
    if (c > thr && fx.slots && mf_index_find(fx, keys[i], &idx, &val)) fv = (int)val;
    out[i] = (c > thr && fv > fthr) ? (uint16_t)c : (uint16_t)0;
}
// IOUtils.filterAndPrintKmers (src/io/IOUtils.java:101-123): selection + ascending order, in HBM
int mf_table_select_filtered_sorted(const mf_table *t, int threshold, mf_table *filter, int filter_threshold, mf_buf<uint64_t> &sk, mf_buf<uint16_t> &sc,
                                    uint64_t *n) {
    mf_ctx *ctx = t->ctx;
    MF_HIP(hipSetDevice(ctx->device));
    *n = 0;
    MF_TRY(sk.alloc(ctx, 0)); MF_TRY(sc.alloc(ctx, 0));
    if (!t->n) return MF_OK;
    if (filter->n) MF_TRY(mf_table_ensure_index(filter));
    mf_buf<uint16_t> masked; MF_TRY(masked.alloc(ctx, t->n));
    mf_index_view fx = mf_view(filter->index);
    if (!filter->n) fx.slots = nullptr;
    k_mask_by_filter<<<(unsigned)((t->n + 255) / 256), 256, 0, ctx->stream>>>(fx, t->d_keys, t->d_counts, t->n, threshold, filter_threshold, masked.p);
    mf_buf<uint64_t> ok; mf_buf<uint16_t> oc; uint64_t m = 0;
    MF_TRY(select_entries<0>(ctx, t->d_keys, masked.p, nullptr, t->n, 0, ok, oc, &m));
    *n = m;
    MF_TRY(sk.alloc(ctx, m)); MF_TRY(sc.alloc(ctx, m));
    return mf_sort_pairs(ctx, ok.p, oc.p, m, 2 * t->k, sk.p, sc.p);
}
extern "C" int mf_table_export(const mf_table *t, int threshold, uint64_t *keys, uint16_t *counts, uint64_t cap, uint64_t *n) {
    if (!t || !n) return mf_set_error("NULL argument");
    mf_ctx *ctx = t->ctx;
    MF_HIP(hipSetDevice(ctx->device));
    if (cap == 0 || !keys || !counts) {               // size query
        mf_buf<uint64_t> ok; mf_buf<uint16_t> oc;
        return select_entries<0>(ctx, t->d_keys, t->d_counts, nullptr, t->n, threshold, ok, oc, n);
    }
    mf_buf<uint64_t> sk; mf_buf<uint16_t> sc; uint64_t m = 0;
    MF_TRY(mf_table_select_sorted(t, threshold, sk, sc, &m));
    *n = m;
    if (cap < m) return mf_set_error("mf_table_export: capacity %llu < %llu", (unsigned long long)cap, (unsigned long long)m);
    if (m) {
        MF_HIP(hipMemcpyAsync(keys, sk.p, m * 8, hipMemcpyDeviceToHost, ctx->stream));
        MF_HIP(hipMemcpyAsync(counts, sc.p, m * 2, hipMemcpyDeviceToHost, ctx->stream));
        MF_HIP(hipStreamSynchronize(ctx->stream));
    }
    return MF_OK;
}

extern "C" int mf_table_lookup(mf_table *t, const uint64_t *keys, uint64_t n, int32_t *values) {
    if (!t || (n && (!keys || !values))) return mf_set_error("NULL argument");
    mf_ctx *ctx = t->ctx;
    MF_HIP(hipSetDevice(ctx->device));
    MF_TRY(mf_table_ensure_index(t));
    if (!n) return MF_OK;
    mf_buf<uint64_t> dk; MF_TRY(dk.alloc(ctx, n));
    mf_buf<int32_t> dv; MF_TRY(dv.alloc(ctx, n));
    MF_HIP(hipMemcpyAsync(dk.p, keys, n * 8, hipMemcpyHostToDevice, ctx->stream));
    k_index_lookup<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(mf_view(t->index), dk.p, n, dv.p);
    MF_HIP(hipMemcpyAsync(values, dv.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    MF_HIP(hipStreamSynchronize(ctx->stream));
    return MF_OK;
}

// histogram of all counts -> host vector[32768]
int mf_table_count_hist(const mf_table *t, std::vector<uint64_t> &hist) {
    mf_ctx *ctx = t->ctx;
    hist.assign(MF_MAX_COUNT + 1, 0);
    if (!t->n) { for (size_t c = 0; c < t->dropped_hist.size() && c < hist.size(); c++) hist[c] += t->dropped_hist[c]; return MF_OK; }
    mf_buf<unsigned long long> dh; MF_TRY(dh.alloc(ctx, MF_MAX_COUNT + 1));
    MF_HIP(hipMemsetAsync(dh.p, 0, dh.bytes(), ctx->stream));
    unsigned grid = (unsigned)std::min<uint64_t>((t->n + 255) / 256, 2048);
    {
        mf_ktimer tm(ctx, "k_count_hist");
        k_count_hist<<<grid, 256, 0, ctx->stream>>>(t->d_counts, t->n, dh.p);
    }
    MF_HIP(hipMemcpyAsync(hist.data(), dh.p, (MF_MAX_COUNT + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    MF_HIP(hipStreamSynchronize(ctx->stream));
    // the k-mers a cut inside the counting pass kept out of the table (count <= cut_thr) were tallied there
    for (size_t c = 0; c < t->dropped_hist.size() && c < hist.size(); c++) hist[c] += t->dropped_hist[c];
    return MF_OK;
}
// QuickQuantitativeStatistics of IOUtils.printKmers (src/io/IOUtils.java:45-71): number of distinct k-mers per count,
// over ALL k-mers that were counted -- also those a cut has dropped from the table
extern "C" int mf_table_hist(const mf_table *t, uint64_t *hist) {
    if (!t || !hist) return mf_set_error("mf_table_hist: NULL argument");
    MF_HIP(hipSetDevice(t->ctx->device));
    std::vector<uint64_t> h;
    MF_TRY(mf_table_count_hist(t, h));
    memcpy(hist, h.data(), (size_t)(MF_MAX_COUNT + 1) * 8);
    return MF_OK;
}

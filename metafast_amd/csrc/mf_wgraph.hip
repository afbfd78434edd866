// mf_wgraph.hip -- NO-REFERENCE EXTENSION: unitigs, components and features for 32 <= k <= 63 (2k-bit k-mers).
//
// The reference rejects k > 31 (src/tools/KmersCounterMain.java:66-73: one Java long per k-mer); BASELINE.json's config 4 names a k = 63
// leg whose metric is "counted + graphed".  Nothing here replaces a reference function or takes part in a parity claim; the checker is
// the test suite's CPU restatement compiled for 128-bit keys (the same text that is pinned at 64 bits).  The definitions are the k <= 31 ones on 2k-bit
// numbers: getLeft/RightNucleotide (src/algo/HashMapOperations.java:13-47), the walk and emission rule of
// AddSequencesShiftingRightTask.java:40-123, possibleNeighbours (src/algo/KmerOperations.java:9-26), ComponentsBuilder.java:58-270,
// FeaturesCalculatorMain.buildAndPrintVector :169-236.
//
// Design: only the NEIGHBOUR LOOK-UP sees 128-bit keys.  A wide table (mf_wide.hip) is ascending, so a k-mer's place in it is a 32-bit
// vertex id that orders like the k-mer: one kernel per stage turns every vertex into the ids of its eight neighbours through an
// open-addressed index in HBM (8-byte slots: position | 32-bit tag, load <= 0.5; a probe reads the two key words only behind a matching
// tag), and from there the k <= 31 machinery runs unchanged on ids -- links, jump words, walks, Wyllie doubling, emission (mf_ut_build,
// mf_unitig.hip), tile + HBM union-find, threshold levels, members (mf_cc_build, mf_cc.hip).  The components come out as ids and are
// turned back into k-mers once, at the end.
#include <algorithm>
#include <memory>
#include "mf_common.h"
#include "mf_wide.h"
#include "mf_unitig.h"
#include "mf_cc.h"


static inline unsigned ggrid(uint64_t n, unsigned bs = 256) { return (unsigned)std::min<uint64_t>((n + bs - 1) / bs, 0x7FFFFFFFull); }

// ---------------------------------------------------------------------------------------------
// index
// ---------------------------------------------------------------------------------------------
__global__ void k_windex_build(unsigned long long *__restrict__ slots, uint64_t mask, const uint64_t *__restrict__ hi, const uint64_t *__restrict__ lo, uint64_t n, int k) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t h = mf_whash_key(((mf_u128)hi[i] << 64) | (mf_u128)lo[i], k);          // (the canonical interior's: mf_wide.h)
    const unsigned long long ent = (h & 0xFFFFFFFF00000000ull) | (unsigned long long)(uint32_t)i;
    uint64_t s = h & mask;
    for (;;) {                                               // (the keys of a table are distinct)
        if (slots[s] == MF_WIDX_EMPTY && atomicCAS(&slots[s], MF_WIDX_EMPTY, ent) == MF_WIDX_EMPTY) break;
        s = (s + 1) & mask;
    }
}
int mf_wtable_ensure_index(mf_wtable *t) {
    MF_TRY(mf_wtable_flatten(t));
    if (t->index.p || !t->n) return MF_OK;
    if (t->n >= 0xFFFFFFFFull) return mf_set_error("wide table: more than 2^32 entries cannot be indexed");
    mf_ctx *ctx = t->ctx;
    uint64_t cap = 1024;
    while (cap < 2 * t->n) cap <<= 1;
    MF_TRY(t->index.alloc(ctx, cap));
    t->index_mask = cap - 1;
    mf_ktimer tm(ctx, "k_windex_build");
    MF_HIP(hipMemsetAsync(t->index.p, 0xFF, cap * 8, ctx->stream));
    auto &pc = *t->pieces[0];
    k_windex_build<<<ggrid(t->n), 256, 0, ctx->stream>>>(t->index.p, t->index_mask, pc.hi.p, pc.lo.p, t->n, t->k);
    return MF_OK;
}
static mf_windex_view wview(const mf_wtable *t) {
    mf_windex_view v{};
    v.k = t->k;
    if (t->n) { auto &pc = *t->pieces[0]; v.slots = t->index.p; v.mask = t->index_mask; v.hi = pc.hi.p; v.lo = pc.lo.p; v.cnt = pc.cnt.p; v.n = t->n; }
    return v;
}

// ---------------------------------------------------------------------------------------------
// the eight neighbours of a vertex: idx[2 nuc] = x[1..] + nuc (right), idx[2 nuc + 1] = nuc + x[..k-2] (left), as mf_nbr.h; bit i of
// *flip: the table holds neighbour i as its reverse complement (a palindrome counts as itself)
// ---------------------------------------------------------------------------------------------
// The neighbours of ONE side of x in one probe sequence (mf_index_walk_side of the k <= 31 tables on two words): the four k-mers y_c share k - 1
// bases with x.  pa = what (K >> 2) of a stored key K is if K is y_c on the right side (x's last k - 1 bases) or rc(y_c) on the left (rc(x)'s
// last k - 1 bases); pb = what K's low k - 1 bases are if K is rc(y_c) on the right / y_c on the left.  side 0: right (y_c = x[1..] + c), 1: left
// (y_c = c + x[..k-2]).  out[c] = table index or 0xFFFFFFFF; bit c of the result: the table holds neighbour c as its reverse complement (a
// palindrome is both and counts as itself).
__device__ __forceinline__ uint32_t w_side_walk(const mf_windex_view &ix, mf_u128 x, mf_u128 rcx, int k, uint32_t side, uint32_t (&out)[4]) {
    const mf_u128 LM = (((mf_u128)1) << (2 * k - 2)) - 1, WM = LM >> 2;
    const mf_u128 pa = side ? (rcx & LM) : (x & LM), pb = side ? (x >> 2) : (rcx >> 2);
    const uint64_t h = mf_whash_interior(side ? (x >> 4) : (x & WM), side ? (rcx & WM) : (rcx >> 4));      // the interior the four share, and its reverse complement
    const uint32_t tag = (uint32_t)(h >> 32);
    out[0] = out[1] = out[2] = out[3] = 0xFFFFFFFFu;
    uint32_t rv = 0;
    uint64_t s = h & ix.mask;
    for (;;) {
        const unsigned long long v = ix.slots[s];
        if (v == MF_WIDX_EMPTY) break;
        if ((uint32_t)(v >> 32) == tag) {
            const uint32_t p = (uint32_t)v;
            const mf_u128 K = ((mf_u128)ix.hi[p] << 64) | (mf_u128)ix.lo[p];
            const bool ma = (K >> 2) == pa, mb = (K & LM) == pb, fwd = side ? mb : ma;
            if (ma || mb) {
                const uint32_t ca = ((uint32_t)K & 3u) ^ (side ? 3u : 0u), cb = (uint32_t)(K >> (2 * k - 2)) ^ (side ? 0u : 3u);
                const uint32_t c = fwd ? (side ? cb : ca) : (side ? ca : cb);
#pragma unroll
                for (uint32_t q = 0; q < 4; q++) if (c == q) out[q] = p;
                rv = (rv & ~(1u << c)) | ((fwd ? 0u : 1u) << c);
            }
        }
        s = (s + 1) & ix.mask;
    }
    return rv;
}
// the eight neighbours of a vertex: idx[2 nuc] = x[1..] + nuc (right), idx[2 nuc + 1] = nuc + x[..k-2] (left), as mf_nbr.h; bit i of *flip: the
// table holds neighbour i as its reverse complement
__device__ __forceinline__ void w_neighbours(const mf_windex_view &ix, mf_u128 x, int k, uint32_t (&idx)[8], uint32_t *flip, bool *pal) {
    const mf_u128 rcx = mf_wrevcomp(x, k);
    *pal = rcx == x;
    uint32_t r4[4], l4[4];
    const uint32_t rr = w_side_walk(ix, x, rcx, k, 0u, r4), rl = w_side_walk(ix, x, rcx, k, 1u, l4);
    uint32_t fl = 0;
#pragma unroll
    for (uint32_t c = 0; c < 4; c++) {
        idx[2 * c] = r4[c]; idx[2 * c + 1] = l4[c];
        fl |= ((rr >> c) & 1u) << (2 * c);
        fl |= ((rl >> c) & 1u) << (2 * c + 1);
    }
    *flip = fl;
}
// U1 (mf_unitig.hip) for a wide table: unique right / left neighbour, its index and strand
#define WUT_NONE 0xFFFFFFFFu
#define WUT_CODE_NONE 4u
#define WUT_CODE_MANY 5u
__global__ __launch_bounds__(256) void k_w_ut_flags(mf_windex_view ix, ut_arrays A) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n) return;
    const mf_u128 x = ((mf_u128)A.ghi[i] << 64) | (mf_u128)A.gk[i];
    uint32_t idx[8], flip; bool pal;
    w_neighbours(ix, x, A.k, idx, &flip, &pal);
    uint32_t rcode = WUT_CODE_NONE, lcode = WUT_CODE_NONE, ridx = WUT_NONE, lidx = WUT_NONE, ror = 0, lor = 0;
#pragma unroll
    for (uint32_t nuc = 0; nuc < 4; nuc++) {
        if (idx[2 * nuc] != WUT_NONE) {
            if (rcode == WUT_CODE_NONE) { rcode = nuc; ridx = idx[2 * nuc]; ror = (flip >> (2 * nuc)) & 1u; }
            else rcode = WUT_CODE_MANY;
        }
        if (idx[2 * nuc + 1] != WUT_NONE) {
            if (lcode == WUT_CODE_NONE) { lcode = nuc; lidx = idx[2 * nuc + 1]; lor = (flip >> (2 * nuc + 1)) & 1u; }
            else lcode = WUT_CODE_MANY;
        }
    }
    A.info[i] = (uint8_t)(rcode | (lcode << 3) | (ror << 6) | (lor << 7));
    if (A.pal) A.pal[i] = (uint8_t)pal;
    A.ridx[i] = ridx;
    A.lidx[i] = lidx;
}
// C1 (mf_cc.hip) for a wide table
__global__ __launch_bounds__(256) void k_w_cc_adjacency(mf_windex_view ix, uint64_t n, int k, uint32_t *__restrict__ nbr) {
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const mf_u128 x = ((mf_u128)ix.hi[v] << 64) | (mf_u128)ix.lo[v];
    uint32_t idx[8], flip; bool pal;
    w_neighbours(ix, x, k, idx, &flip, &pal);
    uint4 *o = reinterpret_cast<uint4 *>(nbr + v * 8);
    o[0] = make_uint4(idx[0], idx[1], idx[2], idx[3]);
    o[1] = make_uint4(idx[4], idx[5], idx[6], idx[7]);
}

// ---------------------------------------------------------------------------------------------
// A7  unitigs
// ---------------------------------------------------------------------------------------------
extern "C" int mf_build_unitigs_wide_device(mf_ctx *ctx, mf_wtable *t, int freq_threshold, int min_len, mf_seqs **out) {
    mf_range rng_("mf:unitigs_wide");
    if (!ctx || !t || !out) return mf_set_error("mf_build_unitigs_wide_device: NULL argument");
    *out = nullptr;
    if (t->ctx != ctx) return mf_set_error("mf_build_unitigs_wide_device: the table belongs to another context");
    MF_HIP(hipSetDevice(ctx->device));
    // nodes = k-mers with value > freqThreshold (task.run :46-48): the table itself when a cut has made sure of it, else a filtered copy
    mf_wtable *g = t;
    std::unique_ptr<mf_wtable> own;
    if (t->cut_thr < freq_threshold) {
        mf_wtable *f = nullptr;
        MF_TRY(mf_wtable_filter(t, freq_threshold, &f));
        own.reset(f); g = f;
    }
    if (g->n && g->n < 0x7FFFFFFFull) MF_TRY(mf_wtable_ensure_index(g));
    const mf_windex_view ix = wview(g);
    // (an empty table: no high words to point at -- mf_ut_build returns its empty result before it looks)
    return mf_ut_build(ctx, ix.lo, ix.hi, ix.cnt, g->n, g->k, 0, nullptr, min_len, [&](const ut_arrays &A) -> int {
        k_w_ut_flags<<<ggrid(A.n), 256, 0, ctx->stream>>>(ix, A);
        return MF_OK;
    }, out);
}

// ---------------------------------------------------------------------------------------------
// A10  components
// ---------------------------------------------------------------------------------------------
struct mf_wcomps {
    mf_ctx *ctx = nullptr;
    int k = 0;
    uint64_t n = 0, n_kmers = 0;
    std::vector<uint64_t> sizes; std::vector<int64_t> weights; std::vector<int32_t> thr;
    // the member k-mers grouped by component (final order), ascending inside a component; comp[j] = component of k-mer j
    mf_buf<uint64_t> hi, lo; mf_buf<uint32_t> comp;
};
__global__ void k_w_gather(const uint64_t *__restrict__ ids, uint64_t n, const uint64_t *__restrict__ thi, const uint64_t *__restrict__ tlo, uint64_t *__restrict__ ohi,
                           uint64_t *__restrict__ olo) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint64_t v = ids[j];
    ohi[j] = thi[v]; olo[j] = tlo[v];
}
__global__ void k_w_comp_of(const uint64_t *__restrict__ off, uint32_t nc, uint64_t n, uint32_t *__restrict__ comp) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    uint32_t lo = 0, hi = nc;                                // the last component with off <= j
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (off[mid] <= j) lo = mid; else hi = mid; }
    comp[j] = lo;
}
extern "C" int mf_cut_components_wide_device(mf_ctx *ctx, mf_wtable *t, int b1, int b2, mf_wcomps **out) {
    mf_range rng_("mf:components_wide");
    if (!ctx || !t || !out) return mf_set_error("mf_cut_components_wide_device: NULL argument");
    *out = nullptr;
    if (t->ctx != ctx) return mf_set_error("mf_cut_components_wide_device: the table belongs to another context");
    MF_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (t->n >= 0xFFFFFFFFull) return mf_set_error("components: more than 2^32 vertices is not supported");
    MF_TRY(mf_wtable_ensure_ascending(t));                // (vertex ids stand for k-mers here: members come out as ids, ties go to the smallest id)
    if (t->n) MF_TRY(mf_wtable_ensure_index(t));
    const mf_windex_view ix = wview(t);
    mf_comps *inner = nullptr;
    // (d_keys = nullptr: the members come out as vertex ids, ties between components are broken by the smallest id = the smallest k-mer)
    MF_TRY(mf_cc_build(ctx, t->n, 16, ix.cnt, nullptr, b1, b2, [&](uint32_t *nbr) -> int {
        k_w_cc_adjacency<<<ggrid(t->n), 256, 0, st>>>(ix, t->n, t->k, nbr);
        return MF_OK;
    }, &inner));
    struct guard { mf_comps *p; ~guard() { mf_comps_destroy(p); } } gg{inner};
    auto C = std::make_unique<mf_wcomps>();
    C->ctx = ctx; C->k = t->k; C->n = inner->n; C->n_kmers = inner->n_kmers;
    C->sizes = inner->sizes; C->weights = inner->weights; C->thr = inner->thr;
    const uint64_t nk = inner->n_kmers;
    MF_TRY(C->hi.alloc(ctx, nk)); MF_TRY(C->lo.alloc(ctx, nk)); MF_TRY(C->comp.alloc(ctx, nk));
    if (nk) {
        mf_buf<uint64_t> sorted, off;
        MF_TRY(sorted.alloc(ctx, nk)); MF_TRY(off.alloc(ctx, C->n + 1));
        MF_TRY(mf_sort_kmers_by_comp(ctx, inner->d_comp, inner->d_kmers, nk, 32, (uint32_t)std::max<uint64_t>(C->n, 1), sorted.p));
        std::vector<uint64_t> h_off(C->n + 1, 0);
        for (uint64_t c = 0; c < C->n; c++) h_off[c + 1] = h_off[c] + C->sizes[c];
        MF_HIP(hipMemcpyAsync(off.p, h_off.data(), (C->n + 1) * 8, hipMemcpyHostToDevice, st));
        k_w_gather<<<ggrid(nk), 256, 0, st>>>(sorted.p, nk, ix.hi, ix.lo, C->hi.p, C->lo.p);
        k_w_comp_of<<<ggrid(nk), 256, 0, st>>>(off.p, (uint32_t)C->n, nk, C->comp.p);
        MF_HIP(hipStreamSynchronize(st));
    }
    *out = C.release();
    return MF_OK;
}
extern "C" void mf_wcomps_destroy(mf_wcomps *c) { delete c; }
extern "C" int mf_wcomps_stats(const mf_wcomps *c, uint64_t *n_comp, uint64_t *n_kmers) {
    if (!c) return mf_set_error("wide comps is NULL");
    if (n_comp) *n_comp = c->n;
    if (n_kmers) *n_kmers = c->n_kmers;
    return MF_OK;
}
extern "C" int mf_wcomps_export(const mf_wcomps *c, uint64_t *sizes, int64_t *weights, int32_t *thr, uint64_t *kmer_offsets, uint64_t *kmers_hi, uint64_t *kmers_lo) {
    if (!c) return mf_set_error("wide comps is NULL");
    MF_HIP(hipSetDevice(c->ctx->device));
    if (sizes && c->n) memcpy(sizes, c->sizes.data(), c->n * 8);
    if (weights && c->n) memcpy(weights, c->weights.data(), c->n * 8);
    if (thr && c->n) memcpy(thr, c->thr.data(), c->n * 4);
    if (kmer_offsets) { kmer_offsets[0] = 0; for (uint64_t i = 0; i < c->n; i++) kmer_offsets[i + 1] = kmer_offsets[i] + c->sizes[i]; }
    if (kmers_hi && c->n_kmers) MF_HIP(hipMemcpy(kmers_hi, c->hi.p, c->n_kmers * 8, hipMemcpyDeviceToHost));
    if (kmers_lo && c->n_kmers) MF_HIP(hipMemcpy(kmers_lo, c->lo.p, c->n_kmers * 8, hipMemcpyDeviceToHost));
    return MF_OK;
}

// ---------------------------------------------------------------------------------------------
// A12  features: a thread per component k-mer probes the sample's index (k_features_rev of mf_cc.hip on wide keys)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_w_features(mf_windex_view ix, const uint64_t *__restrict__ chi, const uint64_t *__restrict__ clo, const uint32_t *__restrict__ comp_of,
                                                    uint64_t nk, int threshold, unsigned long long *__restrict__ vec, unsigned int *__restrict__ found) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t comp = 0xFFFFFFFFu, val = 0, hit = 0;
    if (j < nk) {
        comp = comp_of[j];
        const uint32_t p = mf_windex_find(ix, chi[j], clo[j]);
        if (p != 0xFFFFFFFFu) { const uint32_t v = ix.cnt[p]; if ((int)v > threshold) { val = v; hit = 1; } }
    }
    const uint32_t first = __shfl(comp, 0, 64);
    if (__ballot(comp != first) == 0ull) {
        for (int d = 32; d >= 1; d >>= 1) { val += __shfl_down(val, d, 64); hit += __shfl_down(hit, d, 64); }
        if (mf_lane() == 0 && hit) { atomicAdd(&vec[first], (unsigned long long)val); atomicAdd(&found[first], hit); }
    } else if (hit) {
        atomicAdd(&vec[comp], (unsigned long long)val);
        atomicAdd(&found[comp], 1u);
    }
}
extern "C" int mf_features_wide_device(mf_ctx *ctx, mf_wcomps *c, mf_wtable *sample, int threshold, int64_t *vec, double *breadth) {
    mf_range rng_("mf:features_wide");
    if (!ctx || !c || !sample || !vec) return mf_set_error("mf_features_wide_device: NULL argument");
    if (sample->ctx != ctx || c->ctx != ctx) return mf_set_error("mf_features_wide_device: a handle belongs to another context");
    if (sample->k != c->k) return mf_set_error("mf_features_wide_device: components of %d-mers, sample of %d-mers", c->k, sample->k);
    MF_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint64_t nc = c->n;
    if (!nc) return MF_OK;
    mf_buf<unsigned long long> dvec; mf_buf<unsigned int> dfound;
    MF_TRY(dvec.alloc(ctx, nc)); MF_TRY(dfound.alloc(ctx, nc));
    MF_HIP(hipMemsetAsync(dvec.p, 0, nc * 8, st));
    MF_HIP(hipMemsetAsync(dfound.p, 0, nc * 4, st));
    if (sample->n && c->n_kmers) {
        MF_TRY(mf_wtable_ensure_index(sample));
        mf_ktimer tm(ctx, "k_features");
        k_w_features<<<ggrid(c->n_kmers), 256, 0, st>>>(wview(sample), c->hi.p, c->lo.p, c->comp.p, c->n_kmers, threshold, dvec.p, dfound.p);
    }
    std::vector<unsigned int> hf(nc);
    MF_HIP(hipMemcpyAsync(vec, dvec.p, nc * 8, hipMemcpyDeviceToHost, st));
    MF_HIP(hipMemcpyAsync(hf.data(), dfound.p, nc * 4, hipMemcpyDeviceToHost, st));
    MF_HIP(hipStreamSynchronize(st));
    if (breadth)
        for (uint64_t i = 0; i < nc; i++) {
            // a negative threshold makes absent k-mers (value 0) count as found (value > threshold), as in mf_cc.hip
            const double f = threshold < 0 ? (double)c->sizes[i] : (double)hf[i];
            breadth[i] = f / (double)c->sizes[i];
        }
    return MF_OK;
}
// Long2ShortHashMap.get for a batch of host k-mers (as mf_table_lookup): values[i] = count or -1
__global__ void k_w_lookup(mf_windex_view ix, const uint64_t *__restrict__ hi, const uint64_t *__restrict__ lo, uint64_t n, int32_t *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = ix.n ? mf_windex_find(ix, hi[i], lo[i]) : 0xFFFFFFFFu;
    out[i] = p == 0xFFFFFFFFu ? -1 : (int32_t)ix.cnt[p];
}
extern "C" int mf_wtable_lookup(mf_wtable *t, const uint64_t *keys_hi, const uint64_t *keys_lo, uint64_t n, int32_t *values) {
    if (!t || (n && (!keys_hi || !keys_lo || !values))) return mf_set_error("mf_wtable_lookup: NULL argument");
    if (!n) return MF_OK;
    mf_ctx *ctx = t->ctx;
    MF_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (t->n) MF_TRY(mf_wtable_ensure_index(t));
    mf_buf<uint64_t> dh, dl; mf_buf<int32_t> dv;
    MF_TRY(dh.alloc(ctx, n)); MF_TRY(dl.alloc(ctx, n)); MF_TRY(dv.alloc(ctx, n));
    MF_HIP(hipMemcpyAsync(dh.p, keys_hi, n * 8, hipMemcpyHostToDevice, st));
    MF_HIP(hipMemcpyAsync(dl.p, keys_lo, n * 8, hipMemcpyHostToDevice, st));
    k_w_lookup<<<ggrid(n), 256, 0, st>>>(wview(t), dh.p, dl.p, n, dv.p);
    MF_HIP(hipMemcpyAsync(values, dv.p, n * 4, hipMemcpyDeviceToHost, st));
    MF_HIP(hipStreamSynchronize(st));
    return MF_OK;
}
// the table's lookup index, released (rebuilt when a graph stage needs it again): a rank that holds several samples
extern "C" int mf_wtable_drop_index(mf_wtable *t) {
    if (!t) return mf_set_error("wide table is NULL");
    t->index.reset(); t->index_mask = 0;
    return MF_OK;
}

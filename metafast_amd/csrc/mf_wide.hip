// mf_wide.hip -- NO-REFERENCE EXTENSION: canonical k-mer counts for 32 <= k <= 63 (128-bit k-mers).
//
// The reference stops at k = 31 (one Java long per k-mer; src/tools/KmersCounterMain.java:66-73 rejects k > 31), so nothing here
// replaces a reference function and nothing here takes part in any parity claim: BASELINE.json's config 4 asks for a k = 63 leg,
// SURVEY.md 8 asks for it as a labelled extension with a checker of its own (the test suite's 128-bit CPU restatement, or_count_wide).
// Same definitions as for k <= 31, on 2k-bit numbers: first base most significant (A0 G1 C2 T3), canonical = min(forward,
// reverse complement), counts saturate at 32767, reads shorter than max(k, min_len) give nothing.
//
// Not the hot path: one rolling pass over the reads writes the canonical k-mer of every valid start (two 64-bit words), a
// device-wide LSD radix sort (rocPRIM, low word then high word) orders them, run lengths are the counts.  32 bytes of HBM per
// k-mer occurrence twice over -- the super-k-mer machinery of mf_skm.hip (16-byte records of <= 50 bases) does not carry 63-mers.
#include <cstring>
#include <rocprim/rocprim.hpp>
#include "mf_common.h"
#include "mf_count_dev.h"

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
__global__ void k_wide_popc(const uint32_t *__restrict__ vmask, uint64_t n_words, uint32_t *__restrict__ cnt) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w < n_words) cnt[w] = (uint32_t)__popc(vmask[w]);
}
struct wide128 { uint64_t hi, lo; };
__device__ __forceinline__ bool wide_less(const wide128 &a, const wide128 &b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }
// one thread per 32-position word: rolls the forward and the reverse-complement k-mer over the word's valid starts
__global__ __launch_bounds__(256) void k_wide_kmers(const uint8_t *__restrict__ bases, uint64_t n_bases, const uint32_t *__restrict__ vmask,
                                                    const uint64_t *__restrict__ woff, uint64_t n_words, int k, uint64_t *__restrict__ out_hi,
                                                    uint64_t *__restrict__ out_lo) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    uint32_t m = vmask[w];
    if (!m) return;
    uint64_t o = woff[w];
    const int hb = 2 * k - 64;                                                   // bits of the k-mer in the high word (0 .. 62)
    const uint64_t hmask = hb >= 64 ? ~0ull : ((1ull << hb) - 1ull);
    const int first = __builtin_ctz(m), last = 31 - __builtin_clz(m);
    wide128 fw = {0, 0}, rc = {0, 0};
    const uint64_t b0 = w * 32 + (uint64_t)first;
    int have = 0;                                                                // bases rolled in so far
    for (uint64_t p = b0; p < w * 32 + (uint64_t)last + (uint64_t)k && p < n_bases; p++) {
        const uint32_t t = ((uint32_t)bases[p] >> 1) & 3u, x0 = t & 1u, x1 = t >> 1;
        const uint64_t c = (uint64_t)(((x0 ^ x1) << 1) | x1);
        fw.hi = ((fw.hi << 2) | (fw.lo >> 62)) & hmask; fw.lo = (fw.lo << 2) | c;
        if (hb >= 2) { rc.lo = (rc.lo >> 2) | (rc.hi << 62); rc.hi = (rc.hi >> 2) | ((3ull - c) << (hb - 2)); }
        else rc.lo = (rc.lo >> 2) | ((3ull - c) << 62);                          // (k = 32: the k-mer is the low word)
        have++;
        if (have >= k) {
            const uint32_t pos = (uint32_t)(p - (uint64_t)k + 1 - w * 32);        // start of the k-mer that ends at p
            if (pos < 32u && ((m >> pos) & 1u)) {
                const wide128 &cn = wide_less(rc, fw) ? rc : fw;
                out_hi[o] = cn.hi; out_lo[o] = cn.lo; o++;
            }
        }
    }
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
    mf_buf<uint32_t> vmask, wcnt; mf_buf<uint64_t> woff, tot;
    if (vmask.alloc(ctx, n_words) < 0 || wcnt.alloc(ctx, n_words) < 0 || woff.alloc(ctx, n_words + 1) < 0 || tot.alloc(ctx, 1) < 0) return fail(MF_ERR);
    {
        mf_ktimer tm(ctx, "k_wide_mask");
        k_wide_mask_init<<<wgrid(n_words), 256, 0, st>>>(vmask.p, n_words);
        k_wide_mask_reads<<<wgrid(n_reads), 256, 0, st>>>((const uint64_t *)d_offsets, n_reads, k, min_read_len, vmask.p);
        k_wide_popc<<<wgrid(n_words), 256, 0, st>>>(vmask.p, n_words, wcnt.p);
    }
    if (mf_scan<1>(ctx, wcnt.p, woff.p, n_words, tot.p) < 0) return fail(MF_ERR);
    uint64_t n_occ = 0;
    if (hipMemcpyAsync(&n_occ, tot.p, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail(mf_set_error("mf_count_wide_device: %s", hipGetErrorString(hipGetLastError())));
    t->n_occ = n_occ;
    if (!n_occ) return MF_OK;
    if (n_occ >= (1ull << 32)) return fail(mf_set_error("mf_count_wide_device: more than 2^32 k-mer occurrences is not supported on this path"));
    mf_buf<uint64_t> h0, l0, h1, l1;
    if (h0.alloc(ctx, n_occ) < 0 || l0.alloc(ctx, n_occ) < 0 || h1.alloc(ctx, n_occ) < 0 || l1.alloc(ctx, n_occ) < 0) return fail(MF_ERR);
    {
        mf_ktimer tm(ctx, "k_wide_kmers");
        k_wide_kmers<<<wgrid(n_words), 256, 0, st>>>((const uint8_t *)d_bases, n_bases, vmask.p, woff.p, n_words, k, h0.p, l0.p);
    }
    {   // ascending (hi, lo): LSD -- by the low word, then (stable) by the high word's 2k - 64 bits
        mf_ktimer tm(ctx, "k_wide_sort");
        size_t t1 = 0, t2 = 0;
        const unsigned hb = (unsigned)std::max(1, 2 * k - 64);
        if (rocprim::radix_sort_pairs(nullptr, t1, l0.p, l1.p, h0.p, h1.p, (size_t)n_occ, 0u, 64u, st) != hipSuccess ||
            rocprim::radix_sort_pairs(nullptr, t2, h1.p, h0.p, l1.p, l0.p, (size_t)n_occ, 0u, hb, st) != hipSuccess) return fail(mf_set_error("mf_count_wide_device: sort set-up failed"));
        mf_buf<uint8_t> tmp;
        if (tmp.alloc(ctx, std::max(t1, t2) + 1) < 0) return fail(MF_ERR);
        if (rocprim::radix_sort_pairs((void *)tmp.p, t1, l0.p, l1.p, h0.p, h1.p, (size_t)n_occ, 0u, 64u, st) != hipSuccess ||
            rocprim::radix_sort_pairs((void *)tmp.p, t2, h1.p, h0.p, l1.p, l0.p, (size_t)n_occ, 0u, hb, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) return fail(mf_set_error("mf_count_wide_device: sort failed: %s", hipGetErrorString(hipGetLastError())));
    }
    h1.reset(); l1.reset();
    // run lengths
    mf_buf<uint32_t> flag; mf_buf<uint64_t> idx;
    if (flag.alloc(ctx, n_occ) < 0 || idx.alloc(ctx, n_occ + 1) < 0) return fail(MF_ERR);
    k_wide_flags<<<wgrid(n_occ), 256, 0, st>>>(h0.p, l0.p, n_occ, flag.p);
    if (mf_scan<1>(ctx, flag.p, idx.p, n_occ, tot.p) < 0) return fail(MF_ERR);
    uint64_t nd = 0;
    if (hipMemcpyAsync(&nd, tot.p, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail(mf_set_error("mf_count_wide_device: %s", hipGetErrorString(hipGetLastError())));
    mf_buf<uint64_t> start;
    if (t->hi.alloc(ctx, nd) < 0 || t->lo.alloc(ctx, nd) < 0 || t->cnt.alloc(ctx, nd) < 0 || start.alloc(ctx, nd) < 0) return fail(MF_ERR);
    {
        mf_ktimer tm(ctx, "k_wide_runs");
        k_wide_heads<<<wgrid(n_occ), 256, 0, st>>>(h0.p, l0.p, flag.p, idx.p, n_occ, t->hi.p, t->lo.p, start.p);
        k_wide_counts<<<wgrid(nd), 256, 0, st>>>(start.p, nd, n_occ, t->cnt.p);
    }
    if (hipStreamSynchronize(st) != hipSuccess) return fail(mf_set_error("mf_count_wide_device: %s", hipGetErrorString(hipGetLastError())));
    t->n = nd;
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

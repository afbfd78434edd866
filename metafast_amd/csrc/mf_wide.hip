// mf_wide.hip -- NO-REFERENCE EXTENSION: canonical k-mer counts for 32 <= k <= 63 (128-bit k-mers).
//
// The reference stops at k = 31 (one Java long per k-mer; src/tools/KmersCounterMain.java:66-73 rejects k > 31), so nothing here
// replaces a reference function and nothing here takes part in any parity claim: BASELINE.json's config 4 asks for a k = 63 leg,
// SURVEY.md 8 asks for it as a labelled extension with a checker of its own (the test suite's 128-bit CPU restatement, or_count_wide).
// Same definitions as for k <= 31, on 2k-bit numbers: first base most significant (A0 G1 C2 T3), canonical = min(forward,
// reverse complement), counts saturate at 32767, reads shorter than max(k, min_len) give nothing.
//
// Not the hot path: one rolling pass over the reads writes the canonical k-mer of every valid start (two 64-bit words), a
// device-wide LSD radix sort (mf_sort.hip, low word then high word) orders them, run lengths are the counts.  32 bytes of HBM per
// k-mer occurrence twice over -- the super-k-mer machinery of mf_skm.hip (16-byte records of <= 50 bases) does not carry 63-mers.
#include <cstring>
#include <memory>
#include <vector>
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
// one thread per 32-position word: rolls the forward and the reverse-complement k-mer over the word's valid starts.
// PASSES (round 4): a sample of 200 M reads has 1.8e10 63-mers -- 32 bytes each, twice over, and more than the 2^32 entries the
// sort takes.  With pbits > 0 only the k-mers whose canonical value starts with the bits `pass` take part: ascending passes append
// ascending runs, i.e. the same table.  COUNT: the k-mers of the pass per word (-> wcnt) instead of the k-mers themselves.
template <bool COUNT>
__global__ __launch_bounds__(256) void k_wide_kmers(const uint8_t *__restrict__ bases, uint64_t n_bases, const uint32_t *__restrict__ vmask,
                                                    const uint64_t *__restrict__ woff, uint64_t n_words, int k, uint64_t *__restrict__ out_hi,
                                                    uint64_t *__restrict__ out_lo, int pbits, uint32_t pass, uint32_t *__restrict__ wcnt) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    uint32_t m = vmask[w];
    if (!m) { if (COUNT) wcnt[w] = 0; return; }
    uint64_t o = COUNT ? 0 : woff[w];
    uint32_t mine = 0;
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
                bool in = true;
                if (pbits) {                                                     // the top pbits of the 2k-bit number
                    const uint64_t top = hb >= pbits ? cn.hi >> (hb - pbits) : ((hb ? cn.hi << (pbits - hb) : 0ull) | (cn.lo >> (64 - (pbits - hb))));
                    in = (uint32_t)top == pass;
                }
                if (in) { if (COUNT) mine++; else { out_hi[o] = cn.hi; out_lo[o] = cn.lo; o++; } }
            }
        }
    }
    if (COUNT) wcnt[w] = mine;
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
    // passes: each takes the k-mers whose canonical value starts with its number (pbits bits).  A pass must stay under 2^32 occurrences
    // (the sort) and under a quarter of the device (four 8-byte arrays + the sort's scratch per occurrence); canonical k-mers crowd the low
    // prefixes (the smaller of two strands), hence two bits of margin.  Option wide_passes forces a number (tests).
    int pbits = 0;
    if (ctx->opt_wide_passes > 0) { while ((1 << pbits) < ctx->opt_wide_passes && pbits < 16) pbits++; }
    else {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess) return fail(mf_set_error("mf_count_wide_device: hipMemGetInfo failed"));
        // (every pass rolls over all the reads again -- 0.19 s per pass at 200 M reads --, so as few as fit: 2^31 occurrences = 64 GB of
        // key arrays + the sort's scratch.  The most crowded prefix class of canonical k-mers holds about twice its share: one bit of margin.)
        const uint64_t per_pass = std::min<uint64_t>(1ull << 31, std::max<uint64_t>(1ull << 20, (uint64_t)(((double)fr + (double)mf_arena_idle(ctx)) * 0.6 / 44.0)));
        while (pbits < 16 && (n_occ >> pbits) > per_pass) pbits++;
        if (pbits) pbits = std::min(16, pbits + 1);
    }
    const uint32_t n_pass = 1u << pbits;
    struct piece { mf_buf<uint64_t> hi, lo; mf_buf<uint16_t> cnt; uint64_t n = 0; };
    std::vector<std::unique_ptr<piece>> pieces;
    uint64_t nd_total = 0, occ_seen = 0;
    for (uint32_t pass = 0; pass < n_pass; pass++) {
        uint64_t n_p = n_occ;
        if (pbits) {
            {
                mf_ktimer tm(ctx, "k_wide_kmers");
                k_wide_kmers<true><<<wgrid(n_words), 256, 0, st>>>((const uint8_t *)d_bases, n_bases, vmask.p, nullptr, n_words, k, nullptr, nullptr, pbits, pass, wcnt.p);
            }
            if (mf_scan<1>(ctx, wcnt.p, woff.p, n_words, tot.p) < 0) return fail(MF_ERR);
            if (hipMemcpyAsync(&n_p, tot.p, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail(mf_set_error("mf_count_wide_device: %s", hipGetErrorString(hipGetLastError())));
        }
        occ_seen += n_p;
        if (!n_p) continue;
        if (n_p >= (1ull << 32)) return fail(mf_set_error("mf_count_wide_device: a pass of more than 2^32 k-mer occurrences (option wide_passes: more passes)"));
        mf_buf<uint64_t> h0, l0, h1, l1;
        if (h0.alloc(ctx, n_p) < 0 || l0.alloc(ctx, n_p) < 0 || h1.alloc(ctx, n_p) < 0 || l1.alloc(ctx, n_p) < 0) return fail(MF_ERR);
        {
            mf_ktimer tm(ctx, "k_wide_kmers");
            k_wide_kmers<false><<<wgrid(n_words), 256, 0, st>>>((const uint8_t *)d_bases, n_bases, vmask.p, woff.p, n_words, k, h0.p, l0.p, pbits, pass, nullptr);
        }
        {   // ascending (hi, lo): LSD -- by the low word, then (stable) by the high word's 2k - 64 bits
            mf_ktimer tm(ctx, "k_wide_sort");
            const int hb = std::max(1, 2 * k - 64);
            if (mf_sort_u64_u64(ctx, l0.p, h0.p, n_p, 64, l1.p, h1.p) < 0 || mf_sort_u64_u64(ctx, h1.p, l1.p, n_p, hb, h0.p, l0.p) < 0) return fail(MF_ERR);
        }
        h1.reset(); l1.reset();
        // run lengths
        mf_buf<uint32_t> flag; mf_buf<uint64_t> idx;
        if (flag.alloc(ctx, n_p) < 0 || idx.alloc(ctx, n_p + 1) < 0) return fail(MF_ERR);
        k_wide_flags<<<wgrid(n_p), 256, 0, st>>>(h0.p, l0.p, n_p, flag.p);
        if (mf_scan<1>(ctx, flag.p, idx.p, n_p, tot.p) < 0) return fail(MF_ERR);
        uint64_t nd = 0;
        if (hipMemcpyAsync(&nd, tot.p, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail(mf_set_error("mf_count_wide_device: %s", hipGetErrorString(hipGetLastError())));
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

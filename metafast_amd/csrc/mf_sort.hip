// mf_sort.hip -- ascending (k-mer, count) order for the file writers and mf_table_export.
// The reference dumps its hash map in iteration order (src/io/IOUtils.java:45-71: not reproducible); this
// implementation writes ascending k-mers so that files can be compared byte for byte.  Not on the hot path: rocPRIM's
// device-wide LSD radix sort (AMD's own primitives library) over the 2k significant key bits.
#include <cstring>
#include <rocprim/rocprim.hpp>
#include "mf_common.h"

int mf_sort_pairs(mf_ctx *ctx, const uint64_t *d_keys_in, const uint16_t *d_vals_in, uint64_t n, int key_bits, uint64_t *d_keys_out,
                  uint16_t *d_vals_out) {
    if (!n) return MF_OK;
    if (n >= (1ull << 32)) return mf_set_error("sort: more than 2^32 entries is not supported");
    MF_HIP(hipSetDevice(ctx->device));
    size_t tmp_bytes = 0;
    const unsigned end_bit = (unsigned)std::min(64, std::max(1, key_bits));
    MF_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_keys_in, d_keys_out, d_vals_in, d_vals_out, (size_t)n, 0u, end_bit, ctx->stream));
    mf_buf<uint8_t> tmp; MF_TRY(tmp.alloc(ctx, tmp_bytes ? tmp_bytes : 1));
    MF_HIP(rocprim::radix_sort_pairs((void *)tmp.p, tmp_bytes, d_keys_in, d_keys_out, d_vals_in, d_vals_out, (size_t)n, 0u, end_bit, ctx->stream));
    MF_HIP(hipStreamSynchronize(ctx->stream));
    return MF_OK;
}

// k-mers grouped by component (ascending component id), ascending inside each component: two stable LSD sorts
int mf_sort_kmers_by_comp(mf_ctx *ctx, const uint32_t *d_comp, const uint64_t *d_kmers, uint64_t n, int key_bits, uint32_t n_comps,
                          uint64_t *d_out) {
    if (!n) return MF_OK;
    if (n >= (1ull << 32)) return mf_set_error("sort: more than 2^32 entries is not supported");
    MF_HIP(hipSetDevice(ctx->device));
    mf_buf<uint64_t> k1; mf_buf<uint32_t> c1, c2;
    MF_TRY(k1.alloc(ctx, n)); MF_TRY(c1.alloc(ctx, n)); MF_TRY(c2.alloc(ctx, n));
    const unsigned kb = (unsigned)std::min(64, std::max(1, key_bits));
    unsigned cb = 1; while (cb < 32 && (1ull << cb) < (uint64_t)n_comps) cb++;
    size_t t1 = 0, t2 = 0;
    MF_HIP(rocprim::radix_sort_pairs(nullptr, t1, d_kmers, k1.p, d_comp, c1.p, (size_t)n, 0u, kb, ctx->stream));
    MF_HIP(rocprim::radix_sort_pairs(nullptr, t2, c1.p, c2.p, k1.p, d_out, (size_t)n, 0u, cb, ctx->stream));
    mf_buf<uint8_t> tmp; MF_TRY(tmp.alloc(ctx, std::max(t1, t2) + 1));
    MF_HIP(rocprim::radix_sort_pairs((void *)tmp.p, t1, d_kmers, k1.p, d_comp, c1.p, (size_t)n, 0u, kb, ctx->stream));
    MF_HIP(rocprim::radix_sort_pairs((void *)tmp.p, t2, c1.p, c2.p, k1.p, d_out, (size_t)n, 0u, cb, ctx->stream));
    MF_HIP(hipStreamSynchronize(ctx->stream));
    return MF_OK;
}

// (u32 key, u32 value) pairs, ascending keys of `bits` bits (stable)
int mf_sort_u32_pairs(mf_ctx *ctx, const uint32_t *d_keys_in, const uint32_t *d_vals_in, uint64_t n, int bits, uint32_t *d_keys_out,
                      uint32_t *d_vals_out) {
    if (!n) return MF_OK;
    MF_HIP(hipSetDevice(ctx->device));
    size_t tb = 0;
    const unsigned eb = (unsigned)std::min(32, std::max(1, bits));
    MF_HIP(rocprim::radix_sort_pairs(nullptr, tb, d_keys_in, d_keys_out, d_vals_in, d_vals_out, (size_t)n, 0u, eb, ctx->stream));
    mf_buf<uint8_t> tmp; MF_TRY(tmp.alloc(ctx, tb + 1));
    MF_HIP(rocprim::radix_sort_pairs((void *)tmp.p, tb, d_keys_in, d_keys_out, d_vals_in, d_vals_out, (size_t)n, 0u, eb, ctx->stream));
    return MF_OK;
}

// (u32 key, u64 value) pairs, ascending keys of `bits` bits (stable): the members of the sharded cutter grouped by component
int mf_sort_u32_u64(mf_ctx *ctx, const uint32_t *d_keys_in, const uint64_t *d_vals_in, uint64_t n, int bits, uint32_t *d_keys_out,
                    uint64_t *d_vals_out) {
    if (!n) return MF_OK;
    MF_HIP(hipSetDevice(ctx->device));
    size_t tb = 0;
    const unsigned eb = (unsigned)std::min(32, std::max(1, bits));
    MF_HIP(rocprim::radix_sort_pairs(nullptr, tb, d_keys_in, d_keys_out, d_vals_in, d_vals_out, (size_t)n, 0u, eb, ctx->stream));
    mf_buf<uint8_t> tmp; MF_TRY(tmp.alloc(ctx, tb + 1));
    MF_HIP(rocprim::radix_sort_pairs((void *)tmp.p, tb, d_keys_in, d_keys_out, d_vals_in, d_vals_out, (size_t)n, 0u, eb, ctx->stream));
    MF_HIP(hipStreamSynchronize(ctx->stream));
    return MF_OK;
}

// (u64 key, u32 value) pairs, ascending keys of `bits` bits (stable)
int mf_sort_u64_u32(mf_ctx *ctx, const uint64_t *d_keys_in, const uint32_t *d_vals_in, uint64_t n, int bits, uint64_t *d_keys_out, uint32_t *d_vals_out) {
    if (!n) return MF_OK;
    if (n >= (1ull << 32)) return mf_set_error("sort: more than 2^32 entries is not supported");
    MF_HIP(hipSetDevice(ctx->device));
    size_t tb = 0;
    const unsigned eb = (unsigned)std::min(64, std::max(1, bits));
    MF_HIP(rocprim::radix_sort_pairs(nullptr, tb, d_keys_in, d_keys_out, d_vals_in, d_vals_out, (size_t)n, 0u, eb, ctx->stream));
    mf_buf<uint8_t> tmp; MF_TRY(tmp.alloc(ctx, tb ? tb : 1));
    MF_HIP(rocprim::radix_sort_pairs((void *)tmp.p, tb, d_keys_in, d_keys_out, d_vals_in, d_vals_out, (size_t)n, 0u, eb, ctx->stream));
    MF_HIP(hipStreamSynchronize(ctx->stream));
    return MF_OK;
}

// mf_wide.h -- NO-REFERENCE EXTENSION (32 <= k <= 63): the table of 2k-bit k-mers shared by mf_wide.hip (counting) and
// mf_wgraph.hip (unitigs, components, features on 128-bit keys).  The reference stops at k = 31 (src/tools/KmersCounterMain.java:66-73).
#pragma once
#include <memory>
#include <vector>
#include "mf_common.h"

typedef unsigned __int128 mf_u128;

// A wide table is ASCENDING in (hi, lo) inside a piece and piece after piece -- a k-mer's place in the table then orders like the k-mer itself, which
// is what lets the component cutter work on 32-bit vertex ids and break ties by the smallest id -- or, fresh from the record path (mf_wskm.hip), still
// in the order of its counting units (ascending == false): mf_wtable_ensure_ascending orders it when somebody needs the order.
struct mf_wtable {
    mf_ctx *ctx = nullptr;
    int k = 0;
    uint64_t n = 0, n_occ = 0;
    uint64_t n_all = 0;           // distinct k-mers before a cut inside the counting pass (mf_count_wide_device_above); else = n
    int cut_thr = -1;             // every entry has count > cut_thr
    bool ascending = true;
    struct piece { mf_buf<uint64_t> hi, lo; mf_buf<uint16_t> cnt; uint64_t n = 0; };     // ascending (hi, lo) inside a piece, and piece after piece
    std::vector<std::unique_ptr<piece>> pieces;   // (one per pass: no second copy of a 74 GB table, and none of the 8 ms per GB a first hipMalloc of it takes)
    // lookup index over a ONE-piece table (built on first use by the graph stages, mf_wgraph.hip): open-addressed, 8-byte slots
    // = position (low 32 bits) | tag (the hash's high 32 bits), ~0 = empty, load <= 0.5
    mf_buf<unsigned long long> index; uint64_t index_mask = 0;
};
struct mf_windex_view { const unsigned long long *slots; uint64_t mask; const uint64_t *hi, *lo; const uint16_t *cnt; uint64_t n; int k; };

// the count on the record path (mf_wskm.hip): 0 = *t filled, 1 = not an input for it (the caller counts the old way), < 0 = error
int mf_count_wide_skm(mf_ctx *ctx, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n_reads, uint64_t n_bases, int k, int min_read_len, int threshold,
                      const uint32_t *vmask, uint64_t n_words, mf_wtable *t);
int mf_wide_order(mf_ctx *ctx, int k, const uint64_t *src_hi, const uint64_t *src_lo, const uint16_t *src_cnt, uint64_t nk, mf_wtable::piece *pc);
int mf_wtable_ensure_ascending(mf_wtable *t);             // (drops an index built over the other order)
int mf_wtable_flatten(mf_wtable *t);                      // all pieces into one (no-op for <= 1 piece)
int mf_wtable_ensure_index(mf_wtable *t);                 // flatten + index
// entries of (hi, lo, cnt)[n] with cnt > thr -> a new piece (order kept)
int mf_wide_compact(mf_ctx *ctx, const uint64_t *hi, const uint64_t *lo, const uint16_t *cnt, uint64_t n, int thr, mf_wtable::piece &out);

#ifdef __HIPCC__
#define MF_WIDX_EMPTY 0xFFFFFFFFFFFFFFFFull
__device__ __forceinline__ uint64_t mf_whash(uint64_t hi, uint64_t lo) { return mf_hash64(lo * 0x9E3779B97F4A7C15ull ^ hi); }
// reverse complement of a 2k-bit k-mer, 32 <= k <= 63 (A0 G1 C2 T3: complement = 3 - n)
__device__ __forceinline__ uint64_t mf_wrev2(uint64_t x) {
    const uint64_t y = __brevll(x);
    return ((y & 0x5555555555555555ull) << 1) | ((y >> 1) & 0x5555555555555555ull);
}
__device__ __forceinline__ mf_u128 mf_wrevcomp(mf_u128 x, int k) {
    const mf_u128 r = ((mf_u128)mf_wrev2((uint64_t)x) << 64) | (mf_u128)mf_wrev2((uint64_t)(x >> 64));
    return (~r) >> (128 - 2 * k);
}
// The index is hashed on the k-mer's CANONICAL INTERIOR (its middle k - 2 bases or their reverse complement, whichever is smaller), as the
// k <= 31 tables' is (mf_cidx_hkey): the four k-mers that extend a (k - 1)-mer on one side share it, and so do their reverse complements, so ONE
// probe sequence from that home slot to the next empty slot meets all four neighbours of a side (w_side_walk, mf_wgraph.hip): two probe
// sequences per vertex instead of eight.  h: home slot = low bits, tag = high 32 bits.
__device__ __forceinline__ uint64_t mf_whash_interior(mf_u128 w, mf_u128 rw) {
    const mf_u128 c = w < rw ? w : rw;
    return mf_whash((uint64_t)(c >> 64), (uint64_t)c);
}
__device__ __forceinline__ uint64_t mf_whash_key(mf_u128 x, int k) {
    const mf_u128 WM = (((mf_u128)1) << (2 * k - 4)) - 1;
    const mf_u128 rcx = mf_wrevcomp(x, k);
    return mf_whash_interior((x >> 2) & WM, (rcx >> 2) & WM);
}
// position of the k-mer (hi, lo) in the table or 0xFFFFFFFF
__device__ __forceinline__ uint32_t mf_windex_find(const mf_windex_view &ix, uint64_t hi, uint64_t lo) {
    const uint64_t h = mf_whash_key(((mf_u128)hi << 64) | (mf_u128)lo, ix.k);
    const uint32_t tag = (uint32_t)(h >> 32);
    uint64_t s = h & ix.mask;
    for (;;) {
        const unsigned long long v = ix.slots[s];
        if (v == MF_WIDX_EMPTY) return 0xFFFFFFFFu;
        if ((uint32_t)(v >> 32) == tag) {
            const uint32_t p = (uint32_t)v;
            if (ix.lo[p] == lo && ix.hi[p] == hi) return p;
        }
        s = (s + 1) & ix.mask;
    }
}
#endif

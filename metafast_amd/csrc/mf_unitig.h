// mf_unitig.h -- what the unitig builder (mf_unitig.hip) shares with the 128-bit front end (mf_wgraph.hip, NO-REFERENCE EXTENSION).
#pragma once
#include <functional>
#include "mf_common.h"
struct ut_arrays {
    const uint64_t *gk; const uint16_t *gv; uint64_t n; int k;
    const uint64_t *ghi;      // != nullptr: 2k-bit k-mers, k >= 32 (mf_wgraph.hip): gk = their low words, ghi = their high words, and the table is
                              // ASCENDING, so a k-mer's index orders like the k-mer (k_ut_ends compares indices)
    uint8_t *info; uint32_t *ridx; uint32_t *lidx;
    uint8_t *pal;             // even k only: 1 if the k-mer equals its reverse complement (else nullptr)
    uint64_t *node;           // per oriented node: successor on its path (low 32 bits, UT_NONE = none) | count << 32 |
                              // last base of the oriented k-mer << 48: everything a walk needs per hop in ONE 8-byte load
    const uint64_t *jump;     // per oriented node: where its chain leaves the node's partition + hops (k_ut_contract)
    uint32_t *starts;         // compacted list of start nodes
    unsigned int *n_starts;
};


// U2 .. U5 of mf_unitig.hip on a table of n good k-mers: `flags` launches the kernel(s) that fill A.info / A.ridx / A.lidx (/ A.pal) -- U1, the
// only step that looks k-mers up --, everything after works on node ids.  d_part_off / part_bits: the table's minimizer partitions (0: none).
int mf_ut_build(mf_ctx *ctx, const uint64_t *gk, const uint64_t *ghi, const uint16_t *gv, uint64_t n, int k, int part_bits, const uint64_t *d_part_off,
                int min_len, const std::function<int(const ut_arrays &)> &flags, mf_seqs **out);

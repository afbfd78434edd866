// mf_cc.h -- what the component cutter (mf_cc.hip) shares with the 128-bit front end (mf_wgraph.hip, NO-REFERENCE EXTENSION).
#pragma once
#include <functional>
#include "mf_common.h"
// C2 .. C5 of mf_cc.hip on n vertices: `adjacency` launches the kernel(s) that fill nbr[8 n] (C1, the only step that looks k-mers up: vertex
// ids of the 8 neighbours or 0xFFFFFFFF); d_keys: the vertices' k-mers (members, tie-break by the smallest) -- nullptr: the table is
// ascending, the vertex id stands for the k-mer (the components' d_kmers then hold ids).
int mf_cc_build(mf_ctx *ctx, uint64_t n, int k, const uint16_t *d_counts, const uint64_t *d_keys, int b1, int b2,
                const std::function<int(uint32_t *)> &adjacency, mf_comps **out);

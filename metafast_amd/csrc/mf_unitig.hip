// mf_unitig.hip -- unitigs (non-branching paths of the implicit de Bruijn graph) on the GPU.
//
// Replaces SequencesFinders.thresholdStrategy (src/algo/SequencesFinders.java:13-31), i.e. one
// AddSequencesShiftingRightTask per shard (src/algo/AddSequencesShiftingRightTask.java:40-123).  The
// reference walks every path sequentially (8 hash probes per step); here the walk is replaced by
//
//   U1 k_ut_flags   per good k-mer: getRightNucleotide / getLeftNucleotide (HashMapOperations.java:13-47) for the
//                   canonical orientation (the other orientation follows by symmetry) + the unique neighbours' indices
//   U2 k_ut_links   per ORIENTED k-mer (node = 2*index + strand): link f->g exists iff R(f) is unique and L(g) is
//                   unique; a node without an incoming link is a start (task.run :52-69)
//   U2b k_ut_contract one wave per minimizer partition follows the links that stay inside it by pointer jumping in LDS: per
//                   node ONE jump word (where its chain leaves the partition, hops)
//   U3 k_ut_walk1   one thread per START node follows the jump words to the end of its path (a load per partition crossed,
//                   nothing written per node): path length + end node.  Walks are cut after 32 / 128 / 512 / 4096 ... jumps
//                   and the unfinished ones continue from a compacted work list, so lanes stay busy although path lengths
//                   differ by orders of magnitude (pointer jumping over ALL nodes would cost O(log len) passes: it was
//                   1.06 s of a 3.4 s step)
//   U4 k_ut_ends    per path: length filter and the reference's emission rule canon(start) <= canon(end k-mer), where
//                   the end k-mer is the one BEYOND the path when the walk stopped on a left branch
//                   (processSequence :83-107) -- this is what makes a path come out 0, 1 or 2 times
//   U5 k_ut_segments / k_ut_walk2   only the EMITTED paths (a small fraction of the nodes) are walked again, node by node:
//                   a walk over the jump words cuts every path into segments of ~192 nodes, one thread per segment writes
//                   its bases at offset + distance and adds its weights to the path's sum / min / max
//
// Isolated cycles have no start node and are never emitted (same as the reference).
#include "mf_common.h"
#include "mf_nbr.h"
#include "mf_unitig.h"
#include "mf_wide.h"
#include <algorithm>
#include <numeric>

#define UT_NONE 0xFFFFFFFFu
// info byte: bits 0-2 rcode, bits 3-5 lcode (0..3 = unique nucleotide, 4 = none, 5 = several), bit 6 ror, bit 7 lor
#define UT_CODE_NONE 4u
#define UT_CODE_MANY 5u
#define UT_WALK_CHUNK 4096

__global__ void k_ut_flags(mf_index_view ix, ut_arrays A) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n) return;
    const int k = A.k;
    const uint64_t kmask = (k == 32) ? ~0ull : ((1ull << (2 * k)) - 1);
    const uint64_t x = A.gk[i];
    uint32_t rcode = UT_CODE_NONE, lcode = UT_CODE_NONE, ridx = UT_NONE, lidx = UT_NONE, ror = 0, lor = 0;
    uint32_t m_nf = 0, m_nl = 0;
    if (ix.skm_k) mf_skm_nbr_mins(x, k, &m_nf, &m_nl);
#pragma unroll
    for (uint32_t nuc = 0; nuc < 4; nuc++) {
        uint64_t y = ((x << 2) | nuc) & kmask;                 // ShortKmer.shiftRight
        uint64_t ry = mf_revcomp(y, k);
        uint64_t c = y < ry ? y : ry;
        uint32_t idx, val;
        if (mf_index_find_ph(ix, c, ix.skm_k ? mf_skm_ph_right(y, k, m_nf) : 0u, &idx, &val)) {
            if (rcode == UT_CODE_NONE) { rcode = nuc; ridx = idx; ror = (c != y); }
            else rcode = UT_CODE_MANY;
        }
    }
#pragma unroll
    for (uint32_t nuc = 0; nuc < 4; nuc++) {
        uint64_t y = (x >> 2) | ((uint64_t)nuc << (2 * k - 2));   // ShortKmer.shiftLeft
        uint64_t ry = mf_revcomp(y, k);
        uint64_t c = y < ry ? y : ry;
        uint32_t idx, val;
        if (mf_index_find_ph(ix, c, ix.skm_k ? mf_skm_ph_left(y, k, m_nl) : 0u, &idx, &val)) {
            if (lcode == UT_CODE_NONE) { lcode = nuc; lidx = idx; lor = (c != y); }
            else lcode = UT_CODE_MANY;
        }
    }
    A.info[i] = (uint8_t)(rcode | (lcode << 3) | (ror << 6) | (lor << 7));
    if (A.pal) A.pal[i] = (uint8_t)(mf_revcomp(x, k) == x);
    A.ridx[i] = ridx;
    A.lidx[i] = lidx;
}

// the same for a table with minimizer partitions: partition-local lookups (mf_nbr.h)
// KT: k as a compile-time constant (0: from A.k) -- every shift of the minimizer scan and of the neighbours becomes an immediate
template <int MODE, int KT = 0>
// (six waves per SIMD for k = 31 where the plain build's 85 registers allow five: the kernel waits for memory two thirds of its time,
// every wave more hides some of it -- 24.5 ms with four, 20.6 with five.  The Makefile checks that the limit costs NO scratch
// (check_resources.py): a build of this kernel that spilled hung on the GPU.)
#ifndef NB_NOFORCE
// (k = 25 and k = 29 need 12 bytes of scratch per lane at six waves -- the build's check refused them --: five there, as the generic build)
__attribute__((amdgpu_waves_per_eu((KT == 0 || KT == 25 || KT == 29) ? 5 : 6, 8)))
#endif
__global__ __launch_bounds__(64 * NB_WAVES) void k_ut_flags_part(mf_index_view ix, ut_arrays A, const uint64_t *__restrict__ part_off, uint32_t np) {
    __shared__ nb_lds S;
    const int k = KT ? KT : A.k;
    nb_for_each<MODE>(ix, A.gk, part_off, 0u, np, k, S, 0, 0u, [&](uint64_t i, uint64_t x, const uint32_t (&idx)[8], uint32_t flip, uint32_t, bool have) {
        if (!have) return;
        uint32_t rcode = UT_CODE_NONE, lcode = UT_CODE_NONE, ridx = UT_NONE, lidx = UT_NONE, ror = 0, lor = 0;
#pragma unroll
        for (uint32_t nuc = 0; nuc < 4; nuc++) {
            if (idx[2 * nuc] != NB_NONE) {
                if (rcode == UT_CODE_NONE) { rcode = nuc; ridx = idx[2 * nuc]; ror = (flip >> (2 * nuc)) & 1u; }
                else rcode = UT_CODE_MANY;
            }
            if (idx[2 * nuc + 1] != NB_NONE) {
                if (lcode == UT_CODE_NONE) { lcode = nuc; lidx = idx[2 * nuc + 1]; lor = (flip >> (2 * nuc + 1)) & 1u; }
                else lcode = UT_CODE_MANY;
            }
        }
        A.info[i] = (uint8_t)(rcode | (lcode << 3) | (ror << 6) | (lor << 7));
        if (A.pal) A.pal[i] = (uint8_t)(mf_revcomp(x, k) == x);
        A.ridx[i] = ridx;
        A.lidx[i] = lidx;
    });
}

// helpers on oriented nodes: node = 2*i + o, o = 1 means reverse complement of the canonical k-mer
__device__ __forceinline__ bool ut_r_unique(uint8_t info, uint32_t o) { return ((o ? (info >> 3) : info) & 7u) < 4u; }
__device__ __forceinline__ bool ut_l_unique(uint8_t info, uint32_t o) { return ((o ? info : (info >> 3)) & 7u) < 4u; }
// A palindromic k-mer (even k) is ONE oriented k-mer: only its strand-0 node exists, every reference to it uses strand 0.
__device__ __forceinline__ uint32_t ut_node(const ut_arrays &A, uint32_t idx, uint32_t strand) {
    if (A.pal && A.pal[idx]) strand = 0;
    return idx * 2u + strand;
}
__device__ __forceinline__ uint32_t ut_right_node(const ut_arrays &A, uint32_t i, uint32_t o, uint8_t info) {
    // right neighbour of x is (ridx, ror); right neighbour of rc(x) is rc(left neighbour of x) = (lidx, !lor)
    return o ? ut_node(A, A.lidx[i], ((info >> 7) & 1u) ^ 1u) : ut_node(A, A.ridx[i], (info >> 6) & 1u);
}
__device__ __forceinline__ uint32_t ut_left_node(const ut_arrays &A, uint32_t i, uint32_t o, uint8_t info) {
    return o ? ut_node(A, A.ridx[i], ((info >> 6) & 1u) ^ 1u) : ut_node(A, A.lidx[i], (info >> 7) & 1u);
}

// One thread per K-MER, both of its oriented nodes: the successor of x and the predecessor of rc(x) are the same table entry, so the two
// nodes want the same two info bytes of other k-mers -- the kernel's uncoalesced reads, and what bounds it (a thread per node read
// each of them twice: 8.5 ms; the pair of node words is one 16-byte store).
template <bool W>
__global__ __launch_bounds__(1024) void k_ut_links(ut_arrays A) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool st0 = false, st1 = false;
    if (i < A.n) {
        const uint8_t info = A.info[i];
        const bool pal_i = A.pal && A.pal[i];
        const bool ru = ut_r_unique(info, 0), lu = ut_l_unique(info, 0);
        uint32_t ri = 0, li = 0;
        uint8_t iR = 0, iL = 0;
        bool palR = false, palL = false;
        if (ru) { ri = A.ridx[i]; iR = A.info[ri]; palR = A.pal && A.pal[ri]; }
        if (lu) { li = A.lidx[i]; iL = A.info[li]; palL = A.pal && A.pal[li]; }
        const uint32_t ror = (info >> 6) & 1u, lor = (info >> 7) & 1u;
        // strand 0: right neighbour (ridx, ror), left neighbour (lidx, lor); a palindrome is referred to on strand 0 (ut_node)
        uint32_t succ0 = UT_NONE, succ1 = UT_NONE;
        if (ru) { const uint32_t sd = palR ? 0u : ror; if (ut_l_unique(iR, sd)) succ0 = ri * 2u + sd; }
        { bool has_in = false; if (lu) has_in = ut_r_unique(iL, palL ? 0u : lor); st0 = !has_in; }
        // strand 1 (not a node of its own for a palindrome): right neighbour = rc(left neighbour of x) = (lidx, !lor), left = (ridx, !ror)
        if (!pal_i) {
            if (lu) { const uint32_t sd = palL ? 0u : (lor ^ 1u); if (ut_l_unique(iL, sd)) succ1 = li * 2u + sd; }
            bool has_in = false;
            if (ru) has_in = ut_r_unique(iR, palR ? 0u : (ror ^ 1u));
            st1 = !has_in;
        }
        // (the walks used to read succ[], keys[] and counts[]: three random lines per hop)
        const uint64_t x = A.gk[i];
        const uint64_t cnt = (uint64_t)A.gv[i] << 32;
        const uint64_t w0 = (uint64_t)succ0 | cnt | ((x & 3ull) << 48);
        const int fsh = 2 * A.k - 2;                         // the first base: bits [2k - 2, 2k) (W: in the high word from k = 33 on)
        const uint32_t first = W ? (uint32_t)((fsh >= 64 ? A.ghi[i] >> (fsh - 64) : x >> fsh) & 3ull) : (uint32_t)((x >> fsh) & 3ull);
        const uint64_t w1 = (uint64_t)succ1 | cnt | ((uint64_t)(3u - first) << 48);
        *reinterpret_cast<ulonglong2 *>(&A.node[2 * i]) = make_ulonglong2(w0, w1);
    }
    // block-aggregated append of the start nodes: ONE global atomic per 1024-thread workgroup (a per-wave atomic on the
    // single cursor serialises at ~12 ns each and cost 126 ms on 7.2e8 nodes)
    __shared__ uint32_t wave_base[16];
    __shared__ uint32_t block_base;
    const unsigned long long b0 = __ballot(st0), b1 = __ballot(st1);
    const int wave = threadIdx.x >> 6;
    if (mf_lane() == 0) wave_base[wave] = (uint32_t)(__popcll(b0) + __popcll(b1));
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); w++) { uint32_t c = wave_base[w]; wave_base[w] = acc; acc += c; }
        block_base = acc ? atomicAdd(A.n_starts, acc) : 0u;
    }
    __syncthreads();
    const unsigned long long below = (1ull << mf_lane()) - 1ull;
    const uint32_t base = block_base + wave_base[wave];
    if (st0) A.starts[base + (uint32_t)__popcll(b0 & below)] = (uint32_t)(2 * i);
    if (st1) A.starts[base + (uint32_t)__popcll(b0) + (uint32_t)__popcll(b1 & below)] = (uint32_t)(2 * i + 1);
}

// U2b k_ut_contract: jump words.  A walk that reads one node word per hop misses the cache on nearly every hop (measured:
// 82 GB fetched for 7.2e8 hops, 114 B per hop), although nine links in ten stay inside the node's minimizer partition -- a
// few hundred nodes next to each other in memory.  So one wave per partition loads its nodes' successors (coalesced), follows
// the links that stay inside by pointer jumping in LDS, and writes for every node ONE word: where the chain leaves the
// partition (the successor of its last inside node) and how many hops that is; the length-only walk (U3) then makes one
// load per partition crossed, not per node.
//   jump[f] = target (low 32 bits) | hops << 32 | END << 47:  END: the chain ends in the partition, target = its last node,
//   hops to it; else target = the first node outside, hops to it.
#define UT_J_END (1ull << 47)
#define UT_J_MAXN 1024              // oriented nodes of a partition the LDS arrays take (larger partitions: one hop per word)
#define UT_J_WAVES 4
__global__ __launch_bounds__(64 * UT_J_WAVES) void k_ut_contract(const uint64_t *__restrict__ node, const uint64_t *__restrict__ part_off, uint32_t np,
                                                                 uint64_t *__restrict__ jump) {
    __shared__ uint32_t pj[UT_J_WAVES][UT_J_MAXN];          // local pointer (low 16 bits) | hops (high 16 bits)
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t gw = blockIdx.x * UT_J_WAVES + wave, nw = gridDim.x * UT_J_WAVES;
    uint32_t *P = pj[wave];
    for (uint32_t p = gw; p < np; p += nw) {
        const uint64_t lo = part_off[p], hi = part_off[p + 1];
        const uint32_t m = (uint32_t)(hi - lo) * 2u;                       // oriented nodes [2 lo, 2 hi)
        if (m == 0) continue;                                              // wave-uniform
        const uint64_t f0 = lo * 2;
        if (m > (uint32_t)UT_J_MAXN) {
            for (uint32_t j = lane; j < m; j += 64) {
                const uint32_t g = (uint32_t)node[f0 + j];
                jump[f0 + j] = g == UT_NONE ? ((uint64_t)(uint32_t)(f0 + j) | UT_J_END) : ((uint64_t)g | (1ull << 32));
            }
            continue;
        }
        for (uint32_t j = lane; j < m; j += 64) {
            const uint32_t g = (uint32_t)node[f0 + j];
            const bool inside = g != UT_NONE && (uint64_t)g >= f0 && (uint64_t)g < f0 + m;
            P[j] = inside ? ((g - (uint32_t)f0) | (1u << 16)) : j;          // the last inside node of a chain points at itself, 0 hops
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // pointer jumping; a (pointer, hops) pair is ONE word, so a read always sees a consistent pair: hops = distance to pointer
        for (int round = 0; round < 11; round++) {                         // 2^11 > UT_J_MAXN (cycles inside a partition never settle: nobody walks them)
            bool changed = false;
            for (uint32_t j = lane; j < m; j += 64) {
                const uint32_t a = P[j], t = a & 0xFFFFu;
                if (t == j) continue;
                const uint32_t b = P[t];
                if ((b & 0xFFFFu) == t) continue;                           // already at the chain's last node
                P[j] = (b & 0xFFFFu) | (((a >> 16) + (b >> 16)) << 16);
                changed = true;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (!__any(changed)) break;
        }
        for (uint32_t j = lane; j < m; j += 64) {
            const uint32_t a = P[j], e = a & 0xFFFFu, h = (a >> 16) & 0x7FFFu;
            const uint32_t g = (uint32_t)node[f0 + e];                     // (the chain's last inside node: its word was read a moment ago -- a cache hit;
                                                                            //  a copy of the successors in LDS cost the kernel three of its eight waves per SIMD)
            jump[f0 + j] = g == UT_NONE ? ((uint64_t)((uint32_t)f0 + e) | ((uint64_t)h << 32) | UT_J_END) : ((uint64_t)g | ((uint64_t)(h + 1u) << 32));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}
// (Round 3 tried k_ut_links + k_ut_contract as ONE pass per partition -- the wave that owns a partition stages its info bytes in
// LDS, computes the links from them and contracts the chains it has just computed, start nodes through a bit array: 17.5 ms
// against 8.5 + 5.4.  The links are latency-bound random reads that want the streaming kernel's full occupancy more than they
// want the LDS copy; profiles/r03o_bench_100M_fused_links_contract.json.)
// tables without minimizer partitions: one hop per word
__global__ void k_ut_jump_plain(const uint64_t *__restrict__ node, uint64_t n_nodes, uint64_t *__restrict__ jump) {
    const uint64_t f = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_nodes) return;
    const uint32_t g = (uint32_t)node[f];
    jump[f] = g == UT_NONE ? ((uint64_t)(uint32_t)f | UT_J_END) : ((uint64_t)g | (1ull << 32));
}

// ... and doubled (round 6): out[f] = in[f] followed by in[its target] -- a table without partitions (the 2k-bit tables of mf_wgraph.hip are
// ascending, neighbours far apart) has nothing for k_ut_contract to contract, every walk paid one dependent load per NODE and a path of
// 1e5 k-mers kept a lane busy for 50 ms, twice (200 M reads at k = 63: k_ut_walk1 34 ms + k_ut_double 185 + k_ut_segments 53).  Three rounds
// over all nodes (a streaming read + one random 8-byte read each) make a word span up to 8 hops.
__global__ void k_ut_jump_double(const uint64_t *__restrict__ in, uint64_t *__restrict__ out, uint64_t n_nodes) {
    const uint64_t f = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_nodes) return;
    const uint64_t a = in[f];
    if (a & UT_J_END) { out[f] = a; return; }
    const uint64_t b = in[(uint32_t)a];
    out[f] = (uint64_t)(uint32_t)b | ((((a >> 32) & 0x7FFFull) + ((b >> 32) & 0x7FFFull)) << 32) | (b & UT_J_END);
}

// walk item: a path being followed from `start` (slot = its index in starts[]), currently at `node`, `dist` hops in
struct ut_item { uint32_t node, slot, dist; };
struct ut_walk_out {
    uint32_t *end_node;       // [n_starts] last node of the path
    uint32_t *end_dist;       // [n_starts] its distance from the start
    ut_item *cont; unsigned int *n_cont;
};
template <bool FIRST>
__global__ void k_ut_walk1(ut_arrays A, const ut_item *__restrict__ items, uint32_t n_items, ut_walk_out W, int chunk) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    bool going = t < n_items;
    uint32_t f = 0, slot = 0, d = 0;
    if (going) {
        if (FIRST) { f = A.starts[t]; slot = t; d = 0; }
        else { ut_item it = items[t]; f = it.node; slot = it.slot; d = it.dist; }
        for (int step = 0; step < chunk; step++) {
            const uint64_t w = A.jump[f];
            d += (uint32_t)(w >> 32) & 0x7FFFu;
            f = (uint32_t)w;
            if (w & UT_J_END) { W.end_node[slot] = f; W.end_dist[slot] = d; going = false; break; }
        }
    }
    __shared__ uint32_t rs_scratch[18];
    const uint32_t c = mf_block_reserve(W.n_cont, going ? 1u : 0u, rs_scratch);       // unfinished walks go on in the next round (one atomic per workgroup)
    if (going) { W.cont[c].node = f; W.cont[c].slot = slot; W.cont[c].dist = d; }
}

// ---- U3b: long paths (round 5) ----
// A walk is a chain of dependent loads, ~0.7 us per jump: a unitig of 1e6 k-mers -- what the reference's own example parameters (-k 23 -b 5
// -l 1200, /root/reference/Example.md:18-21) leave of an abundant genome -- kept ONE lane busy for 108 ms, and once more for the segment cuts
// (profiles/r05v_shape_cami_example_k23_b5_l1200.json: k_ut_walk1 108 ms + k_ut_segments 92 ms of a 297 ms step).  When walks are still under
// way after the rounds of 32, 128, 512 and 4096 jumps, the jump words are doubled instead (Wyllie's list ranking) -- over the ENTRY nodes only,
// the nodes some jump lands on (one in ~6): EJ[i] = entry (or, END, the path's last node) | distance << 32 | END << 63.  The rounds run in place
// (a word is always a true statement "t is d nodes ahead"; every round at least doubles what a word spans) until every unfinished walk reads
// its end in one load, and on until every entry does: the segment cuts (U5) are then made by the entries themselves -- the first entry in
// every stretch of UT_SEG nodes of an emitted path, found with one atomicMin per entry -- instead of one more walk per path.
#define UTD_END (1ull << 63)
__global__ void k_utd_mark(const uint64_t *__restrict__ jump, uint64_t n_nodes, uint32_t *__restrict__ flag) {
    const uint64_t f = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_nodes) return;
    const uint64_t w = jump[f];
    if (!(w & UT_J_END)) flag[(uint32_t)w] = 1u;
}
__global__ void k_utd_mark_items(const ut_item *__restrict__ items, uint32_t n_items, uint32_t *__restrict__ flag) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_items) flag[items[t].node] = 1u;
}
__global__ void k_utd_build(const uint64_t *__restrict__ jump, const uint32_t *__restrict__ flag, const uint64_t *__restrict__ idx, uint64_t n_nodes,
                            uint64_t *__restrict__ ej, uint32_t *__restrict__ ent_node) {
    const uint64_t f = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_nodes || !flag[f]) return;
    const uint64_t i = idx[f], w = jump[f], hops = (w >> 32) & 0x7FFFull;
    ent_node[i] = (uint32_t)f;
    ej[i] = (w & UT_J_END) ? ((uint64_t)(uint32_t)w | (hops << 32) | UTD_END) : (idx[(uint32_t)w] | (hops << 32));
}
// out[i] = in[i] followed by in[its target] (in == out: in place)
__global__ void k_utd_double(const uint64_t *in, uint64_t *out, uint64_t n_ent) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_ent) return;
    const uint64_t a = in[i];
    if (a & UTD_END) { if (out != in) out[i] = a; return; }
    const uint64_t b = in[(uint32_t)a];
    out[i] = (uint64_t)(uint32_t)b | ((((a >> 32) & 0x7FFFFFFFull) + ((b >> 32) & 0x7FFFFFFFull)) << 32) | (b & UTD_END);
}
// the unfinished walks: the end of the path, if the word of the walk's node says it already; *n_open: the others
__global__ void k_utd_finish(const ut_item *__restrict__ items, uint32_t n_items, const uint64_t *__restrict__ idx, const uint64_t *__restrict__ ej, ut_walk_out W) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_items) return;
    const ut_item it = items[t];
    const uint64_t a = ej[idx[it.node]];
    if (a & UTD_END) { W.end_node[it.slot] = (uint32_t)a; W.end_dist[it.slot] = it.dist + (uint32_t)((a >> 32) & 0x7FFFFFFFull); }
    else atomicAdd(W.n_cont, 1u);
}

// PASS 0: equal-case arbitration (atomicMin of the start node id per start k-mer), count candidates
// PASS 1: emit: path id from a cursor, pstart / plen / pkey per path (a palindromic start gives two paths)
struct ut_paths {
    uint32_t *eqmin;      // [n]
    uint32_t *pstart;     // [n_paths] start node
    uint32_t *pend;       // [n_paths] last node (nullptr: not wanted)
    uint32_t *plen;       // [n_paths] length in nt
    uint64_t *pkey;       // [n_paths] canonical start k-mer * 2 + strand
    unsigned int *cursor;
};
template <int PASS, bool W>
__global__ void k_ut_ends(ut_arrays A, ut_paths P, const uint32_t *__restrict__ end_node, const uint32_t *__restrict__ end_dist,
                          uint32_t n_starts, int min_len) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t take = 0;                 // paths this start emits (0, 1 or 2); the cursor is advanced once per wave
    uint32_t s = 0; uint64_t len_nt = 0, stc = 0; bool eq = false, twice = false;
    if (t < n_starts) do {
    s = A.starts[t];
    const uint32_t f = end_node[t], dist = end_dist[t];
    uint32_t i = f >> 1, o = f & 1u;
    len_nt = (uint64_t)dist + (uint64_t)A.k;
    if ((int64_t)len_nt < (int64_t)min_len) break;
    uint8_t info = A.info[i];
    // end k-mer: the walk stops either because R(f) < 0 (cur = f) or because the k-mer beyond f has
    // several left neighbours (cur = that k-mer): processSequence :83-92
    // (W: two words per k-mer -- the table need not be ascending (round 6: the record path leaves it in the order of its counting units until somebody
    // asks for the order), so the k-mers themselves are compared; the path's key is the start k-mer's PLACE in the table)
    uint32_t ei = i;
    if (ut_r_unique(info, o)) ei = ut_right_node(A, i, o, info) >> 1;
    if (W) {
        const uint32_t si = s >> 1;
        const uint64_t eh = A.ghi[ei], el = A.gk[ei], sh = A.ghi[si], sl = A.gk[si];
        if (sh > eh || (sh == eh && sl > el)) break;
        eq = sh == eh && sl == el;
        stc = (uint64_t)si;
    } else {
        const uint64_t endc = A.gk[ei];
        stc = A.gk[s >> 1];
        if (stc > endc) break;
        eq = stc == endc;
    }
    // task.run :50-51 processes {kmerF, kmerF.rc()}: for a palindromic start k-mer these are the same oriented k-mer, so
    // the reference walks the identical path twice and prints it twice unless the equal-case `used` set stops the second
    twice = A.pal && A.pal[s >> 1] && !eq;
    if (PASS == 0) { if (eq) atomicMin(&P.eqmin[s >> 1], s); }
    else if (eq && P.eqmin[s >> 1] != s) break;            // "print any sequence, but only one of them" :109-118
    take = twice ? 2u : 1u;
    } while (0);
    __shared__ uint32_t rs_scratch[18];
    const uint32_t pid = mf_block_reserve(P.cursor, take, rs_scratch);        // (one atomic per workgroup: 2e5 per-wave atomics on the cursor cost 2.4 ms a pass)
    if (PASS == 1 && take) {
        P.pstart[pid] = s;
        if (P.pend) { P.pend[pid] = end_node[t]; if (twice) P.pend[pid + 1] = end_node[t]; }
        P.plen[pid] = (uint32_t)len_nt;
        P.pkey[pid] = stc * 2ull + (uint64_t)(s & 1u);
        if (twice) { P.pstart[pid + 1] = s; P.plen[pid + 1] = (uint32_t)len_nt; P.pkey[pid + 1] = stc * 2ull + 1ull; }
    }
}

struct ut_out {
    const uint64_t *off;     // [n_paths+1]
    uint8_t *bases;
    int32_t *wavg, *wmin, *wmax;
};
// second walk, emitted paths only
// A path is written by ONE thread per SEGMENT of about UT_SEG nodes: a thread per path is as slow as the longest path (1.5e4
// hops of ~0.6 us each = 8 of this step's 11 ms at 100 M reads).  The cut points come from a walk over the jump words
// (k_ut_segments: a hop per partition crossed), so a cut is always the first node of a chain inside a partition.
#define UT_SEG 192u
struct ut_seg { uint32_t node, pid, dist, stop; };        // nodes at distance [dist, stop) from the path's start; node = UT_NONE: unused entry
__global__ void k_ut_seg_bound(const uint32_t *__restrict__ plen, uint32_t np, int k, uint32_t *__restrict__ nmax) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < np) nmax[t] = (plen[t] - (uint32_t)k + 1u) / UT_SEG + 1u;      // a cut uses up at least UT_SEG nodes
}
__global__ void k_ut_segments(ut_arrays A, const uint32_t *__restrict__ pstart, uint32_t np, const uint64_t *__restrict__ segoff, ut_seg *__restrict__ seg) {
    const uint32_t pid = blockIdx.x * blockDim.x + threadIdx.x;
    if (pid >= np) return;
    uint64_t idx = segoff[pid];
    uint32_t f = pstart[pid], d = 0, last = 0;
    ut_seg cur; cur.node = f; cur.pid = pid; cur.dist = 0; cur.stop = 0xFFFFFFFFu;
    for (;;) {
        const uint64_t w = A.jump[f];
        if (w & UT_J_END) break;
        d += (uint32_t)(w >> 32) & 0x7FFFu;
        f = (uint32_t)w;
        if (d - last >= UT_SEG) {
            cur.stop = d; seg[idx++] = cur;
            cur.node = f; cur.dist = d; cur.stop = 0xFFFFFFFFu; last = d;
        }
    }
    seg[idx] = cur;
}
// ... with the doubled jump words of U3b: no walk.  pid_of_end[last node of an emitted path] = its path (the first of the two when a path is
// written twice: they follow each other and share the start node); slot b of a path = the first entry at a distance in [b, b + 1) x UT_SEG
__global__ void k_utd_path_ends(const uint32_t *__restrict__ pstart, const uint32_t *__restrict__ pend, uint32_t np, const uint64_t *__restrict__ segoff,
                                uint32_t *__restrict__ pid_of_end, unsigned long long *__restrict__ seg_min) {
    const uint32_t pid = blockIdx.x * blockDim.x + threadIdx.x;
    if (pid >= np) return;
    atomicMin(&pid_of_end[pend[pid]], pid);
    seg_min[segoff[pid]] = (unsigned long long)pstart[pid];                       // (distance 0: the start node, nobody's target)
}
__global__ void k_utd_cuts(const uint64_t *__restrict__ ej, const uint32_t *__restrict__ ent_node, uint64_t n_ent, const uint32_t *__restrict__ pid_of_end,
                           const uint32_t *__restrict__ pstart, const uint32_t *__restrict__ plen, uint32_t np, int k, const uint64_t *__restrict__ segoff,
                           unsigned long long *__restrict__ seg_min) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_ent) return;
    const uint64_t a = ej[i];
    if (!(a & UTD_END)) return;                                                  // (an isolated cycle)
    const uint32_t pid = pid_of_end[(uint32_t)a];
    if (pid == UT_NONE) return;                                                  // a path that is not written
    const uint32_t hops = plen[pid] - (uint32_t)k, dte = (uint32_t)((a >> 32) & 0x7FFFFFFFull);     // start -> last node; this entry -> last node
    const uint32_t d = hops - dte;
    const unsigned long long v = ((unsigned long long)d << 32) | ent_node[i];
    atomicMin(&seg_min[segoff[pid] + d / UT_SEG], v);
    if (pid + 1 < np && pstart[pid + 1] == pstart[pid]) atomicMin(&seg_min[segoff[pid + 1] + d / UT_SEG], v);       // the path's second copy
}
__global__ void k_utd_seg_fill(const unsigned long long *__restrict__ seg_min, uint64_t n_seg, const uint64_t *__restrict__ segoff, uint32_t np, ut_seg *__restrict__ seg) {
    const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_seg) return;
    const unsigned long long v = seg_min[q];
    if (v == ~0ull) return;                                                      // (seg[q].node stays UT_NONE: an unused entry)
    uint32_t lo = 0, hi = np;                                                    // the path of slot q: the last one with segoff <= q
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (segoff[mid] <= q) lo = mid; else hi = mid; }
    ut_seg sg; sg.node = (uint32_t)v; sg.pid = lo; sg.dist = (uint32_t)(v >> 32); sg.stop = 0xFFFFFFFFu;
    for (uint64_t r = q + 1; r < segoff[lo + 1]; r++) { const unsigned long long w = seg_min[r]; if (w != ~0ull) { sg.stop = (uint32_t)(w >> 32); break; } }
    seg[q] = sg;
}
__global__ void k_utd_max(const uint32_t *__restrict__ v, uint32_t n, unsigned int *__restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x = t < n ? v[t] : 0u;
    for (int d = 32; d; d >>= 1) { const uint32_t y = __shfl_xor(x, d); x = y > x ? y : x; }
    if ((threadIdx.x & 63u) == 0 && x) atomicMax(out, x);
}
template <bool W>
__global__ void k_ut_walk2(ut_arrays A, const ut_seg *__restrict__ seg, uint64_t n_seg, ut_out O, unsigned long long *__restrict__ wsum) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_seg) return;
    const ut_seg sg = seg[t];
    if (sg.node == UT_NONE) return;
    const int k = A.k;
    const uint64_t base = O.off[sg.pid];
    const char *NUC = "AGCT";
    uint32_t f = sg.node, d = sg.dist;
    if (d == 0) {                                              // the first k-mer in full
        if (W) {
            const mf_u128 x = ((mf_u128)A.ghi[f >> 1] << 64) | (mf_u128)A.gk[f >> 1];
            const mf_u128 y = (f & 1u) ? mf_wrevcomp(x, k) : x;
            for (int j = 0; j < k - 1; j++) O.bases[base + j] = (uint8_t)NUC[(uint32_t)(y >> (2 * (k - 1 - j))) & 3u];
        } else {
        const uint64_t x = A.gk[f >> 1];
        const uint64_t y = (f & 1u) ? mf_revcomp(x, k) : x;
        for (int j = 0; j < k - 1; j++) O.bases[base + j] = (uint8_t)NUC[(y >> (2 * (k - 1 - j))) & 3u];
        }
    }
    // one base per hop: collected in a register and written as aligned 8-byte words (a byte store per hop is one
    // partial-line write per hop); a word is stored whole only when all its eight bytes are this segment's
    uint64_t acc = 0; uint32_t nacc = 0;
    auto flush_bytes = [&](uint64_t end_pos) {                  // the nacc bytes ending just before end_pos
        for (uint32_t j = 0; j < nacc; j++) O.bases[end_pos - nacc + j] = (uint8_t)(acc >> (8 * (8 - nacc + j)));
        nacc = 0;
    };
    unsigned long long sum = 0; int32_t mn = 0x7FFFFFFF, mx = 0;
    while (d != sg.stop) {
        const uint64_t e = A.node[f];
        const uint64_t pos = base + d + (uint64_t)(k - 1);
        acc = (acc >> 8) | ((uint64_t)(uint8_t)NUC[(e >> 48) & 3u] << 56);
        nacc++;
        if ((pos & 7ull) == 7ull) {
            if (nacc == 8) { *reinterpret_cast<uint64_t *>(O.bases + pos - 7) = acc; nacc = 0; }
            else flush_bytes(pos + 1);
        }
        const int32_t v = (int32_t)((e >> 32) & 0xFFFFull);
        sum += (unsigned long long)v;
        mn = v < mn ? v : mn;
        mx = v > mx ? v : mx;
        d++;
        const uint32_t g = (uint32_t)e;
        if (g == UT_NONE) break;
        f = g;
    }
    flush_bytes(base + d + (uint64_t)(k - 1));
    atomicAdd(&wsum[sg.pid], sum);
    atomicMin(&O.wmin[sg.pid], mn);
    atomicMax(&O.wmax[sg.pid], mx);
}
__global__ void k_ut_wfinal(ut_out O, const unsigned long long *__restrict__ wsum, uint32_t np, int k) {
    const uint32_t pid = blockIdx.x * blockDim.x + threadIdx.x;
    if (pid >= np) return;
    const uint64_t len = O.off[pid + 1] - O.off[pid];
    O.wavg[pid] = (int32_t)(wsum[pid] / (len - (uint64_t)k + 1));          // (int)(seqWeight / (len - k + 1)) :120-121
}
__global__ void k_fill_u32(uint32_t *p, uint64_t n, uint32_t v) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

static inline unsigned grid_for(uint64_t n, unsigned bs = 256) { return (unsigned)((n + bs - 1) / bs); }

int mf_ut_build(mf_ctx *ctx, const uint64_t *gk, const uint64_t *ghi, const uint16_t *gv, uint64_t n, int k, int part_bits, const uint64_t *d_part_off,
                int min_len, const std::function<int(const ut_arrays &)> &flags, mf_seqs **out) {
    hipStream_t st = ctx->stream;
    const bool wide = ghi != nullptr;
    mf_seqs *S = new mf_seqs();
    S->ctx = ctx; S->k = k;
    if (n == 0) {
        void *p = nullptr;
        if (mf_alloc(ctx, 8, &p) < 0) { delete S; return MF_ERR; }
        S->d_offsets = (uint64_t *)p; S->offsets_bytes = 8;
        hipMemsetAsync(S->d_offsets, 0, 8, st);
        *out = S;
        return MF_OK;
    }
    if (n >= 0x7FFFFFFFull) { delete S; return mf_set_error("unitigs: more than 2^31 good k-mers per table is not supported"); }
    int rc = MF_OK;
    do {
        mf_buf<uint8_t> info, pal; mf_buf<uint32_t> ridx, lidx, eqmin, starts; mf_buf<uint64_t> succ;
        mf_buf<unsigned int> ctr;
        if ((rc = info.alloc(ctx, n)) < 0 || (rc = ridx.alloc(ctx, n)) < 0 || (rc = lidx.alloc(ctx, n)) < 0 ||
            (rc = succ.alloc(ctx, 2 * n)) < 0 || (rc = starts.alloc(ctx, 2 * n)) < 0 || (rc = ctr.alloc(ctx, 4)) < 0) break;
        hipMemsetAsync(ctr.p, 0, 16, st);
        ut_arrays A;
        A.gk = gk; A.ghi = ghi; A.gv = gv; A.n = n; A.k = k;
        A.info = info.p; A.ridx = ridx.p; A.lidx = lidx.p; A.node = succ.p; A.starts = starts.p; A.n_starts = &ctr.p[1];
        A.pal = nullptr; A.jump = nullptr;
        if ((k & 1) == 0) { if ((rc = pal.alloc(ctx, n)) < 0) break; A.pal = pal.p; }   // palindromes need an even k
        {
            mf_ktimer tm(ctx, "k_ut_flags");
            if ((rc = flags(A)) < 0) break;
        }
        {
            mf_ktimer tm(ctx, "k_ut_links");
            if (wide) k_ut_links<true><<<grid_for(n, 1024), 1024, 0, st>>>(A);
            else k_ut_links<false><<<grid_for(n, 1024), 1024, 0, st>>>(A);
        }
        mf_buf<uint64_t> jump;
        if ((rc = jump.alloc(ctx, 2 * n)) < 0) break;
        A.jump = jump.p;
        {
            mf_ktimer tm(ctx, "k_ut_contract");
            if (part_bits > 0 && d_part_off) {
                const uint32_t npart = 1u << part_bits;
                const unsigned grid = (unsigned)std::min<uint64_t>((npart + UT_J_WAVES - 1) / UT_J_WAVES, (uint64_t)ctx->n_cu * 32);
                k_ut_contract<<<grid, 64 * UT_J_WAVES, 0, st>>>(succ.p, d_part_off, npart, jump.p);
            } else {
                k_ut_jump_plain<<<grid_for(2 * n), 256, 0, st>>>(succ.p, 2 * n, jump.p);
                // (a target is a node's successor chain 2^r hops on -- or the path's last node; an isolated cycle just goes round)
                const int rounds_j = n >= (1u << 16) ? (int)std::max<int64_t>(0, std::min<int64_t>(ctx->opt_ut_plain_rounds, 8)) : 0;
                if (rounds_j) {
                    mf_buf<uint64_t> tmp;
                    if (tmp.alloc(ctx, 2 * n) == MF_OK) {
                        for (int r = 0; r < rounds_j; r++) { k_ut_jump_double<<<grid_for(2 * n), 256, 0, st>>>(jump.p, tmp.p, 2 * n); jump.swap(tmp); }
                        A.jump = jump.p;
                    } else (void)hipGetLastError();
                }
            }
        }
        unsigned int n_starts = 0;
        if (hipMemcpyAsync(&n_starts, &ctr.p[1], 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
            rc = mf_set_error("unitigs: links pass failed: %s", hipGetErrorString(hipGetLastError())); break;
        }
        auto chunk_of = [](int round) { return round == 0 ? 32 : round == 1 ? 128 : round == 2 ? 512 : UT_WALK_CHUNK; };
        // U3: length-only walks from the start nodes
        mf_buf<uint32_t> end_node, end_dist;
        if ((rc = end_node.alloc(ctx, n_starts)) < 0 || (rc = end_dist.alloc(ctx, n_starts)) < 0) break;
        int rounds = 0;
        mf_buf<uint64_t> d_ej; mf_buf<uint32_t> d_ent;                          // U3b (long paths): the doubled jump words of the entry nodes, entry -> node
        uint64_t d_n_ent = 0;
        int doubled = 0;
        if (n_starts) {
            mf_buf<ut_item> contA, contB;
            if ((rc = contA.alloc(ctx, n_starts)) < 0) break;
            ut_walk_out W; W.end_node = end_node.p; W.end_dist = end_dist.p; W.cont = contA.p; W.n_cont = &ctr.p[2];
            {
                mf_ktimer tm(ctx, "k_ut_walk1");
                k_ut_walk1<true><<<grid_for(n_starts), 256, 0, st>>>(A, nullptr, n_starts, W, chunk_of(0));
            }
            rounds = 1;
            for (;;) {
                unsigned int n_cont = 0;
                if (hipMemcpyAsync(&n_cont, &ctr.p[2], 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
                    rc = mf_set_error("unitigs: walk failed: %s", hipGetErrorString(hipGetLastError())); break;
                }
                if (!n_cont) break;
                if (!contB.p && (rc = contB.alloc(ctx, n_cont)) < 0) break;
                hipMemsetAsync(&ctr.p[2], 0, 4, st);
                if (rounds >= (int)ctx->opt_ut_double_after && A.jump) {
                    // U3b: still walking after the chunked rounds -- double the jump words over the entry nodes (if there is room: else the walks go on)
                    ut_item *items = (rounds & 1) ? contA.p : contB.p;
                    const uint64_t nn = 2 * n;
                    mf_buf<uint32_t> flag; mf_buf<uint64_t> etot, d_idx;
                    uint64_t n_ent = 0;
                    bool ok = flag.alloc(ctx, nn) == MF_OK && d_idx.alloc(ctx, nn + 1) == MF_OK && etot.alloc(ctx, 1) == MF_OK;
                    if (ok) {
                        mf_ktimer tm(ctx, "k_ut_double");
                        hipMemsetAsync(flag.p, 0, nn * 4, st);
                        k_utd_mark<<<grid_for(nn), 256, 0, st>>>(A.jump, nn, flag.p);
                        k_utd_mark_items<<<grid_for(n_cont), 256, 0, st>>>(items, n_cont, flag.p);
                        ok = mf_scan<1>(ctx, flag.p, d_idx.p, nn, etot.p) == MF_OK && hipMemcpyAsync(&n_ent, etot.p, 8, hipMemcpyDeviceToHost, st) == hipSuccess &&
                             hipStreamSynchronize(st) == hipSuccess;
                    }
                    ok = ok && n_ent && d_ej.alloc(ctx, n_ent) == MF_OK && d_ent.alloc(ctx, n_ent) == MF_OK;
                    if (ok) {
                        mf_ktimer tm(ctx, "k_ut_double");
                        k_utd_build<<<grid_for(nn), 256, 0, st>>>(A.jump, flag.p, d_idx.p, nn, d_ej.p, d_ent.p);
                        unsigned int open = n_cont;
                        int done_rounds = 0;
                        for (; open && done_rounds < 40; done_rounds += 3) {      // (three rounds, then a look; isolated cycles never settle: nobody walks them)
                            for (int q = 0; q < 3; q++) k_utd_double<<<grid_for(n_ent), 256, 0, st>>>(d_ej.p, d_ej.p, n_ent);
                            hipMemsetAsync(&ctr.p[2], 0, 4, st);
                            k_utd_finish<<<grid_for(n_cont), 256, 0, st>>>(items, n_cont, d_idx.p, d_ej.p, W);
                            if (hipMemcpyAsync(&open, &ctr.p[2], 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
                                rc = mf_set_error("unitigs: walk failed: %s", hipGetErrorString(hipGetLastError())); break;
                            }
                        }
                        if (rc < 0) break;
                        if (open) { rc = mf_set_error("unitigs: internal error, %u walks without an end after the doubling rounds", open); break; }
                        // ... and on until EVERY entry of a path knows its end (2^rounds >= the longest path): the segment cuts need them all
                        unsigned int longest = 0;
                        hipMemsetAsync(&ctr.p[2], 0, 4, st);
                        k_utd_max<<<grid_for(n_starts), 256, 0, st>>>(end_dist.p, n_starts, &ctr.p[2]);
                        if (hipMemcpyAsync(&longest, &ctr.p[2], 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
                            rc = mf_set_error("unitigs: walk failed: %s", hipGetErrorString(hipGetLastError())); break;
                        }
                        int need = 1; while (need < 32 && (1ull << need) < (unsigned long long)longest + 2ull) need++;
                        for (; done_rounds < need + 1; done_rounds++) k_utd_double<<<grid_for(n_ent), 256, 0, st>>>(d_ej.p, d_ej.p, n_ent);
                        hipMemsetAsync(&ctr.p[2], 0, 4, st);
                        doubled = 1; ctx->n_ut_doubled++; d_n_ent = n_ent;
                        if (ctx->opt_verbose) fprintf(stderr, "[mf] unitigs: %u walks still under way after %d rounds: jump words doubled over %llu entry nodes of %llu in %d rounds (longest path %u nodes)\n",
                                                      n_cont, rounds, (unsigned long long)n_ent, (unsigned long long)nn, done_rounds, longest + 1);
                        break;
                    }
                    (void)hipGetLastError();
                    d_ej.reset(); d_ent.reset();                                 // (no room: the walks go on jump by jump)
                    hipMemsetAsync(&ctr.p[2], 0, 4, st);
                }
                ut_item *in = (rounds & 1) ? contA.p : contB.p;
                W.cont = (rounds & 1) ? contB.p : contA.p;
                {
                    mf_ktimer tm(ctx, "k_ut_walk1");
                    k_ut_walk1<false><<<grid_for(n_cont), 256, 0, st>>>(A, in, n_cont, W, chunk_of(rounds));
                }
                rounds++;
            }
            if (rc < 0) break;
        }
        // U4
        if ((rc = eqmin.alloc(ctx, n)) < 0) break;
        k_fill_u32<<<std::min(grid_for(n), 65536u), 256, 0, st>>>(eqmin.p, n, UT_NONE);
        hipMemsetAsync(ctr.p, 0, 4, st);
        ut_paths P; P.eqmin = eqmin.p; P.pstart = nullptr; P.pend = nullptr; P.plen = nullptr; P.pkey = nullptr; P.cursor = ctr.p;
        if (n_starts) {
            mf_ktimer tm(ctx, "k_ut_ends");
            if (wide) k_ut_ends<0, true><<<grid_for(n_starts, 1024), 1024, 0, st>>>(A, P, end_node.p, end_dist.p, n_starts, min_len);
            else k_ut_ends<0, false><<<grid_for(n_starts, 1024), 1024, 0, st>>>(A, P, end_node.p, end_dist.p, n_starts, min_len);
        }
        unsigned int ncand = 0;
        if (hipMemcpyAsync(&ncand, ctr.p, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
            rc = mf_set_error("unitigs: ends pass failed"); break;
        }
        mf_buf<uint32_t> plen, pstart, pend; mf_buf<uint64_t> pkey;
        if ((rc = plen.alloc(ctx, ncand)) < 0 || (rc = pkey.alloc(ctx, ncand)) < 0 || (rc = pstart.alloc(ctx, ncand)) < 0 || (doubled && (rc = pend.alloc(ctx, ncand)) < 0)) break;
        hipMemsetAsync(ctr.p, 0, 4, st);
        P.plen = plen.p; P.pkey = pkey.p; P.pstart = pstart.p; P.pend = doubled ? pend.p : nullptr;
        if (n_starts) {
            mf_ktimer tm(ctx, "k_ut_ends");
            if (wide) k_ut_ends<1, true><<<grid_for(n_starts, 1024), 1024, 0, st>>>(A, P, end_node.p, end_dist.p, n_starts, min_len);
            else k_ut_ends<1, false><<<grid_for(n_starts, 1024), 1024, 0, st>>>(A, P, end_node.p, end_dist.p, n_starts, min_len);
        }
        unsigned int np = 0;
        if (hipMemcpyAsync(&np, ctr.p, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
            rc = mf_set_error("unitigs: ends pass failed"); break;
        }
        ridx.reset(); lidx.reset(); end_node.reset(); end_dist.reset(); eqmin.reset(); starts.reset();
        // U5: walk the emitted paths again and write them out
        mf_buf<uint64_t> off, tot; mf_buf<int32_t> wmin, wmax, wavg;
        if ((rc = off.alloc(ctx, (size_t)np + 1)) < 0 || (rc = tot.alloc(ctx, 1)) < 0 || (rc = wmin.alloc(ctx, np)) < 0 ||
            (rc = wmax.alloc(ctx, np)) < 0 || (rc = wavg.alloc(ctx, np)) < 0) break;
        if ((rc = mf_scan<1>(ctx, plen.p, off.p, (uint64_t)np, tot.p)) < 0) break;      // (multi-block: np reaches millions)
        uint64_t total = 0;
        if (hipMemcpyAsync(&total, tot.p, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
            rc = mf_set_error("unitigs: scan failed"); break;
        }
        mf_buf<uint8_t> bases;
        if ((rc = bases.alloc(ctx, total + 64)) < 0) break;     // slack for the counting kernels' 16-byte loads
        int rounds2 = 0;
        if (np) {
            ut_out O; O.off = off.p; O.bases = bases.p; O.wavg = wavg.p; O.wmin = wmin.p; O.wmax = wmax.p;
            mf_buf<uint32_t> nmax; mf_buf<uint64_t> segoff, stot; mf_buf<unsigned long long> wsum;
            if ((rc = nmax.alloc(ctx, np)) < 0 || (rc = segoff.alloc(ctx, (size_t)np + 1)) < 0 || (rc = stot.alloc(ctx, 1)) < 0 || (rc = wsum.alloc(ctx, np)) < 0) break;
            k_ut_seg_bound<<<grid_for(np), 256, 0, st>>>(plen.p, np, k, nmax.p);
            if ((rc = mf_scan<1>(ctx, nmax.p, segoff.p, (uint64_t)np, stot.p)) < 0) break;
            uint64_t n_seg = 0;
            if (hipMemcpyAsync(&n_seg, stot.p, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
                rc = mf_set_error("unitigs: scan failed"); break;
            }
            mf_buf<ut_seg> seg;
            if ((rc = seg.alloc(ctx, n_seg)) < 0) break;
            hipMemsetAsync(seg.p, 0xFF, n_seg * sizeof(ut_seg), st);
            hipMemsetAsync(wsum.p, 0, (size_t)np * 8, st);
            k_fill_u32<<<std::min(grid_for(np), 65536u), 256, 0, st>>>(reinterpret_cast<uint32_t *>(wmin.p), np, 0x7FFFFFFFu);
            hipMemsetAsync(wmax.p, 0, (size_t)np * 4, st);
            {
                mf_ktimer tm(ctx, "k_ut_segments");
                if (doubled) {
                    mf_buf<uint32_t> pid_of_end; mf_buf<unsigned long long> seg_min;
                    if ((rc = pid_of_end.alloc(ctx, 2 * n)) < 0 || (rc = seg_min.alloc(ctx, n_seg)) < 0) break;
                    hipMemsetAsync(pid_of_end.p, 0xFF, 2 * n * 4, st);
                    hipMemsetAsync(seg_min.p, 0xFF, n_seg * 8, st);
                    k_utd_path_ends<<<grid_for(np), 256, 0, st>>>(pstart.p, pend.p, np, segoff.p, pid_of_end.p, seg_min.p);
                    k_utd_cuts<<<grid_for(d_n_ent), 256, 0, st>>>(d_ej.p, d_ent.p, d_n_ent, pid_of_end.p, pstart.p, plen.p, np, k, segoff.p, seg_min.p);
                    k_utd_seg_fill<<<grid_for(n_seg), 256, 0, st>>>(seg_min.p, n_seg, segoff.p, np, seg.p);
                    if (hipStreamSynchronize(st) != hipSuccess) { rc = mf_set_error("unitigs: cuts failed: %s", hipGetErrorString(hipGetLastError())); break; }
                } else k_ut_segments<<<grid_for(np, 64), 64, 0, st>>>(A, pstart.p, np, segoff.p, seg.p);
            }
            {
                mf_ktimer tm(ctx, "k_ut_walk2");
                if (wide) k_ut_walk2<true><<<grid_for(n_seg), 256, 0, st>>>(A, seg.p, n_seg, O, wsum.p);
                else k_ut_walk2<false><<<grid_for(n_seg), 256, 0, st>>>(A, seg.p, n_seg, O, wsum.p);
                k_ut_wfinal<<<grid_for(np), 256, 0, st>>>(O, wsum.p, np, k);
            }
            rounds2 = 1;
            if (hipStreamSynchronize(st) != hipSuccess) { rc = mf_set_error("unitigs: emit failed: %s", hipGetErrorString(hipGetLastError())); break; }
        }
        jump.reset();
        if (hipStreamSynchronize(st) != hipSuccess) { rc = mf_set_error("unitigs: emit failed: %s", hipGetErrorString(hipGetLastError())); break; }
        if (ctx->opt_verbose)
            fprintf(stderr, "[mf] unitigs: good=%llu starts=%u walk_rounds=%d+%d candidates=%u paths=%u bases=%llu\n", (unsigned long long)n,
                    n_starts, rounds, rounds2, ncand, np, (unsigned long long)total);
        S->n = np; S->n_bases = total;
        S->bases_bytes = bases.bytes(); S->d_bases = bases.take();
        S->offsets_bytes = off.bytes(); S->d_offsets = off.take();
        S->w_bytes = wavg.bytes();
        S->d_avg = wavg.take(); S->d_min = wmin.take(); S->d_max = wmax.take();
        S->sk_bytes = pkey.bytes(); S->d_startkey = pkey.take();
    } while (0);
    if (rc < 0) { delete S; return rc; }
    *out = S;
    return MF_OK;
}

extern "C" int mf_build_unitigs_device(mf_ctx *ctx, mf_table *t, int freq_threshold, int min_len, mf_seqs **out) {
    mf_range rng_("mf:unitigs");
    if (!ctx || !t || !out) return mf_set_error("mf_build_unitigs_device: NULL argument");
    *out = nullptr;
    MF_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int k = t->k;

    // nodes = k-mers with value > freqThreshold (task.run :46-48)
    mf_table *g = nullptr;
    MF_TRY(mf_table_filter_or_alias(t, freq_threshold, &g));
    struct guard { mf_table *p; ~guard() { mf_table_destroy(p); } } gg{g};
    const uint64_t n = g->n;
    if (n && n < 0x7FFFFFFFull) MF_TRY(mf_table_ensure_index(g));
    return mf_ut_build(ctx, g->d_keys, nullptr, g->d_counts, n, k, g->part_bits, g->d_part_off, min_len, [&](const ut_arrays &A) -> int {
        if (ctx->opt_verbose >= 2 && g->part_bits > 0 && g->d_part_off) {       // how the keys spread over the partitions (the neighbour lookup's LDS table takes NB_CAP)
            const uint32_t npart = 1u << g->part_bits;
            std::vector<uint64_t> po((size_t)npart + 1);
            hipMemcpy(po.data(), g->d_part_off, po.size() * 8, hipMemcpyDeviceToHost);
            uint64_t over = 0, nover = 0, mx = 0, b[6] = {0, 0, 0, 0, 0, 0};
            for (uint32_t p = 0; p < npart; p++) {
                const uint64_t c = po[p + 1] - po[p];
                if (c > (uint64_t)NB_CAP) { over += c; nover++; }
                mx = std::max(mx, c);
                b[c <= 64 ? 0 : c <= 128 ? 1 : c <= 256 ? 2 : c <= 352 ? 3 : c <= 704 ? 4 : 5] += c;
            }
            fprintf(stderr, "[mf] unitigs: %u partitions, %.1f keys each, largest %llu; keys in partitions > %d: %.1f %% (%llu partitions); keys by partition size <=64/128/256/352/704/more: %.1f %.1f %.1f %.1f %.1f %.1f %%\n",
                    npart, (double)n / npart, (unsigned long long)mx, NB_CAP, 100.0 * over / n, (unsigned long long)nover, 100.0 * b[0] / n, 100.0 * b[1] / n, 100.0 * b[2] / n,
                    100.0 * b[3] / n, 100.0 * b[4] / n, 100.0 * b[5] / n);
        }
        {
            if (g->index.skm_k && g->index.part_bits && g->d_part_off && !ctx->opt_nbr_global && (n >> g->part_bits) >= 100) {      // (small partitions: the set-up per partition outweighs the local lookups)
                const uint32_t np = 1u << g->part_bits;
                const unsigned grid = (unsigned)std::min<uint64_t>((np + NB_WAVES - 1) / NB_WAVES, (uint64_t)ctx->n_cu * 64);
                const unsigned grid2 = (unsigned)std::min<uint64_t>(np, (uint64_t)ctx->n_cu * 16);      // partitions of 353 .. 1408 keys: a workgroup each
                // k as a compile-time constant for the k values users run (round 5: 31 alone was specialised, and k = 21 -- BASELINE config 4 --
                // and the CAMI example's 23, Example.md:18-21, paid 2.8 x per k-mer in the generic build); any other k: the generic one
#define UT_FLAGS_K(KK) case KK: k_ut_flags_part<1, KK><<<grid, 64 * NB_WAVES, 0, st>>>(mf_view(g->index), A, g->d_part_off, np); \
                                k_ut_flags_part<2, KK><<<grid2, 64 * NB_WAVES, 0, st>>>(mf_view(g->index), A, g->d_part_off, np); break;
                switch (k) {
                    UT_FLAGS_K(21) UT_FLAGS_K(23) UT_FLAGS_K(25) UT_FLAGS_K(27) UT_FLAGS_K(29) UT_FLAGS_K(31)
                    default:
                        k_ut_flags_part<1><<<grid, 64 * NB_WAVES, 0, st>>>(mf_view(g->index), A, g->d_part_off, np);
                        k_ut_flags_part<2><<<grid2, 64 * NB_WAVES, 0, st>>>(mf_view(g->index), A, g->d_part_off, np);
                }
#undef UT_FLAGS_K
            } else
            k_ut_flags<<<grid_for(n), 256, 0, st>>>(mf_view(g->index), A);
        }
        return MF_OK;
    }, out);
}

extern "C" void mf_seqs_destroy(mf_seqs *s) {
    if (!s) return;
    if (s->d_bases) mf_release(s->ctx, s->d_bases, s->bases_bytes);
    if (s->d_offsets) mf_release(s->ctx, s->d_offsets, s->offsets_bytes);
    if (s->d_avg) mf_release(s->ctx, s->d_avg, s->w_bytes);
    if (s->d_min) mf_release(s->ctx, s->d_min, s->w_bytes);
    if (s->d_max) mf_release(s->ctx, s->d_max, s->w_bytes);
    if (s->d_startkey) mf_release(s->ctx, s->d_startkey, s->sk_bytes);
    delete s;
}
extern "C" int mf_seqs_stats(const mf_seqs *s, uint64_t *n_seqs, uint64_t *total_len) {
    if (!s) return mf_set_error("seqs is NULL");
    if (n_seqs) *n_seqs = s->n;
    if (total_len) *total_len = s->n_bases;
    return MF_OK;
}
extern "C" int mf_seqs_device_view(const mf_seqs *s, const void **d_bases, const void **d_offsets, const void **d_avg,
                                   const void **d_min, const void **d_max, uint64_t *n_seqs, uint64_t *n_bases) {
    if (!s) return mf_set_error("seqs is NULL");
    if (d_bases) *d_bases = s->d_bases;
    if (d_offsets) *d_offsets = s->d_offsets;
    if (d_avg) *d_avg = s->d_avg;
    if (d_min) *d_min = s->d_min;
    if (d_max) *d_max = s->d_max;
    if (n_seqs) *n_seqs = s->n;
    if (n_bases) *n_bases = s->n_bases;
    return MF_OK;
}

// host copy in deterministic order (canonical start k-mer, strand)
// the sequences in their output order (ascending oriented start k-mer), made in HBM: sort of (start k-mer, index), the lengths in that
// order, their prefix sums, a gather of the bases (a wave per sequence) -- the host sorted and gathered 3e5 sequences in 0.1 s
int mf_sort_u64_u32(mf_ctx *ctx, const uint64_t *d_keys_in, const uint32_t *d_vals_in, uint64_t n, int bits, uint64_t *d_keys_out, uint32_t *d_vals_out);
__global__ void k_seq_iota(uint32_t *__restrict__ v, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (uint32_t)i;
}
__global__ void k_seq_ordered_meta(const uint32_t *__restrict__ order, const uint64_t *__restrict__ off, const int32_t *__restrict__ a, const int32_t *__restrict__ mn,
                                   const int32_t *__restrict__ mx, uint64_t n, uint32_t *__restrict__ len, int32_t *__restrict__ oa, int32_t *__restrict__ omn,
                                   int32_t *__restrict__ omx) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = order[i];
    len[i] = (uint32_t)(off[j + 1] - off[j]); oa[i] = a[j]; omn[i] = mn[j]; omx[i] = mx[j];
}
__global__ __launch_bounds__(256) void k_seq_gather(const uint32_t *__restrict__ order, const uint64_t *__restrict__ off, const uint64_t *__restrict__ noff,
                                                    const uint8_t *__restrict__ bases, uint64_t n, uint8_t *__restrict__ out) {
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= n) return;
    const uint32_t j = order[i];
    const uint64_t src = off[j], len = off[j + 1] - src, dst = noff[i];
    for (uint64_t t = threadIdx.x & 63u; t < len; t += 64) out[dst + t] = bases[src + t];
}
int mf_seqs_to_host(const mf_seqs *s, std::vector<uint8_t> &bases, std::vector<uint64_t> &off, std::vector<int32_t> &avg,
                    std::vector<int32_t> &mn, std::vector<int32_t> &mx) {
    mf_ctx *ctx = s->ctx;
    MF_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint64_t n = s->n;
    bases.resize(s->n_bases); off.assign(n + 1, 0); avg.resize(n); mn.resize(n); mx.resize(n);
    if (!n) return MF_OK;
    if (n >= (1ull << 32)) return mf_set_error("sequences: more than 2^32 is not supported");
    mf_buf<uint32_t> idx, order, len; mf_buf<uint64_t> skey, noff, tot; mf_buf<int32_t> oa, omn, omx; mf_buf<uint8_t> ob;
    MF_TRY(idx.alloc(ctx, n)); MF_TRY(order.alloc(ctx, n)); MF_TRY(len.alloc(ctx, n)); MF_TRY(skey.alloc(ctx, n)); MF_TRY(noff.alloc(ctx, n + 1)); MF_TRY(tot.alloc(ctx, 1));
    MF_TRY(oa.alloc(ctx, n)); MF_TRY(omn.alloc(ctx, n)); MF_TRY(omx.alloc(ctx, n)); MF_TRY(ob.alloc(ctx, s->n_bases + 1));
    k_seq_iota<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(idx.p, n);
    MF_TRY(mf_sort_u64_u32(ctx, s->d_startkey, idx.p, n, 64, skey.p, order.p));
    k_seq_ordered_meta<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(order.p, s->d_offsets, s->d_avg, s->d_min, s->d_max, n, len.p, oa.p, omn.p, omx.p);
    MF_TRY(mf_scan<1>(ctx, len.p, noff.p, n, tot.p));
    k_seq_gather<<<(unsigned)((n * 64 + 255) / 256), 256, 0, st>>>(order.p, s->d_offsets, noff.p, s->d_bases, n, ob.p);
    if (s->n_bases) MF_HIP(hipMemcpyAsync(bases.data(), ob.p, s->n_bases, hipMemcpyDeviceToHost, st));
    MF_HIP(hipMemcpyAsync(off.data(), noff.p, (n + 1) * 8, hipMemcpyDeviceToHost, st));
    MF_HIP(hipMemcpyAsync(avg.data(), oa.p, n * 4, hipMemcpyDeviceToHost, st));
    MF_HIP(hipMemcpyAsync(mn.data(), omn.p, n * 4, hipMemcpyDeviceToHost, st));
    MF_HIP(hipMemcpyAsync(mx.data(), omx.p, n * 4, hipMemcpyDeviceToHost, st));
    MF_HIP(hipStreamSynchronize(st));
    return MF_OK;
}
extern "C" int mf_seqs_export(const mf_seqs *s, uint8_t *bases, uint64_t *offsets, int32_t *avg, int32_t *mn, int32_t *mx) {
    if (!s || !offsets) return mf_set_error("mf_seqs_export: NULL argument");
    std::vector<uint8_t> b; std::vector<uint64_t> o; std::vector<int32_t> a, lo, hi;
    MF_TRY(mf_seqs_to_host(s, b, o, a, lo, hi));
    if (bases && !b.empty()) memcpy(bases, b.data(), b.size());
    memcpy(offsets, o.data(), o.size() * 8);
    if (avg && s->n) memcpy(avg, a.data(), s->n * 4);
    if (mn && s->n) memcpy(mn, lo.data(), s->n * 4);
    if (mx && s->n) memcpy(mx, hi.data(), s->n * 4);
    return MF_OK;
}

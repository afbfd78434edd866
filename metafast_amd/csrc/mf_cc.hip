// mf_cc.hip -- connected components of the implicit de Bruijn neighbour graph + features + Bray-Curtis.
//
// Replaces ComponentsBuilder.splitStrategy (src/algo/ComponentsBuilder.java:24-32, run :58-153,
// findAllComponents :198-213, bfs :220-270, Task.run :162-179).  The reference runs one sequential BFS
// over the whole map; here:
//
//   C1 k_cc_adjacency  per vertex (cutter k-mer): the 8 canonical neighbours of KmerOperations.possibleNeighbours
//                      (src/algo/KmerOperations.java:9-26) looked up in the HBM index -> 8 vertex ids
//   C2 k_cc_hook       lock-free union-find: hook the larger root under the smaller with atomicCAS,
//                      path halving on the way (so the root of a component is its smallest vertex id)
//   C3 k_cc_flatten    parent[v] = root(v)
//   C4 k_cc_stats      size / weight per root by atomics
//   C5 k_cc_classify   size < b1 dropped | b1 <= size <= b2 kept | size > b2: vertices with value >= thr+1 stay
//                      alive for the next round with thr+1 (bfs :245-262 builds nextHM the same way)
//
// A round at threshold t is ONE global pass over all still-alive vertices: different oversize components
// are disconnected, so splitting them together is the same as the reference's per-component Tasks.
#include "mf_common.h"
#include "mf_nbr.h"
#include "mf_cc.h"
#include <algorithm>
#include <numeric>
#include <memory>

#define CC_NONE 0xFFFFFFFFu

__global__ void k_cc_adjacency(mf_index_view ix, const uint64_t *__restrict__ keys, uint64_t n,
                               int k, uint32_t *__restrict__ nbr) {
    uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const uint64_t kmask = (1ull << (2 * k)) - 1;
    const uint64_t x = keys[v];
    uint32_t out[8];
    uint32_t m_nf = 0, m_nl = 0;
    if (ix.skm_k) mf_skm_nbr_mins(x, k, &m_nf, &m_nl);
#pragma unroll
    for (uint32_t nuc = 0; nuc < 4; nuc++) {
        uint32_t idx, val;
        uint64_t y = ((x << 2) | nuc) & kmask;
        out[2 * nuc] = mf_index_find_ph(ix, mf_canon(y, k), ix.skm_k ? mf_skm_ph_right(y, k, m_nf) : 0u, &idx, &val) ? idx : CC_NONE;
        y = (x >> 2) | ((uint64_t)nuc << (2 * k - 2));
        out[2 * nuc + 1] = mf_index_find_ph(ix, mf_canon(y, k), ix.skm_k ? mf_skm_ph_left(y, k, m_nl) : 0u, &idx, &val) ? idx : CC_NONE;
    }
    uint4 *o = reinterpret_cast<uint4 *>(nbr + v * 8);
    o[0] = make_uint4(out[0], out[1], out[2], out[3]);
    o[1] = make_uint4(out[4], out[5], out[6], out[7]);
}

// the same for a table with minimizer partitions: partition-local lookups (mf_nbr.h)
template <int MODE, int KT = 0>
__attribute__((amdgpu_waves_per_eu(KT ? 6 : 5, 8)))         // (as k_ut_flags_part, mf_unitig.hip: no scratch, checked by the build)
__global__ __launch_bounds__(64 * NB_WAVES) void k_cc_adjacency_part(mf_index_view ix, const uint64_t *__restrict__ keys, const uint64_t *__restrict__ part_off,
                                                                   uint32_t np, int k_rt, uint32_t *__restrict__ nbr) {
    __shared__ nb_lds S;
    const int k = KT ? KT : k_rt;
    nb_for_each<MODE>(ix, keys, part_off, 0u, np, k, S, 0, 0u, [&](uint64_t v, uint64_t, const uint32_t (&idx)[8], uint32_t, uint32_t, bool have) {
        if (!have) return;
        uint4 *o = reinterpret_cast<uint4 *>(nbr + v * 8);
        o[0] = make_uint4(idx[0], idx[1], idx[2], idx[3]);
        o[1] = make_uint4(idx[4], idx[5], idx[6], idx[7]);
    });
}

__device__ __forceinline__ uint32_t cc_find(uint32_t *parent, uint32_t x) {
    // path halving; plain (relaxed) loads/stores: any stale value is still an ancestor
    for (;;) {
        uint32_t p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p == x) return x;
        uint32_t gp = __hip_atomic_load(&parent[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gp != p) __hip_atomic_store(&parent[x], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x = gp;
    }
}
// The table is ordered by minimizer partition, so most edges join vertices a few hundred places apart: a workgroup first
// solves its TILE of consecutive vertices in LDS (the edges with both ends inside: about five in six), writes every
// vertex's tile root to parent[], and only the edges that leave the tile go through the union-find in HBM (k_cc_hook).
#define CC_TILE 1024
#define CC_LEDGE 2048               // edges that leave a tile, collected in LDS (more: the tile is left to the per-vertex kernel, ecount[1])
// edges != nullptr: the edges from this tile's vertices to SMALLER tiles -- one in six, the ones the tile cannot join in LDS -- go to a
// list (edges[0 .. ecount[0]), block-aggregated), for k_cc_hook_edges: a thread per listed edge.  (The per-vertex kernel that used to
// find them again read the whole adjacency a second time and ran its union-finds with one lane in six busy: 5.5 ms for a sixth of the
// edges where this kernel takes 1.4 for the rest.)
__global__ __launch_bounds__(256) void k_cc_hook_tile(const uint32_t *__restrict__ nbr, const uint8_t *__restrict__ alive, uint32_t *__restrict__ parent,
                                                      uint32_t *__restrict__ csize, unsigned long long *__restrict__ cweight, uint64_t n,
                                                      uint2 *__restrict__ edges, unsigned long long *__restrict__ ecount, uint64_t ecap) {
    __shared__ uint32_t lp[CC_TILE];
    __shared__ uint8_t la[CC_TILE];
    __shared__ uint2 le[CC_LEDGE];
    __shared__ uint32_t ln;
    __shared__ unsigned long long lbase;
    const uint64_t base = (uint64_t)blockIdx.x * CC_TILE;
    if (threadIdx.x == 0) ln = 0;
    for (uint32_t i = threadIdx.x; i < CC_TILE; i += blockDim.x) {
        lp[i] = i;
        const uint8_t al = base + i < n ? alive[base + i] : 0;
        la[i] = al;
        if (al) { csize[base + i] = 0; cweight[base + i] = 0; }          // (dead vertices are never read again: later levels skip 16 B each)
    }
    __syncthreads();
    auto find_l = [&](uint32_t x) -> uint32_t {
        for (;;) {
            const uint32_t p = *(volatile uint32_t *)&lp[x];
            if (p == x) return x;
            const uint32_t gp = *(volatile uint32_t *)&lp[p];
            if (gp != p) *(volatile uint32_t *)&lp[x] = gp;
            x = gp;
        }
    };
    for (uint32_t i = threadIdx.x; i < CC_TILE; i += blockDim.x) {
        if (!la[i]) continue;
        const uint64_t v = base + i;
        const uint4 *q = reinterpret_cast<const uint4 *>(nbr + v * 8);
        const uint4 a = q[0], b = q[1];
        const uint32_t nb[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t u = nb[j];
            if (u == CC_NONE || u >= (uint32_t)v) continue;
            if (u < (uint32_t)base) {
                if (edges) { const uint32_t e = atomicAdd(&ln, 1u); if (e < (uint32_t)CC_LEDGE) le[e] = make_uint2((uint32_t)v, u); }
                continue;
            }
            if (!la[u - (uint32_t)base]) continue;
            uint32_t ra = i, rb = u - (uint32_t)base;
            for (;;) {
                ra = find_l(ra); rb = find_l(rb);
                if (ra == rb) break;
                if (ra < rb) { const uint32_t t = ra; ra = rb; rb = t; }
                if (atomicCAS(&lp[ra], ra, rb) == ra) break;
            }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < CC_TILE; i += blockDim.x) {
        if (base + i >= n) break;
        if (!la[i]) continue;
        uint32_t r = i;
        for (;;) { const uint32_t p = lp[r]; if (p == r) break; r = p; }
        parent[base + i] = (uint32_t)base + r;
    }
    if (edges) {
        const uint32_t cnt = ln < (uint32_t)CC_LEDGE ? ln : (uint32_t)CC_LEDGE;         // (ln: final since the barrier above)
        if (threadIdx.x == 0) {
            lbase = cnt ? atomicAdd(ecount, (unsigned long long)cnt) : 0ull;
            if (ln > (uint32_t)CC_LEDGE || (cnt && lbase + cnt > ecap)) atomicAdd(ecount + 1, 1ull);      // the list is incomplete: k_cc_hook runs
        }
        __syncthreads();
        for (uint32_t e = threadIdx.x; e < cnt; e += blockDim.x) if (lbase + e < ecap) edges[lbase + e] = le[e];
    }
}
// a thread per listed edge (v, u), u in a smaller tile than v: alive[v] is known, alive[u] is looked at here
__global__ void k_cc_hook_edges(const uint2 *__restrict__ edges, const unsigned long long *__restrict__ ecount, uint64_t ecap, const uint8_t *__restrict__ alive, uint32_t *parent) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long ne = ecount[0] < ecap ? ecount[0] : ecap;
    if (e >= ne) return;
    const uint2 ed = edges[e];
    if (!alive[ed.y]) return;
    uint32_t ra = ed.x, rb = ed.y;
    for (;;) {
        ra = cc_find(parent, ra); rb = cc_find(parent, rb);
        if (ra == rb) break;
        if (ra < rb) { const uint32_t t = ra; ra = rb; rb = t; }
        if (atomicCAS(&parent[ra], ra, rb) == ra) break;            // hook larger root under smaller
    }
}
// the edges that leave the tile of their larger end.  list != nullptr (a SPARSE level, below): the vertices list[0 .. n) only, ALL their
// edges to smaller alive vertices (no tile pass ran)
// unless: the counters of an edge list (k_cc_hook_tile): the kernel has nothing to do when the list is complete (unless[1] == 0)
__global__ void k_cc_hook(const uint32_t *__restrict__ nbr, const uint8_t *__restrict__ alive, uint32_t *parent, uint64_t n, const uint32_t *__restrict__ list,
                          const unsigned long long *__restrict__ unless) {
    uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    if (unless && unless[1] == 0ull) return;
    if (list) v = list[v];
    else if (!alive[v]) return;
    const uint4 *q = reinterpret_cast<const uint4 *>(nbr + v * 8);
    uint4 a = q[0], b = q[1];
    uint32_t nb[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    const uint32_t tile0 = list ? (uint32_t)v : (uint32_t)v & ~(uint32_t)(CC_TILE - 1);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        uint32_t u = nb[j];
        if (u == CC_NONE || u >= tile0 || !alive[u]) continue;          // each undirected edge once (adjacency is symmetric)
        uint32_t ra = (uint32_t)v, rb = u;
        for (;;) {
            ra = cc_find(parent, ra); rb = cc_find(parent, rb);
            if (ra == rb) break;
            if (ra < rb) { uint32_t t = ra; ra = rb; rb = t; }
            if (atomicCAS(&parent[ra], ra, rb) == ra) break;            // hook larger root under smaller
        }
    }
}
// The forest is final after k_cc_hook.  root[] is a SEPARATE array and the find below does not write: storing the
// root into parent[] here would race with other threads' path-halving stores (which may put back a non-root ancestor).
// A workgroup takes a tile of CC_TILE consecutive vertices.  After k_cc_hook_tile most of them still point at one of a
// few tile roots, so the tile is first grouped by parent value in an LDS hash table (size and weight summed with LDS
// atomics); the walk to the root and the atomics on the root's two global counters then happen once per distinct parent
// value instead of once per vertex (the giant component's counters are the hottest addresses of the whole step).
// every alive vertex points at its root (a thread per vertex, ancestors read through the cache): k_cc_flatten_stats then finds a tile's roots in one step
// each instead of walking up from every distinct parent value with the 12 waves per CU its LDS tables leave it
__global__ void k_cc_compress(const uint8_t *__restrict__ alive, uint32_t *__restrict__ parent, uint64_t n) {
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n || !alive[v]) return;
    uint32_t r = parent[v];
    if (r == (uint32_t)v) return;
    for (;;) { const uint32_t p = parent[r]; if (p == r) break; r = p; }
    parent[v] = r;
}
__global__ __launch_bounds__(256) void k_cc_flatten_stats(const uint8_t *__restrict__ alive, const uint32_t *__restrict__ parent,
                                                          uint32_t *__restrict__ root, const uint16_t *__restrict__ vals,
                                                          uint32_t *__restrict__ csize, unsigned long long *__restrict__ cweight, uint64_t n) {
    constexpr uint32_t HS = 2 * CC_TILE;
    __shared__ uint32_t hk[HS], hr[HS], hs[HS], hs2[HS], hw[HS], hw2[HS];      // (weights of one tile: <= 1024 x 32767 < 2^32; 48 KiB: three workgroups per CU)
    const uint64_t base = (uint64_t)blockIdx.x * CC_TILE;
    for (uint32_t i = threadIdx.x; i < HS; i += blockDim.x) { hk[i] = CC_NONE; hs[i] = 0; hw[i] = 0; hs2[i] = 0; hw2[i] = 0; }
    __syncthreads();
    uint32_t slot[CC_TILE / 256];
#pragma unroll
    for (int j = 0; j < CC_TILE / 256; j++) {
        const uint64_t v = base + (uint32_t)j * 256u + threadIdx.x;
        slot[j] = CC_NONE;
        if (v >= n || !alive[v]) continue;
        const uint32_t key = parent[v];
        uint32_t s = (key * 0x9E3779B1u) >> (32 - 11);
        for (;;) {
            const uint32_t old = atomicCAS(&hk[s], CC_NONE, key);
            if (old == CC_NONE || old == key) break;
            s = (s + 1) & (HS - 1);
        }
        slot[j] = s;
        atomicAdd(&hs[s], 1u);
        atomicAdd(&hw[s], (uint32_t)vals[v]);
    }
    __syncthreads();
    // (distinct PARENT values -> their roots.  The finds of k_cc_hook leave the vertices of a tile pointing at many different ancestors
    // of ONE root -- a hundred or so per tile for the giant component -- and one pair of atomics per parent value put 2.7e7 of them on
    // the giant root's two counters at the union of 8 x 200 M reads' unitigs: 544 ms of serialised same-address atomics, 88 % of that
    // shape's components stage (round 3 read it as "a hundred threshold levels"; there are five).  The sums are therefore added up
    // once more in LDS, by ROOT, and a tile touches a root's counters once.)
    uint32_t myroot[HS / 256];
#pragma unroll
    for (int j = 0; j < (int)(HS / 256); j++) {
        const uint32_t i = (uint32_t)j * 256u + threadIdx.x;
        uint32_t r = hk[i];
        myroot[j] = CC_NONE;
        if (r == CC_NONE) continue;
        for (;;) { const uint32_t p = parent[r]; if (p == r) break; r = p; }
        hr[i] = r; myroot[j] = r;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < HS; i += blockDim.x) hk[i] = CC_NONE;      // (the parent values have done their job: hk becomes the table of roots)
    __syncthreads();
#pragma unroll
    for (int j = 0; j < (int)(HS / 256); j++) {
        if (myroot[j] == CC_NONE) continue;
        const uint32_t i = (uint32_t)j * 256u + threadIdx.x, key = myroot[j];
        uint32_t s = (key * 0x9E3779B1u) >> (32 - 11);
        for (;;) {
            const uint32_t old = atomicCAS(&hk[s], CC_NONE, key);
            if (old == CC_NONE || old == key) break;
            s = (s + 1) & (HS - 1);
        }
        atomicAdd(&hs2[s], hs[i]);
        atomicAdd(&hw2[s], hw[i]);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < HS; i += blockDim.x) {
        const uint32_t r = hk[i];
        if (r == CC_NONE) continue;
        atomicAdd(&csize[r], hs2[i]);
        atomicAdd(&cweight[r], (unsigned long long)hw2[i]);
    }
#pragma unroll
    for (int j = 0; j < CC_TILE / 256; j++) {
        const uint64_t v = base + (uint32_t)j * 256u + threadIdx.x;
        if (slot[j] != CC_NONE) root[v] = hr[slot[j]];
    }
}
// ---- SPARSE levels (round 4).  After the first threshold level or two only the vertices of the oversize components that reach the next
// threshold are left -- 29 %, 1.4 %, 0.08 % of the union of 8 x 50 M reads' unitigs at levels 3, 4, 5; the union of 8 x 200 M reads goes
// through about a hundred levels, most of them on a sliver of the graph -- while the dense kernels above visit all n vertices five times
// per level to find out that they are dead (961 ms of that shape's 4.3 s step).  k_cc_members therefore lists the survivors (a dense level as far as
// a list of n / 3 has room), and once fewer than a third are left a level runs on the list: set-up, hook (every edge to a smaller alive vertex), size / weight per root
// with the lanes of a wave that share a root adding up first, classification, members -- all launched over the list.
__global__ void k_ccs_init(const uint32_t *__restrict__ list, uint64_t m, uint32_t *__restrict__ parent, uint32_t *__restrict__ csize, unsigned long long *__restrict__ cweight) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const uint32_t v = list[i];
    parent[v] = v; csize[v] = 0; cweight[v] = 0;
}
__global__ void k_ccs_stats(const uint32_t *__restrict__ list, uint64_t m, const uint32_t *__restrict__ parent, uint32_t *__restrict__ root, const uint16_t *__restrict__ vals,
                            uint32_t *__restrict__ csize, unsigned long long *__restrict__ cweight) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool have = i < m;
    uint32_t v = 0, r = 0; unsigned long long w = 0;
    if (have) {
        v = list[i]; r = v;
        for (;;) { const uint32_t p = parent[r]; if (p == r) break; r = p; }        // (read-only: the forest is final)
        root[v] = r; w = vals[v];
    }
    // the lanes of a wave mostly share a root (the list is in table order): one pair of atomics per distinct root and wave
    unsigned long long todo = __ballot(have);
    for (int round = 0; round < 6 && todo; round++) {                   // wave-uniform
        const int lead = __ffsll((long long)todo) - 1;
        const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)r, lead);
        const bool in = have && r == r0;
        const unsigned long long grp = __ballot(in);
        unsigned long long sw = in ? w : 0ull;
        for (int d = 32; d >= 1; d >>= 1) sw += __shfl_xor(sw, d, 64);
        if ((int)mf_lane() == lead) { atomicAdd(&csize[r0], (uint32_t)__popcll(grp)); atomicAdd(&cweight[r0], sw); }
        if (in) have = false;
        todo &= ~grp;
    }
    if (have) { atomicAdd(&csize[r], 1u); atomicAdd(&cweight[r], w); }
}
// per root: classify; kept roots get a slot in the kept list (SoA: root / size / weight / smallest k-mer)
struct cc_kept_arrays { uint32_t *root; uint32_t *size; unsigned long long *weight; unsigned long long *minkey; };
__global__ void k_cc_classify(const uint8_t *__restrict__ alive, const uint32_t *__restrict__ parent,
                              const uint32_t *__restrict__ csize, const unsigned long long *__restrict__ cweight, uint64_t n,
                              uint32_t b1, uint32_t b2, uint32_t *__restrict__ keptslot, cc_kept_arrays K,
                              unsigned int *__restrict__ counters /* [0]=kept comps [1]=kept kmers [2]=big comps */, const uint32_t *__restrict__ list) {
    uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    if (list) v = list[v];
    else if (!alive[v]) return;
    if (parent[v] != (uint32_t)v) return;
    uint32_t s = csize[v];
    if (s < b1) return;
    if (s <= b2) {
        uint32_t slot = atomicAdd(&counters[0], 1u);
        atomicAdd(&counters[1], s);
        keptslot[v] = slot;
        K.root[slot] = (uint32_t)v; K.size[slot] = s; K.weight[slot] = cweight[v]; K.minkey[slot] = ~0ull;
    } else atomicAdd(&counters[2], 1u);
}
// per vertex: members of kept components are appended to the member list (k-mer + temporary component id);
// vertices of oversize components with value >= thr+1 stay alive, everything else dies
__global__ void k_cc_members(uint8_t *__restrict__ alive, const uint32_t *__restrict__ parent, const uint32_t *__restrict__ csize,
                             const uint16_t *__restrict__ vals, const uint64_t *__restrict__ keys, uint64_t n, uint32_t b1,
                             uint32_t b2, uint32_t next_thr, const uint32_t *__restrict__ keptslot,
                             const uint64_t *__restrict__ slot_off, uint32_t *__restrict__ slot_fill, uint32_t comp_base,
                             unsigned long long *__restrict__ minkey, uint64_t *__restrict__ members,
                             uint32_t *__restrict__ member_comp, unsigned int *__restrict__ n_alive, const uint32_t *__restrict__ list,
                             uint32_t *__restrict__ next_list, uint32_t next_cap) {
    // list != nullptr: the vertices list[0 .. n); next_list (may be nullptr): the vertices that stay alive are appended as far as it has
    // room (next_cap; the caller uses the list only if all of them fitted), *n_alive = how many
    __shared__ uint32_t scratch[18];
    uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool put = false, stays = false;
    uint32_t slot = 0; uint64_t key = 0;
    const bool mine = v < n;
    if (mine && list) v = list[v];
    if (mine && alive[v]) {
        const uint32_t r = parent[v];
        const uint32_t s = csize[r];
        if (s > b2) { if ((uint32_t)vals[v] < next_thr) alive[v] = 0; else stays = true; }
        else {
            alive[v] = 0;
            if (s >= b1) { put = true; slot = keptslot[r]; key = keys ? keys[v] : v; }      // (keys == nullptr: an ascending table, the vertex id stands for the k-mer)
        }
    }
    {
        const uint32_t at = mf_block_reserve(n_alive, stays ? 1u : 0u, scratch);       // (one atomic per workgroup)
        if (stays && next_list && at < next_cap) next_list[at] = (uint32_t)v;
    }
    // positions in the member lists: one cursor atomic per distinct component and wave (see k_cc_flatten_stats)
    unsigned long long todo = __ballot(put);
    const uint64_t lt_mask = (1ull << mf_lane()) - 1ull;
    for (int round = 0; round < 4 && todo; round++) {               // wave-uniform
        const int lead = __ffsll((long long)todo) - 1;
        const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)slot, lead);
        const bool in = put && slot == s0;
        const unsigned long long grp = __ballot(in);
        uint64_t mk = in ? key : ~0ull;
        for (int d = 32; d >= 1; d >>= 1) { const uint64_t o = __shfl_xor(mk, d, 64); mk = o < mk ? o : mk; }
        uint32_t base = 0;
        if (mf_lane() == lead) { base = atomicAdd(&slot_fill[s0], (uint32_t)__popcll(grp)); atomicMin(&minkey[s0], (unsigned long long)mk); }
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, lead);
        if (in) {
            const uint64_t at = slot_off[s0] + base + (uint32_t)__popcll(grp & lt_mask);
            members[at] = key;
            member_comp[at] = comp_base + s0;
            put = false;
        }
        todo &= ~grp;
    }
    if (put) {
        const uint32_t pos = atomicAdd(&slot_fill[slot], 1u);
        members[slot_off[slot] + pos] = key;
        member_comp[slot_off[slot] + pos] = comp_base + slot;
        atomicMin(&minkey[slot], (unsigned long long)key);
    }
}
__global__ void k_cc_remap(uint32_t *__restrict__ comp, uint64_t n, const uint32_t *__restrict__ rank) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) comp[i] = rank[comp[i]];
}

// ---- features: one thread per sample record ----
__global__ void k_features(mf_index_view ix, const uint32_t *__restrict__ comp_of,
                           const uint64_t *__restrict__ keys, const uint16_t *__restrict__ cnts, uint64_t n, int threshold,
                           const uint8_t *__restrict__ sel, unsigned long long *__restrict__ vec, unsigned int *__restrict__ found) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        int c = (int)cnts[i];
        if (c <= threshold) continue;                       // value > threshold (buildAndPrintVector :196-199)
        uint32_t idx, val;
        if (!mf_index_find(ix, keys[i], &idx, &val)) continue;   // hm.contains(kmer) (KmersPresenceWorker :583-587)
        if (sel && !sel[idx]) continue;                     // selected.getWithZero(kmer) > 0 (:193)
        uint32_t comp = comp_of[idx];
        atomicAdd(&vec[comp], (unsigned long long)c);
        atomicAdd(&found[comp], 1u);
    }
}

// ---- features, the other way round: one thread per COMPONENT k-mer, looked up in the sample's index (the component
// lists are a few percent of a sample's table: 3e7 probes instead of 3.6e8, and no index over the components).  The
// k-mers of a component are contiguous, so a wave mostly adds to ONE component: wave-level sum, one atomic. ----
__global__ void k_features_rev(mf_index_view ix, const uint64_t *__restrict__ ckeys, const uint32_t *__restrict__ comp_of, uint64_t nk,
                               int threshold, const uint8_t *__restrict__ sel, unsigned long long *__restrict__ vec, unsigned int *__restrict__ found) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t comp = 0xFFFFFFFFu, val = 0, hit = 0;
    if (j < nk) {
        uint32_t idx, v;
        comp = comp_of[j];
        if ((!sel || sel[j]) && mf_index_find(ix, ckeys[j], &idx, &v) && (int)v > threshold) { val = v; hit = 1; }
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

// features from per-k-mer occurrence counters (reads mode): same reduction as k_features_rev
__global__ void k_features_occ(const unsigned long long *__restrict__ occ, const uint32_t *__restrict__ comp_of, uint64_t nk,
                               int threshold, const uint8_t *__restrict__ sel, unsigned long long *__restrict__ vec, unsigned int *__restrict__ found) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t comp = 0xFFFFFFFFu, hit = 0; unsigned long long val = 0;
    if (j < nk) {
        comp = comp_of[j];
        const unsigned long long v = occ[j];
        if ((!sel || sel[j]) && (long long)v > (long long)threshold) { val = v; hit = 1; }
    }
    const uint32_t first = __shfl(comp, 0, 64);
    if (__ballot(comp != first) == 0ull) {
        for (int d = 32; d >= 1; d >>= 1) { val += __shfl_down(val, d, 64); hit += __shfl_down(hit, d, 64); }
        if (mf_lane() == 0 && hit) { atomicAdd(&vec[first], val); atomicAdd(&found[first], hit); }
    } else if (hit) {
        atomicAdd(&vec[comp], val);
        atomicAdd(&found[comp], 1u);
    }
}

// --selected (FeaturesCalculatorMain.java:55-57, 115-117, 193): sel[j] = 1 iff component k-mer j has a value > 0 in the table of
// selected k-mers (selected.getWithZero(kmer) > 0); selcnt[c] = such k-mers of component c (kmersCount, the breadth's denominator)
__global__ void k_features_select(mf_index_view ix, const uint64_t *__restrict__ ckeys, const uint32_t *__restrict__ comp_of, uint64_t nk,
                                  uint8_t *__restrict__ sel, unsigned int *__restrict__ selcnt) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t comp = 0xFFFFFFFFu, hit = 0;
    if (j < nk) {
        uint32_t idx, v;
        comp = comp_of[j];
        hit = mf_index_find(ix, ckeys[j], &idx, &v) && (int)v > 0;
        sel[j] = (uint8_t)hit;
    }
    const uint32_t first = __shfl(comp, 0, 64);
    if (__ballot(comp != first) == 0ull) {
        const uint32_t tot = (uint32_t)__popcll(__ballot(hit != 0));
        if (mf_lane() == 0 && tot) atomicAdd(&selcnt[first], tot);
    } else if (hit) atomicAdd(&selcnt[comp], 1u);
}

static inline unsigned cgrid(uint64_t n, unsigned bs = 256) { return (unsigned)((n + bs - 1) / bs); }

// builds d_kmers / d_comp / index from the host vectors of a finished mf_comps (mf_comps_load path)
static int comps_upload(mf_comps *C) {
    mf_ctx *ctx = C->ctx;
    const uint64_t nk = C->kmers.size();
    C->n_kmers = nk;
    if (nk >= 0xFFFFFFFFull) return mf_set_error("components: too many k-mers");
    void *p = nullptr;
    MF_TRY(mf_alloc(ctx, (nk ? nk : 1) * 8, &p)); C->d_kmers = (uint64_t *)p; C->kmers_bytes = (nk ? nk : 1) * 8;
    MF_TRY(mf_alloc(ctx, (nk ? nk : 1) * 4, &p)); C->d_comp = (uint32_t *)p; C->comp_bytes = (nk ? nk : 1) * 4;
    if (nk) {
        std::vector<uint32_t> comp(nk);
        for (uint64_t c = 0; c < C->n; c++)
            for (uint64_t j = C->offsets[c]; j < C->offsets[c + 1]; j++) comp[j] = (uint32_t)c;
        MF_HIP(hipMemcpyAsync(C->d_kmers, C->kmers.data(), nk * 8, hipMemcpyHostToDevice, ctx->stream));
        MF_HIP(hipMemcpyAsync(C->d_comp, comp.data(), nk * 4, hipMemcpyHostToDevice, ctx->stream));
        MF_HIP(hipStreamSynchronize(ctx->stream));
    }
    C->host_ready = true;
    return MF_OK;
}

// member lists on the host (ascending k-mers inside each component), built on first use
int mf_comps_materialize(mf_comps *C) {
    if (C->host_ready) return MF_OK;
    mf_ctx *ctx = C->ctx;
    MF_HIP(hipSetDevice(ctx->device));
    const uint64_t nk = C->n_kmers;
    C->offsets.assign(C->n + 1, 0);
    for (uint64_t c = 0; c < C->n; c++) C->offsets[c + 1] = C->offsets[c] + C->sizes[c];
    C->kmers.resize(nk);
    if (nk) {
        // grouped by component, ascending inside: sorted in HBM (the host version of this took seconds per 1e7 k-mers)
        mf_buf<uint64_t> sorted; MF_TRY(sorted.alloc(ctx, nk));
        MF_TRY(mf_sort_kmers_by_comp(ctx, C->d_comp, C->d_kmers, nk, 2 * (C->k > 0 ? C->k : 32), (uint32_t)std::max<uint64_t>(C->n, 1), sorted.p));
        MF_HIP(hipMemcpyAsync(C->kmers.data(), sorted.p, nk * 8, hipMemcpyDeviceToHost, ctx->stream));
        MF_HIP(hipStreamSynchronize(ctx->stream));
    }
    C->host_ready = true;
    return MF_OK;
}

int mf_cc_build(mf_ctx *ctx, uint64_t n, int k, const uint16_t *d_counts, const uint64_t *d_keys, int b1, int b2,
                const std::function<int(uint32_t *)> &adjacency, mf_comps **out) {
    hipStream_t st = ctx->stream;
    if (n >= 0xFFFFFFFFull) return mf_set_error("components: more than 2^32 vertices is not supported");
    if (b1 < 0) b1 = 0;
    struct rec { uint64_t size; int64_t weight; int32_t thr; uint64_t minkey; uint32_t temp; };
    std::vector<rec> recs;
    struct level_buf { mf_buf<uint64_t> members; mf_buf<uint32_t> comp; uint64_t n = 0; };
    std::vector<std::unique_ptr<level_buf>> levels;
    uint64_t total_k = 0;
    if (n) {
        mf_buf<uint32_t> nbr, parent, root, csize, keptslot, slot_fill, k_root, k_size; mf_buf<unsigned long long> cweight, k_weight, k_minkey;
        mf_buf<uint8_t> alive; mf_buf<unsigned int> counters; mf_buf<uint64_t> slot_off, tot;
        const uint64_t max_kept = n / (uint64_t)std::max(b1, 1) + 1;
        MF_TRY(nbr.alloc(ctx, n * 8)); MF_TRY(parent.alloc(ctx, n)); MF_TRY(root.alloc(ctx, n)); MF_TRY(csize.alloc(ctx, n));
        MF_TRY(keptslot.alloc(ctx, n)); MF_TRY(cweight.alloc(ctx, n)); MF_TRY(alive.alloc(ctx, n)); MF_TRY(counters.alloc(ctx, 4));
        MF_TRY(k_root.alloc(ctx, max_kept)); MF_TRY(k_size.alloc(ctx, max_kept)); MF_TRY(k_weight.alloc(ctx, max_kept));
        MF_TRY(k_minkey.alloc(ctx, max_kept)); MF_TRY(slot_off.alloc(ctx, max_kept + 1)); MF_TRY(slot_fill.alloc(ctx, max_kept));
        MF_TRY(tot.alloc(ctx, 1));
        // the edges that leave their tile (k_cc_hook_tile): room for one per two vertices, 4 bytes per vertex beside the adjacency's 32
        mf_buf<uint2> edges; mf_buf<unsigned long long> ecount;
        const uint64_t ecap = n / 2 + 1024;
        MF_TRY(edges.alloc(ctx, ecap)); MF_TRY(ecount.alloc(ctx, 2));
        cc_kept_arrays K; K.root = k_root.p; K.size = k_size.p; K.weight = k_weight.p; K.minkey = k_minkey.p;
        {
            mf_ktimer tm(ctx, "k_cc_adjacency");
            MF_TRY(adjacency(nbr.p));
        }
        MF_HIP(hipMemsetAsync(alive.p, 1, n, st));
        // the survivors of a level, listed for the next one (sparse levels: see k_ccs_init); [cur] is read, [cur ^ 1] written
        mf_buf<uint32_t> alist[2];
        uint64_t m = n;                     // alive vertices of the level at hand
        bool sparse = false;                // the level runs on alist[cur]
        int cur = 0;
        for (int thr = 1;; thr++) {
            MF_HIP(hipMemsetAsync(counters.p, 0, 16, st));
            const uint32_t *L = sparse ? alist[cur].p : nullptr;
            const uint64_t span = sparse ? m : n;                              // threads of the per-vertex kernels
            {
                mf_ktimer tm(ctx, "k_cc_hook");
                if (sparse) {
                    if (m) k_ccs_init<<<cgrid(m), 256, 0, st>>>(L, m, parent.p, csize.p, cweight.p);
                    if (m) k_cc_hook<<<cgrid(m), 256, 0, st>>>(nbr.p, alive.p, parent.p, m, L, nullptr);
                } else {
                    MF_HIP(hipMemsetAsync(ecount.p, 0, 16, st));
                    k_cc_hook_tile<<<cgrid(n, CC_TILE), 256, 0, st>>>(nbr.p, alive.p, parent.p, csize.p, cweight.p, n, edges.p, ecount.p, ecap);
                    k_cc_hook_edges<<<cgrid(ecap), 256, 0, st>>>(edges.p, ecount.p, ecap, alive.p, parent.p);
                    k_cc_hook<<<cgrid(n), 256, 0, st>>>(nbr.p, alive.p, parent.p, n, nullptr, ecount.p);      // (only if the list overflowed)
                }
            }
            {
                mf_ktimer tm(ctx, "k_cc_stats");
                if (sparse) { if (m) k_ccs_stats<<<cgrid(m), 256, 0, st>>>(L, m, parent.p, root.p, d_counts, csize.p, cweight.p); }
                else {
                    if (ctx->opt_cc_compress) k_cc_compress<<<cgrid(n), 256, 0, st>>>(alive.p, parent.p, n);        // (5-fold depth: k_cc_stats 51 -> 42 ms)
                    k_cc_flatten_stats<<<cgrid(n, CC_TILE), 256, 0, st>>>(alive.p, parent.p, root.p, d_counts, csize.p, cweight.p, n);
                }
                if (span) k_cc_classify<<<cgrid(span), 256, 0, st>>>(alive.p, root.p, csize.p, cweight.p, span, (uint32_t)b1, (uint32_t)b2, keptslot.p, K, counters.p, L);
            }
            unsigned int cnt[4];
            MF_HIP(hipMemcpyAsync(cnt, counters.p, 16, hipMemcpyDeviceToHost, st));
            MF_HIP(hipStreamSynchronize(st));
            const uint32_t nkept = cnt[0], nkm = cnt[1], nbig = cnt[2];
            auto lv = std::make_unique<level_buf>();
            MF_TRY(lv->members.alloc(ctx, nkm)); MF_TRY(lv->comp.alloc(ctx, nkm));
            lv->n = nkm;
            MF_TRY(mf_scan<1>(ctx, k_size.p, slot_off.p, (uint64_t)nkept, tot.p));      // (multi-block: nkept reaches millions)
            MF_HIP(hipMemsetAsync(slot_fill.p, 0, (size_t)(nkept ? nkept : 1) * 4, st));
            // the survivors are listed from the level on where the list pays: it can hold a third of the vertices (a dense level
            // that leaves more behind writes no list, and the next level is dense again)
            const uint64_t lcap = std::max<uint64_t>(n / 3, 1);
            const bool want_list = nbig != 0 && ctx->opt_cc_sparse != 0;
            uint32_t *NL = nullptr;
            if (want_list) {
                const uint64_t need = std::min<uint64_t>(sparse ? m : n, sparse ? m : lcap);
                if (alist[cur ^ 1].n < need) { alist[cur ^ 1].reset(); MF_TRY(alist[cur ^ 1].alloc(ctx, need)); }
                NL = alist[cur ^ 1].p;
            }
            {
                mf_ktimer tm(ctx, "k_cc_members");
                // (a dense level writes the list as far as it has room and counts the survivors: the list stands if they all fitted)
                if (span) k_cc_members<<<cgrid(span), 256, 0, st>>>(alive.p, root.p, csize.p, d_counts, d_keys, span, (uint32_t)b1, (uint32_t)b2,
                                                                    (uint32_t)(thr + 1), keptslot.p, slot_off.p, slot_fill.p, (uint32_t)recs.size(),
                                                                    k_minkey.p, lv->members.p, lv->comp.p, &counters.p[3], L, NL, (uint32_t)std::min<uint64_t>(sparse ? m : lcap, 0xFFFFFFFFull));
            }
            unsigned int na = 0;
            MF_HIP(hipMemcpyAsync(&na, &counters.p[3], 4, hipMemcpyDeviceToHost, st));
            if (nkept) {
                std::vector<uint32_t> hs(nkept); std::vector<unsigned long long> hw(nkept), hm(nkept);
                MF_HIP(hipMemcpyAsync(hs.data(), k_size.p, (size_t)nkept * 4, hipMemcpyDeviceToHost, st));
                MF_HIP(hipMemcpyAsync(hw.data(), k_weight.p, (size_t)nkept * 8, hipMemcpyDeviceToHost, st));
                MF_HIP(hipMemcpyAsync(hm.data(), k_minkey.p, (size_t)nkept * 8, hipMemcpyDeviceToHost, st));
                MF_HIP(hipStreamSynchronize(st));
                for (uint32_t i = 0; i < nkept; i++)
                    recs.push_back({hs[i], (int64_t)hw[i], thr, hm[i], (uint32_t)recs.size()});
            } else MF_HIP(hipStreamSynchronize(st));
            total_k += nkm;
            levels.push_back(std::move(lv));
            if (ctx->opt_verbose)
                fprintf(stderr, "[mf] components: thr=%d (%s, %llu vertices) kept=%u (%u k-mers) big=%u, %u vertices go on to the next level\n", thr, sparse ? "sparse" : "dense",
                        (unsigned long long)(sparse ? m : n), nkept, nkm, nbig, na);
            if (!nbig) break;
            if (thr > MF_MAX_COUNT) return mf_set_error("components: threshold loop did not terminate");
            if (sparse) { cur ^= 1; m = na; }
            else if ((uint64_t)na <= lcap && NL) { sparse = true; cur ^= 1; m = na; }      // (the dense level's list holds all its survivors)
        }
    }
    // final order: ConnectedComponent.compareTo (src/structures/ConnectedComponent.java:125-136): thr asc, weight desc,
    // size desc; ties (discovery order in the reference = hash / race dependent) broken by the smallest k-mer
    std::sort(recs.begin(), recs.end(), [](const rec &a, const rec &b) {
        if (a.thr != b.thr) return a.thr < b.thr;
        if (a.weight != b.weight) return a.weight > b.weight;
        if (a.size != b.size) return a.size > b.size;
        return a.minkey < b.minkey;
    });
    mf_comps *C = new mf_comps();
    C->ctx = ctx; C->k = k; C->n = recs.size(); C->n_kmers = total_k;
    std::vector<uint32_t> rank(recs.size() ? recs.size() : 1);
    for (size_t i = 0; i < recs.size(); i++) {
        C->sizes.push_back(recs[i].size); C->weights.push_back(recs[i].weight); C->thr.push_back(recs[i].thr);
        rank[recs[i].temp] = (uint32_t)i;
    }
    int rc = MF_OK;
    do {
        void *p = nullptr;
        if ((rc = mf_alloc(ctx, (total_k ? total_k : 1) * 8, &p)) < 0) break;
        C->d_kmers = (uint64_t *)p; C->kmers_bytes = (total_k ? total_k : 1) * 8;
        if ((rc = mf_alloc(ctx, (total_k ? total_k : 1) * 4, &p)) < 0) break;
        C->d_comp = (uint32_t *)p; C->comp_bytes = (total_k ? total_k : 1) * 4;
        uint64_t pos = 0;
        for (auto &lv : levels) {
            if (!lv->n) continue;
            hipMemcpyAsync(C->d_kmers + pos, lv->members.p, lv->n * 8, hipMemcpyDeviceToDevice, st);
            hipMemcpyAsync(C->d_comp + pos, lv->comp.p, lv->n * 4, hipMemcpyDeviceToDevice, st);
            pos += lv->n;
        }
        if (total_k) {
            mf_buf<uint32_t> d_rank;
            if ((rc = d_rank.alloc(ctx, rank.size())) < 0) break;
            hipMemcpyAsync(d_rank.p, rank.data(), rank.size() * 4, hipMemcpyHostToDevice, st);
            k_cc_remap<<<cgrid(total_k), 256, 0, st>>>(C->d_comp, total_k, d_rank.p);
            if (hipStreamSynchronize(st) != hipSuccess) { rc = mf_set_error("components: remap failed"); break; }
        }
    } while (0);
    if (rc < 0) { mf_comps_destroy(C); return rc; }
    *out = C;
    return MF_OK;
}

extern "C" int mf_cut_components_device(mf_ctx *ctx, mf_table *t, int b1, int b2, mf_comps **out) {
    mf_range rng_("mf:components");
    if (!ctx || !t || !out) return mf_set_error("mf_cut_components_device: NULL argument");
    *out = nullptr;
    MF_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint64_t n = t->n;
    const int k = t->k;
    if (n >= 0xFFFFFFFFull) return mf_set_error("components: more than 2^32 vertices is not supported");
    if (n) MF_TRY(mf_table_ensure_index(t));
    return mf_cc_build(ctx, n, k, t->d_counts, t->d_keys, b1, b2, [&](uint32_t *nbr) -> int {
        {
            if (t->index.skm_k && t->index.part_bits && t->d_part_off && !ctx->opt_nbr_global && (n >> t->part_bits) >= 100) {      // (small partitions: the set-up per partition outweighs the local lookups)
                const uint32_t np = 1u << t->part_bits;
                const unsigned grid = (unsigned)std::min<uint64_t>((np + NB_WAVES - 1) / NB_WAVES, (uint64_t)ctx->n_cu * 64);
                const unsigned grid2 = (unsigned)std::min<uint64_t>(np, (uint64_t)ctx->n_cu * 16);
#define CC_ADJ_K(KK) case KK: k_cc_adjacency_part<1, KK><<<grid, 64 * NB_WAVES, 0, st>>>(mf_view(t->index), t->d_keys, t->d_part_off, np, k, nbr); \
                              k_cc_adjacency_part<2, KK><<<grid2, 64 * NB_WAVES, 0, st>>>(mf_view(t->index), t->d_keys, t->d_part_off, np, k, nbr); break;
                switch (k) {                                                    // (k as a compile-time constant: see k_ut_flags_part's dispatch, mf_unitig.hip)
                    CC_ADJ_K(21) CC_ADJ_K(23) CC_ADJ_K(25) CC_ADJ_K(27) CC_ADJ_K(29) CC_ADJ_K(31)
                    default:
                        k_cc_adjacency_part<1><<<grid, 64 * NB_WAVES, 0, st>>>(mf_view(t->index), t->d_keys, t->d_part_off, np, k, nbr);
                        k_cc_adjacency_part<2><<<grid2, 64 * NB_WAVES, 0, st>>>(mf_view(t->index), t->d_keys, t->d_part_off, np, k, nbr);
                }
#undef CC_ADJ_K
            } else
            k_cc_adjacency<<<cgrid(n), 256, 0, st>>>(mf_view(t->index), t->d_keys, n, k, nbr);
        }
        return MF_OK;
    }, out);
}

extern "C" void mf_comps_destroy(mf_comps *c) {
    if (!c) return;
    if (--c->refs > 0) return;
    if (c->d_kmers) mf_release(c->ctx, c->d_kmers, c->kmers_bytes);
    if (c->d_comp) mf_release(c->ctx, c->d_comp, c->comp_bytes);
    if (c->index.slots) mf_release(c->ctx, c->index.slots, c->index_bytes);
    delete c;
}
extern "C" int mf_comps_stats(const mf_comps *c, uint64_t *n_comp, uint64_t *n_kmers) {
    if (!c) return mf_set_error("comps is NULL");
    if (n_comp) *n_comp = c->n;
    if (n_kmers) *n_kmers = c->n_kmers;
    return MF_OK;
}
extern "C" int mf_comps_export(const mf_comps *cc, uint64_t *sizes, int64_t *weights, int32_t *thr, uint64_t *kmer_offsets,
                               uint64_t *kmers) {
    if (!cc) return mf_set_error("comps is NULL");
    mf_comps *c = const_cast<mf_comps *>(cc);
    MF_TRY(mf_comps_materialize(c));
    if (sizes && c->n) memcpy(sizes, c->sizes.data(), c->n * 8);
    if (weights && c->n) memcpy(weights, c->weights.data(), c->n * 8);
    if (thr && c->n) memcpy(thr, c->thr.data(), c->n * 4);
    if (kmer_offsets) memcpy(kmer_offsets, c->offsets.data(), (c->n + 1) * 8);
    if (kmers && !c->kmers.empty()) memcpy(kmers, c->kmers.data(), c->kmers.size() * 8);
    return MF_OK;
}

// used by mf_comps_load (mf_io.hip)
int mf_comps_from_host(mf_ctx *ctx, int k, const std::vector<uint64_t> &sizes, const std::vector<int64_t> &weights,
                       const std::vector<int32_t> &thr, const std::vector<uint64_t> &offsets, const std::vector<uint64_t> &kmers,
                       mf_comps **out) {
    mf_comps *C = new mf_comps();
    C->ctx = ctx; C->k = k; C->n = sizes.size();
    C->sizes = sizes; C->weights = weights; C->thr = thr; C->offsets = offsets; C->kmers = kmers;
    int rc = comps_upload(C);
    if (rc < 0) { mf_comps_destroy(C); return rc; }
    *out = C;
    return MF_OK;
}

// the selection of --selected on the component k-mers: mask per k-mer + count per component (host); selected == NULL: no mask
static int features_selection(mf_ctx *ctx, mf_comps *c, mf_table *selected, mf_buf<uint8_t> &mask, std::vector<unsigned int> &selcnt) {
    if (!selected) return MF_OK;
    if (selected->ctx != ctx) return mf_set_error("features: the table of selected k-mers belongs to another context");
    hipStream_t st = ctx->stream;
    mf_buf<unsigned int> dcnt;
    MF_TRY(mask.alloc(ctx, c->n_kmers ? c->n_kmers : 1)); MF_TRY(dcnt.alloc(ctx, c->n));
    MF_HIP(hipMemsetAsync(mask.p, 0, c->n_kmers ? c->n_kmers : 1, st));
    MF_HIP(hipMemsetAsync(dcnt.p, 0, c->n * 4, st));
    if (selected->n && c->n_kmers) {
        MF_TRY(mf_table_ensure_index(selected));
        mf_ktimer tm(ctx, "k_features_select");
        k_features_select<<<cgrid(c->n_kmers), 256, 0, st>>>(mf_view(selected->index), c->d_kmers, c->d_comp, c->n_kmers, mask.p, dcnt.p);
    }
    selcnt.resize(c->n);
    MF_HIP(hipMemcpyAsync(selcnt.data(), dcnt.p, c->n * 4, hipMemcpyDeviceToHost, st));
    MF_HIP(hipStreamSynchronize(st));
    return MF_OK;
}
// breadth[i] = ((double) kmersFound) / kmersCount (:203); with --selected kmersCount counts the selected k-mers only and a
// component without any gives 0.0 / 0.0 = NaN, as the Java division does
static void features_breadth(const mf_comps *c, const std::vector<unsigned int> &hf, const std::vector<unsigned int> *selcnt, int threshold, double *breadth) {
    for (uint64_t i = 0; i < c->n; i++) {
        const uint64_t cnt = selcnt ? (uint64_t)(*selcnt)[i] : c->sizes[i];
        // a negative threshold makes absent k-mers (value 0) count as found (value > threshold)
        const double f = threshold < 0 ? (double)cnt : (double)hf[i];
        breadth[i] = f / (double)cnt;
    }
}

extern "C" int mf_features_device_selected(mf_ctx *ctx, mf_comps *c, const mf_table *sample, mf_table *selected, int threshold,
                                           int64_t *vec, double *breadth) {
    mf_range rng_("mf:features");
    if (!ctx || !c || !sample || !vec) return mf_set_error("mf_features_device: NULL argument");
    MF_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint64_t nc = c->n;
    if (!nc) return MF_OK;
    mf_buf<uint8_t> mask; std::vector<unsigned int> selcnt;
    MF_TRY(features_selection(ctx, c, selected, mask, selcnt));
    mf_buf<unsigned long long> dvec; mf_buf<unsigned int> dfound;
    MF_TRY(dvec.alloc(ctx, nc)); MF_TRY(dfound.alloc(ctx, nc));
    MF_HIP(hipMemsetAsync(dvec.p, 0, nc * 8, st));
    MF_HIP(hipMemsetAsync(dfound.p, 0, nc * 4, st));
    if (sample->n && c->n_kmers) {
        // probe whichever side has an index already or is cheaper to index: the sample's table (built for the unitigs) ...
        // (components of ANOTHER context of the same device -- pipeline.py's peer context for overlapped samples --: only read here, never indexed:
        // their index would come out of this context's workspace and two contexts' threads would race for it)
        if (sample->index.slots || c->n_kmers < sample->n || c->ctx != ctx) {
            MF_TRY(mf_table_ensure_index(const_cast<mf_table *>(sample)));
            mf_ktimer tm(ctx, "k_features");
            k_features_rev<<<cgrid(c->n_kmers), 256, 0, st>>>(mf_view(sample->index), c->d_kmers, c->d_comp, c->n_kmers, threshold, mask.p, dvec.p, dfound.p);
        } else {          // ... or the components
            if (!c->index.slots) MF_TRY(mf_index_build(ctx, c->d_kmers, nullptr, c->n_kmers, &c->index, &c->index_bytes));
            unsigned grid = (unsigned)std::min<uint64_t>((sample->n + 255) / 256, 65536);
            mf_ktimer tm(ctx, "k_features");
            k_features<<<grid, 256, 0, st>>>(mf_view(c->index), c->d_comp, sample->d_keys,
                                             sample->d_counts, sample->n, threshold, mask.p, dvec.p, dfound.p);
        }
    }
    std::vector<unsigned int> hf(nc);
    MF_HIP(hipMemcpyAsync(vec, dvec.p, nc * 8, hipMemcpyDeviceToHost, st));
    MF_HIP(hipMemcpyAsync(hf.data(), dfound.p, nc * 4, hipMemcpyDeviceToHost, st));
    MF_HIP(hipStreamSynchronize(st));
    if (breadth) features_breadth(c, hf, selected ? &selcnt : nullptr, threshold, breadth);
    return MF_OK;
}
extern "C" int mf_features_device(mf_ctx *ctx, mf_comps *c, const mf_table *sample, int threshold, int64_t *vec, double *breadth) {
    return mf_features_device_selected(ctx, c, sample, nullptr, threshold, vec, breadth);
}

int mf_presence_core(mf_ctx *ctx, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n_reads, uint64_t n_bases, int k,
                     const mf_index &index, unsigned long long *d_occ);
extern "C" int mf_features_reads_device_selected(mf_ctx *ctx, mf_comps *c, const void *d_bases, const void *d_offsets, uint64_t n_reads,
                                                 uint64_t n_bases, int k, mf_table *selected, int threshold, int64_t *vec, double *breadth) {
    mf_range rng_("mf:features_reads");
    if (!ctx || !c || !vec) return mf_set_error("mf_features_reads_device: NULL argument");
    MF_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint64_t nc = c->n;
    if (!nc) return MF_OK;
    mf_buf<uint8_t> mask; std::vector<unsigned int> selcnt;
    MF_TRY(features_selection(ctx, c, selected, mask, selcnt));
    mf_buf<unsigned long long> dvec, occ; mf_buf<unsigned int> dfound;
    MF_TRY(dvec.alloc(ctx, nc)); MF_TRY(dfound.alloc(ctx, nc)); MF_TRY(occ.alloc(ctx, c->n_kmers));
    MF_HIP(hipMemsetAsync(dvec.p, 0, nc * 8, st));
    MF_HIP(hipMemsetAsync(dfound.p, 0, nc * 4, st));
    MF_HIP(hipMemsetAsync(occ.p, 0, (c->n_kmers ? c->n_kmers : 1) * 8, st));
    if (c->n_kmers) {
        if (!c->index.slots) MF_TRY(mf_index_build(ctx, c->d_kmers, nullptr, c->n_kmers, &c->index, &c->index_bytes));   // hm.put(kmer, 0) :99-103
        MF_TRY(mf_presence_core(ctx, (const uint8_t *)d_bases, (const uint64_t *)d_offsets, n_reads, n_bases, k, c->index, occ.p));
        mf_ktimer tm(ctx, "k_features");
        k_features_occ<<<cgrid(c->n_kmers), 256, 0, st>>>(occ.p, c->d_comp, c->n_kmers, threshold, mask.p, dvec.p, dfound.p);
    }
    std::vector<unsigned int> hf(nc);
    MF_HIP(hipMemcpyAsync(vec, dvec.p, nc * 8, hipMemcpyDeviceToHost, st));
    MF_HIP(hipMemcpyAsync(hf.data(), dfound.p, nc * 4, hipMemcpyDeviceToHost, st));
    MF_HIP(hipStreamSynchronize(st));
    if (breadth) features_breadth(c, hf, selected ? &selcnt : nullptr, threshold, breadth);
    return MF_OK;
}
extern "C" int mf_features_reads_device(mf_ctx *ctx, mf_comps *c, const void *d_bases, const void *d_offsets, uint64_t n_reads,
                                        uint64_t n_bases, int k, int threshold, int64_t *vec, double *breadth) {
    return mf_features_reads_device_selected(ctx, c, d_bases, d_offsets, n_reads, n_bases, k, nullptr, threshold, vec, breadth);
}

// DistanceMatrixCalculatorMain.brayCurtisDistance :140-152.  Both sums are exact integers < 2^53 in double
// arithmetic for realistic inputs, so one IEEE division reproduces the Java result bit for bit.
extern "C" int mf_bray_curtis(const int64_t *vecs, int n_samples, int n_comp, double *out) {
    if (!vecs || !out || n_samples < 0 || n_comp < 0) return mf_set_error("mf_bray_curtis: bad argument");
    for (int i = 0; i < n_samples; i++) {
        out[(size_t)i * n_samples + i] = 0.0;
        for (int j = i + 1; j < n_samples; j++) {
            double sumdiff = 0, sum = 0;
            const int64_t *a = vecs + (size_t)i * n_comp, *b = vecs + (size_t)j * n_comp;
            for (int p = 0; p < n_comp; p++) {
                double x = (double)a[p], y = (double)b[p];
                sumdiff += x > y ? x - y : y - x;
                sum += (x < 0 ? -x : x) + (y < 0 ? -y : y);
            }
            out[(size_t)i * n_samples + j] = out[(size_t)j * n_samples + i] = sumdiff / sum;
        }
    }
    return MF_OK;
}

// =============================================================================================
// Distributed component cutter: the cutter table and the components step over ALL samples' unitigs are
// the one global join of the pipeline (ComponentCutterMain.runImpl, src/tools/ComponentCutterMain.java:78-114;
// ComponentsBuilder.run, src/algo/ComponentsBuilder.java:58-153).  Repeating them on every rank costs each rank the work of
// the whole node.  Here every rank OWNS the k-mers whose minimizer-partition hash starts with its rank (top log2(world)
// bits of mf_skm_ph): it holds that shard of the cutter table, finds its vertices' neighbours (its own ones in its index,
// the others' by asking their owners), and solves its part of every threshold level; the ranks exchange
//   * their unitigs (all-gather); every rank then counts the k-mers IT owns (mf_count_device_shard)
//   * neighbour queries and answers                                                   (once, all-to-all both ways)
//   * per level: the edges between fragments of different ranks (all-to-all + all-gather of root pairs), the fragments'
//     sizes / weights (all-gather), the kept components of each owner (all-gather)
//   * the members of the kept components                                              (once, all-gather)
// The collectives themselves are torch.distributed's (metafast_amd/pipeline.py); the entry points below take and fill
// device buffers.  Global vertex id = base[rank] + position in the rank's shard table (< 2^32 over all ranks).
//
// Per level, on every rank: union-find over its own vertices and the edges inside its shard (the kernels above) -> local
// roots ("fragments"); an edge to another rank's vertex becomes a pair (root of this end, root of that end) -- the lower
// rank sends (remote vertex, global id of its own root) to the owner, who answers with the pair; all pairs are gathered on
// every rank, which runs the same union-find over fragment roots (an array over all global ids) and so knows the global root
// of each of its fragments; sizes and weights of the fragments are gathered and summed per global root everywhere, so every
// rank classifies alike.  Whether a remote neighbour is still alive at level t needs no message: the edge was active at level
// t-1, so both ends were in the same component and share its fate; what remains is the neighbour's own value, which the
// answer to the query brought along.
// =============================================================================================
struct dcc_query { uint64_t key; uint64_t src; };
struct dcc_kept_rec { uint32_t g, size; unsigned long long weight; };
#define DCC_NOKEY 0x7FFFFFFFFFFFFFFFull          /* "no member here" (k-mers are < 2^62; signed-safe for a MIN all-reduce on int64) */
struct mf_dcc {
    mf_ctx *ctx = nullptr; mf_table *t = nullptr; int k = 0, rank = 0, world = 1, lw = 0;
    uint32_t n = 0, n_total = 0; std::vector<uint32_t> base;
    mf_buf<uint32_t> nbr, parent, root, csize, groot;        // [8n] local neighbour ids; [n]; [n]; [n]; [n] global root of a LOCAL ROOT
    mf_buf<unsigned long long> cweight; mf_buf<uint8_t> alive;
    mf_buf<uint2> edges; mf_buf<unsigned long long> ecount; uint64_t ecap = 0;      // the edges that leave their tile (k_cc_hook_tile)
    mf_buf<unsigned int> ctr;                                // [64] counters
    // queries (16 bytes: key, source = vertex*8 + neighbour number) grouped by owner
    std::vector<uint64_t> qoff;                              // [world + 1]
    mf_buf<unsigned long long> qcur;                         // [64] cursors
    mf_buf<dcc_query> qflat; uint32_t nq = 0;                // this rank's queries in the order they arose
    // cross edges
    mf_buf<uint32_t> xv, xu; mf_buf<uint8_t> xr, xalive; mf_buf<uint16_t> xval; uint64_t nx = 0;
    std::vector<uint64_t> xoff;                              // [world + 1] (pairs this rank sends per level)
    // contracted graph, replicated
    mf_buf<uint32_t> pg, gsize; mf_buf<unsigned long long> gweight, gmin;     // [n_total]
    // results
    mf_buf<uint64_t> mk; mf_buf<uint32_t> mg; uint64_t nm = 0;          // [n] members of kept components: k-mer, global root
    uint64_t nstat = 0, nkept_owned = 0; int level = 0;
    mf_buf<uint32_t> touched; mf_buf<uint2> hp; std::vector<uint64_t> xstart;     // [n] global roots this rank contributes to; [nx] half pairs; [world + 1] cross edges by rank
    mf_buf<dcc_kept_rec> keptbuf;
    mf_buf<uint64_t> gmk; mf_buf<uint2> gruns; uint64_t n_runs = 0;        // members grouped by component: k-mers, (root, count) runs
    mf_buf<uint64_t> dseg; mf_buf<uint32_t> dbase;         // [world + 1] on the device: first record of every rank (per level), first vertex of every rank
};


static int dcc_log2(int w) { int l = 0; while ((1 << l) < w) l++; return l; }
// ---- neighbours: own ones looked up partition-locally (mf_nbr.h), the others' become queries (flat list, owner in the top
// byte of src; sorted by owner afterwards)
#define DCC_SRC_MASK ((1ull << 56) - 1ull)
// A wave takes room in the query list DCC_QCHUNK entries at a time (one atomic on the list's cursor per chunk, not per 64
// k-mers: 600 k atomics on one address cost 7 ms); what it leaves unused of a chunk is marked void (src = ~0).
#define DCC_QCHUNK 512u
template <int MODE>
__global__ __launch_bounds__(64 * NB_WAVES) void k_dcc_adjacency_part(mf_index_view ix, const uint64_t *__restrict__ keys, const uint64_t *__restrict__ part_off,
                                                                    uint32_t p_lo, uint32_t p_hi, int k, int lw, uint32_t me, uint32_t *__restrict__ nbr,
                                                                    dcc_query *__restrict__ q, uint32_t qcap, unsigned int *__restrict__ qcount) {
    __shared__ nb_lds S;
    const uint64_t kmask = (1ull << (2 * k)) - 1;
    uint32_t qcur = 0, qend = 0;                                         // this wave's chunk (wave-uniform)
    auto void_rest = [&]() {
        for (uint32_t i = qcur + mf_lane(); i < qend; i += 64) if (i < qcap) q[i].src = ~0ull;
    };
    nb_for_each<MODE>(ix, keys, part_off, p_lo, p_hi, k, S, lw, me, [&](uint64_t v, uint64_t x, const uint32_t (&idx)[8], uint32_t, uint32_t foreign, bool have) {
        if (have) {
            uint4 *o = reinterpret_cast<uint4 *>(nbr + v * 8);
            o[0] = make_uint4(idx[0], idx[1], idx[2], idx[3]);
            o[1] = make_uint4(idx[4], idx[5], idx[6], idx[7]);
        }
        const uint32_t cnt = have ? (uint32_t)__popc(foreign) : 0u;
        uint32_t tot;
        const uint32_t ex = mf_wave_excl_scan(cnt, &tot);
        if (!tot) return;                                                // wave-uniform
        if (qcur + tot > qend) {
            void_rest();
            uint32_t base = 0;
            const uint32_t ch = tot > DCC_QCHUNK ? tot : DCC_QCHUNK;
            if (mf_lane() == 0) base = atomicAdd(qcount, ch);
            qcur = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            qend = qcur + ch;
        }
        uint32_t at = qcur + ex;
        qcur += tot;
        if (cnt) {
            uint32_t m_nf = 0, m_nl = 0;
            mf_skm_nbr_mins(x, k, &m_nf, &m_nl);
#pragma unroll
            for (uint32_t i = 0; i < 8; i++) {
                if (!((foreign >> i) & 1u)) continue;
                uint64_t y; uint32_t ph;
                const uint64_t c = nb_neighbour(x, k, kmask, i, m_nf, m_nl, &y, &ph);
                if (at < qcap) { q[at].key = c; q[at].src = (v * 8 + i) | ((uint64_t)(ph >> (32 - lw)) << 56); }
                at++;
            }
        }
    });
    void_rest();
}
// queries by owner: counts, then the grouped copy (a workgroup reserves its share of every owner's range with one atomic each)
template <bool FILL>
__global__ __launch_bounds__(256) void k_dcc_q_group(const dcc_query *__restrict__ q, uint32_t n, unsigned long long *__restrict__ cur, dcc_query *__restrict__ out) {
    __shared__ uint32_t cnt[64]; __shared__ unsigned long long base[64];
    if (threadIdx.x < 64) cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t per = 256 * 8, i0 = blockIdx.x * per;
    uint32_t o[8], pos[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t i = i0 + (uint32_t)j * 256u + threadIdx.x;
        o[j] = 0xFFFFFFFFu;
        if (i < n) { o[j] = (uint32_t)(q[i].src >> 56); if (o[j] < 64u) pos[j] = atomicAdd(&cnt[o[j]], 1u); else o[j] = 0xFFFFFFFFu; }      // (>= 64: void)
    }
    __syncthreads();
    if (threadIdx.x < 64 && cnt[threadIdx.x]) base[threadIdx.x] = atomicAdd(&cur[threadIdx.x], (unsigned long long)cnt[threadIdx.x]);
    if (!FILL) return;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t i = i0 + (uint32_t)j * 256u + threadIdx.x;
        if (o[j] == 0xFFFFFFFFu) continue;
        dcc_query r = q[i];
        r.src &= DCC_SRC_MASK;
        out[base[o[j]] + pos[j]] = r;
    }
}
struct dcc_answer { uint64_t src; uint32_t lid; uint32_t val; };
__global__ void k_dcc_answer(mf_index_view ix, const dcc_query *__restrict__ q, uint64_t n, dcc_answer *__restrict__ a) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t idx, val;
    const bool f = ix.slots && mf_index_find(ix, q[i].key, &idx, &val);
    a[i].src = q[i].src; a[i].lid = f ? idx : CC_NONE; a[i].val = f ? val : 0u;
}
// Every thread of the workgroup calls it; the threads with `act` get distinct positions from cur[r] (r < 64): the workgroup
// counts per rank in LDS and takes its share of every rank's range with one atomic each, instead of one atomic per lane (or
// wave) on a handful of addresses (~12 ns each, serialised).  L: 64 + 64 * 2 uint32 of LDS.
__device__ __forceinline__ uint64_t dcc_reserve_by(unsigned long long *cur, uint32_t r, bool act, uint32_t *L) {
    unsigned long long *base = reinterpret_cast<unsigned long long *>(L + 64);
    __syncthreads();
    if (threadIdx.x < 64) L[threadIdx.x] = 0;
    __syncthreads();
    uint32_t pos = 0;
    if (act) pos = atomicAdd(&L[r], 1u);
    __syncthreads();
    if (threadIdx.x < 64 && L[threadIdx.x]) base[threadIdx.x] = atomicAdd(&cur[threadIdx.x], (unsigned long long)L[threadIdx.x]);
    __syncthreads();
    return act ? base[r] + pos : 0ull;
}
// answers (grouped by answering rank, aoff[]) -> cross edges, grouped by rank too: cur[r] counts (xv == nullptr) or is the
// cursor of rank r's range
__global__ void k_dcc_cross(const dcc_answer *__restrict__ a, uint64_t n, const uint64_t *__restrict__ aoff, int world, unsigned long long *__restrict__ cur,
                            uint32_t *__restrict__ xv, uint32_t *__restrict__ xu, uint8_t *__restrict__ xr, uint16_t *__restrict__ xval) {
    __shared__ uint32_t L[192];
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool act = i < n && a[i].lid != CC_NONE;
    uint32_t r = 0;
    if (act) while ((int)r + 1 < world && i >= aoff[r + 1]) r++;
    const uint64_t at = dcc_reserve_by(cur, r, act, L);
    if (act && xv) { xv[at] = (uint32_t)(a[i].src >> 3); xu[at] = a[i].lid; xr[at] = (uint8_t)r; xval[at] = (uint16_t)a[i].val; }
}

__global__ void k_dcc_fill64(unsigned long long *__restrict__ p, uint64_t n, unsigned long long v) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
extern "C" int mf_dcc_world(const mf_dcc *D) { return D ? D->world : mf_set_error("mf_dcc_world: NULL handle"); }
extern "C" void mf_dcc_destroy(mf_dcc *d) { delete d; }
// shard: this rank's part of the cutter table; base[0..world]: global id of every rank's first vertex
extern "C" int mf_dcc_create(mf_ctx *ctx, mf_table *shard, int rank, int world, const uint32_t *base, mf_dcc **out) {
    if (!ctx || !shard || !base || !out) return mf_set_error("mf_dcc_create: NULL argument");
    *out = nullptr;
    if (world < 1 || world > 64 || (world & (world - 1))) return mf_set_error("mf_dcc_create: the world size must be a power of two <= 64");
    if (base[world] != 0 && (uint64_t)base[rank] + shard->n != base[rank + 1]) return mf_set_error("mf_dcc_create: base[] does not match the shard");
    MF_HIP(hipSetDevice(ctx->device));
    std::unique_ptr<mf_dcc> D(new mf_dcc());
    D->ctx = ctx; D->t = shard; D->k = shard->k; D->rank = rank; D->world = world; D->lw = dcc_log2(world);
    D->n = (uint32_t)shard->n; D->n_total = base[world]; D->base.assign(base, base + world + 1);
    const size_t n1 = D->n ? D->n : 1;
    MF_TRY(D->nbr.alloc(ctx, n1 * 8)); MF_TRY(D->parent.alloc(ctx, n1)); MF_TRY(D->root.alloc(ctx, n1)); MF_TRY(D->csize.alloc(ctx, n1));
    D->ecap = n1 / 2 + 1024; MF_TRY(D->edges.alloc(ctx, D->ecap)); MF_TRY(D->ecount.alloc(ctx, 2));
    MF_TRY(D->groot.alloc(ctx, n1)); MF_TRY(D->cweight.alloc(ctx, n1)); MF_TRY(D->alive.alloc(ctx, n1)); MF_TRY(D->ctr.alloc(ctx, 64));
    MF_TRY(D->qcur.alloc(ctx, 64)); MF_TRY(D->mk.alloc(ctx, n1)); MF_TRY(D->mg.alloc(ctx, n1)); MF_TRY(D->touched.alloc(ctx, n1));
    D->xstart.assign(world + 1, 0);
    if (D->n && !(shard->part_skm && shard->part_bits >= D->lw && shard->d_part_off))
        return mf_set_error("mf_dcc_create: the shard must come from mf_count_device_shard (minimizer partitions)");
    if (D->n) {                                                           // ... and hold this rank's partitions only
        const uint32_t p_lo = (uint32_t)rank << (shard->part_bits - D->lw), p_hi = ((uint32_t)rank + 1u) << (shard->part_bits - D->lw);
        uint64_t lo = 0, hi = 0;
        MF_HIP(hipMemcpy(&lo, shard->d_part_off + p_lo, 8, hipMemcpyDeviceToHost));
        MF_HIP(hipMemcpy(&hi, shard->d_part_off + p_hi, 8, hipMemcpyDeviceToHost));
        if (lo != 0 || hi != shard->n) return mf_set_error("mf_dcc_create: the table holds k-mers of other ranks' partitions (not rank %d's shard of %d)", rank, world);
    }
    const size_t nt = D->n_total ? D->n_total : 1;
    MF_TRY(D->pg.alloc(ctx, nt)); MF_TRY(D->gsize.alloc(ctx, nt)); MF_TRY(D->gweight.alloc(ctx, nt)); MF_TRY(D->gmin.alloc(ctx, nt));
    { mf_ktimer tm_(ctx, "k_dcc_fill64"); k_dcc_fill64<<<cgrid(nt), 256, 0, ctx->stream>>>(D->gmin.p, nt, DCC_NOKEY); }
    if (D->n) MF_TRY(mf_table_ensure_index(shard));
    MF_HIP(hipMemsetAsync(D->alive.p, 1, n1, ctx->stream));
    MF_HIP(hipMemsetAsync(D->ctr.p, 0, D->ctr.bytes(), ctx->stream));
    *out = D.release();
    return MF_OK;
}
// step 1: own neighbours looked up; counts[o] = queries for rank o
extern "C" int mf_dcc_queries(mf_dcc *D, uint64_t *counts) {
    if (!D || !counts) return mf_set_error("mf_dcc_queries: NULL argument");
    mf_ctx *ctx = D->ctx; hipStream_t st = ctx->stream;
    MF_HIP(hipSetDevice(ctx->device));
    D->nq = 0;
    if (D->n) {
        mf_table *t = D->t;
        const uint32_t p_lo = (uint32_t)D->rank << (t->part_bits - D->lw), p_hi = ((uint32_t)D->rank + 1u) << (t->part_bits - D->lw);
        const unsigned grid = (unsigned)std::min<uint64_t>((p_hi - p_lo + NB_WAVES - 1) / NB_WAVES, (uint64_t)ctx->n_cu * 16);
        uint64_t cap = std::min<uint64_t>((uint64_t)D->n + (D->n >> 1) + 2ull * grid * NB_WAVES * DCC_QCHUNK, 0xFFFFFFF0ull);
        for (int attempt = 0; attempt < 2; attempt++) {
            MF_TRY(D->qflat.alloc(ctx, cap));
            MF_HIP(hipMemsetAsync(&D->ctr.p[8], 0, 4, st));
            {
                mf_ktimer tm(ctx, "k_cc_adjacency");
                k_dcc_adjacency_part<1><<<grid, 64 * NB_WAVES, 0, st>>>(mf_view(t->index), t->d_keys, t->d_part_off, p_lo, p_hi, D->k, D->lw, (uint32_t)D->rank, D->nbr.p,
                                                                       D->qflat.p, (uint32_t)cap, &D->ctr.p[8]);
                k_dcc_adjacency_part<2><<<grid, 64 * NB_WAVES, 0, st>>>(mf_view(t->index), t->d_keys, t->d_part_off, p_lo, p_hi, D->k, D->lw, (uint32_t)D->rank, D->nbr.p,
                                                                       D->qflat.p, (uint32_t)cap, &D->ctr.p[8]);
            }
            unsigned int c = 0;
            MF_HIP(hipMemcpyAsync(&c, &D->ctr.p[8], 4, hipMemcpyDeviceToHost, st));
            MF_HIP(hipStreamSynchronize(st));
            D->nq = c;
            if (c <= cap) break;
            if (attempt) return mf_set_error("mf_dcc_queries: query list overflow");
            cap = std::min<uint64_t>((uint64_t)c + (c >> 3) + 2ull * grid * NB_WAVES * DCC_QCHUNK, 0xFFFFFFF0ull);   // (more than 1.5 foreign neighbours per k-mer: once more with room for what was asked for)
        }
    }
    MF_HIP(hipMemsetAsync(D->qcur.p, 0, D->qcur.bytes(), st));
    if (D->nq) { mf_ktimer tm_(ctx, "k_dcc_q_group"); k_dcc_q_group<false><<<cgrid(D->nq, 2048), 256, 0, st>>>(D->qflat.p, D->nq, D->qcur.p, nullptr); }
    std::vector<unsigned long long> h(64);
    MF_HIP(hipMemcpyAsync(h.data(), D->qcur.p, 64 * 8, hipMemcpyDeviceToHost, st));
    MF_HIP(hipStreamSynchronize(st));
    D->qoff.assign(D->world + 1, 0);
    for (int o = 0; o < D->world; o++) { counts[o] = h[o]; D->qoff[o + 1] = D->qoff[o] + h[o]; }
    return MF_OK;
}
// step 2: the queries themselves, grouped by owner, 16 bytes each, into d_q (room for sum(counts))
extern "C" int mf_dcc_queries_fill(mf_dcc *D, void *d_q) {
    if (!D || (D->qoff.back() && !d_q)) return mf_set_error("mf_dcc_queries_fill: NULL argument");
    mf_ctx *ctx = D->ctx; hipStream_t st = ctx->stream;
    MF_HIP(hipSetDevice(ctx->device));
    std::vector<unsigned long long> c0(64, 0);
    for (int o = 0; o < D->world; o++) c0[o] = D->qoff[o];
    MF_HIP(hipMemcpyAsync(D->qcur.p, c0.data(), 64 * 8, hipMemcpyHostToDevice, st));
    if (D->nq) { mf_ktimer tm_(ctx, "k_dcc_q_group"); k_dcc_q_group<true><<<cgrid(D->nq, 2048), 256, 0, st>>>(D->qflat.p, D->nq, D->qcur.p, (dcc_query *)d_q); }
    MF_HIP(hipStreamSynchronize(st));
    D->qflat.reset();
    return MF_OK;
}
// step 3 (owner side): n queries -> n answers (16 bytes each)
extern "C" int mf_dcc_answer(mf_dcc *D, const void *d_q, uint64_t n, void *d_a) {
    if (!D || (n && (!d_q || !d_a))) return mf_set_error("mf_dcc_answer: NULL argument");
    mf_ctx *ctx = D->ctx;
    MF_HIP(hipSetDevice(ctx->device));
    mf_index_view ix = mf_view(D->t->index);
    if (!D->n) ix.slots = nullptr;
    if (n) { mf_ktimer tm_(ctx, "k_dcc_answer"); k_dcc_answer<<<cgrid(n), 256, 0, ctx->stream>>>(ix, (const dcc_query *)d_q, n, (dcc_answer *)d_a); }
    MF_HIP(hipStreamSynchronize(ctx->stream));
    return MF_OK;
}
// step 4: the answers to this rank's queries (grouped by answering rank like the queries were) -> cross edges
extern "C" int mf_dcc_set_answers(mf_dcc *D, const void *d_a, uint64_t n) {
    if (!D || (n && !d_a)) return mf_set_error("mf_dcc_set_answers: NULL argument");
    if (n != D->qoff.back()) return mf_set_error("mf_dcc_set_answers: %llu answers for %llu queries", (unsigned long long)n, (unsigned long long)D->qoff.back());
    mf_ctx *ctx = D->ctx; hipStream_t st = ctx->stream;
    MF_HIP(hipSetDevice(ctx->device));
    mf_buf<uint64_t> aoff; MF_TRY(aoff.alloc(ctx, (size_t)D->world + 1));
    MF_HIP(hipMemcpyAsync(aoff.p, D->qoff.data(), ((size_t)D->world + 1) * 8, hipMemcpyHostToDevice, st));
    MF_HIP(hipMemsetAsync(D->qcur.p, 0, D->qcur.bytes(), st));
    std::vector<unsigned long long> h(64, 0);
    if (n) {
        { mf_ktimer tm_(ctx, "k_dcc_cross"); k_dcc_cross<<<cgrid(n), 256, 0, st>>>((const dcc_answer *)d_a, n, aoff.p, D->world, D->qcur.p, nullptr, nullptr, nullptr, nullptr); }
        MF_HIP(hipMemcpyAsync(h.data(), D->qcur.p, 64 * 8, hipMemcpyDeviceToHost, st));
        MF_HIP(hipStreamSynchronize(st));
    }
    std::vector<unsigned long long> c0(64, 0);
    uint64_t nx = 0;
    for (int o = 0; o < D->world; o++) { c0[o] = nx; D->xstart[o] = nx; nx += h[o]; }
    D->xstart[D->world] = nx;
    if (nx >= 0xFFFFFFFFull) return mf_set_error("mf_dcc_set_answers: too many cross edges");
    D->nx = nx;
    const size_t n1 = nx ? nx : 1;
    MF_TRY(D->hp.alloc(ctx, n1));
    MF_TRY(D->xv.alloc(ctx, n1)); MF_TRY(D->xu.alloc(ctx, n1)); MF_TRY(D->xr.alloc(ctx, n1)); MF_TRY(D->xval.alloc(ctx, n1)); MF_TRY(D->xalive.alloc(ctx, n1));
    if (nx) {
        MF_HIP(hipMemcpyAsync(D->qcur.p, c0.data(), 64 * 8, hipMemcpyHostToDevice, st));
        { mf_ktimer tm_(ctx, "k_dcc_cross"); k_dcc_cross<<<cgrid(n), 256, 0, st>>>((const dcc_answer *)d_a, n, aoff.p, D->world, D->qcur.p, D->xv.p, D->xu.p, D->xr.p, D->xval.p); }
        MF_HIP(hipMemsetAsync(D->xalive.p, 1, nx, st));
    }
    MF_HIP(hipStreamSynchronize(st));
    return MF_OK;
}

// ---- one threshold level
// (a) union-find inside the shard; counts[o] = root pairs this rank asks rank o to complete (edges to higher ranks only).
//     The cross edges are grouped by rank, so the half pairs of rank r are compacted inside r's range of the edge list.
__global__ __launch_bounds__(256) void k_dcc_pairs_out(const uint32_t *__restrict__ xv, const uint32_t *__restrict__ xu, const uint8_t *__restrict__ xr,
                                                       const uint8_t *__restrict__ xalive, uint64_t nx, const uint8_t *__restrict__ alive, const uint32_t *__restrict__ root,
                                                       uint32_t me, uint32_t mybase, unsigned long long *__restrict__ cur, uint2 *__restrict__ out) {
    __shared__ uint32_t L[192];
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t r = 0; bool act = false;
    if (e < nx && xalive[e]) { r = xr[e]; act = r > me && alive[xv[e]]; }
    const uint64_t at = dcc_reserve_by(cur, r, act, L);
    if (act) out[at] = make_uint2(xu[e], mybase + root[xv[e]]);
}
extern "C" int mf_dcc_level_local(mf_dcc *D, uint64_t *counts) {
    if (!D || !counts) return mf_set_error("mf_dcc_level_local: NULL argument");
    mf_ctx *ctx = D->ctx; hipStream_t st = ctx->stream;
    if (ctx->opt_dcc_test_fail / 1000 == 2 && ++ctx->dcc_test_calls[2] == ctx->opt_dcc_test_fail % 1000) return mf_set_error("injected failure (option dcc_test_fail)");
    MF_HIP(hipSetDevice(ctx->device));
    const uint64_t n = D->n;
    if (n) {
        mf_ktimer tm(ctx, "k_cc_hook");
        MF_HIP(hipMemsetAsync(D->ecount.p, 0, 16, st));
        k_cc_hook_tile<<<cgrid(n, CC_TILE), 256, 0, st>>>(D->nbr.p, D->alive.p, D->parent.p, D->csize.p, D->cweight.p, n, D->edges.p, D->ecount.p, D->ecap);
        k_cc_hook_edges<<<cgrid(D->ecap), 256, 0, st>>>(D->edges.p, D->ecount.p, D->ecap, D->alive.p, D->parent.p);
        k_cc_hook<<<cgrid(n), 256, 0, st>>>(D->nbr.p, D->alive.p, D->parent.p, n, nullptr, D->ecount.p);      // (only if the list overflowed)
    }
    if (n) {
        mf_ktimer tm(ctx, "k_cc_stats");
        k_cc_flatten_stats<<<cgrid(n, CC_TILE), 256, 0, st>>>(D->alive.p, D->parent.p, D->root.p, D->t->d_counts, D->csize.p, D->cweight.p, n);
    }
    std::vector<unsigned long long> h(64, 0);
    for (int o = 0; o < D->world; o++) h[o] = D->xstart[o];
    MF_HIP(hipMemcpyAsync(D->qcur.p, h.data(), 64 * 8, hipMemcpyHostToDevice, st));
    if (D->nx) {
        mf_ktimer tm_(ctx, "k_dcc_pairs_out");
        k_dcc_pairs_out<<<cgrid(D->nx), 256, 0, st>>>(D->xv.p, D->xu.p, D->xr.p, D->xalive.p, D->nx, D->alive.p, D->root.p, (uint32_t)D->rank, D->base[D->rank], D->qcur.p, D->hp.p);
    }
    MF_HIP(hipMemcpyAsync(h.data(), D->qcur.p, 64 * 8, hipMemcpyDeviceToHost, st));
    MF_HIP(hipStreamSynchronize(st));
    D->xoff.assign(D->world + 1, 0);
    for (int o = 0; o < D->world; o++) { counts[o] = h[o] - D->xstart[o]; D->xoff[o + 1] = D->xoff[o] + counts[o]; }
    return MF_OK;
}
// (b) the half pairs (remote vertex, global id of this end's root), 8 bytes each, grouped by rank
extern "C" int mf_dcc_pairs_fill(mf_dcc *D, void *d_out) {
    if (!D || (D->xoff.back() && !d_out)) return mf_set_error("mf_dcc_pairs_fill: NULL argument");
    mf_ctx *ctx = D->ctx; hipStream_t st = ctx->stream;
    MF_HIP(hipSetDevice(ctx->device));
    for (int o = 0; o < D->world; o++)
        if (D->xoff[o + 1] > D->xoff[o])
            MF_HIP(hipMemcpyAsync((uint2 *)d_out + D->xoff[o], D->hp.p + D->xstart[o], (D->xoff[o + 1] - D->xoff[o]) * 8, hipMemcpyDeviceToDevice, st));
    MF_HIP(hipStreamSynchronize(st));
    return MF_OK;
}
// (round 5) exchanges of a fixed capacity fill their unused room with the pair (DCC_NO_PAIR, DCC_NO_PAIR), which no kernel takes: the sizes of
// a level's exchanges then need no integer gather (metafast_amd/pipeline.py: distributed_components).  Vertex ids stay below 2^32 - 1.
#define DCC_NO_PAIR 0xFFFFFFFFu
// (c) owner side: (own vertex u, other root) -> (own root of u, other root), in place
__global__ void k_dcc_pairs_complete(uint2 *__restrict__ p, uint64_t n, const uint32_t *__restrict__ root, uint32_t mybase) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && p[i].x != DCC_NO_PAIR) p[i].x = mybase + root[p[i].x];
}
extern "C" int mf_dcc_pairs_complete(mf_dcc *D, void *d_pairs, uint64_t n) {
    if (!D || (n && !d_pairs)) return mf_set_error("mf_dcc_pairs_complete: NULL argument");
    MF_HIP(hipSetDevice(D->ctx->device));
    if (n) { mf_ktimer tm_(D->ctx, "k_dcc_pairs_complete"); k_dcc_pairs_complete<<<cgrid(n), 256, 0, D->ctx->stream>>>((uint2 *)d_pairs, n, D->root.p, D->base[D->rank]); }
    MF_HIP(hipStreamSynchronize(D->ctx->stream));
    return MF_OK;
}
// (d) all ranks' pairs -> union-find over fragment roots (the same on every rank) -> the global root of every own fragment.
//     The arrays over all global ids (pg, gsize, gweight) are set up in full at the first level only; later levels, with a
//     few per cent of the vertices left, touch the ids that occur: the ends of the pairs and this rank's own fragment roots.
__global__ void k_dcc_iota(uint32_t *__restrict__ p, uint32_t *__restrict__ gs, unsigned long long *__restrict__ gw, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { p[i] = (uint32_t)i; gs[i] = 0; gw[i] = 0; }
}
__global__ void k_dcc_init_pairs(const uint2 *__restrict__ pr, uint64_t n, uint32_t *__restrict__ p, uint32_t *__restrict__ gs, unsigned long long *__restrict__ gw) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || pr[i].x == DCC_NO_PAIR) return;
    const uint32_t a = pr[i].x, b = pr[i].y;
    p[a] = a; gs[a] = 0; gw[a] = 0;
    p[b] = b; gs[b] = 0; gw[b] = 0;
}
__global__ void k_dcc_init_roots(const uint8_t *__restrict__ alive, const uint32_t *__restrict__ root, uint32_t n, uint32_t mybase, uint32_t *__restrict__ p,
                                 uint32_t *__restrict__ gs, unsigned long long *__restrict__ gw) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n || !alive[v] || root[v] != v) return;
    const uint32_t g = mybase + v;
    p[g] = g; gs[g] = 0; gw[g] = 0;
}
__global__ void k_dcc_hook_pairs(const uint2 *__restrict__ pr, uint64_t n, uint32_t *parent) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || pr[i].x == DCC_NO_PAIR) return;
    uint32_t ra = pr[i].x, rb = pr[i].y;
    for (;;) {
        ra = cc_find(parent, ra); rb = cc_find(parent, rb);
        if (ra == rb) break;
        if (ra < rb) { const uint32_t t = ra; ra = rb; rb = t; }
        if (atomicCAS(&parent[ra], ra, rb) == ra) break;
    }
}
// own fragments: global root; their sizes / weights are summed per global root HERE first (a large component has millions
// of fragments on every rank: one record per rank and component travels instead of one per fragment).  touched[]: the
// global roots this rank contributes to (the thread that finds the sum at zero lists the root).
struct dcc_stat { uint32_t g, size; unsigned long long weight; };
__global__ __launch_bounds__(256) void k_dcc_groots(const uint8_t *__restrict__ alive, const uint32_t *__restrict__ root, uint32_t n, const uint32_t *__restrict__ pg,
                                                    uint32_t mybase, uint32_t *__restrict__ groot, const uint32_t *__restrict__ csize,
                                                    const unsigned long long *__restrict__ cweight, uint32_t *__restrict__ gsize, unsigned long long *__restrict__ gweight,
                                                    unsigned int *__restrict__ cnt, uint32_t *__restrict__ touched) {
    __shared__ uint32_t scratch[18];
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    const bool is_root = v < n && alive[v] && root[v] == v;
    uint32_t g = 0; bool first = false;
    if (is_root) {
        g = mybase + v;
        for (;;) { const uint32_t p = pg[g]; if (p == g) break; g = p; }
        groot[v] = g;
        first = atomicAdd(&gsize[g], csize[v]) == 0u;                    // (a fragment has at least one vertex)
        atomicAdd(&gweight[g], cweight[v]);
    }
    const uint32_t at = mf_block_reserve(cnt, first ? 1u : 0u, scratch);
    if (first) touched[at] = g;
}
__global__ void k_dcc_stats_emit(const uint32_t *__restrict__ touched, uint32_t n, const uint32_t *__restrict__ gsize, const unsigned long long *__restrict__ gweight,
                                 dcc_stat *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = touched[i];
    out[i].g = g; out[i].size = gsize[g]; out[i].weight = gweight[g];
}
extern "C" int mf_dcc_merge(mf_dcc *D, const void *d_pairs, uint64_t n, uint64_t *n_stats) {
    if (!D || !n_stats || (n && !d_pairs)) return mf_set_error("mf_dcc_merge: NULL argument");
    mf_ctx *ctx = D->ctx; hipStream_t st = ctx->stream;
    if (ctx->opt_dcc_test_fail / 1000 == 1 && ++ctx->dcc_test_calls[1] == ctx->opt_dcc_test_fail % 1000) return mf_set_error("injected failure (option dcc_test_fail)");
    MF_HIP(hipSetDevice(ctx->device));
    if (D->level == 0 || (n * 48 > D->n_total && !ctx->opt_dcc_sparse)) {                           // (random writes: only worth it when few ids occur)
        if (D->n_total) { mf_ktimer tm_(ctx, "k_dcc_iota"); k_dcc_iota<<<cgrid(D->n_total), 256, 0, st>>>(D->pg.p, D->gsize.p, D->gweight.p, D->n_total); }
    } else {
        mf_ktimer tm_(ctx, "k_dcc_iota");
        if (n) k_dcc_init_pairs<<<cgrid(n), 256, 0, st>>>((const uint2 *)d_pairs, n, D->pg.p, D->gsize.p, D->gweight.p);
        if (D->n) k_dcc_init_roots<<<cgrid(D->n), 256, 0, st>>>(D->alive.p, D->root.p, D->n, D->base[D->rank], D->pg.p, D->gsize.p, D->gweight.p);
    }
    D->level++;
    if (n) { mf_ktimer tm(ctx, "k_dcc_hook_pairs"); k_dcc_hook_pairs<<<cgrid(n), 256, 0, st>>>((const uint2 *)d_pairs, n, D->pg.p); }
    MF_HIP(hipMemsetAsync(&D->ctr.p[1], 0, 4, st));
    if (D->n) {
        mf_ktimer tm_(ctx, "k_dcc_groots");
        k_dcc_groots<<<cgrid(D->n), 256, 0, st>>>(D->alive.p, D->root.p, D->n, D->pg.p, D->base[D->rank], D->groot.p, D->csize.p, D->cweight.p, D->gsize.p, D->gweight.p,
                                                  &D->ctr.p[1], D->touched.p);
    }
    unsigned int c = 0;
    MF_HIP(hipMemcpyAsync(&c, &D->ctr.p[1], 4, hipMemcpyDeviceToHost, st));
    MF_HIP(hipStreamSynchronize(st));
    D->nstat = c; *n_stats = c;
    return MF_OK;
}
// (e) this rank's share of the components it touches: (global root, size, weight), 16 bytes each
extern "C" int mf_dcc_stats_fill(mf_dcc *D, void *d_out) {
    if (!D || (D->nstat && !d_out)) return mf_set_error("mf_dcc_stats_fill: NULL argument");
    mf_ctx *ctx = D->ctx; hipStream_t st = ctx->stream;
    MF_HIP(hipSetDevice(ctx->device));
    if (D->nstat) k_dcc_stats_emit<<<cgrid(D->nstat), 256, 0, st>>>(D->touched.p, (uint32_t)D->nstat, D->gsize.p, D->gweight.p, (dcc_stat *)d_out);
    MF_HIP(hipStreamSynchronize(st));
    return MF_OK;
}
// (f) the other ranks' shares -> size / weight per global root complete; classify; members of kept components and the next
//     level's alive set
__global__ void k_dcc_accumulate(const dcc_stat *__restrict__ s, uint64_t n, uint64_t skip_lo, uint64_t skip_hi, uint32_t *__restrict__ gsize,
                                 unsigned long long *__restrict__ gweight) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || (i >= skip_lo && i < skip_hi)) return;
    atomicAdd(&gsize[s[i].g], s[i].size);
    atomicAdd(&gweight[s[i].g], s[i].weight);
}
// gmin[g]: smallest member k-mer of the kept component with global root g among THIS rank's vertices (the ranks' minima
// are combined at the end, mf_dcc_minkeys).
__global__ __launch_bounds__(256) void k_dcc_apply(uint8_t *__restrict__ alive, const uint32_t *__restrict__ root, const uint32_t *__restrict__ groot,
                                                   const uint32_t *__restrict__ gsize, const unsigned long long *__restrict__ gweight, const uint16_t *__restrict__ vals,
                                                   const uint64_t *__restrict__ keys, uint32_t n, uint32_t mybase, uint32_t b1, uint32_t b2, uint32_t next_thr,
                                                   unsigned int *__restrict__ cnt /* [2] members */, uint64_t *__restrict__ mk,
                                                   uint32_t *__restrict__ mg, unsigned long long *__restrict__ gmin) {
    __shared__ uint32_t scratch[18];
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    bool put = false; uint32_t g = 0; uint64_t key = 0;
    if (v < n && alive[v]) {
        const uint32_t r = root[v];
        g = groot[r];
        const uint32_t s = gsize[g];
        if (s > b2) { if ((uint32_t)vals[v] < next_thr) alive[v] = 0; }
        else { alive[v] = 0; put = s >= b1; }
    }
    const uint32_t at = mf_block_reserve(&cnt[2], put ? 1u : 0u, scratch);
    if (put) { key = keys[v]; mk[at] = key; mg[at] = g; }
    // smallest k-mer per component: neighbours in the table mostly share their component -> one atomic per distinct root and wave
    unsigned long long todo = __ballot(put);
    for (int round = 0; round < 4 && todo; round++) {                   // wave-uniform
        const int lead = __ffsll((long long)todo) - 1;
        const uint32_t g0 = (uint32_t)__builtin_amdgcn_readlane((int)g, lead);
        const bool in = put && g == g0;
        const unsigned long long grp = __ballot(in);
        uint64_t m = in ? key : ~0ull;
        for (int d = 32; d >= 1; d >>= 1) { const uint64_t o = __shfl_xor(m, d, 64); m = o < m ? o : m; }
        if ((int)mf_lane() == lead) atomicMin(&gmin[g0], (unsigned long long)m);
        if (in) put = false;
        todo &= ~grp;
    }
    if (put) atomicMin(&gmin[g], (unsigned long long)key);
}
// EVERY rank lists ALL kept components of the level (and counts the oversize ones) from the gathered records: a component's
// records arrive from every rank that touches it; the one from the rank that owns the global root -- it always has one, the
// root vertex itself is in the component -- stands for the component.  (Round 2 had the owner report its components and
// all-gathered those lists and their lengths: two collectives and a host round trip per level for data every rank already holds.)
__global__ void k_dcc_report(const dcc_stat *__restrict__ s, uint64_t n, const uint64_t *__restrict__ seg, const uint32_t *__restrict__ base, int world,
                             const uint32_t *__restrict__ gsize, const unsigned long long *__restrict__ gweight, uint32_t b1, uint32_t b2,
                             unsigned int *__restrict__ cnt /* [0] kept [1] big */, dcc_kept_rec *__restrict__ kept) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int r = 0;
    while (r + 1 < world && i >= seg[r + 1]) r++;                          // (world <= 64)
    const uint32_t g = s[i].g;
    if (g < base[r] || g >= base[r + 1]) return;
    const uint32_t sz = gsize[g];
    if (sz < b1) return;
    if (sz <= b2) { const uint32_t at = atomicAdd(&cnt[0], 1u); kept[at].g = g; kept[at].size = sz; kept[at].weight = gweight[g]; }
    else atomicAdd(&cnt[1], 1u);
}
// size / weight of every root the level's records name go back to zero, the other ranks' roots included: the next level's sparse
// set-up (mf_dcc_merge) resets only the ends of ITS pairs and this rank's own fragment roots, and a component that by then lives on
// one other rank without a cross edge would be added on top of what its root summed up at this level (k_dcc_report reads every rank's)
__global__ void k_dcc_clear(const dcc_stat *__restrict__ s, uint64_t n, uint32_t *__restrict__ gsize, unsigned long long *__restrict__ gweight) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { gsize[s[i].g] = 0; gweight[s[i].g] = 0; }
}
__global__ void k_dcc_cross_next(uint8_t *__restrict__ xalive, const uint32_t *__restrict__ xv, const uint16_t *__restrict__ xval, uint64_t nx,
                                 const uint8_t *__restrict__ alive_after, uint32_t next_thr) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nx || !xalive[e]) return;
    // this end survived (its component is too large and its value reaches the next threshold), and so does the other end
    // if its value does: it is in the same component
    if (!alive_after[xv[e]] || (uint32_t)xval[e] < next_thr) xalive[e] = 0;
}
// all ranks' records (n of them, in rank order: rank r's are [seg_first[r], seg_first[r + 1]); [own_first, own_first + own_n) are
// this rank's own, already counted) -> size / weight per global root; every vertex of this rank is classified: members of kept
// components are remembered (k-mer, global root), vertices of oversize components whose value reaches thr + 1 stay alive.
// *n_kept / *n_big: the kept / oversize components of the level over ALL ranks -- the same numbers and, up to order, the same list
// (mf_dcc_kept_fill) on every rank.
extern "C" int mf_dcc_classify(mf_dcc *D, const void *d_stats, uint64_t n, const uint64_t *seg_first, uint64_t own_first, uint64_t own_n, int b1, int b2,
                               int thr, uint64_t *n_kept, uint64_t *n_big) {
    if (!D || !n_kept || !n_big || !seg_first || (n && !d_stats)) return mf_set_error("mf_dcc_classify: NULL argument");
    if (own_n != D->nstat || own_first + own_n > n) return mf_set_error("mf_dcc_classify: this rank's own records are %llu, not %llu", (unsigned long long)D->nstat, (unsigned long long)own_n);
    if (seg_first[0] != 0 || seg_first[D->world] != n || seg_first[D->rank] != own_first) return mf_set_error("mf_dcc_classify: seg_first does not describe %llu records", (unsigned long long)n);
    mf_ctx *ctx = D->ctx; hipStream_t st = ctx->stream;
    MF_HIP(hipSetDevice(ctx->device));
    if (b1 < 0) b1 = 0;
    if (n > own_n) { mf_ktimer tm_(ctx, "k_dcc_accumulate"); k_dcc_accumulate<<<cgrid(n), 256, 0, st>>>((const dcc_stat *)d_stats, n, own_first, own_first + own_n, D->gsize.p, D->gweight.p); }
    MF_HIP(hipMemsetAsync(&D->ctr.p[2], 0, 8, st));
    if (D->keptbuf.n < n + 1) MF_TRY(D->keptbuf.alloc(ctx, n + (n >> 2) + 1024));                 // (at most one component per record)
    if (n) {
        if (!D->dseg.p) { MF_TRY(D->dseg.alloc(ctx, (size_t)D->world + 1)); MF_TRY(D->dbase.alloc(ctx, (size_t)D->world + 1));
                          MF_HIP(hipMemcpyAsync(D->dbase.p, D->base.data(), ((size_t)D->world + 1) * 4, hipMemcpyHostToDevice, st)); }
        MF_HIP(hipMemcpyAsync(D->dseg.p, seg_first, ((size_t)D->world + 1) * 8, hipMemcpyHostToDevice, st));
        mf_ktimer tm_(ctx, "k_dcc_report");
        k_dcc_report<<<cgrid(n), 256, 0, st>>>((const dcc_stat *)d_stats, n, D->dseg.p, D->dbase.p, D->world, D->gsize.p, D->gweight.p, (uint32_t)b1, (uint32_t)b2,
                                               &D->ctr.p[2], D->keptbuf.p);
    }
    // (every vertex becomes a member at most once: one list of n entries, the cursor ctr[4] runs on from level to level)
    if (D->n) {
        mf_ktimer tm(ctx, "k_cc_members");
        k_dcc_apply<<<cgrid(D->n), 256, 0, st>>>(D->alive.p, D->root.p, D->groot.p, D->gsize.p, D->gweight.p, D->t->d_counts, D->t->d_keys, D->n, D->base[D->rank],
                                                 (uint32_t)b1, (uint32_t)b2, (uint32_t)(thr + 1), &D->ctr.p[2], D->mk.p, D->mg.p, D->gmin.p);
    }
    if (D->nx) { mf_ktimer tm_(ctx, "k_dcc_cross_next"); k_dcc_cross_next<<<cgrid(D->nx), 256, 0, st>>>(D->xalive.p, D->xv.p, D->xval.p, D->nx, D->alive.p, (uint32_t)(thr + 1)); }
    if (n) k_dcc_clear<<<cgrid(n), 256, 0, st>>>((const dcc_stat *)d_stats, n, D->gsize.p, D->gweight.p);      // (after the last reader, k_dcc_apply)
    unsigned int c[4];
    MF_HIP(hipMemcpyAsync(c, &D->ctr.p[2], 12, hipMemcpyDeviceToHost, st));
    MF_HIP(hipStreamSynchronize(st));
    *n_kept = D->nkept_owned = c[0]; *n_big = c[1];
    D->nm = c[2];
    return MF_OK;
}
// the kept components of this level, all ranks' (every rank holds the same set, in no particular order): 16 bytes each
// (root u32, size u32, weight u64)
extern "C" int mf_dcc_kept_fill(mf_dcc *D, void *d_out) {
    if (!D || (D->nkept_owned && !d_out)) return mf_set_error("mf_dcc_kept_fill: NULL argument");
    mf_ctx *ctx = D->ctx; hipStream_t st = ctx->stream;
    MF_HIP(hipSetDevice(ctx->device));
    if (D->nkept_owned) MF_HIP(hipMemcpyAsync(d_out, D->keptbuf.p, D->nkept_owned * sizeof(dcc_kept_rec), hipMemcpyDeviceToDevice, st));
    MF_HIP(hipStreamSynchronize(st));
    return MF_OK;
}
// members of kept components among this rank's vertices, all levels: count, then (k-mer u64[], global root u32[])
extern "C" int mf_dcc_members(mf_dcc *D, uint64_t *n) {
    if (!D || !n) return mf_set_error("mf_dcc_members: NULL argument");
    *n = D->nm;
    return MF_OK;
}
extern "C" int mf_dcc_members_fill(mf_dcc *D, void *d_keys, void *d_roots) {
    if (!D || (D->nm && (!d_keys || !d_roots))) return mf_set_error("mf_dcc_members_fill: NULL argument");
    mf_ctx *ctx = D->ctx; hipStream_t st = ctx->stream;
    MF_HIP(hipSetDevice(ctx->device));
    if (D->nm) {
        MF_HIP(hipMemcpyAsync(d_keys, D->mk.p, D->nm * 8, hipMemcpyDeviceToDevice, st));
        MF_HIP(hipMemcpyAsync(d_roots, D->mg.p, D->nm * 4, hipMemcpyDeviceToDevice, st));
    }
    MF_HIP(hipStreamSynchronize(st));
    return MF_OK;
}
// The members grouped by component: 8 bytes per member on the wire (the k-mer) + one (root, count) record per component and rank,
// instead of 12 bytes per member (k-mer + root): the all-gather of the members is the largest exchange of the sharded cutter
// (2.6 GB per rank at 8 x 50 M reads).  mf_dcc_members_grouped sorts this rank's members by root; the receiver expands the runs
// of all ranks back to one root per member (mf_dcc_finish_grouped).
__global__ void k_dcc_run_flags(const uint32_t *__restrict__ g, uint64_t n, uint32_t *__restrict__ flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (i == 0 || g[i] != g[i - 1]) ? 1u : 0u;
}
__global__ void k_dcc_run_heads(const uint32_t *__restrict__ g, const uint32_t *__restrict__ flag, const uint64_t *__restrict__ idx, uint64_t n, uint64_t n_runs,
                                uint2 *__restrict__ runs, uint64_t *__restrict__ start) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flag[i]) { const uint64_t j = idx[i]; runs[j].x = g[i]; start[j] = i; }
}
__global__ void k_dcc_run_counts(const uint64_t *__restrict__ start, uint64_t n_runs, uint64_t n, uint2 *__restrict__ runs) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n_runs) runs[j].y = (uint32_t)((j + 1 < n_runs ? start[j + 1] : n) - start[j]);
}
__global__ void k_dcc_run_lens(const uint2 *__restrict__ runs, uint64_t n_runs, uint32_t *__restrict__ len) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n_runs) len[j] = runs[j].y;
}
__global__ void k_dcc_run_expand(const uint2 *__restrict__ runs, const uint64_t *__restrict__ off, uint64_t n_runs, uint32_t *__restrict__ roots) {
    // one wave per run
    const uint64_t j = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (j >= n_runs) return;
    const uint32_t g = runs[j].x, c = runs[j].y;
    const uint64_t o = off[j];
    for (uint32_t i = (uint32_t)mf_lane(); i < c; i += 64) roots[o + i] = g;
}
extern "C" int mf_dcc_members_grouped(mf_dcc *D, uint64_t *n_members, uint64_t *n_runs) {
    if (!D || !n_members || !n_runs) return mf_set_error("mf_dcc_members_grouped: NULL argument");
    mf_ctx *ctx = D->ctx; hipStream_t st = ctx->stream;
    MF_HIP(hipSetDevice(ctx->device));
    *n_members = D->nm; *n_runs = D->n_runs = 0;
    if (!D->nm) return MF_OK;
    mf_buf<uint32_t> sg, flag; mf_buf<uint64_t> idx, start, tot;
    MF_TRY(sg.alloc(ctx, D->nm)); MF_TRY(D->gmk.alloc(ctx, D->nm)); MF_TRY(flag.alloc(ctx, D->nm)); MF_TRY(idx.alloc(ctx, D->nm + 1)); MF_TRY(tot.alloc(ctx, 1));
    MF_TRY(mf_sort_u32_u64(ctx, D->mg.p, D->mk.p, D->nm, 32, sg.p, D->gmk.p));
    k_dcc_run_flags<<<cgrid(D->nm), 256, 0, st>>>(sg.p, D->nm, flag.p);
    MF_TRY(mf_scan<1>(ctx, flag.p, idx.p, D->nm, tot.p));
    uint64_t nr = 0;
    MF_HIP(hipMemcpyAsync(&nr, tot.p, 8, hipMemcpyDeviceToHost, st));
    MF_HIP(hipStreamSynchronize(st));
    MF_TRY(D->gruns.alloc(ctx, nr)); MF_TRY(start.alloc(ctx, nr));
    k_dcc_run_heads<<<cgrid(D->nm), 256, 0, st>>>(sg.p, flag.p, idx.p, D->nm, nr, D->gruns.p, start.p);
    k_dcc_run_counts<<<cgrid(nr), 256, 0, st>>>(start.p, nr, D->nm, D->gruns.p);
    MF_HIP(hipStreamSynchronize(st));
    *n_runs = D->n_runs = nr;
    return MF_OK;
}
extern "C" int mf_dcc_members_grouped_fill(mf_dcc *D, void *d_kmers, void *d_runs) {
    if (!D || (D->nm && (!d_kmers || !d_runs))) return mf_set_error("mf_dcc_members_grouped_fill: NULL argument");
    mf_ctx *ctx = D->ctx; hipStream_t st = ctx->stream;
    MF_HIP(hipSetDevice(ctx->device));
    if (D->nm) {
        if (!D->gmk.p) return mf_set_error("mf_dcc_members_grouped_fill: call mf_dcc_members_grouped first");
        MF_HIP(hipMemcpyAsync(d_kmers, D->gmk.p, D->nm * 8, hipMemcpyDeviceToDevice, st));
        MF_HIP(hipMemcpyAsync(d_runs, D->gruns.p, D->n_runs * 8, hipMemcpyDeviceToDevice, st));
    }
    MF_HIP(hipStreamSynchronize(st));
    return MF_OK;
}
// smallest member k-mer of each kept component among THIS rank's vertices (DCC_NOKEY: none here) -> d_out u64[n_kept];
// the caller takes the minimum over the ranks
__global__ void k_dcc_minkeys(const uint32_t *__restrict__ g, uint64_t n, const unsigned long long *__restrict__ gmin, unsigned long long *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = gmin[g[i]];
}
extern "C" int mf_dcc_minkeys(mf_dcc *D, const uint32_t *kept_root, uint64_t n_kept, void *d_out) {
    if (!D || (n_kept && (!kept_root || !d_out))) return mf_set_error("mf_dcc_minkeys: NULL argument");
    mf_ctx *ctx = D->ctx; hipStream_t st = ctx->stream;
    MF_HIP(hipSetDevice(ctx->device));
    for (uint64_t i = 0; i < n_kept; i++) if (kept_root[i] >= D->n_total) return mf_set_error("mf_dcc_minkeys: component root out of range");
    if (n_kept) {
        mf_buf<uint32_t> d_g; MF_TRY(d_g.alloc(ctx, n_kept));
        MF_HIP(hipMemcpyAsync(d_g.p, kept_root, n_kept * 4, hipMemcpyHostToDevice, st));
        { mf_ktimer tm_(ctx, "k_dcc_minkeys"); k_dcc_minkeys<<<cgrid(n_kept), 256, 0, st>>>(d_g.p, n_kept, D->gmin.p, (unsigned long long *)d_out); }
        MF_HIP(hipStreamSynchronize(st));
    }
    return MF_OK;
}
// all ranks' members + all levels' kept components (host arrays, the same on every rank) -> the components object every
// rank holds (order: ConnectedComponent.compareTo, src/structures/ConnectedComponent.java:125-136, ties by the smallest k-mer)
__global__ void k_dcc_slot_map(const uint32_t *__restrict__ g, uint32_t n, uint32_t *__restrict__ map) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) map[g[i]] = i;
}
__global__ void k_dcc_member_slots(const uint32_t *__restrict__ mg, uint64_t n, const uint32_t *__restrict__ map, uint32_t *__restrict__ comp) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) comp[i] = map[mg[i]];
}
// ALL ranks' grouped members (k-mers in rank order, each rank's sorted by component; the ranks' (root, count) runs in the same
// order) -> mf_dcc_finish on the expanded roots
extern "C" int mf_dcc_finish(mf_dcc *D, const void *d_keys, const void *d_roots, uint64_t nm, const uint32_t *kept_root, const uint32_t *kept_size,
                             const int64_t *kept_weight, const int32_t *kept_thr, const uint64_t *minkey, uint64_t n_kept, mf_comps **out);
extern "C" int mf_dcc_finish_grouped(mf_dcc *D, const void *d_keys, uint64_t nm, const void *d_runs, uint64_t n_runs, const uint32_t *kept_root,
                                     const uint32_t *kept_size, const int64_t *kept_weight, const int32_t *kept_thr, const uint64_t *minkey, uint64_t n_kept,
                                     mf_comps **out) {
    if (!D || !out || (nm && (!d_keys || !d_runs))) return mf_set_error("mf_dcc_finish_grouped: NULL argument");
    mf_ctx *ctx = D->ctx; hipStream_t st = ctx->stream;
    MF_HIP(hipSetDevice(ctx->device));
    mf_buf<uint32_t> roots, len; mf_buf<uint64_t> off, tot;
    MF_TRY(roots.alloc(ctx, nm ? nm : 1));
    if (n_runs) {
        MF_TRY(len.alloc(ctx, n_runs)); MF_TRY(off.alloc(ctx, n_runs + 1)); MF_TRY(tot.alloc(ctx, 1));
        k_dcc_run_lens<<<cgrid(n_runs), 256, 0, st>>>((const uint2 *)d_runs, n_runs, len.p);
        MF_TRY(mf_scan<1>(ctx, len.p, off.p, n_runs, tot.p));
        uint64_t total = 0;
        MF_HIP(hipMemcpyAsync(&total, tot.p, 8, hipMemcpyDeviceToHost, st));
        MF_HIP(hipStreamSynchronize(st));
        if (total != nm) return mf_set_error("mf_dcc_finish_grouped: the runs describe %llu members, %llu were passed", (unsigned long long)total, (unsigned long long)nm);
        k_dcc_run_expand<<<cgrid(n_runs * 64), 256, 0, st>>>((const uint2 *)d_runs, off.p, n_runs, roots.p);
    } else if (nm) return mf_set_error("mf_dcc_finish_grouped: members without runs");
    return mf_dcc_finish(D, d_keys, roots.p, nm, kept_root, kept_size, kept_weight, kept_thr, minkey, n_kept, out);
}
extern "C" int mf_dcc_finish(mf_dcc *D, const void *d_keys, const void *d_roots, uint64_t nm, const uint32_t *kept_root, const uint32_t *kept_size,
                             const int64_t *kept_weight, const int32_t *kept_thr, const uint64_t *minkey, uint64_t n_kept, mf_comps **out) {
    if (!D || !out || (nm && (!d_keys || !d_roots)) || (n_kept && (!kept_root || !kept_size || !kept_weight || !kept_thr || !minkey)))
        return mf_set_error("mf_dcc_finish: NULL argument");
    *out = nullptr;
    mf_ctx *ctx = D->ctx; hipStream_t st = ctx->stream;
    MF_HIP(hipSetDevice(ctx->device));
    if (nm >= 0xFFFFFFFFull || n_kept >= 0xFFFFFFFFull) return mf_set_error("components: too many k-mers");
    uint64_t sum = 0;
    for (uint64_t i = 0; i < n_kept; i++) { sum += kept_size[i]; if (kept_root[i] >= D->n_total) return mf_set_error("mf_dcc_finish: component root out of range"); }
    if (sum != nm) return mf_set_error("mf_dcc_finish: %llu members for components of %llu k-mers", (unsigned long long)nm, (unsigned long long)sum);
    std::unique_ptr<mf_comps, void (*)(mf_comps *)> C(new mf_comps(), mf_comps_destroy);
    C->ctx = ctx; C->k = D->k; C->n = n_kept; C->n_kmers = nm;
    void *p = nullptr;
    MF_TRY(mf_alloc(ctx, (nm ? nm : 1) * 8, &p)); C->d_kmers = (uint64_t *)p; C->kmers_bytes = (nm ? nm : 1) * 8;
    MF_TRY(mf_alloc(ctx, (nm ? nm : 1) * 4, &p)); C->d_comp = (uint32_t *)p; C->comp_bytes = (nm ? nm : 1) * 4;
    mf_buf<uint32_t> d_g, d_rank;
    MF_TRY(d_g.alloc(ctx, n_kept ? n_kept : 1)); MF_TRY(d_rank.alloc(ctx, n_kept ? n_kept : 1));
    if (n_kept) {
        MF_HIP(hipMemcpyAsync(d_g.p, kept_root, n_kept * 4, hipMemcpyHostToDevice, st));
        { mf_ktimer tm_(ctx, "k_dcc_slot_map"); k_dcc_slot_map<<<cgrid(n_kept), 256, 0, st>>>(d_g.p, (uint32_t)n_kept, D->pg.p); }
        if (nm) {
            MF_HIP(hipMemcpyAsync(C->d_kmers, d_keys, nm * 8, hipMemcpyDeviceToDevice, st));
            { mf_ktimer tm_(ctx, "k_dcc_member_slots"); k_dcc_member_slots<<<cgrid(nm), 256, 0, st>>>((const uint32_t *)d_roots, nm, D->pg.p, C->d_comp); }
        }
    }
    std::vector<uint32_t> order(n_kept);
    for (uint64_t i = 0; i < n_kept; i++) order[i] = (uint32_t)i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        if (kept_thr[a] != kept_thr[b]) return kept_thr[a] < kept_thr[b];
        if (kept_weight[a] != kept_weight[b]) return kept_weight[a] > kept_weight[b];
        if (kept_size[a] != kept_size[b]) return kept_size[a] > kept_size[b];
        return minkey[a] < minkey[b];
    });
    std::vector<uint32_t> rank(n_kept ? n_kept : 1);
    for (uint64_t i = 0; i < n_kept; i++) {
        const uint32_t s = order[i];
        C->sizes.push_back(kept_size[s]); C->weights.push_back(kept_weight[s]); C->thr.push_back(kept_thr[s]);
        rank[s] = (uint32_t)i;
    }
    if (nm) {
        MF_HIP(hipMemcpyAsync(d_rank.p, rank.data(), n_kept * 4, hipMemcpyHostToDevice, st));
        k_cc_remap<<<cgrid(nm), 256, 0, st>>>(C->d_comp, nm, d_rank.p);
        MF_HIP(hipStreamSynchronize(st));
    }
    *out = C.release();
    return MF_OK;
}

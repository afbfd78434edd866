// mf_nbr.h -- the eight de Bruijn neighbours of every k-mer of a table, looked up PARTITION-LOCALLY.
//
// getLeftNucleotide / getRightNucleotide (src/algo/HashMapOperations.java:13-47) and possibleNeighbours
// (src/algo/KmerOperations.java:9-26) probe the map eight times per k-mer.  Against the HBM index that is eight random
// 16-byte reads per k-mer, each pulling in a whole line: round 1 measured 169 GB fetched for 38 GB asked (k_ut_flags) and
// 60 GB for 10 GB (k_cc_adjacency).  But a table that comes out of the counting pass is ordered by MINIMIZER partition,
// and nine neighbours in ten share their k-mer's minimizer, i.e. its partition -- a few hundred k-mers that sit next to
// each other in the table.  So: one wave per partition.  It reads the partition's keys (coalesced) into LDS, builds a small
// open-addressed index of them there (1024 two-byte slots: tag + position), and looks the neighbours up there; only a
// neighbour whose own minimizer differs (its partition hash says so before any memory is touched) goes to the HBM index.
// Partitions larger than NB_CAP keys use the HBM index for everything.
//
// ONE WALK PER SIDE.  The four k-mers that extend x on one side differ in one base at an end: they share their interior
// (k-2)-mer, and so do their reverse complements.  Both indexes -- the one in LDS and the table's in HBM (mf_cidx_hkey) -- take
// home slot and tag from the hash of the CANONICAL INTERIOR, so one probe sequence from that home slot to the next empty slot
// meets all four; a stored key K is neighbour c of the side if its other k-1 bases are x's (K >> 2 == pa or K's low k-1 bases
// == pb, mf_index_walk_side), and which of the two says on which strand the table holds it.  Neither a neighbour nor a canonical
// form is built anywhere.
#pragma once
#include "mf_common.h"
#ifdef __HIPCC__
#ifndef NB_ABLATE
#define NB_ABLATE 0
#endif
#define NB_SLOTS 1024               // 16-bit slots of a wave's index (tag | position): load <= 0.34, on average half of that
#define NB_CAP 352                 // keys of a partition that go into LDS
#define NB_WAVES 4                 // waves (= partitions in flight) per workgroup
#define NB_NONE 0xFFFFFFFFu

#define NB_RQ 64                   // remote requests a wave collects per 64 k-mers (more: looked up on the spot); a request is the neighbours of ONE side that share
                                   // a minimizer (24 bytes): half a request per k-mer; 25 KiB of LDS per workgroup = six per CU
// The LDS table of a partition is its keys IN TABLE ORDER plus an open-addressed index of 16-bit slots, slot = tag (the hash bits
// after the home slot's) | position: a lookup of a k-mer that is not there -- three neighbours in four -- ends at the first empty
// slot without reading a key, and a key is read only behind a matching tag.  (Until round 4 the slots held the keys themselves, 512
// of them at load <= 0.69: every trip of a probe loop an 8-byte read and two 64-bit compares, and the loops ran ~5 trips because a
// wave probes as long as the slowest of its lanes.)
struct nb_lds {
    uint64_t key[NB_WAVES][NB_CAP];
    uint64_t rq_pa[NB_WAVES][NB_RQ], rq_pb[NB_WAVES][NB_RQ];               // a request: the two patterns of a side (mf_index_walk_side); then its four answers
    uint32_t rq_mn[NB_WAVES][NB_RQ], rq_meta[NB_WAVES][NB_RQ];              // the minimizer hash its neighbours share; side | wanted neighbours << 1
    uint32_t slot[NB_WAVES][NB_SLOTS / 2];                                  // two slots a word (LDS compare-and-swap works on words)
};

// neighbour i of x: i = 2*nuc (append nuc on the right) or 2*nuc+1 (prepend nuc on the left); *ph = its partition hash
// rcx = mf_revcomp(x, k), computed once per k-mer: the reverse complement of a neighbour is a shift of it (three
// instructions instead of the ~25 of a full reverse complement, eight times per k-mer)
__device__ __forceinline__ uint64_t nb_neighbour(uint64_t x, uint64_t rcx, int k, uint64_t kmask, uint32_t i, uint32_t m_nf, uint32_t m_nl, uint64_t *oriented, uint32_t *ph) {
    const uint32_t nuc = i >> 1;
    uint64_t y, r;
    if (i & 1u) { y = (x >> 2) | ((uint64_t)nuc << (2 * k - 2)); r = ((rcx << 2) | (uint64_t)(3u - nuc)) & kmask; *ph = mf_skm_ph_left(y, k, m_nl); }
    else { y = ((x << 2) | nuc) & kmask; r = (rcx >> 2) | ((uint64_t)(3u - nuc) << (2 * k - 2)); *ph = mf_skm_ph_right(y, k, m_nf); }
    *oriented = y;
    return y < r ? y : r;
}
__device__ __forceinline__ uint64_t nb_neighbour(uint64_t x, int k, uint64_t kmask, uint32_t i, uint32_t m_nf, uint32_t m_nl, uint64_t *oriented, uint32_t *ph) {
    return nb_neighbour(x, mf_revcomp(x, k), k, kmask, i, m_nf, m_nl, oriented, ph);
}
// Calls emit(j, x, idx[8], flip [8 bits: the table holds neighbour i, where found, as its reverse complement], foreign [8 bits], have) for every k-mer j of the table (lanes past
// the end of a partition call it with have = false); idx[i] = table index of neighbour i or NB_NONE.
// MODE 0: one wave per partition, partitions dealt round-robin to the waves of the grid; a partition of more than NB_CAP keys
//         looks everything up in the HBM index.
// MODE 1 + MODE 2 (two launches, together = MODE 0's result): 12 % of the benchmark's good k-mers sit in partitions of
//         353 .. 1408 keys, and their 8 lookups each were HALF of all probes that went to HBM.  MODE 1 leaves those partitions
//         out; MODE 2 takes only them, one WORKGROUP per partition: the four waves' LDS tables together (4096 slots) hold the
//         partition, the waves share its k-mers.
// With lw > 0 the table is rank `me`'s SHARD of a larger one (mf_count_device_shard; owner of a k-mer = top lw bits of its
// partition hash): neighbours that other ranks own are not looked up, emit gets their numbers in `foreign` (8 bits).
#define NB_BIGCAP (NB_CAP * NB_WAVES)
__device__ __forceinline__ uint32_t nb_hash(uint64_t key) { return ((uint32_t)key ^ (uint32_t)(key >> 29)) * 0x9E3779B1u; }
// the canonical INTERIOR of a k-mer: its middle k-2 bases or their reverse complement, whichever is smaller (rcx = mf_revcomp(x, k))
__device__ __forceinline__ uint64_t nb_interior(uint64_t x, uint64_t rcx, int k) {
    const uint64_t WM = (1ull << (2 * k - 4)) - 1ull;
    const uint64_t w = (x >> 2) & WM, rw = (rcx >> 2) & WM;
    return w < rw ? w : rw;
}
template <int MODE, typename F>
__device__ __forceinline__ void nb_for_each(const mf_index_view &ix, const uint64_t *__restrict__ keys, const uint64_t *__restrict__ part_off,
                                            uint32_t p_lo, uint32_t np, int k, nb_lds &S, int lw, uint32_t me, F &&emit) {
    constexpr bool BIG = MODE == 2;
    constexpr uint32_t SLOTS = BIG ? (uint32_t)(NB_SLOTS * NB_WAVES) : (uint32_t)NB_SLOTS;
    constexpr int HSHIFT = BIG ? 20 : 22;                                   // 12 / 10 slot bits
    constexpr int POSBITS = BIG ? 11 : 9;                                   // NB_BIGCAP <= 2047, NB_CAP <= 511: a position is never all ones,
    constexpr uint32_t POSMASK = (1u << POSBITS) - 1u;                      // so no entry looks like an empty slot (0xFFFF)
    constexpr int TSHIFT = HSHIFT - (16 - POSBITS);                         // 5 / 7 tag bits, the hash bits below the slot's
    constexpr uint32_t TAGMASK = (1u << (16 - POSBITS)) - 1u;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t first = BIG ? blockIdx.x : blockIdx.x * NB_WAVES + wave, stride = BIG ? gridDim.x : gridDim.x * NB_WAVES;
    const uint64_t kmask = (1ull << (2 * k)) - 1;
    uint64_t *hk = BIG ? &S.key[0][0] : S.key[wave]; uint32_t *hw = BIG ? &S.slot[0][0] : S.slot[wave];
    const uint16_t *hs = reinterpret_cast<const uint16_t *>(hw);
    uint64_t *qa = S.rq_pa[wave], *qb = S.rq_pb[wave]; uint32_t *qm = S.rq_mn[wave], *qt = S.rq_meta[wave];
    const uint32_t tl = BIG ? threadIdx.x : lane, tn = BIG ? (uint32_t)(64 * NB_WAVES) : 64u;      // the team that builds the table
    // (wave level: the barrier builtin orders nothing by itself -- pair it with a wavefront-scope fence so that the LDS hand-offs
    // between lanes are ordered by contract, not by what the alias analysis happens to keep)
    auto wave_sync = [&]() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
    auto team_sync = [&]() { if (BIG) __syncthreads(); else wave_sync(); };
    for (uint32_t p = p_lo + first; p < np; p += stride) {
        const uint64_t lo = part_off[p], hi = part_off[p + 1];
        const uint32_t n = (uint32_t)(hi - lo);
        if (n == 0) continue;                                               // team-uniform
        const bool mid = n > (uint32_t)NB_CAP && n <= (uint32_t)NB_BIGCAP;
        if (MODE == 1 && mid) continue;
        if (MODE == 2 && !mid) continue;
        const bool local = BIG || n <= (uint32_t)NB_CAP;
        if (local) {
            for (uint32_t s = tl; s < SLOTS / 2; s += tn) hw[s] = 0xFFFFFFFFu;
            team_sync();
            for (uint32_t j = tl; j < n; j += tn) {
                const uint64_t x = keys[lo + j];
                hk[j] = x;
                const uint32_t h = nb_hash(nb_interior(x, mf_revcomp(x, k), k));
                const uint32_t ent = (((h >> TSHIFT) & TAGMASK) << POSBITS) | j;
                uint32_t s = h >> HSHIFT;
                for (;;) {                                                  // (keys of a table are distinct)
                    uint32_t *w = &hw[s >> 1];
                    const uint32_t sh = (s & 1u) * 16u;
                    const uint32_t old = __atomic_load_n(w, __ATOMIC_RELAXED);
                    if (((old >> sh) & 0xFFFFu) != 0xFFFFu) { s = (s + 1u) & (SLOTS - 1u); continue; }
                    if (atomicCAS(w, old, (old & ~(0xFFFFu << sh)) | (ent << sh)) == old) break;      // (lost: the word's other half changed, or this one -- look again)
                }
            }
            team_sync();
        }
        constexpr uint32_t JSTEP = BIG ? (uint32_t)(64 * NB_WAVES) : 64u;
        for (uint32_t j0 = BIG ? wave * 64u : 0u; j0 < n; j0 += JSTEP) {      // wave-uniform
            const uint32_t j = j0 + lane;
            const bool have = j < n;
            const uint64_t x = have ? (local ? hk[j] : keys[lo + j]) : 0ull;       // (the copy in LDS: the table is read once)
            uint32_t m_nf = 0, m_nl = 0, m_own = 0;
            mf_skm_nbr_mins(x, k, &m_nf, &m_nl, &m_own);
            // Pass 1: which neighbours live in another partition (about one in ten)?  Those become REQUESTS for the HBM index -- a
            // directory entry, a slot, a key: dependent cache misses.  Looked up where they arise they cost the wave eight such round
            // trips per 64 k-mers with six lanes in 64 busy (ablation in round 2, 100 M reads: 25 of the kernel's 39 ms).  So the wave
            // collects them in LDS, looks them up together, one request per lane, and hands the answers back through LDS.
            uint32_t remote = 0, foreign = 0, flip = 0;
            uint32_t phs[8];
            const uint64_t rcx = mf_revcomp(x, k);
            const int M = mf_skm_m(k);
            const uint32_t MM = (1u << (2 * M)) - 1u;
#pragma unroll
            for (uint32_t i = 0; i < 8; i++) {
                // the neighbour's minimizer hash from x's minima and the ONE M-mer it has that x has not: its last (right side: x's last
                // M-1 bases + the new one) or its first (left side); re-mixed only where a partition hash is needed.  (Neither the
                // neighbour nor its canonical form is built here: a walk compares stored keys with x's own k-1 bases.)
                const uint32_t nuc = i >> 1;
                const uint32_t f = (i & 1u) ? (((uint32_t)(x >> (2 * (k - M) + 2)) & (MM >> 2)) | (nuc << (2 * M - 2))) : ((((uint32_t)x << 2) | nuc) & MM);
                const uint32_t r = mf_mmer_rc(f, M), h = mf_mmer_hash(f < r ? f : r, M), mo = (i & 1u) ? m_nl : m_nf;
                phs[i] = h < mo ? h : mo;
                // the same minimizer = the same partition; another minimizer that lands in this partition all the same (one in
                // 2^part_bits) is simply looked up through the index
                if (have && lw && (mf_remix32(phs[i]) >> (32 - lw)) != me) foreign |= 1u << i;
                else if (have && !(local && phs[i] == m_own)) remote |= 1u << i;
            }
            // Requests.  The remote neighbours of a side that share their minimizer -- all four when x's own minimizer was its first /
            // last M-mer, the usual case -- are ONE request: one directory entry, one probe sequence (the index is hashed on the
            // interior the four share).  A side whose remote neighbours have different minimizers asks for each of them alone.
            uint32_t lead[2], grouped = 0, nreq = 0;
#pragma unroll
            for (uint32_t side = 0; side < 2; side++) {
                const uint32_t rem = (remote >> side) & 0x55u;
                lead[side] = 0;
                bool same = true;
#pragma unroll
                for (int c = 3; c >= 0; c--) if ((rem >> (2 * c)) & 1u) lead[side] = phs[2 * c + side];
#pragma unroll
                for (uint32_t c = 0; c < 4; c++) if (((rem >> (2 * c)) & 1u) && phs[2 * c + side] != lead[side]) same = false;
                if (same) grouped |= 1u << side;
                nreq += rem ? (same ? 1u : (uint32_t)__popc(rem)) : 0u;
            }
            uint32_t R;
            const uint32_t rbase = mf_wave_excl_scan(nreq, &R);
            uint32_t idx[8];
            // (Tried on the per-neighbour probe loops this code had until round 4: two slots per LDS read, k_ut_flags 26.7 -> 30.0 ms; the
            // first probes of eight / four / two neighbours in flight behind one wait, 29.2 / 28.3 against 26.2 -- registers, i.e. waves per
            // SIMD, each time: profiles/r03f_bench_100M_batched_lookups.json, r03j_bench_100M_flags_batch2.json.  And on this code: the next
            // batch's minimizer scan behind the home slot's read, 24.35 against 24.5 ms at four waves.)
#pragma unroll
            for (uint32_t i = 0; i < 8; i++) idx[i] = NB_NONE;
            {
                const uint64_t LM = kmask >> 2;
                uint32_t at = rbase;
#pragma unroll
                for (uint32_t side = 0; side < 2; side++) {
                    const uint32_t rem = (remote >> side) & 0x55u;
                    if (!rem) continue;
                    const uint64_t pa = side ? (rcx & LM) : (x & LM), pb = side ? (x >> 2) : (rcx >> 2);
                    uint32_t want4 = 0;
#pragma unroll
                    for (uint32_t c = 0; c < 4; c++) want4 |= ((rem >> (2 * c)) & 1u) << c;
                    if ((grouped >> side) & 1u) {
                        if (at < (uint32_t)NB_RQ) { qa[at] = pa; qb[at] = pb; qm[at] = lead[side]; qt[at] = side | (want4 << 1); }
                        else {                                                   // (no room: looked up on the spot)
                            uint32_t o4[4], rv;
                            mf_index_walk_side(ix, mf_remix32(lead[side]), pa, pb, side, want4, k, o4, &rv);
#pragma unroll
                            for (uint32_t c = 0; c < 4; c++) if ((want4 >> c) & 1u) { idx[2 * c + side] = o4[c]; flip |= ((rv >> c) & 1u) << (2 * c + side); }
                        }
                        at++;
                    } else {
#pragma unroll
                        for (uint32_t c = 0; c < 4; c++) {
                            if (!((want4 >> c) & 1u)) continue;
                            if (at < (uint32_t)NB_RQ) { qa[at] = pa; qb[at] = pb; qm[at] = phs[2 * c + side]; qt[at] = side | (1u << (c + 1)); }
                            else { uint32_t o4[4], rv; mf_index_walk_side(ix, mf_remix32(phs[2 * c + side]), pa, pb, side, 1u << c, k, o4, &rv); idx[2 * c + side] = o4[c]; flip |= ((rv >> c) & 1u) << (2 * c + side); }
                            at++;
                        }
                    }
                }
            }
            // lane r takes request r (NB_RQ = 64: one each).  Its directory entry -- the first of the two or three dependent reads of a
            // lookup -- is asked for HERE and used after the local walks below: one memory latency of every batch behind LDS work.
            static_assert(NB_RQ <= 64, "one request per lane");
            const uint32_t Rl = R < (uint32_t)NB_RQ ? R : (uint32_t)NB_RQ;
            ulonglong2 dent = make_ulonglong2(0ull, 0ull);
            if (R) {
                wave_sync();
                if (lane < Rl && ix.compact) dent = mf_index_side_dir(ix, mf_remix32(qm[lane]));
            }
#if !(NB_ABLATE & 1)
            if (local) {
                // the four neighbours of a side differ in ONE base at an end: they share their interior (k-2)-mer, the index is
                // hashed on it, so ONE walk from the interior's home slot to the next empty slot meets all of them.  A key K met
                // on the way is neighbour c of the side if its other k-1 bases are x's: K = y (forward) or K = rc(y).
                const uint64_t LM = kmask >> 2, WM = kmask >> 4;
                const uint32_t nloc = ~(remote | foreign);
#pragma unroll
                for (uint32_t side = 0; side < 2; side++) {
                    if (have && (nloc & (0x55u << side))) {
                        const uint64_t w = side ? (x >> 4) : (x & WM), rw = side ? (rcx & WM) : (rcx >> 4);
                        const uint64_t pa = side ? (rcx & LM) : (x & LM);          // against K >> 2: K = y (right side) / K = rc(y) (left side)
                        const uint64_t pb = side ? (x >> 2) : (rcx >> 2);          // against K & LM: K = rc(y) (right side) / K = y (left side)
                        const uint32_t h = nb_hash(w < rw ? w : rw), tag = (h >> TSHIFT) & TAGMASK;
                        uint32_t s = h >> HSHIFT;
                        for (;;) {
                            const uint32_t v = hs[s];
                            if (v == 0xFFFFu) break;
                            if ((v >> POSBITS) == tag) {
                                const uint64_t K = hk[v & POSMASK];
                                // K is the neighbour itself or its reverse complement (a palindrome is both and counts as itself)
                                const bool ma = (K >> 2) == pa, mb = (K & LM) == pb, fwd = side ? mb : ma;
                                const uint32_t ca = ((uint32_t)K & 3u) ^ (side ? 3u : 0u), cb = (uint32_t)(K >> (2 * k - 2)) ^ (side ? 0u : 3u);
                                const uint32_t c = (ma || mb) ? (fwd ? (side ? cb : ca) : (side ? ca : cb)) : 4u;
                                const uint32_t at = (uint32_t)lo + (v & POSMASK);
#pragma unroll
                                for (uint32_t q = 0; q < 4; q++)
                                    if (c == q && ((nloc >> (2 * q + side)) & 1u)) { idx[2 * q + side] = at; flip = (flip & ~(1u << (2 * q + side))) | ((fwd ? 0u : 1u) << (2 * q + side)); }
                            }
                            s = (s + 1u) & (SLOTS - 1u);
                        }
                    }
                }
            }
#endif
            if (R) {
                if (lane < Rl) {
                    uint32_t o4[4], rv = 0;
                    const uint32_t meta = qt[lane];
#if NB_ABLATE & 2
                    o4[0] = o4[1] = o4[2] = o4[3] = (qa[lane] == 12345ull && dent.x == 77ull) ? 5u : NB_NONE;
#else
                    if (ix.compact) mf_index_walk_side_d(ix, dent, qa[lane], qb[lane], meta & 1u, meta >> 1, k, o4, &rv);
                    else mf_index_walk_side(ix, mf_remix32(qm[lane]), qa[lane], qb[lane], meta & 1u, meta >> 1, k, o4, &rv);
#endif
                    uint32_t *ans = reinterpret_cast<uint32_t *>(&qa[lane]), *ans2 = reinterpret_cast<uint32_t *>(&qb[lane]);      // (the request has been read: its answers take its place)
                    ans[0] = o4[0]; ans[1] = o4[1]; ans2[0] = o4[2]; ans2[1] = o4[3]; qt[lane] = rv;
                }
                wave_sync();
                uint32_t at = rbase;
#pragma unroll
                for (uint32_t side = 0; side < 2; side++) {
                    const uint32_t rem = (remote >> side) & 0x55u;
                    if (!rem) continue;
#pragma unroll
                    for (uint32_t c = 0; c < 4; c++) {
                        if (!((rem >> (2 * c)) & 1u)) continue;
                        if (at < (uint32_t)NB_RQ) {
                            idx[2 * c + side] = c < 2 ? reinterpret_cast<const uint32_t *>(&qa[at])[c] : reinterpret_cast<const uint32_t *>(&qb[at])[c - 2];
                            flip |= ((qt[at] >> c) & 1u) << (2 * c + side);
                        }
                        if (!((grouped >> side) & 1u)) at++;
                    }
                    if ((grouped >> side) & 1u) at++;
                }
                wave_sync();
            }
            emit(lo + j, x, idx, flip, foreign, have);                       // (every lane: the callee may use wave-wide operations)
        }
        team_sync();                                                        // (the table is cleared for the next partition)
    }
}
#endif

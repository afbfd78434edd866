// mf_wskm.hip -- NO-REFERENCE EXTENSION (32 <= k <= 63): the canonical counts of 2k-bit k-mers on the RECORD path (round 6).
//
// mf_wide.hip's count writes every k-mer OCCURRENCE as two 64-bit words and orders them with four radix passes over HBM + two LDS passes:
// 36 bytes of HBM per occurrence several times over, 2.0 s for the 1.76e10 63-mers of 200 M reads -- 19 x what the k <= 31 count takes per
// occurrence, because that one moves SUPER-K-MERS (mf_skm.hip: a run of consecutive k-mers of a read that share their minimizer travels as one
// record).  The same idea for wide k-mers, written plainly (no inline assembly, no LDS-staged radix levels -- the existing radix sort orders 8-byte
// (partition, record number) pairs instead of the 32-byte records themselves):
//
//   k_wskm_scan    a workgroup takes a tile of 4096 base positions (+ halo): bases packed two bits each into LDS, the canonical 15-mer hash of
//                  every position (mf_mmer_hash: the k <= 31 path's minimizer order), the minimum over a k-mer's 49 - (63 - k) 15-mers in two
//                  steps (windows of 7, then 7 of those), run starts where the minimum changes (or a read starts, or a run has 48 k-mers), and one
//                  32-BYTE RECORD per run: its bases (<= 110, two bits each) + the number of k-mers; beside it the run's partition = the top bits
//                  of the re-mixed minimizer hash.  ~1.3 bytes per k-mer occurrence instead of 16.
//   mf_sort        (partition, record number) pairs, 8 bytes each, by partition; a binary search per partition gives the units' extents.
//   k_wskm_count   a workgroup per unit: a thread unrolls a record k-mer by k-mer (two rolling 128-bit registers: forward and reverse complement)
//                  into an open-addressed table in LDS -- 4096 slots of (high word, low word, count); a slot is claimed with ONE 64-bit
//                  compare-and-swap on its high word (a k-mer's high word has its top two bits clear: EMPTY and BUSY cannot be keys), the low word
//                  is written behind it and the high word published last; a unit with too many distinct k-mers is counted in 4 / 16 / ... passes
//                  by key hash, as the k <= 31 path does.  The entries with count > threshold leave through a block-wide reservation.
//   order          the kept entries (a twentieth of the occurrences) are put in ascending (high, low) order with the pair sorts: by the low word
//                  carrying the entry number, then -- stable -- by the high word.
// A partition holds ALL occurrences of its k-mers, so the counts are exact; the table that comes out is the one mf_wide.hip's count gives, entry for
// entry (tests/test_wide_gpu.py, tools/fuzz_wide.py run both).  Inputs it does not suit (tiny ones, no memory for the records) return 1: the
// caller counts the old way.
#include <algorithm>
#include <memory>
#include "mf_common.h"
#include "mf_wide.h"

int mf_sort_u64_u32(mf_ctx *ctx, const uint64_t *d_keys_in, const uint32_t *d_vals_in, uint64_t n, int bits, uint64_t *d_keys_out, uint32_t *d_vals_out);

#define WS_T 256                                   // threads of the scan
#define WS_TILE 4096                               // base positions (k-mer starts) of a tile
#define WS_HALO 128                                // positions behind the tile whose bases / hashes the tile's last k-mers need (>= 63 + 15)
#define WS_PK ((WS_TILE + WS_HALO) / 16 + 2)       // packed words (16 bases each)
#define WS_M 15                                    // minimizer length (mf_skm_m for k > 25)
#define WS_RMAX 48                                 // k-mers per record: 48 + 62 = 110 bases = 220 bits of the record's 224
#define WC_T 1024                                  // threads of the count (one workgroup per CU: 16 waves)
#define WC_SLOTS 4096
#define WC_FILL 2800
#ifndef WC_RO
#define WC_RO 4                                    // consecutive k-mers of a thread in the passes over one class of an overflowing unit (k_wskm_count; 1 / 2 / 4 / 8 / 16: 422 / 410 / 404 / 413 / 439 ms)
#endif
#ifndef WC_PROF
#define WC_PROF 0                                  // (1: thread 0 of every workgroup adds up the cycles of the kernel's phases: counters[8 .. 13], printed with verbose)
#endif
#ifndef WC_ABL
#define WC_ABL 0                                   // (timing experiments, results are wrong: 1 no table operations, 3 no fill counter, 4 no fence, 5 no look at the overflow flag)
#endif
#define WC_EMPTY 0xFFFFFFFFFFFFFFFFull
#define WC_BUSY 0xFFFFFFFFFFFFFFFEull

struct wskm_rec { uint32_t w[8]; };                 // w[0..6]: bases, first base in the top bits of w[0]; w[7]: number of k-mers

static inline unsigned wsgrid(uint64_t n, unsigned bs = 256) { return (unsigned)std::min<uint64_t>((n + bs - 1) / bs, 0x7FFFFFFFull); }

__device__ __forceinline__ uint32_t ws_pack16(const uint8_t *__restrict__ bases, uint64_t q, uint64_t n_bases, bool aligned) {
    uint32_t w[4];
    if (aligned && q + 16 <= n_bases) { const uint4 v = *reinterpret_cast<const uint4 *>(bases + q); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
    else {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            w[j] = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) { const uint64_t a = q + 4 * j + b; if (a < n_bases) w[j] |= (uint32_t)bases[a] << (8 * b); }
        }
    }
    uint32_t out = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {                                                // A0 G1 C2 T3 of four bytes at once; the first base ends up on top
        const uint32_t x = (w[j] >> 1) & 0x03030303u, x0 = x & 0x01010101u, x1 = (x >> 1) & 0x01010101u;
        const uint32_t c = ((x0 ^ x1) << 1) | x1;
        out = (out << 8) | ((c * 0x40100401u) >> 24);
    }
    return out;
}
// 16 bases from base position p of the packed tile (first base on top)
__device__ __forceinline__ uint32_t ws_word_at(const uint32_t *pk, uint32_t p) {
    const uint32_t q = p >> 4, o = (p & 15u) * 2u;
    const uint64_t W = ((uint64_t)pk[q] << 32) | (uint64_t)pk[q + 1];
    return (uint32_t)(W >> (32u - o));
}

// occurrences = sum over the reads of max(0, len - k + 1), reads shorter than max(k, min_len) give nothing
__global__ void k_wskm_nocc(const uint64_t *__restrict__ off, uint64_t n_reads, int k, int min_len, unsigned long long *__restrict__ out) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long v = 0;
    if (r < n_reads) { const uint64_t len = off[r + 1] - off[r]; if (len >= (uint64_t)k && (int64_t)len >= (int64_t)min_len) v = len - (uint64_t)k + 1; }
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, 64);
    if ((threadIdx.x & 63u) == 0 && v) atomicAdd(out, v);
}

__global__ __launch_bounds__(WS_T) void k_wskm_scan(const uint8_t *__restrict__ bases, uint64_t n_bases, const uint32_t *__restrict__ vmask, uint64_t n_words, int k, int pbits,
                                                   wskm_rec *__restrict__ recs, uint32_t *__restrict__ part, uint64_t cap, unsigned long long *__restrict__ cursor) {
    __shared__ uint32_t pk[WS_PK];
    __shared__ uint32_t h[WS_TILE + WS_HALO];           // M-mer hashes, then (in place) the k-mers' minimizer hashes
    __shared__ uint32_t m7[WS_TILE + WS_HALO];
    __shared__ uint32_t vw[WS_TILE / 32 + 1], sw[WS_TILE / 32 + 1];      // bitmaps: a k-mer starts here; a run starts here
    __shared__ uint32_t scratch[18];
    __shared__ unsigned long long s_base;
    __shared__ uint8_t rl[WS_TILE];
    const uint32_t tid = threadIdx.x;
    const bool aligned = (reinterpret_cast<uintptr_t>(bases) & 15u) == 0;
    const int W = k - WS_M + 1;                          // M-mers of a k-mer (18 .. 49)
    const uint64_t n_tiles = (n_words * 32 + WS_TILE - 1) / WS_TILE;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        __syncthreads();
        const uint64_t q0 = tile * WS_TILE;
        for (uint32_t i = tid; i < WS_PK; i += WS_T) pk[i] = ws_pack16(bases, q0 + (uint64_t)i * 16, n_bases, aligned);
        for (uint32_t i = tid; i < WS_TILE / 32 + 1; i += WS_T) { const uint64_t w = q0 / 32 + i; vw[i] = (i < WS_TILE / 32 && w < n_words) ? vmask[w] : 0u; sw[i] = 0u; }
        __syncthreads();
        bool any = false;
        for (uint32_t i = 0; i < WS_TILE / 32; i++) any |= vw[i] != 0u;
        if (!any) continue;                              // (uniform: everybody has read the same words)
        // canonical M-mer hash of every position
        for (uint32_t i = tid; i < WS_TILE + WS_HALO - 16; i += WS_T) {
            const uint32_t f = ws_word_at(pk, i) >> (32 - 2 * WS_M);
            const uint32_t r = mf_mmer_rc(f, WS_M);
            h[i] = mf_mmer_hash(f < r ? f : r, WS_M);
        }
        __syncthreads();
        // minimum over W consecutive hashes: windows of 7, then as many of those as cover W (the last one overlapping)
        for (uint32_t i = tid; i < WS_TILE + WS_HALO - 24; i += WS_T) {
            uint32_t m = h[i];
#pragma unroll
            for (int j = 1; j < 7; j++) { const uint32_t v = h[i + j]; m = v < m ? v : m; }
            m7[i] = m;
        }
        __syncthreads();
        for (uint32_t p = tid; p < WS_TILE; p += WS_T) {
            uint32_t m = m7[p];
            for (int j = 7; j + 7 <= W; j += 7) { const uint32_t v = m7[p + j]; m = v < m ? v : m; }
            { const uint32_t v = m7[p + W - 7]; m = v < m ? v : m; }
            h[p] = m;                                    // (h[p] is read by m7 only: all m7 are made)
        }
        __syncthreads();
        // run starts: a k-mer whose minimizer differs from the one before it (or that has no k-mer before it) -- a ballot per 64 positions
        for (uint32_t p0 = 0; p0 < WS_TILE; p0 += WS_T) {
            const uint32_t p = p0 + tid;
            const bool v = (vw[p >> 5] >> (p & 31u)) & 1u;
            const bool pv = p > 0 && ((vw[(p - 1) >> 5] >> ((p - 1) & 31u)) & 1u);
            const unsigned long long bal = __ballot(v && (!pv || h[p] != h[p - 1]));
            if ((tid & 63u) == 0) { sw[(p >> 5)] = (uint32_t)bal; sw[(p >> 5) + 1] = (uint32_t)(bal >> 32); }
        }
        __syncthreads();
        // records: position p starts one if it is a k-mer start whose distance from its run's start is a multiple of WS_RMAX; it holds the k-mers up
        // to the run's end (the next position that is no k-mer start or starts another run), WS_RMAX at most.  Two rounds: count, then write.
        auto rec_len = [&](uint32_t p) -> uint32_t {      // 0: no record starts here
            if (!((vw[p >> 5] >> (p & 31u)) & 1u)) return 0u;
            uint32_t w = p >> 5;
            uint32_t m = sw[w] & (0xFFFFFFFFu >> (31u - (p & 31u)));              // run starts at or before p (there is one: a k-mer start after a gap is one)
            while (!m) m = sw[--w];
            const uint32_t ls = w * 32u + 31u - (uint32_t)__builtin_clz(m);
            if ((p - ls) % (uint32_t)WS_RMAX) return 0u;
            w = p >> 5;
            m = (~vw[w] | sw[w]) & ~((2u << (p & 31u)) - 1u);                     // stops behind p
            uint32_t e = WS_TILE;
            for (;;) {
                if (m) { e = w * 32u + (uint32_t)__builtin_ctz(m); break; }
                if (++w >= (uint32_t)(WS_TILE / 32)) break;
                m = ~vw[w] | sw[w];
            }
            if (e > (uint32_t)WS_TILE) e = WS_TILE;
            return e - p < (uint32_t)WS_RMAX ? e - p : (uint32_t)WS_RMAX;
        };
        uint32_t mine = 0;
        for (uint32_t p = tid; p < WS_TILE; p += WS_T) {  // (rec_len is 40 instructions and ran twice: its answers wait in LDS for the second round)
            const uint32_t l = rec_len(p);
            rl[p] = (uint8_t)l;
            mine += l ? 1u : 0u;
        }
        uint32_t tot;
        uint32_t at = mf_block_excl_scan(mine, scratch, &tot);
        if (tid == 0) s_base = tot ? atomicAdd(cursor, (unsigned long long)tot) : 0ull;
        __syncthreads();
        for (uint32_t p = tid; p < WS_TILE; p += WS_T) {
            const uint32_t len = rl[p];                  // (written by this thread)
            if (!len) continue;
            const unsigned long long g = s_base + at;
            at++;
            if (g < cap) {
                uint32_t R[7];
#pragma unroll
                for (int j = 0; j < 7; j++) R[j] = ws_word_at(pk, p + 16u * (uint32_t)j);
                *reinterpret_cast<uint4 *>(&recs[g].w[0]) = make_uint4(R[0], R[1], R[2], R[3]);
                *reinterpret_cast<uint4 *>(&recs[g].w[4]) = make_uint4(R[4], R[5], R[6], len);
                // (the records' CONTENT as low bits of the sort key -- identical records next to each other inside their partition, for k_wskm_pack's windows -- was
                // measured: k_wskm_count 412 -> 408 ms, k_wskm_scan 177 -> 194: the windows meet most repeats as it is)
                part[g] = pbits ? mf_remix32(h[p]) >> (32 - pbits) : 0u;
            }
        }
    }
}

__global__ void k_wskm_iota(uint32_t *__restrict__ v, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (uint32_t)i;
}
// Counting units = RUNS OF WHOLE PARTITIONS of about G records: unit j starts at the first partition border at or behind record j G (a partition of
// more than G records makes the units it covers empty but the first).  The partitions are very unequal -- 200 M reads at k = 63, 2^23 of them: 77 %
// hold fewer than 128 records (a round of 1024 k-mers half empty, behind a table clear, two barriers' worth of scans and two sweeps of 4096 slots)
// and 5 % hold 44 % of all records -- so small neighbours share a table (their k-mers differ: a k-mer has one partition) and large ones stay alone.
__global__ void k_wskm_units(const uint64_t *__restrict__ poff, uint32_t np, uint64_t n_rec, uint64_t G, uint32_t nu, uint64_t *__restrict__ uoff) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > nu) return;
    const uint64_t t = (uint64_t)j * G;
    if (j == nu || t >= n_rec) { uoff[j] = n_rec; return; }
    uint32_t lo = 0, hi = np;                            // the first partition whose first record is at or behind t (poff[np] = n_rec >= t)
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (poff[mid] < t) lo = mid + 1; else hi = mid; }
    uoff[j] = poff[lo];
}
__global__ void k_wskm_offsets(const uint32_t *__restrict__ part_sorted, uint64_t n, uint32_t np, uint64_t *__restrict__ off) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p > np) return;
    uint64_t lo = 0, hi = n;                             // first position with part >= p
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (part_sorted[mid] < p) lo = mid + 1; else hi = mid; }
    off[p] = lo;
}

// two 32-bit hashes of a k-mer from 32-bit multiplies (a 64-bit multiply is a sequence of quarter-rate instructions, and this runs per occurrence):
// .x the table slot's, .y the refinement classes'
__device__ __forceinline__ uint2 wc_hash(uint64_t hi, uint64_t lo) {
    const uint32_t a = (uint32_t)lo, b = (uint32_t)(lo >> 32), c = (uint32_t)hi, d = (uint32_t)(hi >> 32);
    uint32_t x = a ^ (b * 0x9E3779B1u) ^ (c * 0x85EBCA6Bu) ^ (d * 0xC2B2AE35u);
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12;
    uint32_t y = b ^ (a * 0xC2B2AE35u) ^ (d * 0x9E3779B1u) ^ (c * 0x27D4EB2Fu);
    y ^= y >> 16; y *= 0x85EBCA6Bu; y ^= y >> 13;
    return make_uint2(x, y);
}

// counters: [0] kept entries written, [1] distinct k-mers before the cut, [2] units that needed more than one pass, [3] failures
// A THREAD PER K-MER: a unit has a hundred records or so of ~18 k-mers each -- a thread per record left three quarters of the workgroup idle and the
// rest rolling through their records one k-mer after the other (1.7 s for 1.76e10 k-mers).  The unit's records are parked in LDS a chunk at a time
// (WC_CH records), an exclusive scan of their k-mer counts numbers the chunk's k-mers, and thread i takes k-mer i: it finds its record in the scan
// (a binary search), cuts the k-mer out of the record's words (two funnelled 128-bit shifts) and reverses it for the other strand.
#define WC_CH 1024                                 // records of a chunk (= the threads: one loads a record each)
__global__ __launch_bounds__(WC_T) void k_wskm_count(const wskm_rec *__restrict__ recs, const uint32_t *__restrict__ order, const uint64_t *__restrict__ uoff, uint32_t n_units, int k,
                                                    int thr, uint64_t *__restrict__ out_hi, uint64_t *__restrict__ out_lo, uint16_t *__restrict__ out_cnt, uint64_t cap,
                                                    unsigned long long *__restrict__ counters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wc_smem[];       // 80 KiB table + 36 KiB of parked records: one workgroup of 1024 threads per CU
    unsigned long long *thi = reinterpret_cast<unsigned long long *>(wc_smem), *tlo = thi + WC_SLOTS;
    uint32_t *tcnt = reinterpret_cast<uint32_t *>(tlo + WC_SLOTS);
    uint32_t *srec = tcnt + WC_SLOTS;                                             // [WC_CH][8]
    uint32_t *soff = srec + WC_CH * 8;                                            // [WC_CH + 1]
    __shared__ uint32_t scratch[18];
    __shared__ uint32_t s_claims, s_over, s_sp;
    __shared__ uint32_t stk_v[64], stk_b[64];
    __shared__ unsigned long long s_base;
    const uint32_t tid = threadIdx.x;
#if WC_PROF
    unsigned long long pc[6] = {0, 0, 0, 0, 0, 0}, pt = clock64();           // (thread 0: cycles per phase -- 0 top + clear, 1 load + scan, 2 rounds, 3 sweep + write; 4: all of them in passes after an overflow; 5: rounds)
#define WC_TICK(i) do { if (tid == 0) { const unsigned long long n_ = clock64(); pc[i] += n_ - pt; if (!first) pc[4] += n_ - pt; pt = n_; } } while (0)
#else
#define WC_TICK(i) do { } while (0)
#endif
    unsigned long long dist_acc = 0;                     // (this thread's share of the distinct k-mers: ONE atomic per wave at the end -- a wave's atomic per
                                                         // unit on one address was 1.3e8 serialised atomics of 12 ns each: 1.6 of the kernel's 1.7 s)
    for (uint32_t u = blockIdx.x; u < n_units; u += gridDim.x) {
        const uint64_t a = uoff[u], b = uoff[u + 1];
        if (a == b) continue;                            // (uniform)
        // The unit's k-mers are counted as a whole, or -- when their distinct k-mers do not fit the table -- class by class of their hash: a class
        // (value v of the hash's `nb` low class bits) that overflows is split into four (two more bits), the classes that fitted are done: a stack
        // of classes, at most three per level of refinement.
        if (tid == 0) { s_sp = 1; stk_v[0] = 0; stk_b[0] = 0; }
        bool first = true;
        for (;;) {
            __syncthreads();
            if (s_sp == 0) break;                        // (uniform)
            const uint32_t cv = stk_v[s_sp - 1], cb = stk_b[s_sp - 1];
            const uint32_t cmask = (1u << cb) - 1u;
            __syncthreads();
            for (uint32_t s = tid; s < WC_SLOTS; s += WC_T) { thi[s] = WC_EMPTY; tcnt[s] = 0; }
            if (tid == 0) { s_sp--; s_claims = 0; s_over = 0; }
            for (uint64_t c0 = a; c0 < b; c0 += WC_CH) {                          // (uniform)
                __syncthreads();
                WC_TICK(c0 == a ? 0 : 2);
                if (s_over) break;                       // (uniform: read behind the barrier)
                const uint32_t nr = (uint32_t)(b - c0 < (uint64_t)WC_CH ? b - c0 : (uint64_t)WC_CH);
                uint32_t T;
                if (first || b - a > (uint64_t)WC_CH) {
                    uint32_t len = 0;
                    if (tid < nr) {
                        const wskm_rec *rp = order ? &recs[order[c0 + tid]] : &recs[c0 + tid];      // (order == nullptr: the records lie in their units' order, k_wskm_pack)
                        const uint4 A = *reinterpret_cast<const uint4 *>(&rp->w[0]), B = *reinterpret_cast<const uint4 *>(&rp->w[4]);
                        *reinterpret_cast<uint4 *>(&srec[tid * 8]) = A;
                        *reinterpret_cast<uint4 *>(&srec[tid * 8 + 4]) = B;
                        len = B.w & 0xFFu;                           // (w[7] = k-mers | the records this one stands for << 8: k_wskm_pack; 0 there = 1)
                    }
                    const uint32_t ex = mf_block_excl_scan(len, scratch, &T);
                    soff[tid] = ex;
                    if (tid == 0) soff[WC_CH] = T;
                    __syncthreads();
                } else T = soff[WC_CH];                  // (a unit of one chunk: its records and their scan are still parked from the first pass)
                WC_TICK(1);
#if WC_PROF
                if (tid == 0) pc[5] += (T + WC_T - 1) / WC_T;
#endif
                // A thread takes R CONSECUTIVE k-mers: it finds the record of the first (a binary search in the scan), cuts the k-mer out of the record's
                // words (two funnelled 128-bit shifts) and reverses it for the other strand; the following ones of the same record roll in a base at a
                // time (forward: shift left; the other strand: shift right).  R = 1 in a unit's first pass -- every k-mer goes into the table, and the
                // inserts of a thread run one after the other: R = 2, 4, 8 there measured 679, 770, 984 ms against 647 --, R = WC_RO in the passes over
                // ONE CLASS of an overflowing unit (half of the kernel's time: 2.5 % of the units, counted 5, 21, 85 ... times over), where most k-mers
                // are only made, looked at and dropped.
                const mf_u128 kmask = (~(mf_u128)0) >> (128 - 2 * k);
                const uint32_t R = cb == 0 ? 1u : (uint32_t)WC_RO;
                for (uint32_t g0 = tid * R; g0 < T; g0 += WC_T * R) {
                    if (*(volatile uint32_t *)&s_over) break;                   // (the table is filling up: nobody adds to it any more)
                    const uint32_t iend = g0 + R < T ? g0 + R : T;
                    uint32_t lo_r = 0, hi_r = nr;                                // the last record with soff <= g0
                    while (hi_r - lo_r > 1) { const uint32_t mid = (lo_r + hi_r) >> 1; if (soff[mid] <= g0) lo_r = mid; else hi_r = mid; }
                    uint32_t j = g0 - soff[lo_r], rend = soff[lo_r + 1];
                    uint32_t wb = lo_r * 8u;                                     // (the record's words: srec[wb ..])
                    mf_u128 fw = 0, rc = 0;
                    bool have = false;
                    uint32_t wgt = 1u;                                           // identical records this one stands for (k_wskm_pack)
                    for (uint32_t i = g0; i < iend; i++, j++) {
                        if (i >= rend) {                                         // the next record (a record without k-mers -- a repeat -- has no place in the scan)
                            do { lo_r++; rend = soff[lo_r + 1]; } while (rend <= i);
                            j = 0; wb = lo_r * 8u; have = false;
                        }
                        if (!have) {
                            const uint32_t q = wb + (j >> 4), o = (j & 15u) * 2u;   // the k-mer starts at bit 32 (j / 16) + o of the record's 224
                            const mf_u128 X0 = ((mf_u128)(((uint64_t)srec[q] << 32) | srec[q + 1]) << 64) | (mf_u128)(((uint64_t)srec[q + 2] << 32) | srec[q + 3]);
                            const mf_u128 X = o ? ((X0 << o) | (mf_u128)(srec[q + 4] >> (32u - o))) : X0;      // (j / 16 <= 2: that word is a base word, word 7 is never reached)
                            fw = X >> (128 - 2 * k);
                            rc = mf_wrevcomp(fw, k);
                            have = true;
                            wgt = srec[wb + 7u] >> 8;
                            wgt = wgt ? wgt : 1u;
                        } else {
                            const uint32_t bi = j + (uint32_t)k - 1u;            // the base that comes in
                            const uint32_t nb = (srec[wb + (bi >> 4)] >> (30u - 2u * (bi & 15u))) & 3u;
                            fw = ((fw << 2) | (mf_u128)nb) & kmask;
                            rc = (rc >> 2) | ((mf_u128)(3u - nb) << (2 * k - 2));
                        }
                        const mf_u128 cn = fw < rc ? fw : rc;
                        const unsigned long long hi = (unsigned long long)(cn >> 64), lo = (unsigned long long)cn;
                        const uint2 hh = wc_hash(hi, lo);
                        if ((hh.y & cmask) != cv) continue;
#if WC_ABL == 1
                        if (hh.x == 0x12345u && lo == 77ull) atomicAdd(&tcnt[hh.x & (WC_SLOTS - 1)], 1u);
                        continue;
#endif
                        uint32_t s = hh.x & (WC_SLOTS - 1);
                        for (uint32_t probes = 0;; probes++) {
                            if (probes >= (uint32_t)WC_SLOTS) { s_over = 1u; break; }
                            const unsigned long long old = atomicCAS(&thi[s], WC_EMPTY, WC_BUSY);
                            if (old == WC_EMPTY) {                               // the slot is this k-mer's: low word first, the high word publishes it
                                tlo[s] = lo;
                                __threadfence_block();
                                atomicExch(&thi[s], hi);
                                atomicAdd(&tcnt[s], wgt);
                                if (atomicAdd(&s_claims, 1u) + 1u > (uint32_t)WC_FILL) s_over = 1u;
                                break;
                            }
                            if (old == WC_BUSY) continue;                        // (being written: look again)
                            if (old == hi && *(volatile unsigned long long *)&tlo[s] == lo) { atomicAdd(&tcnt[s], wgt); break; }
                            s = (s + 1u) & (WC_SLOTS - 1);
                        }
                    }
                }
            }
            __syncthreads();
            WC_TICK(2);
            if (s_over) {                                // too many distinct k-mers for the table: the class in four
                if (cb + 2 > 24) { if (tid == 0) atomicAdd(&counters[3], 1ull); break; }      // (2800 x 2^24 distinct k-mers in one minimizer partition: not with 2k >= 64 bits of key)
                if (tid == 0) {
                    if (first) atomicAdd(&counters[2], 1ull);
                    // (as many classes at once as the part of the unit seen so far asks for -- distinct k-mers found / the share of the records seen, 16 to
                    // 256 classes -- was measured: 665 ms against 621; the estimate is too high where the k-mers repeat, and every class too many is a pass)
                    for (uint32_t c = 0; c < 4; c++) { stk_v[s_sp] = cv | (c << cb); stk_b[s_sp] = cb + 2; s_sp++; }
                }
                first = false;
                continue;
            }
            first = false;
            // the entries that pass the cut
            uint32_t keep = 0, dist = 0;
            for (uint32_t s = tid; s < WC_SLOTS; s += WC_T) if (thi[s] != WC_EMPTY) { dist++; if ((int)tcnt[s] > thr) keep++; }
            uint32_t tot;
            const uint32_t ex = mf_block_excl_scan(keep, scratch, &tot);
            dist_acc += dist;
            if (tid == 0) s_base = tot ? atomicAdd(&counters[0], (unsigned long long)tot) : 0ull;
            __syncthreads();
            unsigned long long at = s_base + ex;
            for (uint32_t s = tid; s < WC_SLOTS; s += WC_T)
                if (thi[s] != WC_EMPTY && (int)tcnt[s] > thr) {
                    if (at < cap) { out_hi[at] = thi[s]; out_lo[at] = tlo[s]; out_cnt[at] = (uint16_t)(tcnt[s] > (uint32_t)MF_MAX_COUNT ? (uint32_t)MF_MAX_COUNT : tcnt[s]); }
                    at++;
                }
            WC_TICK(3);
        }
    }
#if WC_PROF
    if (tid == 0) for (int i = 0; i < 6; i++) atomicAdd(&counters[8 + i], pc[i]);
#endif
    for (int d = 32; d >= 1; d >>= 1) dist_acc += __shfl_down(dist_acc, d, 64);
    if ((tid & 63u) == 0 && dist_acc) atomicAdd(&counters[1], dist_acc);
}

// The records in the order of their partitions, IDENTICAL ONES ONCE (as the k <= 31 count does: reads of the same place give the same super-k-mers -- 21 % of
// the 200 M-read sample's records repeat an earlier one, with 27 % of the k-mers): a window of 1024 sorted records is parked in LDS, a record that finds an
// identical one in the window's set before it gives up its k-mers (w[7] = 0 k-mers) and the first one's weight grows (w[7] = k-mers | weight << 8).
// Identical records have the same minimizer, so the same partition and unit: wherever the windows' borders fall, the counts stay right.  The counting
// kernel then reads its records one after the other instead of through the sort's index.
__global__ __launch_bounds__(1024) void k_wskm_pack(const wskm_rec *__restrict__ recs, const uint32_t *__restrict__ order, uint64_t n, uint32_t dedupe, wskm_rec *__restrict__ out) {
    __shared__ uint32_t srec[1024 * 8];
    __shared__ uint32_t hset[2048];
    __shared__ uint32_t rw[1024];
    const uint32_t tid = threadIdx.x;
    const uint64_t i = (uint64_t)blockIdx.x * 1024u + tid;
    for (uint32_t q = tid; q < 2048u; q += 1024u) hset[q] = 0xFFFFFFFFu;
    uint4 A = make_uint4(0, 0, 0, 0), B = A;
    uint32_t h = 0;
    if (i < n) {
        const wskm_rec *rp = &recs[order[i]];
        A = *reinterpret_cast<const uint4 *>(&rp->w[0]); B = *reinterpret_cast<const uint4 *>(&rp->w[4]);
        *reinterpret_cast<uint4 *>(&srec[tid * 8]) = A;
        *reinterpret_cast<uint4 *>(&srec[tid * 8 + 4]) = B;
        h = (A.x * 0x9E3779B1u) ^ (A.y * 0x85EBCA6Bu) ^ (A.z * 0xC2B2AE35u) ^ (A.w * 0x27D4EB2Fu) ^ (B.x * 0x165667B1u) ^ (B.y * 0xD3A2646Cu) ^ (B.z * 0xFD7046C5u) ^ B.w;
        h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
    }
    rw[tid] = 1u;
    __syncthreads();
    bool repeat = false;
    if (i < n && dedupe) {
        uint32_t sl = h & 2047u;
        for (;;) {
            const uint32_t old = atomicCAS(&hset[sl], 0xFFFFFFFFu, tid);
            if (old == 0xFFFFFFFFu) break;
            bool same = true;
#pragma unroll
            for (int w = 0; w < 8; w++) same &= srec[old * 8 + w] == srec[tid * 8 + w];
            if (same) { atomicAdd(&rw[old], 1u); repeat = true; break; }
            sl = (sl + 1u) & 2047u;
        }
    }
    __syncthreads();
    if (i < n) {
        B.w = repeat ? 0u : ((B.w & 0xFFu) | (rw[tid] << 8));
        *reinterpret_cast<uint4 *>(&out[i].w[0]) = A;
        *reinterpret_cast<uint4 *>(&out[i].w[4]) = B;
    }
}

__global__ void k_wskm_gather64(const uint64_t *__restrict__ src, const uint32_t *__restrict__ idx, uint64_t n, uint64_t *__restrict__ dst) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}
__global__ void k_wskm_gather16(const uint16_t *__restrict__ src, const uint32_t *__restrict__ idx, uint64_t n, uint16_t *__restrict__ dst) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

// ---- the kept entries in ascending order.  The k-mers are distinct and spread evenly, so the LEADING 32 BITS nearly order them: 7.5e8 kept entries
// of the 200 M-read sample fall into 2^32 values, 0.17 per value.  (leading bits, entry number) pairs are sorted -- 4 passes over 8 bytes instead
// of the 16 passes over 12 bytes a full sort of 126 bits takes --, the entries are gathered in that order, and an entry finds its place inside its
// RUN of equal leading bits by counting the run's smaller entries.  A run of more than WO_RUN entries (k-mers that share their first 16 bases by
// the thousand: poly-A tails, satellites) goes ASIDE with its place: all bits of those entries are sorted, and the i-th smallest of them takes the i-th smallest
// of their places -- the runs are ranges of places, and a run with smaller leading bits holds smaller k-mers.  More than an eighth of the entries aside: all
// bits of everything are sorted (option wide_skm_lead = 0: always; 2: tests -- no room aside).
#define WO_RUN 64
__global__ void k_wskm_lead(const uint64_t *__restrict__ hi, const uint64_t *__restrict__ lo, uint64_t n, int k, uint32_t *__restrict__ lead, uint32_t *__restrict__ idx) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const mf_u128 x = ((mf_u128)hi[i] << 64) | (mf_u128)lo[i];
    lead[i] = (uint32_t)(x >> (2 * k - 32));
    idx[i] = (uint32_t)i;
}
// hi / lo / cnt: the entries in the order of their leading bits already (gathered: the run's other entries are then neighbours in memory, not random lines)
__global__ void k_wskm_place(const uint32_t *__restrict__ lead, const uint64_t *__restrict__ hi, const uint64_t *__restrict__ lo,
                             const uint16_t *__restrict__ cnt, uint64_t n, uint64_t *__restrict__ ohi, uint64_t *__restrict__ olo, uint16_t *__restrict__ ocnt,
                             unsigned int *__restrict__ side_n, uint32_t *__restrict__ side_pos, uint64_t *__restrict__ shi, uint64_t *__restrict__ slo,
                             uint16_t *__restrict__ scnt, uint64_t scap) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t t = lead[i];
    uint64_t a = i, b = i + 1;
    while (a > 0 && i - a < (uint64_t)WO_RUN && lead[a - 1] == t) a--;
    while (b < n && b - i <= (uint64_t)WO_RUN && lead[b] == t) b++;
    const uint64_t h = hi[i], l = lo[i];
    if (b - a > (uint64_t)WO_RUN) {                                        // a long run (every entry of it sees that: a search that stops at its bound has seen WO_RUN others): aside, with its place
        const unsigned int at = atomicAdd(side_n, 1u);
        if ((uint64_t)at < scap) { side_pos[at] = (uint32_t)i; shi[at] = h; slo[at] = l; scnt[at] = cnt[i]; }
        return;
    }
    uint64_t rank = 0;
    for (uint64_t j = a; j < b; j++) {
        const uint64_t h2 = hi[j], l2 = lo[j];
        rank += (h2 < h || (h2 == h && l2 < l)) ? 1u : 0u;
    }
    ohi[a + rank] = h; olo[a + rank] = l; ocnt[a + rank] = cnt[i];
}
__global__ void k_wskm_scatter_side(const uint32_t *__restrict__ pos, const uint64_t *__restrict__ hi, const uint64_t *__restrict__ lo, const uint16_t *__restrict__ cnt, uint64_t n,
                                    uint64_t *__restrict__ ohi, uint64_t *__restrict__ olo, uint16_t *__restrict__ ocnt) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const uint32_t p = pos[i]; ohi[p] = hi[i]; olo[p] = lo[i]; ocnt[p] = cnt[i]; }
}

// nk distinct k-mers (src_hi, src_lo, src_cnt) in any order -> *pc: ascending (high, low)
int mf_wide_order(mf_ctx *ctx, int k, const uint64_t *src_hi, const uint64_t *src_lo, const uint16_t *src_cnt, uint64_t nk, mf_wtable::piece *pc) {
    hipStream_t st = ctx->stream;
    MF_TRY(pc->hi.alloc(ctx, nk)); MF_TRY(pc->lo.alloc(ctx, nk)); MF_TRY(pc->cnt.alloc(ctx, nk));
    if (nk) {
        mf_ktimer tm(ctx, "k_wskm_order");
        // all bits of n entries sorted: by the low word carrying the entry number, then -- stable -- by the high word's 2k - 64 bits
        auto full_order = [&](const uint64_t *shi, const uint64_t *slo, const uint16_t *scnt, uint64_t n, uint64_t *dhi, uint64_t *dlo, uint16_t *dcnt) -> int {
            mf_buf<uint32_t> i0, i1, i2; mf_buf<uint64_t> k1, k2, h1;
            MF_TRY(i0.alloc(ctx, n)); MF_TRY(i1.alloc(ctx, n)); MF_TRY(k1.alloc(ctx, n));
            k_wskm_iota<<<wsgrid(n), 256, 0, st>>>(i0.p, n);
            MF_TRY(mf_sort_u64_u32(ctx, slo, i0.p, n, 64, k1.p, i1.p));
            i0.reset();
            const int hb = 2 * k - 64;
            if (hb > 0) {
                MF_TRY(h1.alloc(ctx, n)); MF_TRY(k2.alloc(ctx, n)); MF_TRY(i2.alloc(ctx, n));
                k_wskm_gather64<<<wsgrid(n), 256, 0, st>>>(shi, i1.p, n, h1.p);
                MF_TRY(mf_sort_u64_u32(ctx, h1.p, i1.p, n, hb, k2.p, i2.p));
            } else i2.swap(i1);
            k_wskm_gather64<<<wsgrid(n), 256, 0, st>>>(shi, i2.p, n, dhi);
            k_wskm_gather64<<<wsgrid(n), 256, 0, st>>>(slo, i2.p, n, dlo);
            k_wskm_gather16<<<wsgrid(n), 256, 0, st>>>(scnt, i2.p, n, dcnt);
            MF_HIP(hipStreamSynchronize(st));
            return MF_OK;
        };
        bool placed = false;
        if (ctx->opt_wide_skm_lead) {
            mf_buf<uint32_t> l0, j0, l1, j1; mf_buf<unsigned int> side_n;
            MF_TRY(l0.alloc(ctx, nk)); MF_TRY(j0.alloc(ctx, nk)); MF_TRY(l1.alloc(ctx, nk)); MF_TRY(j1.alloc(ctx, nk)); MF_TRY(side_n.alloc(ctx, 1));
            MF_HIP(hipMemsetAsync(side_n.p, 0, 4, st));
            k_wskm_lead<<<wsgrid(nk), 256, 0, st>>>(src_hi, src_lo, nk, k, l0.p, j0.p);
            MF_TRY(mf_sort_u32_pairs(ctx, l0.p, j0.p, nk, 32, l1.p, j1.p));
            j0.reset();
            // (the entries of long runs: their places in l0 -- free again --, themselves aside; room for an eighth of all, or all bits of everything are sorted)
            const uint64_t scap = ctx->opt_wide_skm_lead == 2 ? 0 : nk / 8 + 1024;
            mf_buf<uint64_t> shi, slo; mf_buf<uint16_t> scnt;
            MF_TRY(shi.alloc(ctx, scap + 1)); MF_TRY(slo.alloc(ctx, scap + 1)); MF_TRY(scnt.alloc(ctx, scap + 1));
            mf_buf<uint64_t> ghi, glo; mf_buf<uint16_t> gcnt;
            MF_TRY(ghi.alloc(ctx, nk)); MF_TRY(glo.alloc(ctx, nk)); MF_TRY(gcnt.alloc(ctx, nk));
            k_wskm_gather64<<<wsgrid(nk), 256, 0, st>>>(src_hi, j1.p, nk, ghi.p);
            k_wskm_gather64<<<wsgrid(nk), 256, 0, st>>>(src_lo, j1.p, nk, glo.p);
            k_wskm_gather16<<<wsgrid(nk), 256, 0, st>>>(src_cnt, j1.p, nk, gcnt.p);
            k_wskm_place<<<wsgrid(nk), 256, 0, st>>>(l1.p, ghi.p, glo.p, gcnt.p, nk, pc->hi.p, pc->lo.p, pc->cnt.p, side_n.p, l0.p, shi.p, slo.p, scnt.p, scap);
            unsigned int ns = 0;
            MF_HIP(hipMemcpyAsync(&ns, side_n.p, 4, hipMemcpyDeviceToHost, st));
            MF_HIP(hipStreamSynchronize(st));
            l1.reset(); j1.reset();
            if ((uint64_t)ns <= scap) {
                placed = true;
                if (ns) {
                    // ascending places x ascending k-mers: the runs are ranges of places, and the k-mers of a run with smaller leading bits are smaller
                    mf_buf<uint32_t> pos, zero, dummy; mf_buf<uint64_t> thi, tlo; mf_buf<uint16_t> tcnt;
                    MF_TRY(pos.alloc(ctx, ns)); MF_TRY(zero.alloc(ctx, ns)); MF_TRY(dummy.alloc(ctx, ns)); MF_TRY(thi.alloc(ctx, ns)); MF_TRY(tlo.alloc(ctx, ns)); MF_TRY(tcnt.alloc(ctx, ns));
                    MF_HIP(hipMemsetAsync(zero.p, 0, (size_t)ns * 4, st));
                    MF_TRY(mf_sort_u32_pairs(ctx, l0.p, zero.p, ns, 32, pos.p, dummy.p));
                    MF_TRY(full_order(shi.p, slo.p, scnt.p, ns, thi.p, tlo.p, tcnt.p));
                    k_wskm_scatter_side<<<wsgrid(ns), 256, 0, st>>>(pos.p, thi.p, tlo.p, tcnt.p, ns, pc->hi.p, pc->lo.p, pc->cnt.p);
                    MF_HIP(hipStreamSynchronize(st));
                }
                if (ctx->opt_verbose) fprintf(stderr, "[mf] count_wide (records): %u of %llu kept k-mers stand in runs of more than %d with the same leading 16 bases (all their bits are sorted)\n", ns, (unsigned long long)nk, WO_RUN);
            }
        }
        if (!placed) MF_TRY(full_order(src_hi, src_lo, src_cnt, nk, pc->hi.p, pc->lo.p, pc->cnt.p));
    }
    pc->n = nk;
    return MF_OK;
}

// 0: *t filled (one ascending piece, the cut made); 1: not an input for this path (nothing changed); < 0: error
int mf_count_wide_skm(mf_ctx *ctx, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n_reads, uint64_t n_bases, int k, int min_read_len, int threshold,
                      const uint32_t *vmask, uint64_t n_words, mf_wtable *t) {
    hipStream_t st = ctx->stream;
    mf_buf<unsigned long long> ctr; MF_TRY(ctr.alloc(ctx, 16));
    MF_HIP(hipMemsetAsync(ctr.p, 0, 128, st));
    k_wskm_nocc<<<wsgrid(n_reads), 256, 0, st>>>(d_offsets, n_reads, k, min_read_len, &ctr.p[4]);
    unsigned long long n_occ = 0;
    MF_HIP(hipMemcpyAsync(&n_occ, &ctr.p[4], 8, hipMemcpyDeviceToHost, st));
    MF_HIP(hipStreamSynchronize(st));
    if (n_occ < (uint64_t)std::max<int64_t>(1, ctx->opt_wide_skm_min)) return 1;
    // (the kept entries are ordered with 32-bit entry numbers: an UNCUT table of a sample this large -- 4.1e9 distinct 63-mers at 200 M reads -- goes
    // the old way, which leaves it in per-pass pieces; with the cut of the k-mer counter, count > b, a twentieth is kept)
    if (threshold < 1 && n_occ >= 3000000000ull) return 1;
    const uint64_t n_tiles = (n_words * 32 + WS_TILE - 1) / WS_TILE;
    // partitions: units of about 4000 k-mer occurrences (a unit whose distinct k-mers do not fit the LDS table is counted in passes)
    int pbits = 0;
    // (partitions of a share of a unit's occurrences where neighbours are merged into units: what is heavy because several minimizers met in it comes apart)
    const int64_t per_part = ctx->opt_wide_skm_merge ? std::max<int64_t>(64, ctx->opt_wide_skm_unit / std::max<int64_t>(1, ctx->opt_wide_skm_fine)) : std::max<int64_t>(256, ctx->opt_wide_skm_unit);
    while (pbits < 28 && (n_occ >> pbits) > (uint64_t)per_part) pbits++;
    // records: a run is (k - 13) / 2 k-mers long on average where reads, tiles and the record format do not cut it shorter
    uint64_t cap = n_occ / 10 + n_tiles * 4 + 4096;
    mf_buf<wskm_rec> recs; mf_buf<uint32_t> part;
    unsigned long long n_rec = 0;
    const unsigned sgrid = (unsigned)std::min<uint64_t>(n_tiles, (uint64_t)ctx->n_cu * 4);
    for (int attempt = 0; attempt < 2; attempt++) {
        if (recs.alloc(ctx, cap) != MF_OK || part.alloc(ctx, cap) != MF_OK) { (void)hipGetLastError(); return 1; }
        MF_HIP(hipMemsetAsync(&ctr.p[5], 0, 8, st));
        {
            mf_ktimer tm(ctx, "k_wskm_scan");
            k_wskm_scan<<<sgrid, WS_T, 0, st>>>(d_bases, n_bases, vmask, n_words, k, pbits, recs.p, part.p, cap, &ctr.p[5]);
        }
        MF_HIP(hipMemcpyAsync(&n_rec, &ctr.p[5], 8, hipMemcpyDeviceToHost, st));
        MF_HIP(hipStreamSynchronize(st));
        if (n_rec <= cap) break;
        if (attempt == 1) return mf_set_error("count (wide records): internal error, %llu records for a room of %llu", n_rec, (unsigned long long)cap);
        recs.reset(); part.reset();
        cap = n_rec + 64;                                                        // (the cursor went on counting: the exact number)
    }
    if (n_rec >= (1ull << 32)) return 1;                                         // (the pair sort takes 2^32 entries; the old way cuts its passes by class)
    t->n_occ = n_occ;
    if (!n_rec) { t->n = 0; t->n_all = 0; t->cut_thr = threshold >= 1 ? threshold : 0; return MF_OK; }
    // by partition
    const uint32_t n_parts = 1u << pbits;
    uint32_t n_units = n_parts;
    mf_buf<uint32_t> order; mf_buf<uint64_t> uoff;
    {
        mf_buf<uint32_t> idx, part_s;
        if (idx.alloc(ctx, n_rec) != MF_OK || part_s.alloc(ctx, n_rec) != MF_OK || order.alloc(ctx, n_rec) != MF_OK) { (void)hipGetLastError(); return 1; }
        mf_buf<uint64_t> poff;
        MF_TRY(poff.alloc(ctx, (size_t)n_parts + 1));
        k_wskm_iota<<<wsgrid(n_rec), 256, 0, st>>>(idx.p, n_rec);
        if (pbits) MF_TRY(mf_sort_u32_pairs(ctx, part.p, idx.p, n_rec, pbits, part_s.p, order.p));
        else { MF_HIP(hipMemcpyAsync(order.p, idx.p, n_rec * 4, hipMemcpyDeviceToDevice, st)); MF_HIP(hipMemcpyAsync(part_s.p, part.p, n_rec * 4, hipMemcpyDeviceToDevice, st)); }
        k_wskm_offsets<<<(n_parts + 1 + 255) / 256, 256, 0, st>>>(part_s.p, n_rec, n_parts, poff.p);
        if (ctx->opt_wide_skm_merge && pbits) {
            // (records per unit: the unit's k-mer occurrences / the k-mers a record holds)
            const uint64_t G = std::max<uint64_t>(8, (uint64_t)((double)std::max<int64_t>(ctx->opt_wide_skm_unit, 64) * (double)n_rec / (double)n_occ));
            const uint64_t nu64 = (n_rec + G - 1) / G;
            if (nu64 + 1 < (1ull << 31)) {
                n_units = (uint32_t)nu64;
                MF_TRY(uoff.alloc(ctx, (size_t)n_units + 1));
                k_wskm_units<<<(n_units + 1 + 255) / 256, 256, 0, st>>>(poff.p, n_parts, n_rec, G, n_units, uoff.p);
            }
        }
        if (!uoff.p) uoff.swap(poff);
        if (ctx->opt_verbose >= 2) {                      // the units' sizes in records: how heavy the heavy ones are
            std::vector<uint64_t> h((size_t)n_units + 1);
            MF_HIP(hipMemcpyAsync(h.data(), uoff.p, ((size_t)n_units + 1) * 8, hipMemcpyDeviceToHost, st));
            MF_HIP(hipStreamSynchronize(st));
            uint64_t bins[12] = {0}, recs_in[12] = {0}, mx = 0;
            for (uint32_t u = 0; u < n_units; u++) {
                const uint64_t r = h[u + 1] - h[u]; mx = std::max(mx, r);
                int b = 0; while (b < 11 && r >= (128ull << b)) b++;
                bins[b]++; recs_in[b] += r;
            }
            fprintf(stderr, "[mf] count_wide (records): units by records (largest %llu):", (unsigned long long)mx);
            for (int b = 0; b < 12; b++) fprintf(stderr, " <%llu: %llu units, %.1f %% of the records;", (unsigned long long)(128ull << b), (unsigned long long)bins[b], 100.0 * recs_in[b] / (double)n_rec);
            fprintf(stderr, "\n");
        }
        MF_HIP(hipStreamSynchronize(st));
    }
    part.reset();
    // the records in their units' order, identical ones once (option wide_skm_pack: 0 -- through the sort's index, every record for itself; 1; 2 -- in order, no weights)
    if (ctx->opt_wide_skm_pack) {
        mf_buf<wskm_rec> packed;
        if (packed.alloc(ctx, n_rec) == MF_OK) {
            {
                mf_ktimer tm(ctx, "k_wskm_pack");
                k_wskm_pack<<<(unsigned)((n_rec + 1023) / 1024), 1024, 0, st>>>(recs.p, order.p, n_rec, ctx->opt_wide_skm_pack == 1 ? 1u : 0u, packed.p);
            }
            MF_HIP(hipStreamSynchronize(st));
            recs.swap(packed); packed.reset(); order.reset();
        } else (void)hipGetLastError();
    }
    // count: the kept entries' number is not known before -- a room sized from the cut, and once more with the exact size if it was too small
    uint64_t ocap = std::max<uint64_t>(1 << 20, threshold >= 1 ? n_occ / 12 : (n_occ < 1000000000ull ? n_occ : n_occ / 2));
    mf_buf<uint64_t> ohi, olo; mf_buf<uint16_t> ocnt;
    unsigned long long res[4] = {0, 0, 0, 0};
    const unsigned cgrid = (unsigned)std::min<uint64_t>(n_units, (uint64_t)ctx->n_cu);
    const size_t wc_lds = (size_t)WC_SLOTS * 20 + (size_t)1024 * 32 + (size_t)1025 * 4;
    MF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_wskm_count), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wc_lds));
    for (int attempt = 0; attempt < 2; attempt++) {
        if (ohi.alloc(ctx, ocap) != MF_OK || olo.alloc(ctx, ocap) != MF_OK || ocnt.alloc(ctx, ocap) != MF_OK) { (void)hipGetLastError(); return 1; }
        MF_HIP(hipMemsetAsync(ctr.p, 0, 32, st));
        {
            mf_ktimer tm(ctx, "k_wskm_count");
            k_wskm_count<<<cgrid, WC_T, wc_lds, st>>>(recs.p, order.p, uoff.p, n_units, k, threshold >= 1 ? threshold : 0, ohi.p, olo.p, ocnt.p, ocap, ctr.p);
        }
        MF_HIP(hipMemcpyAsync(res, ctr.p, 32, hipMemcpyDeviceToHost, st));
        MF_HIP(hipStreamSynchronize(st));
        if (res[3]) return mf_set_error("count (wide records): a unit did not fit the table in 2^24 classes");
        if (res[0] <= ocap) break;
        if (attempt == 1) return mf_set_error("count (wide records): internal error, %llu kept entries for a room of %llu", res[0], (unsigned long long)ocap);
        ohi.reset(); olo.reset(); ocnt.reset();
        ocap = res[0] + 64;
    }
#if WC_PROF
    {
        unsigned long long pc[6];
        MF_HIP(hipMemcpy(pc, ctr.p + 8, 48, hipMemcpyDeviceToHost));
        const double tot = (double)(pc[0] + pc[1] + pc[2] + pc[3]);
        fprintf(stderr, "[mf] k_wskm_count workgroup cycles: top + clear %.1f %%, load + scan %.1f %%, rounds %.1f %%, sweep + write %.1f %%; in passes behind an overflow %.1f %%; %llu rounds of %d k-mers for %llu k-mers\n",
                100.0 * pc[0] / tot, 100.0 * pc[1] / tot, 100.0 * pc[2] / tot, 100.0 * pc[3] / tot, 100.0 * pc[4] / tot, pc[5], WC_T, n_occ);
    }
#endif
    recs.reset(); order.reset(); uoff.reset();
    const uint64_t nk = res[0];
    if (ctx->opt_verbose) fprintf(stderr, "[mf] count_wide (records): %llu k-mers in %llu records (%.1f per record), %u units, %llu counted in several passes; %llu distinct, %llu kept\n", n_occ, n_rec,
                                  (double)n_occ / (double)n_rec, n_units, res[2], res[1], (unsigned long long)nk);
    if (nk >= (1ull << 32)) { t->n_occ = 0; return 1; }                         // (more kept k-mers than entry numbers: the old way)
    // The table stays in the order of the counting units (option wide_skm_lazy_order, the default) until somebody needs it ascending -- an export, the
    // pieces' views, the component cutter (mf_wtable_ensure_ascending): unitigs and features find k-mers through the index and compare k-mers, not places,
    // and ordering 7.5e8 kept entries is 0.2 s of the 200 M-read sample's 1.4 s.
    auto pc = std::make_unique<mf_wtable::piece>();
    if (ctx->opt_wide_skm_lazy_order) {
        pc->hi.swap(ohi); pc->lo.swap(olo); pc->cnt.swap(ocnt);
        t->ascending = false;
    } else {
        MF_TRY(mf_wide_order(ctx, k, ohi.p, olo.p, ocnt.p, nk, pc.get()));
        t->ascending = true;
    }
    pc->n = nk;
    t->pieces.clear();
    if (nk) t->pieces.push_back(std::move(pc));
    t->n = nk; t->n_all = res[1];
    t->cut_thr = threshold >= 1 ? threshold : 0;
    return MF_OK;
}

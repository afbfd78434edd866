// mf_dparse.hip -- read files parsed ON THE DEVICE (round 5).
//
// The reference parses reads serially under a monitor (src/io/ReadsDispatcher.java:34-53: BufferedReader.readLine + a switch per character) --
// its Amdahl limit --, and rounds 1-4 of this build parsed on up to 64 host threads at 15 GB/s of FASTA while the raw bytes cross to the
// device at 56 GB/s.  Here the file's bytes go to HBM as they are (8 MB pieces, pread + hipMemcpyAsync, no host thread looks at them) and
// kernels do what FastaReader / FastqReader do:
//
//   FASTA (itmo!/io/readers/FastaReader.java:53-104): lines end at '\n' (a '\r' before it is dropped); a line that starts with '>' or ';' is a
//     header / comment and ends the record before it; a record = the concatenation of the lines between two such lines; a record with an
//     N / n is skipped (:53-76); characters through DnaTools.fromChar (itmo!/dna/DnaTools.java:46-64, IUPAC codes -> first listed base).
//   FASTQ (FastqReader.java:53-115, FastaReaderFromXQSource.java:66-70, ReadersUtils.java:63-77): records of four lines; a read with an
//     N / n / '.' or a base of phred 0 (quality char == offset; the offset is sniffed on the host from the first 1000 records) is dropped.
//
// Everything is byte-parallel: a workgroup takes a 256 KB chunk, a thread 16 bytes at a time; line starts, header lines and sequence bytes are
// counted per chunk (pass 1), a one-workgroup scan turns the counts into prefixes, pass 2 writes what a record needs (FASTA: the number of
// sequence bytes before each header line, N flags; FASTQ: the start of every line, drop flags), a thread per record decides keep / drop and a
// scan gives every kept record its place, pass 3 moves the sequence bytes of the kept records -- translated to upper-case A / C / G / T --
// to their place through LDS (64-byte-coalesced stores).  The result is the (bases, offsets) pair the host parsers make, byte for byte.
//
// The device parser takes the files it is SURE about.  Anything else -- a lone '\r', an empty line inside a FASTQ file, a character that is no
// nucleotide, a quality outside [offset, 126], a truncated record, lengths that differ -- makes it step back (return 1) and the host parser
// reads the file, with the reference's error messages.  .gz / .bz2 / .binq inputs never come here.
#include "mf_common.h"
#include "mf_parse.h"
#include <fcntl.h>
#include <unistd.h>
#include <sys/stat.h>
#include <atomic>
#include <mutex>
#include <thread>

#define DP_T 256                           // threads per workgroup
#define DP_V 16                            // bytes per thread and tile
#define DP_TILE (DP_T * DP_V)              // 4096
#define DP_TPC 64                          // tiles per chunk
#define DP_CHUNK ((uint64_t)DP_TILE * DP_TPC)   // 256 KB
#define DP_DROP (~0ull)
#define DP_RPB 2048                        // records per workgroup of the per-record kernels (8 per thread)

// anomaly bits (any of them: the host parser takes the file)
#define DP_A_LONE_CR 1u
#define DP_A_BAD_CHAR 2u
#define DP_A_EMPTY_LINE 4u
#define DP_A_BAD_QUAL 8u
#define DP_A_STRUCTURE 16u
#define DP_A_LENGTHS 32u

// DnaTools.fromChar as a table (mf_parse.h: BASE_LUT): upper-case base, 0xFE = N / n, 0xFF = no nucleotide
__device__ __forceinline__ uint32_t dp_lut(const uint8_t *lut, uint32_t c) { return lut[c]; }
__device__ __forceinline__ void dp_fill_lut(uint8_t *lut) {
    // "ACGTacgtRrYyMmKkSsWwHhBbVvDd" -> "ACGTACGTGGTTAAGGGGAAAAGGAAAA" (nucleotide_of, mf_parse.h)
    for (uint32_t c = threadIdx.x; c < 256; c += blockDim.x) {
        uint8_t v = 0xFF;
        switch (c) {
        case 'A': case 'a': case 'M': case 'm': case 'W': case 'w': case 'H': case 'h': case 'V': case 'v': case 'D': case 'd': v = 'A'; break;
        case 'C': case 'c': v = 'C'; break;
        case 'G': case 'g': case 'R': case 'r': case 'K': case 'k': case 'S': case 's': case 'B': case 'b': v = 'G'; break;
        case 'T': case 't': case 'Y': case 'y': v = 'T'; break;
        case 'N': case 'n': v = 0xFE; break;
        default: break;
        }
        lut[c] = v;
    }
}
// the 16 bytes of a thread + the byte before and the byte behind them ('\n' stands for "outside the file": position 0 starts a line, and a
// '\r' that ends the file ends a line)
struct dp_bytes { uint32_t w[4]; uint32_t prev, next; uint32_t cnt; };
__device__ __forceinline__ dp_bytes dp_load(const uint8_t *__restrict__ raw, uint64_t n, uint64_t base) {
    dp_bytes b;
    b.cnt = base >= n ? 0u : (uint32_t)(n - base < DP_V ? n - base : DP_V);
    if (b.cnt) {
        const uint4 v = *reinterpret_cast<const uint4 *>(raw + base);           // (the buffer is padded: whole vectors can be read)
        b.w[0] = v.x; b.w[1] = v.y; b.w[2] = v.z; b.w[3] = v.w;
        b.prev = base ? raw[base - 1] : (uint32_t)'\n';
        b.next = base + DP_V < n ? raw[base + DP_V] : (uint32_t)'\n';
    } else { b.w[0] = b.w[1] = b.w[2] = b.w[3] = 0x0A0A0A0Au; b.prev = b.next = '\n'; }
    return b;
}
__device__ __forceinline__ uint32_t dp_byte(const dp_bytes &b, int j) { return (b.w[j >> 2] >> (8 * (j & 3))) & 0xFFu; }
__device__ __forceinline__ uint32_t dp_after(const dp_bytes &b, int j) { return j + 1 < (int)b.cnt ? dp_byte(b, j + 1) : (j + 1 < DP_V ? (uint32_t)'\n' : b.next); }

// max over the threads BEFORE this one (0 = none) and over the whole workgroup; scratch: 8 uint32 in LDS
__device__ __forceinline__ uint32_t dp_block_excl_max(uint32_t v, uint32_t *scratch, uint32_t *block_max) {
    uint32_t x = v;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d, 64); if (mf_lane() >= d) x = x > y ? x : y; }
    uint32_t ex = __shfl_up(x, 1, 64);
    if (mf_lane() == 0) ex = 0;
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if (mf_lane() == 63) scratch[wave] = x;
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); i++) { const uint32_t t = scratch[i]; if (i < wave) before = before > t ? before : t; all = all > t ? all : t; }
    *block_max = all;
    return ex > before ? ex : before;
}

// ---------------------------------------------------------------------------------------------
// FASTA
// ---------------------------------------------------------------------------------------------
struct dp_fa_chunk { uint32_t n_hdr, seq_head, seq_rest, state; };      // state: 2 = no line start in the chunk, else 1 = its last line is a header line, 0 = it is not
struct dp_fa_prefix { uint64_t hdr, seq; uint32_t in_hdr, pad; };       // before the chunk: header lines, sequence bytes; is its first byte inside a header line?

// what a thread's 16 bytes say without knowing the line they start in: header-line starts, valid sequence bytes before / after the first line
// start, and the state they leave behind (2: no line start among them)
struct dp_fa_local { uint32_t n_hdr, seq_before, seq_after, state, anomaly; };
__device__ __forceinline__ dp_fa_local dp_fa_scan16(const dp_bytes &b, const uint8_t *lut) {
    dp_fa_local L = {0, 0, 0, 2u, 0};
    uint32_t p = b.prev;
#pragma unroll
    for (int j = 0; j < DP_V; j++) {
        if (j < (int)b.cnt) {
            const uint32_t c = dp_byte(b, j);
            if (p == '\n') { L.state = (c == '>' || c == ';') ? 1u : 0u; L.n_hdr += L.state; }
            if (c == '\r') { if (dp_after(b, j) != '\n') L.anomaly |= DP_A_LONE_CR; }
            else if (c != '\n' && L.state != 1u) { if (dp_lut(lut, c) < 0xFEu) { if (L.state == 2u) L.seq_before++; else L.seq_after++; } }
            p = c;
        }
    }
    return L;
}

// pass 1: per chunk
__global__ __launch_bounds__(DP_T) void k_dp_fa_count(const uint8_t *__restrict__ raw, uint64_t n, dp_fa_chunk *__restrict__ chunks, unsigned int *__restrict__ anomaly) {
    __shared__ uint8_t lut[256];
    __shared__ uint32_t scratch[8], acc[3];
    dp_fill_lut(lut);
    if (threadIdx.x < 3) acc[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t c0 = (uint64_t)blockIdx.x * DP_CHUNK;
    uint32_t run = 2u, bad = 0;                       // state at the tile's start (2: still the chunk's unknown one)
    for (int t = 0; t < DP_TPC; t++) {
        const uint64_t base = c0 + (uint64_t)t * DP_TILE + (uint64_t)threadIdx.x * DP_V;
        if (c0 + (uint64_t)t * DP_TILE >= n) break;
        const dp_bytes b = dp_load(raw, n, base);
        const dp_fa_local L = dp_fa_scan16(b, lut);
        bad |= L.anomaly;
        uint32_t all;
        const uint32_t e = dp_block_excl_max(L.state != 2u ? (threadIdx.x + 1u) * 2u + L.state : 0u, scratch, &all);
        const uint32_t in = e ? (e & 1u) : run;
        uint32_t head = 0, rest = L.seq_after;
        if (in == 0u) rest += L.seq_before; else if (in == 2u) head = L.seq_before;
        // (wave sums, one LDS atomic per wave and quantity)
        uint32_t th, tr, tn;
        mf_wave_excl_scan(head, &th); mf_wave_excl_scan(rest, &tr); mf_wave_excl_scan(L.n_hdr, &tn);
        if (mf_lane() == 0) { if (th) atomicAdd(&acc[1], th); if (tr) atomicAdd(&acc[2], tr); if (tn) atomicAdd(&acc[0], tn); }
        if (all) run = all & 1u;
    }
    if (bad) atomicOr(anomaly, bad);
    __syncthreads();
    if (threadIdx.x == 0) chunks[blockIdx.x] = dp_fa_chunk{acc[0], acc[1], acc[2], run};
}
// the chunks' counts -> prefixes (one workgroup walks the array: 60 000 chunks for a 15 GB file)
__global__ __launch_bounds__(1024) void k_dp_fa_scan(const dp_fa_chunk *__restrict__ chunks, uint64_t nc, dp_fa_prefix *__restrict__ pre, uint64_t *__restrict__ totals) {
    __shared__ uint32_t scratch[20];
    __shared__ unsigned long long carry[2];
    __shared__ uint32_t carry_state;
    if (threadIdx.x == 0) { carry[0] = carry[1] = 0; carry_state = 0; }     // (the file's first byte starts a line: whatever stands here is overwritten by it)
    __syncthreads();
    for (uint64_t i0 = 0; i0 < nc; i0 += 1024) {
        const uint64_t i = i0 + threadIdx.x;
        dp_fa_chunk c = {0, 0, 0, 2u};
        if (i < nc) c = chunks[i];
        uint32_t all;
        const uint32_t e = dp_block_excl_max(c.state != 2u ? (threadIdx.x + 1u) * 2u + c.state : 0u, scratch, &all);
        const uint32_t in = e ? (e & 1u) : carry_state;
        const uint32_t seq = c.seq_rest + (in == 0u ? c.seq_head : 0u);
        uint32_t th, ts;
        const uint32_t xh = mf_block_excl_scan(c.n_hdr, scratch, &th);
        const uint32_t xs = mf_block_excl_scan(seq, scratch, &ts);
        if (i < nc) pre[i] = dp_fa_prefix{carry[0] + xh, carry[1] + xs, in, 0};
        __syncthreads();
        if (threadIdx.x == 0) { carry[0] += th; carry[1] += ts; if (all) carry_state = all & 1u; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { totals[0] = carry[0]; totals[1] = carry[1]; }
}
// per tile, for the passes that know the chunk's prefix: every thread's state on entry and the header lines / sequence bytes before its bytes
struct dp_fa_pos { uint32_t state; uint64_t hdr, seq; };
__device__ __forceinline__ dp_fa_pos dp_fa_locate(const dp_fa_local &L, uint32_t &run, uint64_t &run_hdr, uint64_t &run_seq, uint32_t *scratch) {
    uint32_t all;
    const uint32_t e = dp_block_excl_max(L.state != 2u ? (threadIdx.x + 1u) * 2u + L.state : 0u, scratch, &all);
    dp_fa_pos P;
    P.state = e ? (e & 1u) : run;
    const uint32_t seq = L.seq_after + (P.state == 0u ? L.seq_before : 0u);
    uint32_t th, ts;
    const uint32_t xh = mf_block_excl_scan(L.n_hdr, scratch, &th);
    const uint32_t xs = mf_block_excl_scan(seq, scratch, &ts);
    P.hdr = run_hdr + xh; P.seq = run_seq + xs;
    run_hdr += th; run_seq += ts;
    if (all) run = all & 1u;
    return P;
}
// pass 2: rec_seq[r] = sequence bytes before header line r (r = 1 ..; record 0 = what stands before the first header line, rec_seq[0] = 0),
// rec_flag[r] bit 0: the record holds an N, bit 1: it holds a character that is no nucleotide (rare: one atomic OR on the flags' word each)
__global__ __launch_bounds__(DP_T) void k_dp_fa_records(const uint8_t *__restrict__ raw, uint64_t n, const dp_fa_prefix *__restrict__ pre,
                                                        uint64_t *__restrict__ rec_seq, unsigned int *__restrict__ rec_flag_words) {
    __shared__ uint8_t lut[256];
    __shared__ uint32_t scratch[20];
    dp_fill_lut(lut);
    __syncthreads();
    const uint64_t c0 = (uint64_t)blockIdx.x * DP_CHUNK;
    const dp_fa_prefix P0 = pre[blockIdx.x];
    uint32_t run = P0.in_hdr; uint64_t run_hdr = P0.hdr, run_seq = P0.seq;
    for (int t = 0; t < DP_TPC; t++) {
        if (c0 + (uint64_t)t * DP_TILE >= n) break;
        const uint64_t base = c0 + (uint64_t)t * DP_TILE + (uint64_t)threadIdx.x * DP_V;
        const dp_bytes b = dp_load(raw, n, base);
        const dp_fa_local L = dp_fa_scan16(b, lut);
        const dp_fa_pos P = dp_fa_locate(L, run, run_hdr, run_seq, scratch);
        uint32_t st = P.state, p = b.prev; uint64_t h = P.hdr, s = P.seq;
#pragma unroll
        for (int j = 0; j < DP_V; j++) {
            if (j < (int)b.cnt) {
                const uint32_t c = dp_byte(b, j);
                if (p == '\n') { st = (c == '>' || c == ';') ? 1u : 0u; if (st) { h++; rec_seq[h] = s; } }
                if (c != '\n' && c != '\r' && st == 0u) {
                    const uint32_t v = dp_lut(lut, c);
                    if (v < 0xFEu) s++;
                    else atomicOr(&rec_flag_words[h >> 2], (v == 0xFEu ? 1u : 2u) << (8 * (h & 3)));
                }
                p = c;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// records (both formats): keep / drop, places
// ---------------------------------------------------------------------------------------------
// FASTA: the length of record r is rec_seq[r + 1] - rec_seq[r] (rec_seq[n_rec] = all sequence bytes); kept iff it has bases and no N.
// A record with a wrong character and no N is an error in the reference (FastaReader -> DnaTools.fromChar throws): anomaly.
__device__ __forceinline__ void dp_fa_record(const uint64_t *rec_seq, const uint8_t *rec_flag, uint64_t r, uint64_t *len, bool *keep, uint32_t *anomaly) {
    const uint64_t a = rec_seq[r], b = rec_seq[r + 1];
    const uint32_t f = rec_flag[r];
    *len = b - a;
    *keep = b > a && !(f & 1u);
    if ((f & 2u) && !(f & 1u)) *anomaly |= DP_A_BAD_CHAR;
    if (b - a >= ((uint64_t)1 << 24)) *anomaly |= DP_A_LENGTHS;                  // (assembled sequences: the host parser; k_dp_rec_place sums 256 lengths in 32 bits)
}
// FASTQ: record r = lines 4r .. 4r+3 (line L = [ls[L], ls[L + 1] - 1), a '\r' at its end dropped)
__device__ __forceinline__ void dp_fq_record(const uint8_t *raw, const uint64_t *ls, const uint8_t *drop, uint64_t r, uint64_t *len, bool *keep, uint32_t *anomaly) {
    const uint64_t l0 = ls[4 * r], l1 = ls[4 * r + 1], l2 = ls[4 * r + 2], l3 = ls[4 * r + 3], l4 = ls[4 * r + 4];
    uint64_t dl = l2 - 1 - l1, ql = l4 - 1 - l3;
    if (dl && raw[l2 - 2] == '\r') dl--;
    if (ql && raw[l4 - 2] == '\r') ql--;
    const uint32_t h0 = raw[l0], h2 = raw[l2];
    if ((h0 != '@' && h0 != '+') || (h2 != '@' && h2 != '+')) *anomaly |= DP_A_STRUCTURE;          // FastqReader: "Unknown structure of fastq file!"
    if (dl != ql || dl == 0 || dl >= ((uint64_t)1 << 24)) *anomaly |= DP_A_LENGTHS;
    *len = dl;
    *keep = !drop[r];
}
template <int FMT>
__global__ __launch_bounds__(DP_T) void k_dp_rec_count(const uint8_t *__restrict__ raw, const uint64_t *__restrict__ rec_a, const uint8_t *__restrict__ rec_b,
                                                       uint64_t n_rec, unsigned long long *__restrict__ blk_reads, unsigned long long *__restrict__ blk_bases,
                                                       unsigned int *__restrict__ anomaly) {
    __shared__ unsigned long long acc[2];
    if (threadIdx.x < 2) acc[threadIdx.x] = 0;
    __syncthreads();
    unsigned long long nr = 0, nb = 0; uint32_t bad = 0;
    for (int q = 0; q < DP_RPB / DP_T; q++) {
        const uint64_t r = (uint64_t)blockIdx.x * DP_RPB + (uint64_t)q * DP_T + threadIdx.x;
        if (r < n_rec) {
            uint64_t len; bool keep;
            if (FMT == 1) dp_fa_record(rec_a, rec_b, r, &len, &keep, &bad); else dp_fq_record(raw, rec_a, rec_b, r, &len, &keep, &bad);
            if (keep) { nr++; nb += len; }
        }
    }
    if (nr) { atomicAdd(&acc[0], nr); atomicAdd(&acc[1], nb); }
    if (bad) atomicOr(anomaly, bad);
    __syncthreads();
    if (threadIdx.x == 0) { blk_reads[blockIdx.x] = acc[0]; blk_bases[blockIdx.x] = acc[1]; }
}
// exclusive scan of two arrays of 64-bit sums in place (one workgroup), totals to out[0..1]
__global__ __launch_bounds__(1024) void k_dp_scan2(unsigned long long *__restrict__ a, unsigned long long *__restrict__ b, uint64_t n, uint64_t *__restrict__ totals) {
    __shared__ unsigned long long wsum[2][16], carry[2];
    if (threadIdx.x < 2) carry[threadIdx.x] = 0;
    __syncthreads();
    const int lane = mf_lane(), wave = threadIdx.x >> 6;
    for (uint64_t i0 = 0; i0 < n; i0 += 1024) {
        const uint64_t i = i0 + threadIdx.x;
        unsigned long long v[2] = {i < n ? a[i] : 0ull, i < n ? b[i] : 0ull}, x[2];
        for (int q = 0; q < 2; q++) {
            unsigned long long s = v[q];
            for (int d = 1; d < 64; d <<= 1) { const unsigned long long y = __shfl_up(s, d, 64); if (lane >= d) s += y; }
            x[q] = s;
            if (lane == 63) wsum[q][wave] = s;
        }
        __syncthreads();
        unsigned long long before[2] = {0, 0}, all[2] = {0, 0};
        for (int q = 0; q < 2; q++) for (int w = 0; w < 16; w++) { if (w < wave) before[q] += wsum[q][w]; all[q] += wsum[q][w]; }
        if (i < n) { a[i] = carry[0] + before[0] + x[0] - v[0]; b[i] = carry[1] + before[1] + x[1] - v[1]; }
        __syncthreads();
        if (threadIdx.x == 0) { carry[0] += all[0]; carry[1] += all[1]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { totals[0] = carry[0]; totals[1] = carry[1]; }
}
// every kept record: its read number and the place of its first base; rec_place[r] = FASTA: sequence bytes of DROPPED records before r (what
// a byte's rank among all sequence bytes is reduced by), FASTQ: the place of the record's first base; DP_DROP for a dropped record
template <int FMT>
__global__ __launch_bounds__(DP_T) void k_dp_rec_place(const uint8_t *__restrict__ raw, const uint64_t *__restrict__ rec_a, const uint8_t *__restrict__ rec_b, uint64_t n_rec,
                                                       const unsigned long long *__restrict__ blk_reads, const unsigned long long *__restrict__ blk_bases,
                                                       uint64_t *__restrict__ rec_place, uint64_t *__restrict__ offsets) {
    __shared__ uint32_t scratch[20];
    unsigned long long read0 = blk_reads[blockIdx.x], base0 = blk_bases[blockIdx.x];
    for (int q = 0; q < DP_RPB / DP_T; q++) {
        const uint64_t r = (uint64_t)blockIdx.x * DP_RPB + (uint64_t)q * DP_T + threadIdx.x;
        uint64_t len = 0; bool keep = false; uint32_t bad = 0;
        if (r < n_rec) { if (FMT == 1) dp_fa_record(rec_a, rec_b, r, &len, &keep, &bad); else dp_fq_record(raw, rec_a, rec_b, r, &len, &keep, &bad); }
        uint32_t tk, tb;
        const uint32_t xk = mf_block_excl_scan(keep ? 1u : 0u, scratch, &tk);
        const uint32_t xb = mf_block_excl_scan(keep ? (uint32_t)len : 0u, scratch, &tb);          // (256 records of < 2^24 bases each: the device parser is for READ files)
        if (r < n_rec) {
            if (keep) { offsets[read0 + xk] = base0 + xb; rec_place[r] = FMT == 1 ? rec_a[r] - (base0 + xb) : base0 + xb; }
            else rec_place[r] = DP_DROP;
        }
        read0 += tk; base0 += tb;
    }
}

// ---------------------------------------------------------------------------------------------
// pass 3 (both formats): the kept records' bases to their places, through LDS
// ---------------------------------------------------------------------------------------------
// A tile's emitted bytes are consecutive in the output (places grow with the file position and every byte between two emitted ones is
// dropped with its whole record): lds[(first place & 15) + rank] <-> out[first place + rank]; whole 16-byte words go out as such, the two
// ragged ends byte by byte (the neighbouring tiles write the rest of those words).
struct dp_emit_lds { alignas(16) uint8_t buf[DP_TILE + 32]; unsigned long long first; uint32_t scratch[20]; };
__device__ __forceinline__ void dp_emit_flush(dp_emit_lds &S, uint32_t n_e, uint8_t *__restrict__ out) {
    __syncthreads();
    if (n_e) {
        const uint64_t first = S.first;
        const uint32_t a = (uint32_t)(first & 15u);
        uint8_t *g0 = out + (first - a);
        const uint32_t nw = (a + n_e + 15u) >> 4;
        for (uint32_t w = threadIdx.x; w < nw; w += DP_T) {
            const uint32_t lo = w * 16u, hi = lo + 16u;
            if (lo >= a && hi <= a + n_e) *reinterpret_cast<uint4 *>(g0 + lo) = *reinterpret_cast<const uint4 *>(&S.buf[lo]);
            else for (uint32_t q = (lo > a ? lo : a); q < (hi < a + n_e ? hi : a + n_e); q++) g0[q] = S.buf[q];
        }
    }
    __syncthreads();
}
__global__ __launch_bounds__(DP_T) void k_dp_fa_emit(const uint8_t *__restrict__ raw, uint64_t n, const dp_fa_prefix *__restrict__ pre,
                                                     const uint64_t *__restrict__ rec_place, uint8_t *__restrict__ out) {
    __shared__ uint8_t lut[256];
    __shared__ dp_emit_lds S;
    dp_fill_lut(lut);
    __syncthreads();
    const uint64_t c0 = (uint64_t)blockIdx.x * DP_CHUNK;
    const dp_fa_prefix P0 = pre[blockIdx.x];
    uint32_t run = P0.in_hdr; uint64_t run_hdr = P0.hdr, run_seq = P0.seq;
    for (int t = 0; t < DP_TPC; t++) {
        if (c0 + (uint64_t)t * DP_TILE >= n) break;
        const uint64_t base = c0 + (uint64_t)t * DP_TILE + (uint64_t)threadIdx.x * DP_V;
        const dp_bytes b = dp_load(raw, n, base);
        const dp_fa_local L = dp_fa_scan16(b, lut);
        const dp_fa_pos P = dp_fa_locate(L, run, run_hdr, run_seq, S.scratch);
        // first walk: how many bytes this thread emits
        uint32_t st = P.state, p = b.prev, mine = 0; uint64_t h = P.hdr;
        uint64_t place = rec_place[h];
#pragma unroll
        for (int j = 0; j < DP_V; j++) {
            if (j < (int)b.cnt) {
                const uint32_t c = dp_byte(b, j);
                if (p == '\n') { st = (c == '>' || c == ';') ? 1u : 0u; if (st) { h++; place = rec_place[h]; } }
                if (c != '\n' && c != '\r' && st == 0u && place != DP_DROP && dp_lut(lut, c) < 0xFEu) mine++;
                p = c;
            }
        }
        uint32_t n_e;
        const uint32_t rank0 = mf_block_excl_scan(mine, S.scratch, &n_e);
        // second walk: the bytes into LDS, in line with the output (every emitting thread knows where the tile's first emitted byte goes: its own
        // first one's place minus its rank); the thread with rank 0 says it to the flush
        st = P.state; p = b.prev; h = P.hdr; place = rec_place[h];
        uint64_t s = P.seq; uint32_t rank = rank0, a = 0;
        bool first = true;
#pragma unroll
        for (int j = 0; j < DP_V; j++) {
            if (j < (int)b.cnt) {
                const uint32_t c = dp_byte(b, j);
                if (p == '\n') { st = (c == '>' || c == ';') ? 1u : 0u; if (st) { h++; place = rec_place[h]; } }
                if (c != '\n' && c != '\r' && st == 0u) {
                    const uint32_t v = dp_lut(lut, c);
                    if (v < 0xFEu) {
                        if (place != DP_DROP) {
                            if (first) { const uint64_t tile_first = (s - place) - (uint64_t)rank0; a = (uint32_t)(tile_first & 15u); if (rank0 == 0u) S.first = tile_first; first = false; }
                            S.buf[a + rank] = (uint8_t)v;
                            rank++;
                        }
                        s++;
                    }
                }
                p = c;
            }
        }
        dp_emit_flush(S, n_e, out);
    }
}

// ---------------------------------------------------------------------------------------------
// FASTQ
// ---------------------------------------------------------------------------------------------
// pass 1: '\n' per chunk; structure checks that need no line numbers
__global__ __launch_bounds__(DP_T) void k_dp_fq_count(const uint8_t *__restrict__ raw, uint64_t n, unsigned long long *__restrict__ chunk_nl, unsigned int *__restrict__ anomaly) {
    __shared__ uint32_t acc;
    if (threadIdx.x == 0) acc = 0;
    __syncthreads();
    const uint64_t c0 = (uint64_t)blockIdx.x * DP_CHUNK;
    uint32_t cnt = 0, bad = 0;
    for (int t = 0; t < DP_TPC; t++) {
        if (c0 + (uint64_t)t * DP_TILE >= n) break;
        const dp_bytes b = dp_load(raw, n, c0 + (uint64_t)t * DP_TILE + (uint64_t)threadIdx.x * DP_V);
        uint32_t p = b.prev;
#pragma unroll
        for (int j = 0; j < DP_V; j++) {
            if (j < (int)b.cnt) {
                const uint32_t c = dp_byte(b, j);
                if (c == '\n') { cnt++; if (p == '\n') bad |= DP_A_EMPTY_LINE; }
                else if (c == '\r') { const uint32_t x = dp_after(b, j); if (x != '\n') bad |= DP_A_LONE_CR; else if (p == '\n') bad |= DP_A_EMPTY_LINE; }
                p = c;
            }
        }
    }
    uint32_t tot;
    mf_wave_excl_scan(cnt, &tot);
    if (mf_lane() == 0 && tot) atomicAdd(&acc, tot);
    if (bad) atomicOr(anomaly, bad);
    __syncthreads();
    if (threadIdx.x == 0) chunk_nl[blockIdx.x] = acc;
}
// pass 2: the start of every line; per-character checks by the line's role (line number mod 4)
__global__ __launch_bounds__(DP_T) void k_dp_fq_lines(const uint8_t *__restrict__ raw, uint64_t n, const unsigned long long *__restrict__ chunk_nl, uint32_t qoff,
                                                      uint64_t *__restrict__ ls, uint8_t *__restrict__ drop, unsigned int *__restrict__ anomaly) {
    __shared__ uint8_t lut[256];
    __shared__ uint32_t scratch[20];
    dp_fill_lut(lut);
    __syncthreads();
    const uint64_t c0 = (uint64_t)blockIdx.x * DP_CHUNK;
    uint64_t run = chunk_nl[blockIdx.x];
    uint32_t bad = 0;
    for (int t = 0; t < DP_TPC; t++) {
        if (c0 + (uint64_t)t * DP_TILE >= n) break;
        const uint64_t base = c0 + (uint64_t)t * DP_TILE + (uint64_t)threadIdx.x * DP_V;
        const dp_bytes b = dp_load(raw, n, base);
        uint32_t cnt = 0;
#pragma unroll
        for (int j = 0; j < DP_V; j++) if (j < (int)b.cnt && dp_byte(b, j) == '\n') cnt++;
        uint32_t tot;
        const uint32_t ex = mf_block_excl_scan(cnt, scratch, &tot);
        uint64_t line = run + ex;
        run += tot;
        uint32_t p = b.prev;
#pragma unroll
        for (int j = 0; j < DP_V; j++) {
            if (j < (int)b.cnt) {
                const uint32_t c = dp_byte(b, j);
                if (p == '\n') ls[line] = base + (uint64_t)j;
                const uint32_t role = (uint32_t)line & 3u;
                if (c == '\n') line++;
                else if (c != '\r') {
                    if (role == 1u) {
                        if (c == 'N' || c == 'n' || c == '.') drop[line >> 2] = 1;
                        else if (dp_lut(lut, c) >= 0xFEu) bad |= DP_A_BAD_CHAR;
                    } else if (role == 3u) {
                        if (c < qoff || c > 126u) bad |= DP_A_BAD_QUAL;
                        else if (c == qoff) drop[line >> 2] = 1;                                  // phred 0 (FastaReaderFromXQSource.java:66-70)
                    }
                }
                p = c;
            }
        }
    }
    if (bad) atomicOr(anomaly, bad);
}
__global__ __launch_bounds__(DP_T) void k_dp_fq_emit(const uint8_t *__restrict__ raw, uint64_t n, const unsigned long long *__restrict__ chunk_nl,
                                                     const uint64_t *__restrict__ ls, const uint64_t *__restrict__ rec_place, uint8_t *__restrict__ out) {
    __shared__ uint8_t lut[256];
    __shared__ dp_emit_lds S;
    dp_fill_lut(lut);
    __syncthreads();
    const uint64_t c0 = (uint64_t)blockIdx.x * DP_CHUNK;
    uint64_t run = chunk_nl[blockIdx.x];
    for (int t = 0; t < DP_TPC; t++) {
        if (c0 + (uint64_t)t * DP_TILE >= n) break;
        const uint64_t base = c0 + (uint64_t)t * DP_TILE + (uint64_t)threadIdx.x * DP_V;
        const dp_bytes b = dp_load(raw, n, base);
        uint32_t cnt = 0;
#pragma unroll
        for (int j = 0; j < DP_V; j++) if (j < (int)b.cnt && dp_byte(b, j) == '\n') cnt++;
        uint32_t tot;
        const uint32_t ex = mf_block_excl_scan(cnt, S.scratch, &tot);
        const uint64_t line0 = run + ex;
        run += tot;
        // place = where the byte at hand goes if it is a base of a kept record's sequence line, else DP_DROP
        uint64_t place0 = DP_DROP;
        if ((line0 & 3u) == 1u) { const uint64_t pl = rec_place[line0 >> 2]; if (pl != DP_DROP) place0 = pl + (base - ls[line0]); }
        uint32_t mine = 0;
        {
            uint64_t line = line0; bool kept = place0 != DP_DROP;
#pragma unroll
            for (int j = 0; j < DP_V; j++) {
                if (j < (int)b.cnt) {
                    const uint32_t c = dp_byte(b, j);
                    if (c == '\n') { line++; kept = (line & 3u) == 1u && rec_place[line >> 2] != DP_DROP; }
                    else if (c != '\r' && kept) mine++;
                }
            }
        }
        uint32_t n_e;
        const uint32_t rank0 = mf_block_excl_scan(mine, S.scratch, &n_e);
        {
            uint64_t line = line0, place = place0; uint32_t rank = rank0, a = 0; bool first = true;
#pragma unroll
            for (int j = 0; j < DP_V; j++) {
                if (j < (int)b.cnt) {
                    const uint32_t c = dp_byte(b, j);
                    if (c == '\n') { line++; place = (line & 3u) == 1u ? rec_place[line >> 2] : DP_DROP; }
                    else if (c != '\r' && place != DP_DROP) {
                        if (first) { const uint64_t tile_first = place - (uint64_t)rank0; a = (uint32_t)(tile_first & 15u); if (rank0 == 0u) S.first = tile_first; first = false; }
                        S.buf[a + rank] = (uint8_t)dp_lut(lut, c);
                        rank++; place++;
                    }
                }
            }
        }
        dp_emit_flush(S, n_e, out);
    }
}
__global__ void k_dp_set_u64(uint64_t *p, uint64_t v) { *p = v; }

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
int mf_ensure_pin_pool(mf_ctx *ctx, size_t want);
// the file's bytes to d_raw as they are: W threads pread 8 MB pieces into their two staging chunks and copy them up
// (copies from a mapping of the page cache instead of pread + staging were measured: 11 GB/s whatever the number of threads -- the runtime stages
// a pageable source through its own buffers on the calling thread, page fault by page fault; gpurun_out/r05d_upload_rate.txt)
// (one upload at a time per DEVICE: two contexts of a process that share a device -- the driver's two-contexts-per-device mode -- would halve each
// other's PCIe rate and finish together; taking turns, the first one is counting and writing while the second one's bytes cross)
static std::mutex g_upload_mutex[64];
static int ensure_up_pool(mf_ctx *ctx, size_t want) {
    if (ctx->up_pool_bytes < want) {
        if (ctx->up_pool) { if (ctx->up_pool_pinned) hipHostFree(ctx->up_pool); else free(ctx->up_pool); ctx->up_pool = nullptr; ctx->up_pool_bytes = 0; }
        if (hipHostMalloc(&ctx->up_pool, want, hipHostMallocDefault) == hipSuccess) ctx->up_pool_pinned = true;
        else {
            (void)hipGetLastError();
            ctx->up_pool = nullptr; ctx->up_pool_pinned = false;
            if (posix_memalign(&ctx->up_pool, 2 << 20, want) != 0) { ctx->up_pool = nullptr; return 1; }
        }
        ctx->up_pool_bytes = want;
    }
    return 0;
}
// mem != nullptr: the bytes come from host memory (an inflated .gz file) instead of the file: the same staging, memcpy for pread
// file_off: where in the file the fsize bytes start; up: the stream the copies go on (the streamed count, mf_stream.hip: a stream of its own, so that they
// cross PCIe while the context's stream computes); take_turn = false: the caller holds the device's turn already
static int upload_bytes(mf_ctx *ctx, int fd, const uint8_t *mem, size_t fsize, uint8_t *d_raw, size_t file_off = 0, hipStream_t up = nullptr, bool take_turn = true) {
    if (!up) up = ctx->stream;
    std::unique_lock<std::mutex> turn(g_upload_mutex[(unsigned)ctx->device & 63u], std::defer_lock);
    if (take_turn) turn.lock();
    const size_t PIECE = (size_t)std::max<int64_t>(ctx->opt_device_parse_piece, 1 << 16);
    const size_t np = (fsize + PIECE - 1) / PIECE;
    const size_t WMAX = (size_t)std::min<int64_t>(std::max<int64_t>(ctx->opt_device_parse_threads, 1), std::max(ctx->host_threads, 1));
    const int W = (int)std::min<size_t>(WMAX, np);
    // staging: two chunks per thread, PINNED (the copies are then true DMA from where pread has put the bytes; from plain memory the runtime
    // stages every byte a second time).  128 MB by default: hipHostMalloc takes ~30 ms for them once per context, and a short process -- the
    // drop-in's -- is still ahead (first load of a 3 GB file 0.102 s against 0.116 - 0.135 s with plain chunks of any size: profiles/r05j_upload_rate.txt).
    const size_t want = (size_t)2 * WMAX * PIECE;
    if (ensure_up_pool(ctx, want) != 0) return 1;
    std::atomic<size_t> next{0};
    std::atomic<int> state{0};
    std::vector<std::thread> th;
    for (int w = 0; w < W; w++)
        th.emplace_back([&, w]() {
            (void)hipSetDevice(ctx->device);
            uint8_t *pin[2] = {(uint8_t *)ctx->up_pool + (size_t)(2 * w) * PIECE, (uint8_t *)ctx->up_pool + (size_t)(2 * w + 1) * PIECE};
            hipEvent_t ev[2]; bool busy[2] = {false, false};
            (void)hipEventCreateWithFlags(&ev[0], hipEventDisableTiming); (void)hipEventCreateWithFlags(&ev[1], hipEventDisableTiming);
            int cur = 0;
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= np || state.load() != 0) break;
                const size_t lo = i * PIECE, len = std::min(PIECE, fsize - lo);
                if (busy[cur]) { (void)hipEventSynchronize(ev[cur]); busy[cur] = false; }
                size_t got = 0;
                if (mem) { memcpy(pin[cur], mem + lo, len); got = len; }
                else while (got < len) { const ssize_t r = pread(fd, pin[cur] + got, len - got, (off_t)(file_off + lo + got)); if (r <= 0) break; got += (size_t)r; }
                if (got != len) { state = -1; break; }
                if (hipMemcpyAsync(d_raw + lo, pin[cur], len, hipMemcpyHostToDevice, up) != hipSuccess || hipEventRecord(ev[cur], up) != hipSuccess) { state = -2; break; }
                busy[cur] = true;
                cur ^= 1;
            }
            for (int j = 0; j < 2; j++) { if (busy[j]) (void)hipEventSynchronize(ev[j]); (void)hipEventDestroy(ev[j]); }
        });
    for (auto &x : th) x.join();
    if (state.load() == -1) return mf_set_error("short read on the reads file");
    if (state.load() == -2) { (void)hipGetLastError(); return mf_set_error("H2D copy of the reads file failed"); }
    return 0;
}

int mf_upload_file(mf_ctx *ctx, int fd, size_t fsize, uint8_t *d_raw) { return upload_bytes(ctx, fd, nullptr, fsize, d_raw); }
// ---- for the streamed count (mf_stream.hip): a range of a file / a host buffer to HBM on a stream of the caller's; the device's upload turn as a lock object
int mf_upload_range(mf_ctx *ctx, int fd, size_t file_off, size_t len, uint8_t *d_dst, hipStream_t up) { return upload_bytes(ctx, fd, nullptr, len, d_dst, file_off, up, false); }
int mf_upload_mem(mf_ctx *ctx, const uint8_t *mem, size_t len, uint8_t *d_dst, hipStream_t up) { return upload_bytes(ctx, -1, mem, len, d_dst, 0, up, false); }
std::mutex &mf_upload_turn(mf_ctx *ctx) { return g_upload_mutex[(unsigned)ctx->device & 63u]; }
int mf_upload_pool(mf_ctx *ctx) {
    const size_t PIECE = (size_t)std::max<int64_t>(ctx->opt_device_parse_piece, 1 << 16);
    const size_t WMAX = (size_t)std::min<int64_t>(std::max<int64_t>(ctx->opt_device_parse_threads, 1), std::max(ctx->host_threads, 1));
    return ensure_up_pool(ctx, (size_t)2 * WMAX * PIECE);
}

// ---- a .gz file: the host inflates it piece by piece (mf_inflate.h, one member on many threads) INTO the pinned staging chunks, and the chunks go
// up as they fill -- the inflated text is never whole in host memory (first-touch page faults on those gigabytes were a third of the load) and is
// parsed where it lands.  1: not a file for this way (several members, BGZF, an irregular stream, more text than the buffer was sized for): the
// caller inflates it on the host.
static int dparse_source(mf_ctx *ctx, const char *path, const uint8_t *mem, size_t mem_n, int fmt, mf_buf<uint8_t> &bases, mf_buf<uint64_t> &offsets, uint64_t *n_reads, uint64_t *n_bases,
                         mf_buf<uint8_t> *filled, int filled_qoff);
struct dp_gz_sink : mfz::byte_sink {
    mf_ctx *ctx = nullptr; uint8_t *d_raw = nullptr; size_t cap = 0;
    std::vector<char> head;                                                    // the first bytes of the text (FASTQ: the quality offset is decided from them)
    uint8_t *pin[64][2]; hipEvent_t ev[64][2]; bool busy[64][2]; int cur[64];
    std::atomic<int> failed{0};
    uint8_t *acquire(int w) override {
        (void)hipSetDevice(ctx->device);
        const int c = cur[w];
        if (busy[w][c]) { (void)hipEventSynchronize(ev[w][c]); busy[w][c] = false; }
        return pin[w][c];
    }
    bool commit(int w, size_t offset, size_t len) override {
        const int c = cur[w];
        if (offset + len > cap) return false;
        if (offset < head.size()) memcpy(head.data() + offset, pin[w][c], std::min(len, head.size() - offset));
        if (hipMemcpyAsync(d_raw + offset, pin[w][c], len, hipMemcpyHostToDevice, ctx->stream) != hipSuccess || hipEventRecord(ev[w][c], ctx->stream) != hipSuccess) { failed = 1; return false; }
        busy[w][c] = true;
        cur[w] = c ^ 1;
        return true;
    }
};
int mf_dparse_gz(mf_ctx *ctx, const char *path, const void *packed, size_t packed_n, int fmt, mf_buf<uint8_t> &bases, mf_buf<uint64_t> &offsets, uint64_t *n_reads, uint64_t *n_bases) {
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    MF_HIP(hipSetDevice(ctx->device));
    size_t fr = 0, tot = 0;
    MF_HIP(hipMemGetInfo(&fr, &tot));
    // room for six times the compressed bytes (reads compress three- to fivefold), a quarter of what is free at most
    size_t cap = std::min<size_t>(packed_n * 6 + ((size_t)64 << 20), (size_t)(((double)fr + (double)mf_arena_idle(ctx)) / 4.0));
    cap = cap / DP_CHUNK * DP_CHUNK;
    if (cap < packed_n * 2) return 1;
    mf_buf<uint8_t> raw;
    if (raw.alloc(ctx, cap + 64) != MF_OK) return 1;
    size_t total = 0;
    dp_gz_sink sink;
    {
        // (no turn-taking with the device's other contexts here, unlike mf_upload_file: the text goes up at the inflater's 4 GB/s, a tenth of what PCIe
        // carries -- two libraries inflate side by side)
        const size_t PIECE = (size_t)std::max<int64_t>(ctx->opt_device_parse_piece, 1 << 16);
        // (twice the file uploader's threads: a slot is resolved AND summed -- zlib's crc32, 1.5 GB/s a thread -- before it goes up)
        const size_t W = (size_t)std::min<int64_t>(std::min<int64_t>(2 * std::max<int64_t>(ctx->opt_device_parse_threads, 1), std::max(ctx->host_threads, 1)), 64);
        if (ensure_up_pool(ctx, 2 * W * PIECE) != 0) return 1;
        sink.ctx = ctx; sink.d_raw = raw.p; sink.cap = cap; sink.workers = (int)W; sink.slot_bytes = PIECE;
        sink.head.assign((size_t)4 << 20, 0);
        for (size_t w = 0; w < W; w++)
            for (int c = 0; c < 2; c++) {
                sink.pin[w][c] = (uint8_t *)ctx->up_pool + (2 * w + (size_t)c) * PIECE; sink.busy[w][c] = false; sink.cur[w] = 0;
                (void)hipEventCreateWithFlags(&sink.ev[w][c], hipEventDisableTiming);
            }
        const bool ok = mfz::gunzip_to_sink((const uint8_t *)packed, packed_n, ctx->host_threads, &sink, &total, (size_t)ctx->opt_gz_piece);
        for (size_t w = 0; w < W; w++)
            for (int c = 0; c < 2; c++) { if (sink.busy[w][c]) (void)hipEventSynchronize(sink.ev[w][c]); (void)hipEventDestroy(sink.ev[w][c]); }
        if (sink.failed.load()) { (void)hipGetLastError(); return mf_set_error("H2D copy of the inflated reads failed ('%s')", path); }
        if (!ok || total == 0) return 1;
    }
    int qoff = 64;
    if (fmt == 2) {
        read_batch tmp;
        qoff = parse_fastq_pass(sink.head.data(), std::min(total, sink.head.size()), path, 0, 0, tmp);
        if (qoff < 0) return 1;
    }
    if (getenv("MF_IO_TIMING")) fprintf(stderr, "[mf] %s: %.2f GB inflated into HBM (%.2f GB of text) in %.3f s\n", path, packed_n / 1e9, total / 1e9, now() - t0);
    return dparse_source(ctx, path, nullptr, total, fmt, bases, offsets, n_reads, n_bases, &raw, qoff);
}

// one FASTA (fmt 1) / FASTQ (fmt 2) file -> (bases, offsets) in HBM.  0 = done, 1 = not a file for the device parser (the caller takes the host
// parser: nothing was produced), < 0 = error
// mem != nullptr: the file's content is in host memory already (mem_n bytes: a compressed file the host has inflated); path names it in messages
static int dparse_source(mf_ctx *ctx, const char *path, const uint8_t *mem, size_t mem_n, int fmt, mf_buf<uint8_t> &bases, mf_buf<uint64_t> &offsets, uint64_t *n_reads, uint64_t *n_bases,
                         mf_buf<uint8_t> *filled = nullptr, int filled_qoff = 64);
int mf_dparse_file(mf_ctx *ctx, const char *path, int fmt, mf_buf<uint8_t> &bases, mf_buf<uint64_t> &offsets, uint64_t *n_reads, uint64_t *n_bases) {
    return dparse_source(ctx, path, nullptr, 0, fmt, bases, offsets, n_reads, n_bases);
}
// the text is in HBM already (n bytes at d_raw, in a buffer of at least the whole 256 KB chunks they span + 64 bytes, which stays the caller's): a piece
// of a file that starts and ends at record borders (mf_stream.hip).  qoff: the FASTQ quality offset the caller has decided from the file's head
int mf_dparse_device(mf_ctx *ctx, const char *path, uint8_t *d_raw, size_t room, size_t n, int fmt, int qoff, mf_buf<uint8_t> &bases, mf_buf<uint64_t> &offsets, uint64_t *n_reads, uint64_t *n_bases) {
    mf_buf<uint8_t> view; view.borrow(ctx, d_raw, room);
    return dparse_source(ctx, path, nullptr, n, fmt, bases, offsets, n_reads, n_bases, &view, qoff);
}
int mf_dparse_mem(mf_ctx *ctx, const char *path, const void *mem, size_t mem_n, int fmt, mf_buf<uint8_t> &bases, mf_buf<uint64_t> &offsets, uint64_t *n_reads, uint64_t *n_bases) {
    if (!mem || !mem_n) return 1;
    return dparse_source(ctx, path, (const uint8_t *)mem, mem_n, fmt, bases, offsets, n_reads, n_bases);
}
// filled: the content is in HBM already (mem_n bytes in *filled, which holds at least the whole chunks they span + 64: a .gz file inflated straight into it)
static int dparse_source(mf_ctx *ctx, const char *path, const uint8_t *mem, size_t mem_n, int fmt, mf_buf<uint8_t> &bases, mf_buf<uint64_t> &offsets, uint64_t *n_reads, uint64_t *n_bases,
                         mf_buf<uint8_t> *filled, int filled_qoff) {
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    int fd = -1;
    size_t n = mem_n;
    if (!mem && !filled) {
        fd = open(path, O_RDONLY);
        if (fd < 0) return mf_set_error("can't open '%s'", path);
        struct stat sb;
        if (fstat(fd, &sb) != 0) { close(fd); return mf_set_error("can't stat '%s'", path); }
        n = (size_t)sb.st_size;
    }
    auto shut = [&]() { if (fd >= 0) close(fd); fd = -1; };
    if (n == 0) { shut(); return 1; }
    int qoff = filled ? filled_qoff : 64;
    if (fmt == 2 && !filled) {                       // quality offset: the first 1000 records (ReadersUtils.java:63-77), on the host
        std::vector<char> head(std::min<size_t>(n, 4u << 20));
        if (mem) memcpy(head.data(), mem, head.size());
        else if (pread(fd, head.data(), head.size(), 0) != (ssize_t)head.size()) { shut(); return mf_set_error("short read on '%s'", path); }
        read_batch tmp;
        qoff = parse_fastq_pass(head.data(), head.size(), path, 0, 0, tmp);
        if (qoff < 0) { shut(); return 1; }           // (whatever it is: the host parser says it in the reference's words)
    }
    MF_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const uint64_t nc = (n + DP_CHUNK - 1) / DP_CHUNK;
    mf_buf<uint8_t> raw;
    if (filled) raw.swap(*filled);
    else {
        if (raw.alloc(ctx, nc * DP_CHUNK + 64) != MF_OK) { shut(); return 1; }
        const int rc = upload_bytes(ctx, fd, mem, n, raw.p);
        shut();
        if (rc != 0) return rc;
    }
    const double t1 = now();
    mf_buf<unsigned int> flags; MF_TRY(flags.alloc(ctx, 4));
    mf_buf<uint64_t> totals; MF_TRY(totals.alloc(ctx, 4));
    MF_HIP(hipMemsetAsync(flags.p, 0, 16, st));
    uint64_t h_tot[4] = {0, 0, 0, 0}; unsigned int h_flags[4] = {0, 0, 0, 0};
    auto anomaly = [&]() -> int {                   // (synchronises)
        if (hipMemcpyAsync(h_flags, flags.p, 16, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -1;
        return (int)h_flags[0];
    };
    mf_buf<uint64_t> rec_a, rec_place; mf_buf<uint8_t> rec_b;
    mf_buf<unsigned long long> blk_r, blk_b;
    uint64_t n_rec = 0;
    mf_buf<dp_fa_prefix> pre; mf_buf<unsigned long long> chunk_nl;
    if (fmt == 1) {
        mf_buf<dp_fa_chunk> chunks; MF_TRY(chunks.alloc(ctx, nc)); MF_TRY(pre.alloc(ctx, nc));
        {
            mf_ktimer tm(ctx, "k_dparse");
            k_dp_fa_count<<<(unsigned)nc, DP_T, 0, st>>>(raw.p, n, chunks.p, flags.p);
            k_dp_fa_scan<<<1, 1024, 0, st>>>(chunks.p, nc, pre.p, totals.p);
        }
        MF_HIP(hipMemcpyAsync(h_tot, totals.p, 16, hipMemcpyDeviceToHost, st));
        const int a = anomaly();
        if (a < 0) return mf_set_error("device parser: %s", hipGetErrorString(hipGetLastError()));
        if (a) return 1;
        n_rec = h_tot[0] + 1;                           // header lines + what stands before the first one
        if (n_rec >= ((uint64_t)1 << 40)) return 1;
        MF_TRY(rec_a.alloc(ctx, n_rec + 1)); MF_TRY(rec_b.alloc(ctx, (n_rec + 8) & ~(uint64_t)3));
        MF_HIP(hipMemsetAsync(rec_b.p, 0, (n_rec + 8) & ~(uint64_t)3, st));
        MF_HIP(hipMemsetAsync(rec_a.p, 0, 8, st));
        {
            mf_ktimer tm(ctx, "k_dparse");
            k_dp_set_u64<<<1, 1, 0, st>>>(rec_a.p + n_rec, h_tot[1]);
            k_dp_fa_records<<<(unsigned)nc, DP_T, 0, st>>>(raw.p, n, pre.p, rec_a.p, reinterpret_cast<unsigned int *>(rec_b.p));
        }
    } else {
        MF_TRY(chunk_nl.alloc(ctx, nc + 1));
        mf_buf<unsigned long long> dummy; MF_TRY(dummy.alloc(ctx, nc + 1));
        MF_HIP(hipMemsetAsync(dummy.p, 0, (nc + 1) * 8, st));
        {
            mf_ktimer tm(ctx, "k_dparse");
            k_dp_fq_count<<<(unsigned)nc, DP_T, 0, st>>>(raw.p, n, chunk_nl.p, flags.p);
            k_dp_scan2<<<1, 1024, 0, st>>>(chunk_nl.p, dummy.p, nc, totals.p);
        }
        MF_HIP(hipMemcpyAsync(h_tot, totals.p, 16, hipMemcpyDeviceToHost, st));
        uint8_t last = 0;
        MF_HIP(hipMemcpyAsync(&last, raw.p + (n - 1), 1, hipMemcpyDeviceToHost, st));
        const int a = anomaly();
        if (a < 0) return mf_set_error("device parser: %s", hipGetErrorString(hipGetLastError()));
        if (a) return 1;
        const uint64_t n_lines = h_tot[0] + (last != '\n' ? 1 : 0);
        if (n_lines % 4) return 1;                      // ("Unexpected end of file. File is corrupted/Format mismatch.": the host parser says it)
        n_rec = n_lines / 4;
        if (n_rec == 0) return 1;
        MF_TRY(rec_a.alloc(ctx, n_lines + 1)); MF_TRY(rec_b.alloc(ctx, n_rec + 1));
        MF_HIP(hipMemsetAsync(rec_b.p, 0, n_rec + 1, st));
        {
            mf_ktimer tm(ctx, "k_dparse");
            k_dp_set_u64<<<1, 1, 0, st>>>(rec_a.p + n_lines, last != '\n' ? (uint64_t)n + 1 : (uint64_t)n);
            k_dp_fq_lines<<<(unsigned)nc, DP_T, 0, st>>>(raw.p, n, chunk_nl.p, (uint32_t)qoff, rec_a.p, rec_b.p, flags.p);
        }
    }
    // records: keep / drop, places
    const uint64_t nb = (n_rec + DP_RPB - 1) / DP_RPB;
    MF_TRY(blk_r.alloc(ctx, nb)); MF_TRY(blk_b.alloc(ctx, nb)); MF_TRY(rec_place.alloc(ctx, n_rec + 1));
    {
        mf_ktimer tm(ctx, "k_dparse");
        if (fmt == 1) k_dp_rec_count<1><<<(unsigned)nb, DP_T, 0, st>>>(raw.p, rec_a.p, rec_b.p, n_rec, blk_r.p, blk_b.p, flags.p);
        else k_dp_rec_count<2><<<(unsigned)nb, DP_T, 0, st>>>(raw.p, rec_a.p, rec_b.p, n_rec, blk_r.p, blk_b.p, flags.p);
        k_dp_scan2<<<1, 1024, 0, st>>>(blk_r.p, blk_b.p, nb, totals.p);
    }
    MF_HIP(hipMemcpyAsync(h_tot, totals.p, 16, hipMemcpyDeviceToHost, st));
    {
        const int a = anomaly();
        if (a < 0) return mf_set_error("device parser: %s", hipGetErrorString(hipGetLastError()));
        if (a) return 1;
    }
    const uint64_t nr = h_tot[0], nbases = h_tot[1];
    MF_TRY(bases.alloc(ctx, nbases + 64)); MF_TRY(offsets.alloc(ctx, nr + 1));
    {
        mf_ktimer tm(ctx, "k_dparse_emit");                        // (only a file the device parser has accepted gets here)
        MF_HIP(hipMemsetAsync(rec_place.p + n_rec, 0xFF, 8, st));
        k_dp_set_u64<<<1, 1, 0, st>>>(offsets.p + nr, nbases);
        if (fmt == 1) {
            k_dp_rec_place<1><<<(unsigned)nb, DP_T, 0, st>>>(raw.p, rec_a.p, rec_b.p, n_rec, blk_r.p, blk_b.p, rec_place.p, offsets.p);
            k_dp_fa_emit<<<(unsigned)nc, DP_T, 0, st>>>(raw.p, n, pre.p, rec_place.p, bases.p);
        } else {
            k_dp_rec_place<2><<<(unsigned)nb, DP_T, 0, st>>>(raw.p, rec_a.p, rec_b.p, n_rec, blk_r.p, blk_b.p, rec_place.p, offsets.p);
            k_dp_fq_emit<<<(unsigned)nc, DP_T, 0, st>>>(raw.p, n, chunk_nl.p, rec_a.p, rec_place.p, bases.p);
        }
    }
    MF_HIP(hipGetLastError());
    MF_HIP(hipStreamSynchronize(st));
    *n_reads = nr; *n_bases = nbases;
    static const bool env = getenv("MF_IO_TIMING") != nullptr;
    if (env) fprintf(stderr, "[mf] device parser (%s, %.2f GB): read + H2D %.3f s, kernels %.3f s; %llu records, %llu reads kept\n", fmt == 1 ? "FASTA" : "FASTQ", n / 1e9,
                     t1 - t0, now() - t1, (unsigned long long)n_rec, (unsigned long long)nr);
    return 0;
}

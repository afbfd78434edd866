// mf_inflate.h -- DEFLATE (RFC 1951) in gzip members (RFC 1952), whole buffers in memory, host code.
//
// Why: reads arrive as .fastq.gz.  A gzip stream inflates on ONE thread, and that thread is the whole kmer-counter step for a compressed
// library (tools/gz_rate.py: 2.2 of 2.3 s for 5 M reads through zlib's inflate(), 0.34 GB/s of FASTA; the reference reads through
// java.util.zip.GZIPInputStream -- zlib as well -- FastaGZReader.java / FastqGZReader.java).  zlib's decoder is written for streaming: a state
// machine that can stop after any byte of input or output.  With the whole member in memory none of that is needed: a 64-bit bit buffer refilled
// by one unaligned load, one table look-up per symbol or pair of literals (12 bits for literals / lengths, 8 for distances, sub-tables behind them), copies in
// 8-byte steps, and bounds checked once per symbol against slack at both ends.
//
// Nothing depends on this decoder being right: gunzip() returns false on ANYTHING it does not like -- a malformed header, an invalid code, a
// distance beyond the output, a CRC-32 or length that does not match the member's trailer -- and the caller (mf_parse.h: inflate_gz) then
// inflates the file with zlib, which also words the error for a corrupt file.  tests/test_inflate_cpu.py drives both against Python's zlib.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>
#include <zlib.h>

namespace mfz {

struct entry { uint16_t val; uint8_t len; uint8_t op; };   // len: bits to take; op: see below; val: literal / base / sub-table offset
enum : uint8_t { OP_LIT = 0x80, OP_EOB = 0x40, OP_SUB = 0x20, OP_BAD = 0x10 };   // else: op = number of extra bits (0 .. 13), val = base
// OP_LIT | 1: TWO literals (val = first | second << 8, len = the bits of both codes): reads are literals mostly -- quality strings, bases that
// repeat nothing within 32 KB -- and a base costs 2 .. 3 bits, so one look-up of 12 bits often holds two symbols (build_table, kind 0)

static const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

#define MFZ_LBITS 12
#define MFZ_DBITS 8
#define MFZ_LTAB (1 << MFZ_LBITS)
#define MFZ_DTAB (1 << MFZ_DBITS)
#define MFZ_LTAB_MAX (MFZ_LTAB + 288 * 8)       // primary + the sub-tables (<= 2^(15 - 12) entries behind any primary slot that needs one)
#define MFZ_DTAB_MAX (MFZ_DTAB + 32 * 128)

static inline uint32_t rev_bits(uint32_t code, int len) {
    uint32_t r = 0;
    for (int i = 0; i < len; i++) { r = (r << 1) | (code & 1u); code >>= 1; }
    return r;
}
// kind 0: literal / length alphabet, 1: distances, 2: the code-length alphabet (7-bit table, symbols as literals).  false: over-subscribed or
// (but for one lone distance code, which zlib allows too) incomplete code lengths, or a symbol that does not exist.
static bool build_table(const uint8_t *lens, int n, int kind, int pbits, entry *tab, int tab_max) {
    int count[16] = {0};
    for (int i = 0; i < n; i++) count[lens[i]]++;
    if (count[0] == n) {                                                     // no code at all: every look-up is an error (a block of literals only: no distances)
        for (int i = 0; i < (1 << pbits); i++) tab[i] = entry{0, 1, OP_BAD};
        return kind == 1;
    }
    long left = 1;
    for (int l = 1; l <= 15; l++) { left = (left << 1) - count[l]; if (left < 0) return false; }
    const bool incomplete = left > 0;
    if (incomplete && !(kind == 1 && n - count[0] == 1)) return false;
    uint32_t next[16]; next[0] = 0; next[1] = 0;
    for (int l = 1; l < 15; l++) next[l + 1] = (next[l] + (uint32_t)count[l]) << 1;
    auto make = [&](int sym, int len) {
        entry e; e.len = (uint8_t)len;
        if (kind == 2) { e.val = (uint16_t)sym; e.op = OP_LIT; }
        else if (kind == 0) {
            if (sym < 256) { e.val = (uint16_t)sym; e.op = OP_LIT; }
            else if (sym == 256) { e.val = 0; e.op = OP_EOB; }
            else if (sym <= 285) { e.val = LEN_BASE[sym - 257]; e.op = LEN_EXTRA[sym - 257]; }
            else { e.val = 0; e.op = OP_BAD; }
        } else {
            if (sym < 30) { e.val = DIST_BASE[sym]; e.op = DIST_EXTRA[sym]; } else { e.val = 0; e.op = OP_BAD; }
        }
        return e;
    };
    const int psize = 1 << pbits;
    for (int i = 0; i < psize; i++) tab[i] = entry{0, 1, OP_BAD};            // (slots no code reaches: an incomplete distance code)
    // pass 1: the longest code behind every primary slot that needs a sub-table
    uint8_t sub_bits[MFZ_LTAB];                                              // (pbits <= MFZ_LBITS)
    memset(sub_bits, 0, (size_t)psize);
    uint32_t code_of[320];
    {
        uint32_t nx[16]; memcpy(nx, next, sizeof nx);
        for (int s = 0; s < n; s++) {
            const int l = lens[s];
            if (!l) continue;
            const uint32_t r = rev_bits(nx[l]++, l);
            code_of[s] = r;
            if (l > pbits) { uint8_t &b = sub_bits[r & (uint32_t)(psize - 1)]; if (l - pbits > b) b = (uint8_t)(l - pbits); }
        }
    }
    int used = psize;
    for (int i = 0; i < psize; i++)
        if (sub_bits[i]) {
            if (used + (1 << sub_bits[i]) > tab_max) return false;
            tab[i] = entry{(uint16_t)used, sub_bits[i], OP_SUB};             // len: bits of the sub-table's index
            for (int j = 0; j < (1 << sub_bits[i]); j++) tab[used + j] = entry{0, 1, OP_BAD};
            used += 1 << sub_bits[i];
        }
    for (int s = 0; s < n; s++) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t r = code_of[s];
        if (l <= pbits) { const entry e = make(s, l); for (uint32_t i = r; i < (uint32_t)psize; i += 1u << l) tab[i] = e; }
        else {
            const entry &P = tab[r & (uint32_t)(psize - 1)];
            const int sb = P.len, rest = l - pbits;
            const entry e = make(s, rest);                                   // (the primary bits are taken when the pointer is followed)
            for (uint32_t i = r >> pbits; i < (1u << sb); i += 1u << rest) tab[P.val + i] = e;
        }
    }
    if (kind == 0) {
        // pairs of literals: slot i = code of literal a in its low bits; if the bits above hold all of a second literal's code, the slot decodes both
        static thread_local entry single[MFZ_LTAB];
        memcpy(single, tab, sizeof(entry) * (size_t)psize);
        for (int i = 0; i < psize; i++) {
            const entry a = single[i];
            if (a.op != OP_LIT || a.len >= pbits) continue;
            const entry b = single[(uint32_t)i >> a.len];
            if (b.op != OP_LIT || a.len + b.len > pbits) continue;
            tab[i] = entry{(uint16_t)(a.val | (b.val << 8)), (uint8_t)(a.len + b.len), (uint8_t)(OP_LIT | 1)};
        }
    }
    return true;
}

struct out_buf {
    uint8_t *p = nullptr; size_t n = 0, cap = 0;
    bool reserve(size_t more) {                                              // room for `more` bytes behind n
        if (n + more <= cap) return true;
        size_t c = cap ? cap : ((size_t)1 << 20);
        while (c < n + more) c += c / 2;
        uint8_t *q = (uint8_t *)realloc(p, c);
        if (!q) return false;
        p = q; cap = c;
        return true;
    }
};

static inline uint64_t load64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline void store64(uint8_t *p, uint64_t v) { memcpy(p, &v, 8); }
static inline void store16(uint8_t *p, uint16_t v) { memcpy(p, &v, 2); }

// One raw DEFLATE stream from in[pos ..) (the caller guarantees 16 readable bytes behind in + n) appended to out; *pos: the first byte
// after the stream.  false: anything irregular.
static bool inflate_raw(const uint8_t *in, size_t n, size_t *pos, out_buf &out) {
    const uint8_t *p = in + *pos, *const end = in + n;
    uint64_t buf = 0; int cnt = 0;
    const size_t out_start = out.n;                                           // (distances may not reach before the member's first byte)
    static thread_local entry lt[MFZ_LTAB_MAX], dt[MFZ_DTAB_MAX];
    static thread_local entry fixed_lt[MFZ_LTAB_MAX], fixed_dt[MFZ_DTAB_MAX];
    static thread_local bool fixed_ready = false;
#define MFZ_REFILL() do { buf |= load64(p) << cnt; p += (63 - cnt) >> 3; cnt |= 56; } while (0)
#define MFZ_OVERRUN() ((size_t)(p - in) > n + (size_t)(cnt >> 3))           /* bits taken from beyond the input's end */
    for (;;) {
        MFZ_REFILL();
        const int final_block = (int)(buf & 1u), type = (int)((buf >> 1) & 3u);
        buf >>= 3; cnt -= 3;
        if (type == 0) {                                                     // stored: to the byte boundary, LEN, ~LEN, bytes
            const int drop = cnt & 7;
            buf >>= drop; cnt -= drop;
            p -= cnt >> 3; buf = 0; cnt = 0;                                 // (back to the first unused byte)
            if (p + 4 > end) return false;
            const uint32_t len = (uint32_t)p[0] | ((uint32_t)p[1] << 8), nlen = (uint32_t)p[2] | ((uint32_t)p[3] << 8);
            if ((len ^ nlen) != 0xFFFFu) return false;
            p += 4;
            if ((size_t)(end - p) < len) return false;
            if (!out.reserve((size_t)len + 512)) return false;
            memcpy(out.p + out.n, p, len);
            out.n += len; p += len;
        } else if (type == 3) return false;
        else {
            const entry *L, *D;
            if (type == 1) {
                if (!fixed_ready) {
                    uint8_t ll[288], dl[32];
                    for (int i = 0; i < 144; i++) ll[i] = 8;
                    for (int i = 144; i < 256; i++) ll[i] = 9;
                    for (int i = 256; i < 280; i++) ll[i] = 7;
                    for (int i = 280; i < 288; i++) ll[i] = 8;
                    for (int i = 0; i < 32; i++) dl[i] = 5;
                    if (!build_table(ll, 288, 0, MFZ_LBITS, fixed_lt, MFZ_LTAB_MAX) || !build_table(dl, 32, 1, MFZ_DBITS, fixed_dt, MFZ_DTAB_MAX)) return false;
                    fixed_ready = true;
                }
                L = fixed_lt; D = fixed_dt;
            } else {
                const int hlit = (int)(buf & 31u) + 257, hdist = (int)((buf >> 5) & 31u) + 1, hclen = (int)((buf >> 10) & 15u) + 4;
                buf >>= 14; cnt -= 14;
                if (hlit > 286 || hdist > 30) return false;
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                uint8_t cl[19] = {0};
                for (int i = 0; i < hclen; i++) {
                    if (cnt < 3) MFZ_REFILL();
                    cl[order[i]] = (uint8_t)(buf & 7u); buf >>= 3; cnt -= 3;
                }
                entry ct[128];
                if (!build_table(cl, 19, 2, 7, ct, 128)) return false;
                uint8_t lens[288 + 32];
                int i = 0;
                while (i < hlit + hdist) {
                    if (cnt < 7 + 7) MFZ_REFILL();
                    const entry e = ct[buf & 127u];
                    if (e.op != OP_LIT) return false;
                    buf >>= e.len; cnt -= e.len;
                    const int sym = e.val;
                    if (sym < 16) lens[i++] = (uint8_t)sym;
                    else {
                        int rep; uint8_t v = 0;
                        if (sym == 16) { if (!i) return false; v = lens[i - 1]; rep = 3 + (int)(buf & 3u); buf >>= 2; cnt -= 2; }
                        else if (sym == 17) { rep = 3 + (int)(buf & 7u); buf >>= 3; cnt -= 3; }
                        else { rep = 11 + (int)(buf & 127u); buf >>= 7; cnt -= 7; }
                        if (i + rep > hlit + hdist) return false;
                        while (rep--) lens[i++] = v;
                    }
                }
                if (MFZ_OVERRUN()) return false;
                if (lens[256] == 0) return false;                             // no end-of-block code
                if (!build_table(lens, hlit, 0, MFZ_LBITS, lt, MFZ_LTAB_MAX) || !build_table(lens + hlit, hdist, 1, MFZ_DBITS, dt, MFZ_DTAB_MAX)) return false;
                L = lt; D = dt;
            }
            // the symbols of the block
            for (;;) {
                if (out.cap - out.n < 1024) { if (!out.reserve((size_t)1 << 20)) return false; }
                uint8_t *o = out.p + out.n, *const o_lim = out.p + out.cap - 600;       // (a symbol writes <= 258 + 7 bytes; checked every 256 literals at most)
                bool eob = false;
                while (o < o_lim) {
                    if ((size_t)(p - in) > n + 8) return false;                // far beyond the input: a stream without an end
                    MFZ_REFILL();
                    entry e = L[buf & (MFZ_LTAB - 1)];
                    if (e.op & OP_SUB) { buf >>= MFZ_LBITS; cnt -= MFZ_LBITS; e = L[e.val + (buf & ((1u << e.len) - 1u))]; }
                    buf >>= e.len; cnt -= e.len;
                    if (e.op & OP_LIT) {
                        // one or two literals per look-up, up to four look-ups per refill (15 + 3 x 12 bits <= 56)
                        store16(o, e.val); o += 1 + (e.op & 1);
                        entry f = L[buf & (MFZ_LTAB - 1)];
                        if (f.op & OP_LIT) {
                            buf >>= f.len; cnt -= f.len; store16(o, f.val); o += 1 + (f.op & 1);
                            f = L[buf & (MFZ_LTAB - 1)];
                            if (f.op & OP_LIT) {
                                buf >>= f.len; cnt -= f.len; store16(o, f.val); o += 1 + (f.op & 1);
                                f = L[buf & (MFZ_LTAB - 1)];
                                if (f.op & OP_LIT) { buf >>= f.len; cnt -= f.len; store16(o, f.val); o += 1 + (f.op & 1); }
                            }
                        }
                        continue;
                    }
                    if (e.op & (OP_EOB | OP_BAD | OP_SUB)) { if (e.op & OP_EOB) { eob = true; break; } return false; }
                    const uint32_t len = e.val + (uint32_t)(buf & ((1u << e.op) - 1u));
                    buf >>= e.op; cnt -= e.op;
                    if (cnt < 32) MFZ_REFILL();
                    entry d = D[buf & (MFZ_DTAB - 1)];
                    if (d.op & OP_SUB) { buf >>= MFZ_DBITS; cnt -= MFZ_DBITS; d = D[d.val + (buf & ((1u << d.len) - 1u))]; }
                    buf >>= d.len; cnt -= d.len;
                    if (d.op & (OP_LIT | OP_EOB | OP_BAD | OP_SUB)) return false;
                    const uint32_t dist = d.val + (uint32_t)(buf & ((1u << d.op) - 1u));
                    buf >>= d.op; cnt -= d.op;
                    if ((size_t)dist > (size_t)(o - (out.p + out_start))) return false;
                    const uint8_t *s = o - dist;
                    uint8_t *const o_end = o + len;
                    if (dist >= 8) { do { store64(o, load64(s)); o += 8; s += 8; } while (o < o_end); }
                    else if (dist == 1) { const uint64_t v = 0x0101010101010101ull * s[0]; do { store64(o, v); o += 8; } while (o < o_end); }
                    else { do { *o++ = *s++; } while (o < o_end); }
                    o = o_end;
                }
                out.n = (size_t)(o - out.p);
                if (eob) break;
            }
            if (MFZ_OVERRUN()) return false;
        }
        if (final_block) break;
    }
    // back to the byte after the last one a bit was taken from
    p -= cnt >> 3;
    if (p > end) return false;
    *pos = (size_t)(p - in);
    return true;
#undef MFZ_REFILL
#undef MFZ_OVERRUN
}

// CRC-32 of a large buffer on several threads (zlib's crc32 + crc32_combine)
static uint32_t crc32_parallel(const uint8_t *p, size_t n, int threads) {
    const size_t T = (size_t)std::max(1, std::min<int>(threads, (int)(n / ((size_t)8 << 20)) + 1));
    if (T == 1) {
        uLong c = crc32(0L, Z_NULL, 0);
        for (size_t at = 0; at < n; at += (size_t)1 << 30) c = crc32(c, p + at, (uInt)std::min<size_t>(n - at, (size_t)1 << 30));
        return (uint32_t)c;
    }
    std::vector<uLong> part(T);
    std::vector<std::thread> th;
    for (size_t t = 0; t < T; t++)
        th.emplace_back([&, t]() {
            const size_t lo = n * t / T, hi = n * (t + 1) / T;
            uLong c = crc32(0L, Z_NULL, 0);
            for (size_t at = lo; at < hi; at += (size_t)1 << 30) c = crc32(c, p + at, (uInt)std::min<size_t>(hi - at, (size_t)1 << 30));
            part[t] = c;
        });
    for (auto &x : th) x.join();
    uLong c = part[0];
    for (size_t t = 1; t < T; t++) c = crc32_combine(c, part[t], (z_off_t)(n * (t + 1) / T - n * t / T));
    return (uint32_t)c;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// ONE member on SEVERAL threads.  A DEFLATE stream has no index, but its blocks can be found: a thread given a place in the middle of the file
// tries every bit position for the start of a dynamic block -- a header whose three Huffman codes are complete, symbols that decode to the
// block's end, a sane header behind it -- and decodes from there WITHOUT the 32 KB of output that came before: output elements are 16 bits
// wide, and what a match copies from before the thread's start is written as 256 + position in that unknown window.  Afterwards the windows are
// handed from piece to piece (the last 32 KB of each, a short serial pass) and every piece replaces its markers and lands in its place, with
// its CRC-32.  Each piece must END exactly where the next one was found to START, the last block must end at the member's trailer, and CRC-32
// and length must match: anything else is false, and the caller falls back to one thread.  (pugz, Kerbiriou & Chikhi 2019, is this idea for
// FASTQ; rapidgzip, Knespel & Brunst 2023, the general form.)  Files of several members (bgzip, `cat a.gz b.gz`) stay on one thread.
// where the several-thread form may put a member's bytes instead of one host buffer (mf_dparse.hip: pinned staging chunks on their way to HBM --
// the inflated file is never whole in host memory, and first-touch page faults on it were a third of the time)
struct byte_sink {
    int workers = 1;                  // threads that may fill slots at the same time
    size_t slot_bytes = 0;            // what acquire() gives
    virtual uint8_t *acquire(int worker) = 0;                                  // a slot to fill (waits until the worker's earlier use of it is over)
    virtual bool commit(int worker, size_t offset, size_t len) = 0;            // the slot holds bytes [offset, offset + len) of the member
    virtual ~byte_sink() {}
};
struct bitrd {
    const uint8_t *in; size_t n; const uint8_t *p; uint64_t buf; int cnt;
    void seek(size_t bit) { const size_t b = bit >> 3; const int s = (int)(bit & 7); buf = (uint64_t)in[b] >> s; cnt = 8 - s; p = in + b + 1; }
    size_t pos() const { return (size_t)(p - in) * 8 - (size_t)cnt; }
    void refill() { buf |= load64(p) << cnt; p += (63 - cnt) >> 3; cnt |= 56; }
    void take(int k) { buf >>= k; cnt -= k; }
    bool beyond() const { return (size_t)(p - in) > n + 8; }                   // (the caller's 64 readable bytes behind the input cover this much)
};
// the header of a dynamic block behind its three first bits -> the two tables
static bool read_dynamic(bitrd &r, entry *lt, entry *dt) {
    r.refill();
    const int hlit = (int)(r.buf & 31u) + 257, hdist = (int)((r.buf >> 5) & 31u) + 1, hclen = (int)((r.buf >> 10) & 15u) + 4;
    r.take(14);
    if (hlit > 286 || hdist > 30) return false;
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t cl[19] = {0};
    for (int i = 0; i < hclen; i++) { if (r.cnt < 3) r.refill(); cl[order[i]] = (uint8_t)(r.buf & 7u); r.take(3); }
    entry ct[128];
    if (!build_table(cl, 19, 2, 7, ct, 128)) return false;
    uint8_t lens[288 + 32];
    int i = 0;
    while (i < hlit + hdist) {
        if (r.cnt < 14) r.refill();
        const entry e = ct[r.buf & 127u];
        if (e.op != OP_LIT) return false;
        r.take(e.len);
        const int sym = e.val;
        if (sym < 16) lens[i++] = (uint8_t)sym;
        else {
            int rep; uint8_t v = 0;
            if (sym == 16) { if (!i) return false; v = lens[i - 1]; rep = 3 + (int)(r.buf & 3u); r.take(2); }
            else if (sym == 17) { rep = 3 + (int)(r.buf & 7u); r.take(3); }
            else { rep = 11 + (int)(r.buf & 127u); r.take(7); }
            if (i + rep > hlit + hdist) return false;
            while (rep--) lens[i++] = v;
        }
        if (r.beyond()) return false;
    }
    if (lens[256] == 0) return false;
    return build_table(lens, hlit, 0, MFZ_LBITS, lt, MFZ_LTAB_MAX) && build_table(lens + hlit, hdist, 1, MFZ_DBITS, dt, MFZ_DTAB_MAX);
}
static bool fixed_tables(const entry **L, const entry **D) {
    static thread_local entry flt[MFZ_LTAB_MAX], fdt[MFZ_DTAB_MAX];
    static thread_local bool ready = false;
    if (!ready) {
        uint8_t ll[288], dl[32];
        for (int i = 0; i < 144; i++) ll[i] = 8;
        for (int i = 144; i < 256; i++) ll[i] = 9;
        for (int i = 256; i < 280; i++) ll[i] = 7;
        for (int i = 280; i < 288; i++) ll[i] = 8;
        for (int i = 0; i < 32; i++) dl[i] = 5;
        if (!build_table(ll, 288, 0, MFZ_LBITS, flt, MFZ_LTAB_MAX) || !build_table(dl, 32, 1, MFZ_DBITS, fdt, MFZ_DTAB_MAX)) return false;
        ready = true;
    }
    *L = flt; *D = fdt;
    return true;
}
// does a dynamic block start at this bit?  (header, every symbol to the end of the block, the header of what follows)
static bool block_starts_at(const uint8_t *in, size_t n, size_t bit, entry *lt, entry *dt, entry *lt2, entry *dt2) {
    bitrd r{in, n, nullptr, 0, 0};
    r.seek(bit); r.refill();
    if ((r.buf & 7u) != 4u) return false;                                      // not final, dynamic
    r.take(3);
    if (!read_dynamic(r, lt, dt)) return false;
    size_t produced = 0;
    for (uint32_t syms = 0;; syms++) {
        if (syms > (1u << 22) || r.beyond()) return false;
        r.refill();
        entry e = lt[r.buf & (MFZ_LTAB - 1)];
        if (e.op & OP_SUB) { r.take(MFZ_LBITS); e = lt[e.val + (r.buf & ((1u << e.len) - 1u))]; }
        r.take(e.len);
        if (e.op & OP_LIT) { produced += 1 + (e.op & 1); continue; }
        if (e.op & OP_EOB) break;
        if (e.op & (OP_BAD | OP_SUB)) return false;
        r.take(e.op);
        if (r.cnt < 32) r.refill();
        entry d = dt[r.buf & (MFZ_DTAB - 1)];
        if (d.op & OP_SUB) { r.take(MFZ_DBITS); d = dt[d.val + (r.buf & ((1u << d.len) - 1u))]; }
        r.take(d.len);
        if (d.op & (OP_LIT | OP_EOB | OP_BAD | OP_SUB)) return false;
        r.take(d.op);
        produced += 3;
    }
    if (produced < 64 || r.pos() > n * 8) return false;                        // (a block of next to nothing: not where pieces are cut)
    r.refill();
    const uint32_t type = (uint32_t)(r.buf >> 1) & 3u;
    if (type == 3) return false;
    if (type == 2) { r.take(3); return read_dynamic(r, lt2, dt2); }
    if (type == 0) {
        r.take(3); r.take(r.cnt & 7);
        const uint8_t *q = r.p - (r.cnt >> 3);
        if (q + 4 > in + n) return false;
        return (((uint32_t)q[0] | ((uint32_t)q[1] << 8)) ^ ((uint32_t)q[2] | ((uint32_t)q[3] << 8))) == 0xFFFFu;
    }
    return true;
}
static size_t find_block(const uint8_t *in, size_t n, size_t from_bit, size_t to_bit) {       // the first bit in [from, to) where a block starts; SIZE_MAX: none
    static thread_local entry a[MFZ_LTAB_MAX], b[MFZ_DTAB_MAX], c[MFZ_LTAB_MAX], d[MFZ_DTAB_MAX];
    for (size_t bit = from_bit; bit < to_bit && (bit >> 3) + 8 < n; bit++) {
        const uint64_t w = load64(in + (bit >> 3)) >> (bit & 7);
        if ((w & 7u) != 4u || ((w >> 3) & 31u) > 29u || ((w >> 8) & 31u) > 29u) continue;
        if (block_starts_at(in, n, bit, a, b, c, d)) return bit;
    }
    return (size_t)-1;
}
struct marked_out {
    uint16_t *p = nullptr; size_t n = 0, cap = 0;
    ~marked_out() { free(p); }
    bool reserve(size_t more) {
        if (n + more <= cap) return true;
        size_t c = cap ? cap : ((size_t)1 << 20);
        while (c < n + more) c += c / 2;
        uint16_t *q = (uint16_t *)realloc(p, c * 2);
        if (!q) return false;
        p = q; cap = c;
        return true;
    }
};
// blocks from bit `start` until one would start at `stop` (-> *end_bit = stop) or the final block has ended (-> *final_seen, *end_bit behind it).
// first: nothing precedes `start` (a distance beyond the output is an error); else such distances become markers 256 + window position
static bool inflate_piece(const uint8_t *in, size_t n, size_t start, size_t stop, bool first, marked_out &out, size_t *end_bit, bool *final_seen) {
    static thread_local entry lt[MFZ_LTAB_MAX], dt[MFZ_DTAB_MAX];
    bitrd r{in, n, nullptr, 0, 0};
    r.seek(start);
    *final_seen = false;
    for (;;) {
        const size_t at = r.pos();
        if (at == stop) { *end_bit = at; return true; }
        if (at > stop) return false;
        r.refill();
        const int final_block = (int)(r.buf & 1u), type = (int)((r.buf >> 1) & 3u);
        r.take(3);
        if (type == 3) return false;
        if (type == 0) {
            r.take(r.cnt & 7);
            const uint8_t *q = r.p - (r.cnt >> 3);
            if (q + 4 > in + n) return false;
            const uint32_t len = (uint32_t)q[0] | ((uint32_t)q[1] << 8), nlen = (uint32_t)q[2] | ((uint32_t)q[3] << 8);
            if ((len ^ nlen) != 0xFFFFu) return false;
            q += 4;
            if ((size_t)(in + n - q) < len) return false;
            if (!out.reserve((size_t)len + 1024)) return false;
            for (uint32_t i = 0; i < len; i++) out.p[out.n + i] = q[i];
            out.n += len;
            r.seek((size_t)(q + len - in) * 8);
        } else {
            const entry *L, *D;
            if (type == 1) { if (!fixed_tables(&L, &D)) return false; }
            else { if (!read_dynamic(r, lt, dt)) return false; L = lt; D = dt; }
            for (bool eob = false; !eob;) {
                if (out.cap - out.n < 2048 && !out.reserve((size_t)1 << 20)) return false;
                uint16_t *o = out.p + out.n, *const o_lim = out.p + out.cap - 600;
                while (o < o_lim) {
                    if (r.beyond()) return false;
                    r.refill();
                    entry e = L[r.buf & (MFZ_LTAB - 1)];
                    if (e.op & OP_SUB) { r.take(MFZ_LBITS); e = L[e.val + (r.buf & ((1u << e.len) - 1u))]; }
                    r.take(e.len);
                    if (e.op & OP_LIT) {
                        o[0] = (uint16_t)(e.val & 255u); o[1] = (uint16_t)(e.val >> 8); o += 1 + (e.op & 1);
                        for (int more = 0; more < 3; more++) {
                            const entry f = L[r.buf & (MFZ_LTAB - 1)];
                            if (!(f.op & OP_LIT)) break;
                            r.take(f.len);
                            o[0] = (uint16_t)(f.val & 255u); o[1] = (uint16_t)(f.val >> 8); o += 1 + (f.op & 1);
                        }
                        continue;
                    }
                    if (e.op & OP_EOB) { eob = true; break; }
                    if (e.op & (OP_BAD | OP_SUB)) return false;
                    const uint32_t len = e.val + (uint32_t)(r.buf & ((1u << e.op) - 1u));
                    r.take(e.op);
                    if (r.cnt < 32) r.refill();
                    entry d = D[r.buf & (MFZ_DTAB - 1)];
                    if (d.op & OP_SUB) { r.take(MFZ_DBITS); d = D[d.val + (r.buf & ((1u << d.len) - 1u))]; }
                    r.take(d.len);
                    if (d.op & (OP_LIT | OP_EOB | OP_BAD | OP_SUB)) return false;
                    const uint32_t dist = d.val + (uint32_t)(r.buf & ((1u << d.op) - 1u));
                    r.take(d.op);
                    const size_t have = (size_t)(o - out.p);
                    uint32_t j = 0;
                    if ((size_t)dist > have) {                                 // (part of) the source lies before this piece
                        if (first) return false;
                        const uint32_t before = (uint32_t)std::min<size_t>(len, (size_t)dist - have);
                        const uint32_t w0 = 32768u - (uint32_t)((size_t)dist - have);          // window position of the first source element
                        for (; j < before; j++) o[j] = (uint16_t)(256u + w0 + j);
                    }
                    if (dist >= len && j == 0) memcpy(o, o - dist, (size_t)len * 2);
                    else for (; j < len; j++) o[j] = *(o + j - dist);
                    o += len;
                }
                out.n = (size_t)(o - out.p);
            }
            if (r.pos() > n * 8) return false;
        }
        if (final_block) { *final_seen = true; *end_bit = r.pos(); return true; }
    }
}
// a member's raw stream in [d0, n - 8) on `threads` threads, appended to out.  false: one thread has to do it
static bool inflate_parallel(const uint8_t *in, size_t n, size_t d0, int threads, out_buf &out, uint32_t *crc_out, size_t piece_min, byte_sink *sink = nullptr,
                             size_t *sink_total = nullptr) {
    // (not more: every piece in flight holds its output twice over in 16-bit elements, and memory touched for the first time costs more than
    // decoding it -- 0.5 s per GB on this pool's boxes, whatever the number of threads: tools/page_fault_rate.cpp)
    const size_t T = (size_t)std::max(2, std::min(threads, 32));
    const size_t body = n - 8 - d0;
    const bool dbg = getenv("MF_INFLATE_DEBUG") != nullptr;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_0 = now();
    double t_dec = 0, t_win = 0, t_res = 0;
    const size_t piece = std::max<size_t>(piece_min, std::min<size_t>(body / (T * 4) + 1, (size_t)8 << 20));
    const size_t np0 = (body + piece - 1) / piece;
    if (np0 < 2) return false;
    // where the pieces start: piece i at the first block found at or behind d0 + i * piece (none found before the next one's place: no cut there)
    std::vector<size_t> found(np0, (size_t)-1);
    found[0] = d0 * 8;
    {
        std::atomic<size_t> next{1};
        std::vector<std::thread> th;
        for (size_t t = 0; t < std::min(T, np0); t++)
            th.emplace_back([&]() { for (;;) { const size_t i = next++; if (i >= np0) break; found[i] = find_block(in, n - 8, (d0 + i * piece) * 8, (d0 + std::min(body, (i + 1) * piece)) * 8); } });
        for (auto &x : th) x.join();
    }
    std::vector<size_t> starts;
    for (size_t i = 0; i < np0; i++) if (found[i] != (size_t)-1) starts.push_back(found[i]);
    const size_t np = starts.size();
    if (np < 2) return false;
    const double t_find = now() - t_0;
    // waves of pieces: decode (parallel), hand the windows on (serial), resolve the markers into place with a CRC each (parallel)
    std::vector<uint8_t> window(32768, 0);
    size_t total = 0;                                                           // bytes of this member so far
    uLong crc = crc32(0L, Z_NULL, 0);
    const size_t out_start = out.n;
    const size_t wave = T;
    std::vector<marked_out> mo(std::min(wave, np));                             // (reused wave after wave: memory touched once)
    for (auto &M : mo) if (!M.reserve(piece * 7 / 2)) return false;
    for (size_t w0 = 0; w0 < np; w0 += wave) {
        const size_t w1 = std::min(np, w0 + wave), m = w1 - w0;
        for (size_t j = 0; j < m; j++) mo[j].n = 0;
        const double t_a = now();
        std::vector<size_t> endb(m, 0);
        std::vector<char> ok(m, 0), fin(m, 0);
        {
            std::atomic<size_t> next{0};
            std::vector<std::thread> th;
            for (size_t t = 0; t < std::min(T, m); t++)
                th.emplace_back([&]() {
                    for (;;) {
                        const size_t j = next++;
                        if (j >= m) break;
                        const size_t i = w0 + j;
                        bool f = false;
                        ok[j] = inflate_piece(in, n - 8, starts[i], i + 1 < np ? starts[i + 1] : (size_t)-1, i == 0, mo[j], &endb[j], &f) ? 1 : 0;
                        fin[j] = f ? 1 : 0;
                    }
                });
            for (auto &x : th) x.join();
        }
        for (size_t j = 0; j < m; j++) {
            const size_t i = w0 + j;
            if (!ok[j]) return false;
            if (i + 1 < np) { if (fin[j] || endb[j] != starts[i + 1]) return false; }
            else if (!fin[j] || (endb[j] + 7) / 8 != n - 8) return false;         // the last block ends at the trailer
        }
        const double t_b = now(); t_dec += t_b - t_a;
        // windows: wins[j] = the 32 KB before piece w0 + j
        std::vector<std::vector<uint8_t>> wins(m);
        std::vector<size_t> offs(m);
        size_t grow = 0;
        for (size_t j = 0; j < m; j++) {
            wins[j] = window;
            offs[j] = total;
            const marked_out &M = mo[j];
            // markers must not reach before the member's first byte
            const size_t known = std::min<size_t>(total, 32768);
            std::vector<uint8_t> nw(32768, 0);
            const size_t tail = std::min<size_t>(M.n, 32768);
            for (size_t q = 0; q < 32768 - tail; q++) nw[q] = window[q + tail];
            for (size_t q = 0; q < tail; q++) {
                const uint16_t v = M.p[M.n - tail + q];
                if (v >= 256) { const uint32_t wp = (uint32_t)v - 256u; if (wp < 32768 - known) return false; nw[32768 - tail + q] = window[wp]; }
                else nw[32768 - tail + q] = (uint8_t)v;
            }
            window.swap(nw);
            total += M.n; grow += M.n;
        }
        if (!sink && !out.reserve(grow + 64)) return false;
        const double t_c = now(); t_win += t_c - t_b;
        if (sink) {
            // items: a piece's output in runs of <= slot_bytes, resolved into the sink's slots; CRCs joined in order afterwards
            struct item { size_t j, q0, len; };
            std::vector<item> items;
            for (size_t j = 0; j < m; j++) for (size_t q0 = 0; q0 < mo[j].n; q0 += sink->slot_bytes) items.push_back(item{j, q0, std::min(sink->slot_bytes, mo[j].n - q0)});
            std::vector<uLong> ic(items.size(), 0);
            std::atomic<size_t> next{0};
            std::atomic<int> bad{0};
            std::vector<std::thread> th;
            for (int t = 0; t < std::max(1, sink->workers); t++)
                th.emplace_back([&, t]() {
                    for (;;) {
                        const size_t a = next++;
                        if (a >= items.size() || bad.load()) break;
                        const item &I = items[a];
                        const marked_out &M = mo[I.j];
                        uint8_t *dst = sink->acquire(t);
                        if (!dst) { bad = 1; break; }
                        const uint8_t *W = wins[I.j].data();
                        const size_t known = std::min<size_t>(offs[I.j], 32768);
                        const uint16_t *src = M.p + I.q0;
                        bool g = true;
                        for (size_t q = 0; q < I.len; q++) {
                            const uint16_t v = src[q];
                            if (v >= 256) { const uint32_t wp = (uint32_t)v - 256u; if (wp < 32768 - known) { g = false; break; } dst[q] = W[wp]; }
                            else dst[q] = (uint8_t)v;
                        }
                        if (!g) { bad = 1; break; }
                        ic[a] = crc32(crc32(0L, Z_NULL, 0), dst, (uInt)I.len);
                        if (!sink->commit(t, offs[I.j] + I.q0, I.len)) { bad = 1; break; }
                    }
                });
            for (auto &x : th) x.join();
            if (bad.load()) return false;
            for (size_t a = 0; a < items.size(); a++) crc = crc32_combine(crc, ic[a], (z_off_t)items[a].len);
            t_res += now() - t_c;
            continue;
        }
        std::vector<uLong> crcs(m, 0);
        std::vector<char> good(m, 1);
        {
            std::atomic<size_t> next{0};
            std::vector<std::thread> th;
            for (size_t t = 0; t < std::min(T, m); t++)
                th.emplace_back([&]() {
                    for (;;) {
                        const size_t j = next++;
                        if (j >= m) break;
                        const marked_out &M = mo[j];
                        uint8_t *dst = out.p + out_start + offs[j];
                        const uint8_t *W = wins[j].data();
                        const size_t known = std::min<size_t>(offs[j], 32768);
                        bool g = true;
                        for (size_t q = 0; q < M.n; q++) {
                            const uint16_t v = M.p[q];
                            if (v >= 256) { const uint32_t wp = (uint32_t)v - 256u; if (wp < 32768 - known) { g = false; break; } dst[q] = W[wp]; }
                            else dst[q] = (uint8_t)v;
                        }
                        good[j] = g ? 1 : 0;
                        uLong c = crc32(0L, Z_NULL, 0);
                        for (size_t at = 0; g && at < M.n; at += (size_t)1 << 30) c = crc32(c, dst + at, (uInt)std::min<size_t>(M.n - at, (size_t)1 << 30));
                        crcs[j] = c;
                    }
                });
            for (auto &x : th) x.join();
        }
        for (size_t j = 0; j < m; j++) { if (!good[j]) return false; crc = crc32_combine(crc, crcs[j], (z_off_t)mo[j].n); }
        out.n += grow;
        t_res += now() - t_c;
    }
    if (dbg) fprintf(stderr, "[mf] inflate: %zu pieces of %zu bytes on %zu threads: block search %.3f s, decoding %.3f s, windows %.3f s, markers + CRC %.3f s\n", np, piece, T, t_find, t_dec, t_win, t_res);
    *crc_out = (uint32_t)crc;
    if (sink_total) *sink_total = total;
    return true;
}

// BGZF (bgzip, the blocked gzip of htslib): members of <= 64 KB whose FEXTRA field "BC" says how long each one is -- the members can be
// counted off by their headers alone and inflate independently, every thread a run of them, each checked against its own CRC-32 and length.
// false: not (entirely) BGZF, or anything irregular.
static bool gunzip_bgzf(const uint8_t *in, size_t n, int threads, char **out_p, size_t *out_n) {
    struct member { size_t at, csize; uint32_t isize; };
    std::vector<member> ms;
    size_t pos = 0, total = 0;
    while (pos < n) {
        if (n - pos < 28 || in[pos] != 0x1F || in[pos + 1] != 0x8B || in[pos + 2] != 8 || in[pos + 3] != 4) return false;      // FEXTRA and nothing else
        const size_t xlen = (size_t)in[pos + 10] | ((size_t)in[pos + 11] << 8);
        if (pos + 12 + xlen + 8 > n) return false;
        size_t bsize = 0;
        for (size_t q = pos + 12; q + 4 <= pos + 12 + xlen;) {
            const size_t sl = (size_t)in[q + 2] | ((size_t)in[q + 3] << 8);
            if (in[q] == 'B' && in[q + 1] == 'C' && sl == 2 && q + 6 <= pos + 12 + xlen) bsize = ((size_t)in[q + 4] | ((size_t)in[q + 5] << 8)) + 1;
            q += 4 + sl;
        }
        if (bsize < 12 + xlen + 8 + 2 || pos + bsize > n) return false;
        const uint8_t *tr = in + pos + bsize - 8;
        const uint32_t isize = (uint32_t)tr[4] | ((uint32_t)tr[5] << 8) | ((uint32_t)tr[6] << 16) | ((uint32_t)tr[7] << 24);
        if (isize > 65536) return false;
        ms.push_back(member{pos + 12 + xlen, bsize - 12 - xlen - 8, isize});
        total += isize; pos += bsize;
    }
    if (ms.size() < 2) return false;
    uint8_t *out = (uint8_t *)malloc(total + 64);
    if (!out) return false;
    std::vector<size_t> off(ms.size());
    { size_t a = 0; for (size_t i = 0; i < ms.size(); i++) { off[i] = a; a += ms[i].isize; } }
    const size_t T = (size_t)std::max(1, std::min<int>(threads, 64));
    std::atomic<size_t> next{0};
    std::atomic<int> bad{0};
    std::vector<std::thread> th;
    for (size_t t = 0; t < std::min(T, ms.size() / 64 + 1); t++)
        th.emplace_back([&]() {
            out_buf scratch;                                                  // (the decoder writes a little past what it returns: not into a neighbour's place)
            for (;;) {
                const size_t i0 = next.fetch_add(64);
                if (i0 >= ms.size() || bad.load()) break;
                for (size_t i = i0; i < std::min(ms.size(), i0 + 64); i++) {
                    const member &m = ms[i];
                    scratch.n = 0;
                    size_t q = m.at;
                    // (the member's stream must end where its trailer begins, and give what the trailer says)
                    if (!inflate_raw(in, m.at + m.csize, &q, scratch) || q != m.at + m.csize || scratch.n != m.isize) { bad = 1; break; }
                    const uint8_t *tr = in + m.at + m.csize;
                    const uint32_t want = (uint32_t)tr[0] | ((uint32_t)tr[1] << 8) | ((uint32_t)tr[2] << 16) | ((uint32_t)tr[3] << 24);
                    if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), scratch.p, (uInt)scratch.n) != want) { bad = 1; break; }
                    memcpy(out + off[i], scratch.p, scratch.n);
                }
            }
            free(scratch.p);
        });
    for (auto &x : th) x.join();
    if (bad.load()) { free(out); return false; }
    *out_p = (char *)out; *out_n = total;
    return true;
}

// ONE member on several threads straight into a sink: true and *total when all of it went there and CRC-32 and length match; false -- several
// members, BGZF, anything irregular, a sink that refuses -- when the caller has to take the file some other way (what the sink got is void then)
static bool gunzip_to_sink(const uint8_t *in, size_t n, int threads, byte_sink *sink, size_t *total, size_t piece_min = (size_t)2 << 20) {
    if (n < 18 + 8 || in[0] != 0x1F || in[1] != 0x8B || in[2] != 8 || (in[3] & 0xE0)) return false;
    const uint8_t flg = in[3];
    size_t q = 10;
    if (flg & 4) { const size_t xl = (size_t)in[q] | ((size_t)in[q + 1] << 8); q += 2 + xl; }
    if (flg & 8) { while (q < n && in[q]) q++; q++; }
    if (flg & 16) { while (q < n && in[q]) q++; q++; }
    if (flg & 2) q += 2;
    if (q + 8 >= n) return false;
    out_buf none;
    uint32_t crc = 0;
    size_t tot = 0;
    if (!inflate_parallel(in, n, q, threads, none, &crc, piece_min, sink, &tot)) return false;
    const uint8_t *tr = in + n - 8;
    const uint32_t want_crc = (uint32_t)tr[0] | ((uint32_t)tr[1] << 8) | ((uint32_t)tr[2] << 16) | ((uint32_t)tr[3] << 24);
    const uint32_t want_len = (uint32_t)tr[4] | ((uint32_t)tr[5] << 8) | ((uint32_t)tr[6] << 16) | ((uint32_t)tr[7] << 24);
    if ((uint32_t)tot != want_len || crc != want_crc) return false;
    *total = tot;
    return true;
}

// A file of one or more gzip members, whole in memory with 16 readable bytes behind in + n -> *out_p (malloc), *out_n.  false (nothing
// allocated is left behind): the caller inflates with zlib instead.
static bool gunzip(const uint8_t *in, size_t n, int threads, char **out_p, size_t *out_n, size_t parallel_min = (size_t)32 << 20, size_t piece_min = (size_t)2 << 20,
                   bool *went_parallel = nullptr) {
    if (n >= 28 && in[3] == 4 && gunzip_bgzf(in, n, threads, out_p, out_n)) return true;
    out_buf out;
    if (!out.reserve(std::max<size_t>(n * 5, (size_t)1 << 20))) return false;
    size_t pos = 0;
    bool ok = n > 0;
    while (ok && pos < n) {
        // member header (RFC 1952)
        if (n - pos < 18 || in[pos] != 0x1F || in[pos + 1] != 0x8B || in[pos + 2] != 8) { ok = false; break; }
        const uint8_t flg = in[pos + 3];
        if (flg & 0xE0) { ok = false; break; }
        size_t q = pos + 10;
        if (flg & 4) { if (q + 2 > n) { ok = false; break; } const size_t xl = (size_t)in[q] | ((size_t)in[q + 1] << 8); q += 2 + xl; }
        if (flg & 8) { while (q < n && in[q]) q++; q++; }
        if (flg & 16) { while (q < n && in[q]) q++; q++; }
        if (flg & 2) q += 2;
        if (q >= n) { ok = false; break; }
        const size_t member_start = out.n;
        uint32_t have_crc = 0;
        bool have = false;
        if (pos == 0 && threads >= 2 && n - q >= parallel_min) {
            // the file as ONE member on several threads (whatever goes wrong in there -- a second member among it -- leaves out as it was)
            if (inflate_parallel(in, n, q, threads, out, &have_crc, piece_min)) { have = true; q = n - 8; } else out.n = member_start;
            if (went_parallel) *went_parallel = have;
            if (getenv("MF_INFLATE_DEBUG")) fprintf(stderr, "[mf] inflate: %zu bytes on %d threads: %s\n", n, threads, have ? "done" : "refused, one thread");
        }
        if (!have && !inflate_raw(in, n, &q, out)) { ok = false; break; }
        if (q + 8 > n) { ok = false; break; }
        const uint32_t want_crc = (uint32_t)in[q] | ((uint32_t)in[q + 1] << 8) | ((uint32_t)in[q + 2] << 16) | ((uint32_t)in[q + 3] << 24);
        const uint32_t want_len = (uint32_t)in[q + 4] | ((uint32_t)in[q + 5] << 8) | ((uint32_t)in[q + 6] << 16) | ((uint32_t)in[q + 7] << 24);
        if ((uint32_t)(out.n - member_start) != want_len) { ok = false; break; }
        if ((have ? have_crc : crc32_parallel(out.p + member_start, out.n - member_start, threads)) != want_crc) { ok = false; break; }
        pos = q + 8;
    }
    if (!ok) { free(out.p); return false; }
    *out_p = (char *)out.p; *out_n = out.n;
    return true;
}

}   // namespace mfz

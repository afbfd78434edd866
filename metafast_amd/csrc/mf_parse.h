// mf_parse.h -- the host-side parsers of the file seams: FASTA / FASTQ / .gz / .bz2 / .binq readers, the record cutters of the parallel
// and the streaming reader, the header walk of components.bin.  Plain C++ (no HIP): mf_io.hip includes it, and so does the sanitizer
// harness tests/host/parse_harness.cpp (g++ -fsanitize=address,undefined; tests/test_host_sanitized_cpu.py) -- these functions read
// untrusted files.
// Reference: itmo!/io/readers/FastaReader.java:53-104, FastqReader.java:53-82, BinqReader.java, src/structures/ConnectedComponent.java:95-122.
#pragma once
#include <zlib.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include "mf_inflate.h"
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#ifndef MF_OK
#define MF_OK 0
#define MF_ERR (-1)
#endif
#ifndef MF_TRY
#define MF_TRY(x) do { int rc__ = (x); if (rc__ < 0) return rc__; } while (0)
#endif
int mf_set_error(const char *fmt, ...);
extern "C" const char *mf_last_error(void);

static bool ends_with_nocase(const std::string &s, const char *suf) {
    size_t m = strlen(suf);
    if (m > s.size()) return false;
    for (size_t i = 0; i < m; i++) if (tolower((unsigned char)s[s.size() - m + i]) != suf[i]) return false;
    return true;
}
static void be_put(uint8_t *p, uint64_t v, int nb) { for (int i = 0; i < nb; i++) p[i] = (uint8_t)(v >> (8 * (nb - 1 - i))); }
static uint64_t be_get(const uint8_t *p, int nb) { uint64_t v = 0; for (int i = 0; i < nb; i++) v = (v << 8) | p[i]; return v; }

// ---------------------------------------------------------------------------------------------
// A1 readers
// ---------------------------------------------------------------------------------------------
// growable byte buffer WITHOUT value-initialisation (a std::vector would zero gigabytes before every parse)
struct byte_buf {
    uint8_t *p = nullptr; size_t n = 0, cap = 0;
    bool ext = false;                // p is somebody else's fixed region (a pinned staging chunk): never reallocated or freed
    byte_buf() {}
    void use_external(uint8_t *q, size_t c) { if (!ext) free(p); p = q; n = 0; cap = c; ext = true; }
    byte_buf(const byte_buf &) = delete;
    byte_buf &operator=(const byte_buf &) = delete;
    byte_buf(byte_buf &&o) noexcept : p(o.p), n(o.n), cap(o.cap), ext(o.ext) { o.p = nullptr; o.n = o.cap = 0; o.ext = false; }
    byte_buf &operator=(byte_buf &&o) noexcept { if (!ext) free(p); p = o.p; n = o.n; cap = o.cap; ext = o.ext; o.p = nullptr; o.n = o.cap = 0; o.ext = false; return *this; }
    ~byte_buf() { if (!ext) free(p); }
    void reserve(size_t c) { if (c > cap && !ext) { p = (uint8_t *)realloc(p, c); cap = c; } }     // (an external region is as large as its input)
    uint8_t *grow(size_t add) { if (n + add > cap) reserve(std::max(cap * 2, n + add + 4096)); return p + n; }   // room for `add` more bytes
    void push_back(uint8_t c) { *grow(1) = c; n++; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    void resize(size_t m) { n = m; }                 // shrink only
    const uint8_t *data() const { return p; }
};
struct read_batch {
    byte_buf bases;                  // upper-case ACGT
    std::vector<uint64_t> offsets;   // [n+1]
    read_batch() { offsets.push_back(0); }
    void end_read() { offsets.push_back(bases.size()); }
    void drop_read() { bases.resize(offsets.back()); }
};
// byte -> base (DnaTools.fromChar, itmo!/dna/DnaTools.java:46-64; IUPAC codes: first listed choice, see nucleotide_of),
// 0xFE = N/n (the whole read is skipped), 0xFF = not a nucleotide
struct base_lut {
    uint8_t t[256];
    base_lut() {
        for (int c = 0; c < 256; c++) t[c] = 0xFF;
        const char *from = "ACGTacgtRrYyMmKkSsWwHhBbVvDd", *to = "ACGTACGTGGTTAAGGGGAAAAGGAAAA";
        for (int i = 0; from[i]; i++) t[(unsigned char)from[i]] = (uint8_t)to[i];
        t['N'] = t['n'] = 0xFE;
    }
};
static const base_lut BASE_LUT;
// one text line at a time; BufferedReader.readLine semantics (\n, \r or \r\n)
struct line_reader {
    const char *b; size_t n, pos;
    bool next(const char **ln, size_t *len) {
        if (pos >= n) return false;
        size_t s = pos, e = s;
        while (e < n && b[e] != '\n' && b[e] != '\r') e++;
        *ln = b + s; *len = e - s;
        if (e < n) e += (b[e] == '\r' && e + 1 < n && b[e + 1] == '\n') ? 2 : 1;
        pos = e;
        return true;
    }
};
// DnaTools.fromChar (itmo!/dna/DnaTools.java:46-64).  IUPAC codes: the reference picks one of the allowed bases at RANDOM
// (:66-117); this implementation takes the first one listed there, which is one of the reference's possible outcomes.
static int nucleotide_of(int c) {
    switch (c) {
    case 'A': case 'a': return 'A'; case 'C': case 'c': return 'C'; case 'G': case 'g': return 'G'; case 'T': case 't': return 'T';
    case 'R': case 'r': return 'G'; case 'Y': case 'y': return 'T'; case 'M': case 'm': return 'A'; case 'K': case 'k': return 'G';
    case 'S': case 's': return 'G'; case 'W': case 'w': return 'A'; case 'H': case 'h': return 'A'; case 'B': case 'b': return 'G';
    case 'V': case 'v': return 'A'; case 'D': case 'd': return 'A';
    default: return -1;
    }
}
// Is the line upper-case A / C / G / T and nothing else?  Then it is its own translation (one memcpy) -- what nearly every line of a
// read file is; everything else (lower case, N, IUPAC codes, a stray CR, a wrong character) takes the byte-by-byte table below.
// 32 bytes per step where the host has AVX2 (checked once at run time), 8 per step otherwise.
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx2"))) static bool all_acgt_avx2(const char *p, size_t n) {
    const __m256i A = _mm256_set1_epi8('A'), C = _mm256_set1_epi8('C'), G = _mm256_set1_epi8('G'), T = _mm256_set1_epi8('T');
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        const __m256i v = _mm256_loadu_si256((const __m256i *)(p + i));
        const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(v, A), _mm256_cmpeq_epi8(v, C)), _mm256_or_si256(_mm256_cmpeq_epi8(v, G), _mm256_cmpeq_epi8(v, T)));
        if ((uint32_t)_mm256_movemask_epi8(ok) != 0xFFFFFFFFu) return false;
    }
    for (; i < n; i++) { const char c = p[i]; if (c != 'A' && c != 'C' && c != 'G' && c != 'T') return false; }
    return true;
}
#endif
#if defined(__x86_64__)
__attribute__((target("avx2"))) static bool all_in_range_avx2(const char *p, size_t n, uint8_t lo, uint8_t hi) {
    const __m256i L = _mm256_set1_epi8((char)lo), H = _mm256_set1_epi8((char)hi);
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        const __m256i v = _mm256_loadu_si256((const __m256i *)(p + i));
        const __m256i c = _mm256_min_epu8(_mm256_max_epu8(v, L), H);                   // v clamped to [lo, hi]: unchanged iff inside
        if ((uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(c, v)) != 0xFFFFFFFFu) return false;
    }
    for (; i < n; i++) { const uint8_t c = (uint8_t)p[i]; if (c < lo || c > hi) return false; }
    return true;
}
#endif
static bool all_in_range(const char *p, size_t n, uint8_t lo, uint8_t hi) {          // every byte in [lo, hi]?
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) return all_in_range_avx2(p, n, lo, hi);
#endif
    for (size_t i = 0; i < n; i++) { const uint8_t c = (uint8_t)p[i]; if (c < lo || c > hi) return false; }
    return true;
}
static bool all_acgt(const char *p, size_t n) {
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) return all_acgt_avx2(p, n);
#endif
    // eight bytes at a time: x ^ "AAAAAAAA" leaves 0x00 (A), 0x02 (C), 0x06 (G), 0x15 (T); a byte b of those is valid iff
    // b & 0xE8 == 0 (b in 0..7 or 0x10..0x17) and the bit pattern 0x00200045 has bit (b & 7) | (b >> 1 & 8) set -- by table instead:
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t x; memcpy(&x, p + i, 8);
        x ^= 0x4141414141414141ull;
        for (int j = 0; j < 8; j++) { const uint8_t b = (uint8_t)(x >> (8 * j)); if (b != 0x00 && b != 0x02 && b != 0x06 && b != 0x15) return false; }
    }
    for (; i < n; i++) { const char c = p[i]; if (c != 'A' && c != 'C' && c != 'G' && c != 'T') return false; }
    return true;
}
// FastaReader (itmo!/io/readers/FastaReader.java:53-104): records = concatenation of the non-comment lines between
// '>'/';' lines; a record containing N/n is skipped
static int parse_fasta(const char *data, size_t size, const char *path, read_batch &rb) {
    size_t pos = 0;
    bool have = false, has_n = false;
    int bad = -1;
    for (;;) {
        // one line: [ln, ln+len), BufferedReader.readLine line ends (\n, \r\n; a lone \r also ends a line)
        bool got = pos < size;
        const char *ln = data + pos; size_t len = 0;
        if (got) {
            const char *nl = (const char *)memchr(ln, '\n', size - pos);
            size_t e = nl ? (size_t)(nl - data) : size;
            len = e - pos;
            pos = nl ? e + 1 : size;
            if (len && ln[len - 1] == '\r') len--;
            if (len && all_acgt(ln, len)) {                // the usual line: its own translation
                memcpy(rb.bases.grow(len), ln, len);
                rb.bases.n += len;
                have = true;
                continue;
            }
            if (len && memchr(ln, '\r', len)) {            // rare: lone CR inside -> let the generic reader split it
                line_reader lr{data, size, (size_t)(ln - data)};
                const char *l2; size_t n2;
                lr.next(&l2, &n2);
                len = n2; pos = lr.pos;
            }
        }
        bool comment = got && len > 0 && (ln[0] == '>' || ln[0] == ';');
        if (!got || comment) {
            if (have) {
                if (has_n) rb.drop_read();
                else if (bad >= 0) return mf_set_error("Incorrect nucleotide char: \"%c\" (%s)", bad, path);
                else rb.end_read();
            }
            have = has_n = false; bad = -1;
            if (!got) break;
            continue;
        }
        uint8_t *w = rb.bases.grow(len);
        size_t k = 0;
        for (size_t i = 0; i < len; i++) {
            uint8_t b = BASE_LUT.t[(unsigned char)ln[i]];
            if (b >= 0xFE) { if (b == 0xFE) has_n = true; else if (bad < 0) bad = (unsigned char)ln[i]; continue; }
            w[k++] = b;
        }
        rb.bases.n += k;
        if (len) have = true;
    }
    return MF_OK;
}
// FastqReader (itmo!/io/readers/FastqReader.java:53-115) + quality sniffing (ReadersUtils.java:63-77: Illumina +64 on the
// first 1000 records, any char outside [64,126] -> Sanger +33) + phred-0 drop (FastaReaderFromXQSource.java:66-70)
static int fastq_line(line_reader &lr, const char *path, const char **ln, size_t *len) {   // 1 = line, 0 = EOF
    do { if (!lr.next(ln, len)) return 0; } while (*len == 0);
    if ((*ln)[0] != '@' && (*ln)[0] != '+') return mf_set_error("Unknown structure of fastq file! (%s)", path);
    if (!lr.next(ln, len)) return mf_set_error("Unexpected end of file. File is corrupted/Format mismatch. (%s)", path);
    return 1;
}
// pass 0: quality sniffing only (returns the offset, 64 or 33); pass 1: parse [data, data+size) with `offset`
static int parse_fastq_pass(const char *data, size_t size, const char *path, int pass, int offset, read_batch &rb) {
    line_reader lr{data, size, 0};
    const char *d, *q; size_t dl, ql;
    long rec = 0;
    for (;;) {
        int g = fastq_line(lr, path, &d, &dl);
        if (g < 0) return g;
        if (!g) break;
        g = fastq_line(lr, path, &q, &ql);
        if (g < 0) return g;
        if (!g) return mf_set_error("Unexpected end of file. File is corrupted/Format mismatch. (%s)", path);
        if (dl != ql) return mf_set_error("Bad DnaQ record: length of chars and quality is not the same. (%s)", path);
        // the usual record: upper-case A / C / G / T with every quality above phred 0 -- its own translation
        if (pass == 1 && dl && offset < 126 && all_acgt(d, dl) && all_in_range(q, ql, (uint8_t)(offset + 1), 126)) {
            memcpy(rb.bases.grow(dl), d, dl);
            rb.bases.n += dl;
            rb.end_read();
            continue;
        }
        bool good = true;
        for (size_t i = 0; i < dl; i++) {
            int c = (unsigned char)d[i];
            if (c == 'N' || c == 'n' || c == '.') { good = false; continue; }
            int b = nucleotide_of(c);
            if (b < 0) return mf_set_error("Incorrect nucleotide char: \"%c\" (%s)", c, path);
            int qc = (unsigned char)q[i];
            if (pass == 0) { if (qc < 64 || qc > 126) return 33; }
            else {
                if (qc < offset || qc > 126) return mf_set_error("Invalid quality code char: \"%c\" char code = %d (%s)", qc, qc, path);
                if (qc == offset) good = false;
                rb.bases.push_back((uint8_t)b);
            }
        }
        if (pass == 1) { if (good) rb.end_read(); else rb.drop_read(); }
        if (pass == 0 && ++rec >= 1000) break;
    }
    return pass == 0 ? 64 : MF_OK;
}
// .binq (BinqReader.java:52-88 + FastaReaderFromXQSource.java:62-76): records of a 4-byte big-endian length followed by one byte
// per base, nucleotide in bits 0-1 (A0 G1 C2 T3), phred in bits 2-7; bytes of 255 before a record are padding; a read with a
// phred-0 base is dropped (that is how N is stored)
static int parse_binq(const char *b, size_t n, const char *path, read_batch &rb) {
    const unsigned char *u = (const unsigned char *)b;
    size_t pos = 0;
    for (;;) {
        while (pos < n && u[pos] == 255) pos++;
        if (pos >= n) return MF_OK;
        if (pos + 4 > n) return mf_set_error("Unexpected end of file %s", path);
        const size_t len = ((size_t)u[pos] << 24) | ((size_t)u[pos + 1] << 16) | ((size_t)u[pos + 2] << 8) | (size_t)u[pos + 3];
        pos += 4;
        if (pos + len > n) return mf_set_error("Unexpected end of file %s", path);
        bool good = true;
        uint8_t *dst = rb.bases.grow(len);
        for (size_t i = 0; i < len; i++) { const unsigned v = u[pos + i]; dst[i] = (uint8_t)"AGCT"[v & 3u]; good &= (v >> 2) != 0; }
        pos += len;
        if (good) { rb.bases.n += len; rb.end_read(); }
    }
}
// ---- parallel host parsing: the file is cut at record starts, each host thread parses its piece with the serial
// parser above (so the semantics are the serial ones by construction), pieces are concatenated in order.
// The reference parses serially under a monitor (src/io/ReadersDispatcher.java:34-53) -- its Amdahl limit; here the
// parser has to keep a 40 GB/s PCIe link busy. ----
struct raw_file {            // uninitialised heap buffer (std::vector would zero 15 GB serially first)
    char *p = nullptr; size_t n = 0;
    ~raw_file() { free(p); }
    const char *data() const { return p; }
    char *data() { return p; }
    size_t size() const { return n; }
    bool alloc_bytes(size_t m) { free(p); p = (char *)malloc(m ? m : 1); n = p ? m : 0; return p != nullptr; }
};
static int read_file_parallel(const char *path, raw_file &buf, int threads) {
    int fd = open(path, O_RDONLY);
    if (fd < 0) return mf_set_error("can't open '%s'", path);
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return mf_set_error("can't stat '%s'", path); }
    const size_t n = (size_t)st.st_size;
    buf.p = (char *)malloc(n + 64); buf.n = n;                 // (64 readable bytes behind the file: mf_inflate.h loads eight at a time)
    if (!buf.p) { close(fd); return mf_set_error("out of host memory reading '%s'", path); }
    memset(buf.p + n, 0, 64);
    int T = (int)std::min<size_t>((size_t)std::max(threads, 1), n / (16u << 20) + 1);
    std::vector<std::thread> th;
    std::vector<int> ok(T, 1);
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t]() {
            size_t lo = n * t / T, hi = n * (t + 1) / T;
            while (lo < hi) { ssize_t r = pread(fd, buf.p + lo, hi - lo, (off_t)lo); if (r <= 0) { ok[t] = 0; break; } lo += (size_t)r; }
        });
    for (auto &x : th) x.join();
    close(fd);
    for (int t = 0; t < T; t++) if (!ok[t]) return mf_set_error("short read on '%s'", path);
    return MF_OK;
}
// start of the first record at or after `pos` (or n): FASTA = a line starting with '>' or ';';
// FASTQ = a line starting with '@' whose second-next line starts with '+' (a quality line may start with '@' too,
// but then the line two below it is a sequence line)
static size_t next_record_start(const char *b, size_t n, size_t pos, int fmt) {
    if (pos == 0) return 0;
    while (pos < n) {
        const void *nl = memchr(b + pos, '\n', n - pos);
        if (!nl) return n;
        pos = (size_t)((const char *)nl - b) + 1;
        if (pos >= n) return n;
        if (fmt == 1) { if (b[pos] == '>' || b[pos] == ';') return pos; }
        else if (b[pos] == '@') {
            const void *l1 = memchr(b + pos, '\n', n - pos);
            if (!l1) return n;
            size_t p2 = (size_t)((const char *)l1 - b) + 1;
            const void *l2 = p2 < n ? memchr(b + p2, '\n', n - p2) : nullptr;
            if (!l2) return n;
            size_t p3 = (size_t)((const char *)l2 - b) + 1;
            if (p3 < n && b[p3] == '+') return pos;
        }
    }
    return n;
}
static int parse_buffer_parallel(const raw_file &buf, int fmt, const char *path, int threads, std::vector<read_batch> &out_parts) {
    const char *b = buf.data();
    const size_t n = buf.size();
    int offset = 64;
    if (fmt == 2) { read_batch tmp; offset = parse_fastq_pass(b, n, path, 0, 0, tmp); if (offset < 0) return offset; }
    int T = (int)std::min<size_t>((size_t)std::max(threads, 1), n / (4u << 20) + 1);
    // a FASTQ file with empty lines, or a file without '\n' line ends, is parsed serially (cut points would be unsafe)
    if (T > 1 && !memchr(b, '\n', std::min<size_t>(n, 1u << 20))) T = 1;
    if (T > 1 && fmt == 2 && (memmem(b, n, "\n\n", 2) || memmem(b, n, "\n\r\n", 3))) T = 1;
    std::vector<size_t> cut(T + 1, n);
    cut[0] = 0;
    for (int t = 1; t < T; t++) cut[t] = next_record_start(b, n, n * t / T, fmt);
    for (int t = 1; t <= T; t++) if (cut[t] < cut[t - 1]) cut[t] = cut[t - 1];
    std::vector<read_batch> parts(T);
    std::vector<int> rc(T, MF_OK);
    std::vector<std::string> err(T);
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
        th.emplace_back([&, t]() {
            parts[t].bases.reserve(cut[t + 1] - cut[t]);
            rc[t] = fmt == 1 ? parse_fasta(b + cut[t], cut[t + 1] - cut[t], path, parts[t])
                             : parse_fastq_pass(b + cut[t], cut[t + 1] - cut[t], path, 1, offset, parts[t]);
            if (rc[t] < 0) err[t] = mf_last_error();         // thread-local message -> carry it to the caller
        });
    for (auto &x : th) x.join();
    for (int t = 0; t < T; t++) if (rc[t] < 0) return mf_set_error("%s", err[t].c_str());
    for (auto &p : parts) out_parts.push_back(std::move(p));      // pieces stay separate: they go to the device one by one
    return MF_OK;
}
// .gz inputs (FastaGZReader.java:22-30, FastqGZReader.java:24-32: a GZIPInputStream over the file, which also reads
// CONCATENATED gzip members): the compressed file is read whole, inflated into one host buffer (zlib; one stream cannot be
// inflated in parallel) and then parsed by the same parallel parser as a plain file.
static int inflate_gz_zlib(const raw_file &in, raw_file &out, const char *path) {
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, 15 + 32) != Z_OK) return mf_set_error("zlib: inflateInit2 failed");
    size_t cap = std::max<size_t>(in.n * 5, (size_t)1 << 20), have = 0, fed = 0;
    out.p = (char *)malloc(cap);
    if (!out.p) { inflateEnd(&zs); return mf_set_error("out of host memory inflating '%s'", path); }
    int ret = Z_OK;
    for (;;) {
        if (zs.avail_in == 0 && fed < in.n) {
            const size_t chunk = std::min<size_t>(in.n - fed, (size_t)1 << 30);
            zs.next_in = (Bytef *)(in.p + fed); zs.avail_in = (uInt)chunk; fed += chunk;
        }
        if (have == cap) {
            cap *= 2;
            char *np = (char *)realloc(out.p, cap);
            if (!np) { inflateEnd(&zs); return mf_set_error("out of host memory inflating '%s'", path); }
            out.p = np;
        }
        const size_t room = std::min<size_t>(cap - have, (size_t)1 << 30);
        zs.next_out = (Bytef *)(out.p + have); zs.avail_out = (uInt)room;
        ret = inflate(&zs, Z_NO_FLUSH);
        have += room - zs.avail_out;
        if (ret == Z_STREAM_END) {
            if (zs.avail_in == 0 && fed == in.n) break;          // end of the last member
            if (inflateReset(&zs) != Z_OK) { ret = Z_DATA_ERROR; break; }      // next member of a concatenated file
            continue;
        }
        if (ret == Z_BUF_ERROR && zs.avail_in == 0 && fed == in.n) { ret = Z_DATA_ERROR; break; }   // truncated stream
        if (ret != Z_OK && ret != Z_BUF_ERROR) break;
    }
    inflateEnd(&zs);
    if (ret != Z_STREAM_END) return mf_set_error("Not in GZIP format or corrupt stream: '%s'", path);
    out.n = have;
    return MF_OK;
}
// the whole-buffer decoder first (mf_inflate.h: about three times zlib's pace; it checks every member's CRC-32 and length); whatever it does not
// like -- and every corrupt file, for the wording of the error -- goes through zlib.  MF_FAST_INFLATE=0 in the environment: zlib only
static int inflate_gz(const raw_file &in, raw_file &out, const char *path, int threads = 8) {
    static const bool fast = !(getenv("MF_FAST_INFLATE") && atoi(getenv("MF_FAST_INFLATE")) == 0);
    if (fast) {
        char *q = nullptr; size_t m = 0;
        if (mfz::gunzip(reinterpret_cast<const uint8_t *>(in.p), in.n, threads, &q, &m)) { free(out.p); out.p = q; out.n = m; return MF_OK; }
    }
    return inflate_gz_zlib(in, out, path);
}
// .bz2 inputs (FastaBZ2Reader.java:26-30, FastqBZ2Reader: Hadoop's BZip2Codec, which reads concatenated streams): libbz2 is
// loaded at run time (the build image carries libbz2.so.1.0 but not its header; the four entry points and bz_stream below are
// the library's stable public ABI since 1.0).
struct mf_bz_stream {
    char *next_in; unsigned int avail_in, total_in_lo32, total_in_hi32;
    char *next_out; unsigned int avail_out, total_out_lo32, total_out_hi32;
    void *state; void *(*bzalloc)(void *, int, int); void (*bzfree)(void *, void *); void *opaque;
};
static int inflate_bz2(const raw_file &in, raw_file &out, const char *path) {
    typedef int (*init_fn)(mf_bz_stream *, int, int); typedef int (*step_fn)(mf_bz_stream *);
    static void *lib = nullptr;
    static init_fn bz_init = nullptr; static step_fn bz_step = nullptr, bz_end = nullptr;
    static std::once_flag once;               // (contexts on several threads may meet their first .bz2 file together)
    std::call_once(once, [] {
        for (const char *n : {"libbz2.so.1.0", "libbz2.so.1", "libbz2.so"}) if ((lib = dlopen(n, RTLD_NOW))) break;
        if (lib) {
            bz_init = (init_fn)dlsym(lib, "BZ2_bzDecompressInit"); bz_step = (step_fn)dlsym(lib, "BZ2_bzDecompress"); bz_end = (step_fn)dlsym(lib, "BZ2_bzDecompressEnd");
        }
    });
    if (!lib || !bz_init || !bz_step || !bz_end) return mf_set_error("bzip2 input needs libbz2 at run time (not found): '%s'", path);
    size_t cap = std::max<size_t>(in.n * 6, (size_t)1 << 20), have = 0, fed = 0;
    out.p = (char *)malloc(cap);
    if (!out.p) return mf_set_error("out of host memory inflating '%s'", path);
    mf_bz_stream zs;
    memset(&zs, 0, sizeof zs);
    if (bz_init(&zs, 0, 0) != 0) return mf_set_error("bzip2: init failed");
    int ret = 0;                              // BZ_OK 0, BZ_STREAM_END 4
    for (;;) {
        if (zs.avail_in == 0 && fed < in.n) {
            const size_t chunk = std::min<size_t>(in.n - fed, (size_t)1 << 30);
            zs.next_in = in.p + fed; zs.avail_in = (unsigned int)chunk; fed += chunk;
        }
        if (have == cap) {
            cap *= 2;
            char *np = (char *)realloc(out.p, cap);
            if (!np) { bz_end(&zs); return mf_set_error("out of host memory inflating '%s'", path); }
            out.p = np;
        }
        const size_t room = std::min<size_t>(cap - have, (size_t)1 << 30);
        zs.next_out = out.p + have; zs.avail_out = (unsigned int)room;
        const unsigned int in_before = zs.avail_in;
        ret = bz_step(&zs);
        have += room - zs.avail_out;
        if (ret == 4) {
            if (zs.avail_in == 0 && fed == in.n) break;                    // end of the last stream
            bz_end(&zs);
            char *ni = zs.next_in; unsigned int ai = zs.avail_in;
            memset(&zs, 0, sizeof zs);
            if (bz_init(&zs, 0, 0) != 0) return mf_set_error("bzip2: init failed");
            zs.next_in = ni; zs.avail_in = ai;                             // next stream of a concatenated file
            continue;
        }
        if (ret != 0) break;
        if (zs.avail_in == 0 && fed == in.n && room == zs.avail_out && in_before == 0) { ret = -7; break; }   // truncated
    }
    bz_end(&zs);
    if (ret != 4) return mf_set_error("Not in BZIP2 format or corrupt stream: '%s'", path);
    out.n = have;
    return MF_OK;
}
// ReadersUtils.detectFileFormat (itmo!/io/ReadersUtils.java:27-54): ".gz" / ".bz2" is stripped first, then the format extension.
// .binq: the reference's own binary read format, parsed serially.
// the file's content in memory (compressed files inflated), and its format: 1 FASTA, 2 FASTQ, 3 binq
static int read_file_content(const char *path, int threads, raw_file &buf, int *fmt_out) {
    std::string p(path);
    int fmt = 0;
    bool gz = false, bz = false;
    if (ends_with_nocase(p, ".gz")) { gz = true; p.resize(p.size() - 3); }
    if (ends_with_nocase(p, ".bz2")) { bz = true; p.resize(p.size() - 4); }
    if (ends_with_nocase(p, ".binq")) fmt = 3;
    else if (ends_with_nocase(p, ".fastq") || ends_with_nocase(p, ".fq")) fmt = 2;
    else if (ends_with_nocase(p, ".fasta") || ends_with_nocase(p, ".fa") || ends_with_nocase(p, ".fn") || ends_with_nocase(p, ".fna")) fmt = 1;
    if (!fmt) return mf_set_error("Can't detect file format for file '%s'", path);
    *fmt_out = fmt;
    if (gz || bz) {
        raw_file packed;
        MF_TRY(read_file_parallel(path, packed, threads));
        MF_TRY(gz ? inflate_gz(packed, buf, path, threads) : inflate_bz2(packed, buf, path));
    } else MF_TRY(read_file_parallel(path, buf, threads));
    return MF_OK;
}
static int parse_file_content(const raw_file &buf, int fmt, const char *path, int threads, std::vector<read_batch> &parts) {
    if (fmt == 3) { read_batch rb; const int rc = parse_binq(buf.data(), buf.size(), path, rb); if (rc == MF_OK) parts.push_back(std::move(rb)); return rc; }
    return parse_buffer_parallel(buf, fmt, path, threads, parts);
}
static int parse_reads_file(const char *path, int threads, std::vector<read_batch> &parts) {
    raw_file buf;
    int fmt = 0;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    MF_TRY(read_file_content(path, threads, buf, &fmt));
    const double t1 = now();
    const int rc = parse_file_content(buf, fmt, path, threads, parts);
    if (getenv("MF_IO_TIMING")) fprintf(stderr, "[mf] %s: read %.3f s, parse %.3f s\n", path, t1 - t0, now() - t1);
    return rc;
}

static size_t sr_record_start(const char *b, size_t n, size_t from, bool file_start, int fmt) {
    if (from == 0 && file_start) return 0;
    size_t pos = from ? from - 1 : 0;
    while (pos < n) {
        const void *nl = memchr(b + pos, '\n', n - pos);
        if (!nl) return n;
        pos = (size_t)((const char *)nl - b) + 1;
        if (pos >= n) return n;
        if (fmt == 1) { if (b[pos] == '>' || b[pos] == ';') return pos; }
        else if (b[pos] == '@') {
            const void *l1 = memchr(b + pos, '\n', n - pos);
            if (!l1) return n;
            const size_t p2 = (size_t)((const char *)l1 - b) + 1;
            const void *l2 = p2 < n ? memchr(b + p2, '\n', n - p2) : nullptr;
            if (!l2) return n;
            const size_t p3 = (size_t)((const char *)l2 - b) + 1;
            if (p3 < n && b[p3] == '+') return pos;
        }
    }
    return n;
}

// ---- the streamed count (mf_stream.hip): where the files are cut into pieces and what the sample holds.  Host work on bytes nobody has checked yet.
#define ST_WINDOW ((size_t)1 << 20)       // how far a record border is looked for behind a nominal cut
#define ST_SAMPLE ((size_t)128 << 10)     // bytes per sample chunk
// the first record start in [w, w + n) AFTER position 0 (the window begins anywhere inside a record): sr_record_start's rule -- FASTA: a '>' or ';'
// behind a line feed; FASTQ: a line that starts with '@' whose next line but one starts with '+' (a QUALITY line may start with '@' too, but the line
// after it is then the next record's '@' header and the one after that its bases, which never start with '+').  (size_t)-1: none in the window
static size_t st_record_start(const char *w, size_t n, int fmt) {
    if (n < 2) return (size_t)-1;
    const size_t r = sr_record_start(w, n, 1, false, fmt);
    return r >= n ? (size_t)-1 : r;
}
// the last record start in the window (the END of a sample chunk): searched in its second half
static size_t st_last_record_start(const char *w, size_t n, int fmt) {
    size_t best = (size_t)-1, from = n / 2;
    for (;;) {
        if (from + 2 >= n) break;
        const size_t r = st_record_start(w + from, n - from, fmt);
        if (r == (size_t)-1) break;
        best = from + r;
        from = best + 1;
    }
    return best;
}
struct st_piece { int file; size_t off, len; };
// pieces of about P bytes, every one starting and ending at a record border of its file (the first at byte 0, the last at the file's end; no small last
// piece).  0: done; 1: a border was not found within ST_WINDOW bytes (records longer than that: assembled sequences, not reads) or a read failed
static int st_plan_pieces(const std::vector<int> &fds, const std::vector<size_t> &sizes, int fmt, size_t P, std::vector<st_piece> &pieces, size_t *piece_max) {
    std::vector<char> w(ST_WINDOW);
    *piece_max = 0;
    for (size_t i = 0; i < fds.size(); i++) {
        const size_t n = sizes[i];
        size_t at = 0;
        while (at < n) {
            size_t end = at + P;
            if (end + P / 4 >= n) end = n;
            else {
                // (16 KB first: reads are a few hundred bytes, and 58 windows of 1 MB were 10 ms of preads in front of the first upload)
                size_t r = (size_t)-1;
                for (size_t win : {(size_t)16 << 10, ST_WINDOW}) {
                    const size_t m = std::min(win, n - end);
                    if (pread(fds[i], w.data(), m, (off_t)end) != (ssize_t)m) return 1;
                    r = st_record_start(w.data(), m, fmt);
                    if (r != (size_t)-1) break;
                }
                if (r == (size_t)-1) return 1;
                end += r;
            }
            pieces.push_back(st_piece{(int)i, at, end - at});
            *piece_max = std::max(*piece_max, end - at);
            at = end;
        }
    }
    return 0;
}
// n_chunks chunks of ST_SAMPLE bytes spread evenly over the files' bytes, each trimmed to whole records, one after the other in sample[n_chunks * ST_SAMPLE];
// *s_bytes = what they hold.  0: done; 1: a chunk without two record starts, a read that failed, a file shorter than a chunk
static int st_sample(const std::vector<int> &fds, const std::vector<size_t> &sizes, size_t total, int fmt, size_t n_chunks, int threads, char *sample, size_t *s_bytes) {
    for (size_t sz : sizes) if (sz < ST_SAMPLE) return 1;
    std::vector<size_t> s_len(n_chunks, 0);
    std::atomic<size_t> next{0}; std::atomic<int> bad{0};
    std::vector<std::thread> th;
    for (int t = 0; t < std::max(1, threads); t++)
        th.emplace_back([&]() {
            std::vector<char> w(ST_SAMPLE);
            for (;;) {
                const size_t c = next.fetch_add(1);
                if (c >= n_chunks || bad.load()) break;
                // chunk c stands at byte c / n_chunks of all the files' bytes
                size_t g = (size_t)((double)total * ((double)c / (double)n_chunks));
                size_t f = 0; while (f + 1 < fds.size() && g >= sizes[f]) { g -= sizes[f]; f++; }
                if (g + ST_SAMPLE > sizes[f]) g = sizes[f] - ST_SAMPLE;
                if (pread(fds[f], w.data(), ST_SAMPLE, (off_t)g) != (ssize_t)ST_SAMPLE) { bad = 1; break; }
                const size_t a = g == 0 ? 0 : st_record_start(w.data(), ST_SAMPLE, fmt);
                const size_t b = st_last_record_start(w.data(), ST_SAMPLE, fmt);
                if (a == (size_t)-1 || b == (size_t)-1 || b <= a) { bad = 1; break; }
                memcpy(sample + c * ST_SAMPLE, w.data() + a, b - a);
                s_len[c] = b - a;
            }
        });
    for (auto &x : th) x.join();
    if (bad.load()) return 1;
    size_t at = 0;
    for (size_t c = 0; c < n_chunks; c++) {            // close the gaps
        if (at != c * ST_SAMPLE) memmove(sample + at, sample + c * ST_SAMPLE, s_len[c]);
        at += s_len[c];
    }
    *s_bytes = at;
    return 0;
}

// ConnectedComponent.loadComponents (src/structures/ConnectedComponent.java:95-122): the component headers of a components.bin image
// (count, then per component: size u32, weight i64, size k-mers of 8 bytes; big-endian).  foff[i] = byte offset of component i's
// k-mers, koff = prefix sums of the sizes.  A count or a size the image cannot hold is a wrong file, not a reason to allocate for it.
static int comps_walk_headers(const uint8_t *p, size_t n, std::vector<uint64_t> &sizes, std::vector<int64_t> &weights, std::vector<uint64_t> &foff,
                              std::vector<uint64_t> &koff) {
    const char *bad = "Can't load components: file corrupted or format mismatch! Do you set a wrong file?";
    if (n < 4) return mf_set_error("%s", bad);
    const uint64_t cnt = be_get(p, 4);
    if (4 + 12 * cnt > n) return mf_set_error("%s", bad);
    sizes.clear(); weights.clear(); foff.assign(cnt + 1, 0); koff.assign(cnt + 1, 0);
    size_t pos = 4;
    for (uint64_t i = 0; i < cnt; i++) {
        if (pos + 12 > n) return mf_set_error("%s", bad);
        const uint64_t sz = be_get(p + pos, 4);
        weights.push_back((int64_t)be_get(p + pos + 4, 8));
        pos += 12;
        if (8 * sz > n - pos) return mf_set_error("%s", bad);
        foff[i] = pos; koff[i + 1] = koff[i] + sz;
        pos += 8 * sz;
        sizes.push_back(sz);
    }
    return MF_OK;
}

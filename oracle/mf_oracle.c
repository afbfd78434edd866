/*
 * mf_oracle.c -- CPU ORACLE for the MetaFast hot path.  TEST INFRASTRUCTURE ONLY
 * (see mf_oracle.h for the rules on who may load it and for the parity pin).
 *
 * Every function cites the reference lines it restates:
 *   src/...  = /root/reference/src/...
 *   itmo!/.. = /root/reference/lib/itmo-assembler-src.jar!/ru/ifmo/genetics/..
 *
 * Deliberate, documented differences from the reference (none changes counts,
 * component membership or distances):
 *  - iteration order: the reference iterates shard x slot order, which depends
 *    on fastutil's murmurHash3 and the thread count (BigLong2ShortHashMap.java:
 *    64-88, 216-253); the oracle iterates ascending key order.
 *  - IUPAC ambiguity codes are resolved RANDOMLY by the reference
 *    (DnaTools.java:66-117); the oracle takes the first listed nucleotide.
 */
#define _GNU_SOURCE
#include "mf_oracle.h"
#include <ctype.h>
#include <math.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static __thread char g_err[512];
const char *or_last_error(void) { return g_err; }
static int fail(const char *fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
    return -1;
}
void or_free(void *p) { free(p); }

#define SHORT_MAX 32767

/* ------------------------------------------------------------------ */
/* k-mer arithmetic                                                    */
/* ------------------------------------------------------------------ */

/* itmo!/utils/KmerUtils.java:12-22 (same bit trick at src/algo/KmerOperations.java:62-72) */
uint64_t or_revcomp(uint64_t x, int k) {
    x = ((x & 0x3333333333333333ULL) << 2) | ((x & 0xccccccccccccccccULL) >> 2);
    x = ((x & 0x0f0f0f0f0f0f0f0fULL) << 4) | ((x & 0xf0f0f0f0f0f0f0f0ULL) >> 4);
    x = ((x & 0x00ff00ff00ff00ffULL) << 8) | ((x & 0xff00ff00ff00ff00ULL) >> 8);
    x = ((x & 0x0000ffff0000ffffULL) << 16) | ((x & 0xffff0000ffff0000ULL) >> 16);
    x = (x << 32) | (x >> 32);
    x = ~x;
    return x >> (64 - 2 * k);
}

/* itmo!/dna/kmers/ShortKmer.java: fw/rc pair */
typedef struct { uint64_t fw, rc; } skmer;

static inline skmer sk_make(uint64_t kmer, int k) {          /* ShortKmer.java:19-36 */
    skmer s; s.fw = kmer; s.rc = or_revcomp(kmer, k); return s;
}
static inline uint64_t sk_canon(skmer s) {                    /* ShortKmer.java:54-56 toLong */
    return s.fw < s.rc ? s.fw : s.rc;                         /* k<=31: top bits 0, signed==unsigned */
}
static inline void sk_shift_right(skmer *s, int nuc, int k) { /* ShortKmer.java:68-71 */
    uint64_t mask = (1ULL << (2 * k)) - 1;
    s->fw = ((s->fw << 2) | (uint64_t)nuc) & mask;
    s->rc = (s->rc >> 2) | ((uint64_t)(3 - nuc) << (2 * k - 2));
}
static inline void sk_shift_left(skmer *s, int nuc, int k) {  /* ShortKmer.java:89-92 */
    uint64_t mask = (1ULL << (2 * k)) - 1;
    s->fw = (s->fw >> 2) | ((uint64_t)nuc << (2 * k - 2));
    s->rc = ((s->rc << 2) | (uint64_t)(3 - nuc)) & mask;
}
static inline int sk_nuc_at(skmer s, int i, int k) {          /* ShortKmer.java:58-61 */
    return (int)((s.fw >> (2 * (k - 1 - i))) & 3);
}
uint64_t or_canonical(uint64_t kmer, int k) { return sk_canon(sk_make(kmer, k)); }

/* itmo!/dna/DnaTools.java:31,46-64: A=0 G=1 C=2 T=3 */
static const char NUC_CHARS[4] = {'A', 'G', 'C', 'T'};
static inline int nuc_code(int c) {
    switch (c) {
    case 'A': case 'a': return 0;
    case 'G': case 'g': return 1;
    case 'C': case 'c': return 2;
    case 'T': case 't': return 3;
    default: return -1;
    }
}

/* ------------------------------------------------------------------ */
/* map  uint64 -> int64  (open addressing; stands in for                */
/* BigLong2ShortHashMap / BigLong2LongHashMap; only the key->value      */
/* function is restated, not the slot order)                            */
/* ------------------------------------------------------------------ */
#define EMPTY_KEY UINT64_MAX   /* unreachable: k<=31 keys are < 2^62 */

struct or_table { uint64_t *keys; int64_t *vals; uint64_t cap, size; };

static inline uint64_t mix64(uint64_t h) {
    h ^= h >> 33; h *= 0xff51afd7ed558ccdULL; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ULL; h ^= h >> 33;
    return h;
}
static void table_alloc(or_table *t, uint64_t cap) {
    t->cap = cap; t->size = 0;
    t->keys = (uint64_t *)malloc(cap * sizeof(uint64_t));
    t->vals = (int64_t *)malloc(cap * sizeof(int64_t));
    for (uint64_t i = 0; i < cap; i++) t->keys[i] = EMPTY_KEY;
}
or_table *or_table_new(void) {
    or_table *t = (or_table *)calloc(1, sizeof *t);
    table_alloc(t, 1024);
    return t;
}
void or_table_free(or_table *t) { if (t) { free(t->keys); free(t->vals); free(t); } }
uint64_t or_table_size(const or_table *t) { return t->size; }

static inline uint64_t table_pos(const or_table *t, uint64_t key) {
    uint64_t m = t->cap - 1, p = mix64(key) & m;
    while (t->keys[p] != EMPTY_KEY && t->keys[p] != key) p = (p + 1) & m;
    return p;
}
static void table_grow(or_table *t) {
    or_table old = *t;
    table_alloc(t, old.cap * 2);
    for (uint64_t i = 0; i < old.cap; i++)
        if (old.keys[i] != EMPTY_KEY) {
            uint64_t p = table_pos(t, old.keys[i]);
            t->keys[p] = old.keys[i]; t->vals[p] = old.vals[i]; t->size++;
        }
    free(old.keys); free(old.vals);
}
/* Long2ShortHashMap.get :160-175 -> -1 when absent */
int64_t or_table_get(const or_table *t, uint64_t key) {
    uint64_t p = table_pos(t, key);
    return t->keys[p] == key ? t->vals[p] : -1;
}
static void table_put(or_table *t, uint64_t key, int64_t v) {
    uint64_t p = table_pos(t, key);
    if (t->keys[p] == key) { t->vals[p] = v; return; }
    t->keys[p] = key; t->vals[p] = v; t->size++;
    if (t->size * 4 >= t->cap * 3) table_grow(t);
}
/* Long2ShortHashMap.addAndBound :119-157 + NumUtils.addAndBound(short,short) :21-26 */
static void table_add_bound(or_table *t, uint64_t key, int64_t inc, int64_t bound) {
    uint64_t p = table_pos(t, key);
    if (t->keys[p] == key) {
        int64_t v = t->vals[p];
        t->vals[p] = (v > bound - inc) ? bound : v + inc;
        return;
    }
    t->keys[p] = key; t->vals[p] = inc > bound ? bound : inc; t->size++;
    if (t->size * 4 >= t->cap * 3) table_grow(t);
}
int or_table_add(or_table *t, uint64_t key, int inc) { table_add_bound(t, key, inc, SHORT_MAX); return 0; }

static or_table *table_clone(const or_table *t) {
    or_table *c = (or_table *)calloc(1, sizeof *c);
    c->cap = t->cap; c->size = t->size;
    c->keys = (uint64_t *)malloc(t->cap * sizeof(uint64_t));
    c->vals = (int64_t *)malloc(t->cap * sizeof(int64_t));
    memcpy(c->keys, t->keys, t->cap * sizeof(uint64_t));
    memcpy(c->vals, t->vals, t->cap * sizeof(int64_t));
    return c;
}

typedef struct { uint64_t key; int64_t val; } kv_t;
static int kv_cmp(const void *a, const void *b) {
    uint64_t x = ((const kv_t *)a)->key, y = ((const kv_t *)b)->key;
    return x < y ? -1 : x > y;
}
/* all entries with val > threshold, ascending key; caller frees */
static kv_t *table_sorted(const or_table *t, int64_t threshold, uint64_t *n_out) {
    kv_t *a = (kv_t *)malloc((t->size + 1) * sizeof(kv_t));
    uint64_t n = 0;
    for (uint64_t i = 0; i < t->cap; i++)
        if (t->keys[i] != EMPTY_KEY && t->vals[i] > threshold) { a[n].key = t->keys[i]; a[n].val = t->vals[i]; n++; }
    qsort(a, n, sizeof(kv_t), kv_cmp);
    *n_out = n;
    return a;
}
uint64_t or_table_export(const or_table *t, int threshold, uint64_t *keys, int32_t *vals, uint64_t cap) {
    uint64_t n; kv_t *a = table_sorted(t, threshold, &n);
    for (uint64_t i = 0; i < n && i < cap; i++) { keys[i] = a[i].key; vals[i] = (int32_t)a[i].val; }
    free(a);
    return n;
}

/* ------------------------------------------------------------------ */
/* A1 readers                                                          */
/* ------------------------------------------------------------------ */
typedef struct { uint8_t *b; uint64_t nb, cb; uint64_t *off; uint64_t nr, cr; } readbuf;

static void rb_init(readbuf *r) {
    memset(r, 0, sizeof *r);
    r->cb = 1 << 16; r->b = (uint8_t *)malloc(r->cb);
    r->cr = 1 << 10; r->off = (uint64_t *)malloc(r->cr * sizeof(uint64_t));
    r->off[0] = 0;
}
static void rb_push_base(readbuf *r, uint8_t c) {
    if (r->nb == r->cb) { r->cb *= 2; r->b = (uint8_t *)realloc(r->b, r->cb); }
    r->b[r->nb++] = c;
}
static void rb_end_read(readbuf *r) {
    if (r->nr + 2 > r->cr) { r->cr *= 2; r->off = (uint64_t *)realloc(r->off, r->cr * sizeof(uint64_t)); }
    r->off[++r->nr] = r->nb;
}
static void rb_drop_read(readbuf *r) { r->nb = r->off[r->nr]; }

/* DnaTools.fromChar :46-64 via singleLetterCodeToNucleotide :66-113 (first listed choice) */
static int base_from_char(int c) {
    switch (toupper(c)) {
    case 'A': case 'C': case 'G': case 'T': return toupper(c);
    case 'R': return 'G'; case 'Y': return 'T'; case 'M': return 'A'; case 'K': return 'G';
    case 'S': return 'G'; case 'W': return 'A'; case 'H': return 'A'; case 'B': return 'G';
    case 'V': return 'A'; case 'D': return 'A';
    default: return -1;       /* IllegalArgumentException("Incorrect nucleotide char") */
    }
}

static char *slurp(const char *path, size_t *n) {
    FILE *f = fopen(path, "rb");
    if (!f) { fail("can't open %s", path); return NULL; }
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    char *buf = (char *)malloc((size_t)sz + 1);
    if (fread(buf, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); free(buf); fail("short read %s", path); return NULL; }
    fclose(f); buf[sz] = 0; *n = (size_t)sz;
    return buf;
}
/* BufferedReader.readLine: lines end at \n, \r or \r\n. Returns 0 at EOF. */
static int next_line(const char *buf, size_t n, size_t *pos, const char **line, size_t *len) {
    if (*pos >= n) return 0;
    size_t s = *pos, e = s;
    while (e < n && buf[e] != '\n' && buf[e] != '\r') e++;
    *line = buf + s; *len = e - s;
    if (e < n) { if (buf[e] == '\r' && e + 1 < n && buf[e + 1] == '\n') e += 2; else e += 1; }
    *pos = e;
    return 1;
}

static int ends_with_ci(const char *s, const char *suf) {
    size_t n = strlen(s), m = strlen(suf);
    if (m > n) return 0;
    for (size_t i = 0; i < m; i++) if (tolower((unsigned char)s[n - m + i]) != suf[i]) return 0;
    return 1;
}
/* ReadersUtils.detectFileFormat :27-54 (plain fasta / fastq only; gz/bz2/binq are out of scope) */
static int detect_format(const char *path) {
    if (ends_with_ci(path, ".fastq") || ends_with_ci(path, ".fq")) return 2;
    if (ends_with_ci(path, ".fasta") || ends_with_ci(path, ".fa") || ends_with_ci(path, ".fn") ||
        ends_with_ci(path, ".fna")) return 1;
    return 0;
}

/* FastaReader.MyIterator.readNext :53-76 / readNextDataLine :78-104 */
static int read_fasta(const char *buf, size_t n, readbuf *r) {
    size_t pos = 0; const char *ln; size_t len;
    int have = 0, hasN = 0, bad = 0;
    for (;;) {
        int got = next_line(buf, n, &pos, &ln, &len);
        int comment = got && len > 0 && (ln[0] == '>' || ln[0] == ';');
        if (!got || comment) {
            if (have) {                         /* record complete */
                if (hasN) rb_drop_read(r);      /* s.contains("N")||s.contains("n") -> skipped */
                else if (bad) return fail("Incorrect nucleotide char in FASTA");
                else rb_end_read(r);
                have = hasN = bad = 0;
            }
            if (!got) break;
            continue;
        }
        for (size_t i = 0; i < len; i++) {
            int c = (unsigned char)ln[i];
            if (c == 'N' || c == 'n') hasN = 1;
            int b = base_from_char(c);
            if (b < 0) { if (c != 'N' && c != 'n') bad = 1; b = 'A'; }
            rb_push_base(r, (uint8_t)b);
            have = 1;
        }
    }
    return 0;
}

/* FastqReader.MyIterator :53-115.  pass==0: quality sniffing with Illumina (+64) on the first
 * 1000 records (ReadersUtils.determineQualityFormat :63-77) -> returns 64 or 33.
 * pass==1: parse with `offset`; drop reads with any phred==0 (FastaReaderFromXQSource :66-70). */
static int fastq_data_line(const char *buf, size_t n, size_t *pos, const char **ln, size_t *len) {
    /* readNextDataLine :85-110: skip empty lines, expect @/+ header, then the data line */
    for (;;) {
        if (!next_line(buf, n, pos, ln, len)) return 0;
        if (*len != 0) break;
    }
    if (!((*ln)[0] == '@' || (*ln)[0] == '+')) return fail("Unknown structure of fastq file");
    if (!next_line(buf, n, pos, ln, len)) return fail("Unexpected end of fastq file");
    return 1;
}
static int read_fastq(const char *buf, size_t n, readbuf *r, int pass, int offset) {
    size_t pos = 0; const char *d, *q; size_t dl, ql;
    long rec = 0;
    for (;;) {
        int g = fastq_data_line(buf, n, &pos, &d, &dl);
        if (g < 0) return -1;
        if (g == 0) break;
        g = fastq_data_line(buf, n, &pos, &q, &ql);
        if (g <= 0) return g < 0 ? -1 : fail("Unexpected end of fastq file");
        if (dl != ql) return fail("Bad DnaQ record: length of chars and quality is not the same");
        int good = 1;
        for (size_t i = 0; i < dl; i++) {
            int c = (unsigned char)d[i];
            if (c == 'N' || c == 'n' || c == '.') { good = 0; if (pass) rb_push_base(r, 'A'); continue; }
            int b = base_from_char(c);
            if (b < 0) return fail("Incorrect nucleotide char in FASTQ");
            int qc = (unsigned char)q[i];
            if (pass == 0) { if (qc < 64 || qc > 126) return 33; }        /* Illumina.getPhred throws -> Sanger */
            else {
                if (qc < offset || qc > 126) return fail("Invalid quality code char");
                if (qc - offset == 0) good = 0;
                rb_push_base(r, (uint8_t)b);
            }
        }
        if (pass) { if (good) rb_end_read(r); else rb_drop_read(r); }
        if (pass == 0 && ++rec >= 1000) break;
    }
    return pass == 0 ? 64 : 0;
}

int or_read_file(const char *path, uint8_t **bases, uint64_t **offsets, uint64_t *n_reads, uint64_t *n_bases) {
    int fmt = detect_format(path);
    if (!fmt) return fail("Can't detect file format for file '%s'", path);
    size_t n; char *buf = slurp(path, &n);
    if (!buf) return -1;
    readbuf r; rb_init(&r);
    int rc;
    if (fmt == 1) rc = read_fasta(buf, n, &r);
    else {
        int off = read_fastq(buf, n, &r, 0, 0);
        rc = off < 0 ? off : read_fastq(buf, n, &r, 1, off);
    }
    free(buf);
    if (rc < 0) { free(r.b); free(r.off); return rc; }
    *bases = r.b; *offsets = r.off; *n_reads = r.nr; *n_bases = r.nb;
    return 0;
}

/* ------------------------------------------------------------------ */
/* A2-A4 counting                                                      */
/* ------------------------------------------------------------------ */
/* ReadsLoadWorker.process (src/io/IOUtils.java:756-768) + ShortKmer.kmersOf (:104-150) */
int or_count_buffer(or_table *t, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads, int k, int min_len) {
    if (k < 1 || k > 31) return fail("k must be in [1,31]");   /* KmersCounterMain.java:66-73 */
    for (uint64_t r = 0; r < n_reads; r++) {
        const uint8_t *s = bases + offsets[r];
        uint64_t len = offsets[r + 1] - offsets[r];
        if ((int64_t)len < (int64_t)min_len) continue;         /* dna.length() >= minDnaLen */
        if (len < (uint64_t)k) continue;                       /* kmersOf: i=k-1 >= length -> empty */
        uint64_t fw = 0;
        for (int i = 0; i < k; i++) {
            int c = nuc_code(s[i]);
            if (c < 0) return fail("bad base in buffer");
            fw = (fw << 2) | (uint64_t)c;
        }
        skmer km = sk_make(fw, k);
        table_add_bound(t, sk_canon(km), 1, SHORT_MAX);
        for (uint64_t i = (uint64_t)k; i < len; i++) {
            int c = nuc_code(s[i]);
            if (c < 0) return fail("bad base in buffer");
            sk_shift_right(&km, c, k);
            table_add_bound(t, sk_canon(km), 1, SHORT_MAX);
        }
    }
    return 0;
}
/* IOUtils.loadReads :772-803 / run :838-865: files sequentially into one map */
int or_count_files(or_table *t, const char *const *files, int nfiles, int k, int min_len) {
    for (int f = 0; f < nfiles; f++) {
        uint8_t *b; uint64_t *off; uint64_t nr, nb;
        if (or_read_file(files[f], &b, &off, &nr, &nb) < 0) return -1;
        int rc = or_count_buffer(t, b, off, nr, k, min_len);
        free(b); free(off);
        if (rc < 0) return rc;
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* A5/A6 files                                                         */
/* ------------------------------------------------------------------ */
static void put_be(uint8_t *p, uint64_t v, int nbytes) {
    for (int i = 0; i < nbytes; i++) p[i] = (uint8_t)(v >> (8 * (nbytes - 1 - i)));
}
static uint64_t get_be(const uint8_t *p, int nbytes) {
    uint64_t v = 0; for (int i = 0; i < nbytes; i++) v = (v << 8) | p[i]; return v;
}
/* IOUtils.printKmers :45-71; QuickQuantitativeStatistics.toString/printToFile :38-72 */
int or_write_kmers(const or_table *t, int threshold, const char *kmers_bin, const char *stat_txt, uint64_t *n_good) {
    uint64_t n; kv_t *a = table_sorted(t, INT64_MIN, &n);
    FILE *f = fopen(kmers_bin, "wb");
    if (!f) { free(a); return fail("can't write %s", kmers_bin); }
    uint64_t *hist = (uint64_t *)calloc(SHORT_MAX + 1, sizeof(uint64_t));
    uint64_t good = 0;
    for (uint64_t i = 0; i < n; i++) {
        hist[a[i].val]++;                                      /* stats.add(value) for ALL entries */
        if (a[i].val > threshold) {
            uint8_t rec[10]; put_be(rec, a[i].key, 8); put_be(rec + 8, (uint64_t)a[i].val, 2);
            fwrite(rec, 1, 10, f); good++;
        }
    }
    fclose(f); free(a);
    if (stat_txt) {
        f = fopen(stat_txt, "w");
        if (!f) { free(hist); return fail("can't write %s", stat_txt); }
        fprintf(f, "# k-mer frequency\tnumber of such k-mers\n");
        for (int v = 0; v <= SHORT_MAX; v++) if (hist[v]) fprintf(f, "%d\t%llu\n", v, (unsigned long long)hist[v]);
        fprintf(f, "\n");                                      /* out.println(toString()) */
        fclose(f);
    }
    free(hist);
    if (n_good) *n_good = good;
    return 0;
}
/* IOUtils.loadKmers :369-401; Kmers2HMWorker.processKmer :249-257; KmersLoadWorker.process :16-34 */
int or_load_kmers(or_table *t, const char *const *files, int nfiles, int freq_threshold) {
    for (int fi = 0; fi < nfiles; fi++) {
        size_t n; uint8_t *buf = (uint8_t *)slurp(files[fi], &n);
        if (!buf) return -1;
        if (n % 10) { free(buf); return fail("BAD division by work range (%s)", files[fi]); }
        for (size_t i = 0; i < n; i += 10) {
            uint64_t key = get_be(buf + i, 8);
            int freq = (int16_t)get_be(buf + i + 8, 2);
            if (freq > freq_threshold) table_add_bound(t, key, freq, SHORT_MAX);
        }
        free(buf);
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* A7/A8 unitigs                                                       */
/* ------------------------------------------------------------------ */
typedef struct { char *s; uint64_t len; int avg, mn, mx; } seq_t;
struct or_seqs { seq_t *a; uint64_t n, cap; };

/* HashMapOperations.getLeftNucleotide :13-29 */
static int get_left(const or_table *t, skmer km, int k, int thr) {
    int right_nuc = sk_nuc_at(km, k - 1, k), ans = -1;
    for (int nuc = 0; nuc <= 3; nuc++) {
        sk_shift_left(&km, nuc, k);
        uint64_t rep = sk_canon(km);
        sk_shift_right(&km, right_nuc, k);
        if (or_table_get(t, rep) > thr) { if (ans > -1) return -2; ans = nuc; }
    }
    return ans;
}
/* HashMapOperations.getRightNucleotide :31-47 */
static int get_right(const or_table *t, skmer km, int k, int thr) {
    int left_nuc = sk_nuc_at(km, 0, k), ans = -1;
    for (int nuc = 0; nuc <= 3; nuc++) {
        sk_shift_right(&km, nuc, k);
        uint64_t rep = sk_canon(km);
        sk_shift_left(&km, left_nuc, k);
        if (or_table_get(t, rep) > thr) { if (ans > -1) return -2; ans = nuc; }
    }
    return ans;
}
static inline int64_t get_with_zero(const or_table *t, uint64_t key) {   /* Long2ShortHashMap.getWithZero :178-183 */
    int64_t v = or_table_get(t, key); return v == -1 ? 0 : v;
}

/* AddSequencesShiftingRightTask.processSequence :74-123 */
/* census of the last or_build_unitigs call (tests: the 0 / 1 / 2-emission rule must be exercised):
 * [0] walks started, [1] walks of at least min_len nucleotides, [2] walks emitted */
static uint64_t g_census[3];
void or_unitig_census(uint64_t out[3]) { out[0] = g_census[0]; out[1] = g_census[1]; out[2] = g_census[2]; }
static void process_sequence(const or_table *t, skmer start, int k, int thr, int min_len, or_table *used, or_seqs *out) {
    g_census[0]++;
    int64_t value = get_with_zero(t, sk_canon(start));
    uint64_t cap = 256, len = 0;
    char *sb = (char *)malloc(cap);
    for (int i = 0; i < k; i++) sb[len++] = NUC_CHARS[sk_nuc_at(start, i, k)];   /* startKmer.toString() */
    int64_t w = value; int mn = (int)value, mx = (int)value;
    skmer km = start;
    for (;;) {
        int rn = get_right(t, km, k, thr);
        if (rn < 0) break;
        sk_shift_right(&km, rn, k);
        int ln = get_left(t, km, k, thr);
        if (ln < 0) break;                                /* km stays advanced */
        if (len == cap) { cap *= 2; sb = (char *)realloc(sb, cap); }
        sb[len++] = NUC_CHARS[rn];
        value = get_with_zero(t, sk_canon(km));
        w += value;
        if (value < mn) mn = (int)value;
        if (value > mx) mx = (int)value;
    }
    if ((int64_t)len >= (int64_t)min_len) {
        g_census[1]++;
        uint64_t st = sk_canon(start), en = sk_canon(km);
        if (st > en) { free(sb); return; }
        if (st == en) {                                   /* print only one of them */
            if (or_table_get(used, st) != -1) { free(sb); return; }
            table_put(used, st, 1);
        }
        if (out->n == out->cap) { out->cap = out->cap ? out->cap * 2 : 64; out->a = (seq_t *)realloc(out->a, out->cap * sizeof(seq_t)); }
        g_census[2]++;
        seq_t *q = &out->a[out->n++];
        q->s = sb; q->len = len; q->avg = (int)(w / (int64_t)(len - (uint64_t)k + 1)); q->mn = mn; q->mx = mx;
        return;
    }
    free(sb);
}

/* SequencesFinders.thresholdStrategy :13-31 + AddSequencesShiftingRightTask.run :40-71 */
or_seqs *or_build_unitigs(const or_table *t, int k, int thr, int min_len) {
    g_census[0] = g_census[1] = g_census[2] = 0;
    or_seqs *out = (or_seqs *)calloc(1, sizeof *out);
    or_table *used = or_table_new();
    uint64_t n; kv_t *a = table_sorted(t, thr, &n);       /* value <= freqThreshold -> continue */
    for (uint64_t i = 0; i < n; i++) {
        skmer kf = sk_make(a[i].key, k);
        skmer both[2]; both[0] = kf; both[1] = sk_make(kf.rc, k);   /* {kmerF, kmerF.rc()} */
        for (int o = 0; o < 2; o++) {
            skmer km = both[o];
            int is_left = 0;
            int nuc = get_left(t, km, k, thr);
            if (nuc < 0) is_left = 1;
            else {
                int right_nuc = sk_nuc_at(km, k - 1, k);
                sk_shift_left(&km, nuc, k);
                if (get_right(t, km, k, thr) < 0) is_left = 1;
                sk_shift_right(&km, right_nuc, k);
            }
            if (is_left) process_sequence(t, km, k, thr, min_len, used, out);
        }
    }
    free(a); or_table_free(used);
    return out;
}
void or_seqs_free(or_seqs *s) { if (!s) return; for (uint64_t i = 0; i < s->n; i++) free(s->a[i].s); free(s->a); free(s); }
uint64_t or_seqs_count(const or_seqs *s) { return s->n; }
uint64_t or_seqs_total_len(const or_seqs *s) { uint64_t t = 0; for (uint64_t i = 0; i < s->n; i++) t += s->a[i].len; return t; }
int or_seqs_get(const or_seqs *s, uint64_t i, const char **seq, uint64_t *len, int *avg_w, int *min_w, int *max_w) {
    if (i >= s->n) return fail("index");
    *seq = s->a[i].s; *len = s->a[i].len; *avg_w = s->a[i].avg; *min_w = s->a[i].mn; *max_w = s->a[i].mx;
    return 0;
}
/* Sequence.printSequences :26-37; FastaDedicatedWriter.writeData :33-49; TextUtils.printWithLineLimit :35-45 */
int or_seqs_write_fasta(const or_seqs *s, const char *path) {
    FILE *f = fopen(path, "w");
    if (!f) return fail("can't write %s", path);
    for (uint64_t i = 0; i < s->n; i++) {
        const seq_t *q = &s->a[i];
        fprintf(f, ">%llu length=%llu av_weight=%d min_weight=%d max_weight=%d\n",
                (unsigned long long)(i + 1), (unsigned long long)q->len, q->avg, q->mn, q->mx);
        uint64_t j = 0;
        while ((j + 1) * 70 < q->len) { fwrite(q->s + j * 70, 1, 70, f); fputc('\n', f); j++; }
        fwrite(q->s + j * 70, 1, q->len - j * 70, f); fputc('\n', f);
    }
    fclose(f);
    return 0;
}
/* SeqBuilderMain.runImpl :84-98 + dumpStat :170-176 */
int or_write_distribution(const or_table *t, const char *path) {
    uint64_t stat[1024]; memset(stat, 0, sizeof stat);
    for (uint64_t i = 0; i < t->cap; i++)
        if (t->keys[i] != EMPTY_KEY) { int64_t v = t->vals[i]; if (v >= 1024) v = 1023; if (v >= 0) stat[v]++; }
    FILE *f = fopen(path, "w");
    if (!f) return fail("can't write %s", path);
    for (int i = 1; i < 1024; i++) fprintf(f, "%d %llu\n", i, (unsigned long long)stat[i]);
    fclose(f);
    return 0;
}
/* ComponentCutterMain.runImpl :81-82 -> IOUtils.loadReads(seq files, k, minLen) */
int or_count_seqs(or_table *t, const or_seqs *s, int k, int min_len) {
    for (uint64_t i = 0; i < s->n; i++) {
        uint64_t off[2] = {0, s->a[i].len};
        int rc = or_count_buffer(t, (const uint8_t *)s->a[i].s, off, 1, k, min_len);
        if (rc < 0) return rc;
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* A10/A11 components                                                  */
/* ------------------------------------------------------------------ */
typedef struct {
    uint64_t *kmers; uint64_t nk, ck;   /* ConnectedComponent.kmers (null once big) */
    uint64_t size; int64_t weight; int thr;
    or_table *next_hm;                  /* ConnectedComponent.nextHM */
    uint64_t min_key;
} comp_t;
struct or_comps { comp_t *a; uint64_t n, cap; };

static void comp_add(comp_t *c, uint64_t kmer, int64_t w) {   /* ConnectedComponent.add :70-74 */
    if (c->nk == c->ck) { c->ck = c->ck ? c->ck * 2 : 64; c->kmers = (uint64_t *)realloc(c->kmers, c->ck * sizeof(uint64_t)); }
    c->kmers[c->nk++] = kmer; c->size++; c->weight += w;
}
/* KmerOperations.possibleNeighbours :9-26 */
static void possible_neighbours(uint64_t kmer, int k, uint64_t ans[8]) {
    skmer go_right = sk_make(kmer, k), go_left = sk_make(kmer, k);
    sk_shift_right(&go_right, 0, k); ans[0] = sk_canon(go_right);
    sk_shift_left(&go_left, 0, k);   ans[1] = sk_canon(go_left);
    for (int nuc = 1; nuc <= 3; nuc++) {
        /* updateAt(k-1, nuc) / updateAt(0, nuc): ShortKmer.java:94-102 */
        skmer r = sk_make((go_right.fw & ~3ULL) | (uint64_t)nuc, k);
        ans[nuc * 2] = sk_canon(r);
        uint64_t top = 3ULL << (2 * k - 2);
        skmer l = sk_make((go_left.fw & ~top) | ((uint64_t)nuc << (2 * k - 2)), k);
        ans[nuc * 2 + 1] = sk_canon(l);
    }
}
/* ComponentsBuilder.bfs :220-270; hm is mutated (visited = negated value) */
static comp_t bfs(or_table *hm, uint64_t start, int k, int b2, int thr, uint64_t **queue, uint64_t *qcap) {
    comp_t comp; memset(&comp, 0, sizeof comp);
    comp.thr = thr;
    uint64_t qh = 0, qt = 0;
#define ENQ(x) do { if (qt == *qcap) { *qcap *= 2; *queue = (uint64_t *)realloc(*queue, *qcap * sizeof(uint64_t)); } (*queue)[qt++] = (x); } while (0)
    ENQ(start);
    int64_t value = or_table_get(hm, start);
    table_put(hm, start, -value);
    comp_add(&comp, start, value);
    int already_big = 0;
    while (qh < qt) {
        uint64_t kmer = (*queue)[qh++];
        uint64_t nb[8]; possible_neighbours(kmer, k, nb);
        for (int j = 0; j < 8; j++) {
            uint64_t nbr = nb[j];
            value = or_table_get(hm, nbr);
            if (value > 0) {
                ENQ(nbr);
                table_put(hm, nbr, -value);
                if (!already_big) {
                    comp_add(&comp, nbr, value);
                    if (comp.size > (uint64_t)b2) {
                        already_big = 1;
                        comp.next_hm = or_table_new();
                        for (uint64_t q = 0; q < comp.nk; q++) {
                            int64_t v = -or_table_get(hm, comp.kmers[q]);
                            if (v >= thr + 1) table_put(comp.next_hm, comp.kmers[q], v);
                        }
                        free(comp.kmers); comp.kmers = NULL; comp.nk = comp.ck = 0;
                    }
                } else {
                    if (value >= thr + 1) table_put(comp.next_hm, nbr, value);
                    comp.size++;
                }
            }
        }
    }
#undef ENQ
    return comp;
}
static int u64_cmp(const void *a, const void *b) { uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b; return x < y ? -1 : x > y; }
static void comps_push(or_comps *c, comp_t x) {
    if (c->n == c->cap) { c->cap = c->cap ? c->cap * 2 : 16; c->a = (comp_t *)realloc(c->a, c->cap * sizeof(comp_t)); }
    c->a[c->n++] = x;
}
/* ConnectedComponent.compareTo :125-136 (+ min k-mer as a deterministic tie-break; the reference
 * leaves ties in discovery order, which is hash/race dependent) */
static int comp_cmp(const void *pa, const void *pb) {
    const comp_t *a = (const comp_t *)pa, *b = (const comp_t *)pb;
    if (a->thr != b->thr) return a->thr < b->thr ? -1 : 1;
    if (a->weight != b->weight) return a->weight > b->weight ? -1 : 1;
    if (a->size != b->size) return a->size > b->size ? -1 : 1;
    return a->min_key < b->min_key ? -1 : a->min_key > b->min_key;
}
/* ComponentsBuilder.findAllComponents :198-213 + run :58-153 + Task.run :162-179 */
or_comps *or_cut_components(const or_table *t, int k, int b1, int b2) {
    or_comps *ans = (or_comps *)calloc(1, sizeof *ans);
    uint64_t qcap = 1 << 16; uint64_t *queue = (uint64_t *)malloc(qcap * sizeof(uint64_t));
    /* work list of (hm, thr): first the whole map at thr=1, then each big component's nextHM at thr+1 */
    typedef struct { or_table *hm; int thr; } work_t;
    uint64_t wn = 0, wc = 16; work_t *work = (work_t *)malloc(wc * sizeof(work_t));
    work[wn].hm = table_clone(t); work[wn].thr = 1; wn++;
    while (wn) {
        work_t w = work[--wn];
        uint64_t n; kv_t *a = table_sorted(w.hm, 0, &n);          /* value > 0 i.e. not processed */
        for (uint64_t i = 0; i < n; i++) {
            if (or_table_get(w.hm, a[i].key) <= 0) continue;
            comp_t c = bfs(w.hm, a[i].key, k, b2, w.thr, &queue, &qcap);
            if (c.size < (uint64_t)b1) { free(c.kmers); }
            else if (c.size <= (uint64_t)b2) {
                qsort(c.kmers, c.nk, sizeof(uint64_t), u64_cmp);
                c.min_key = c.kmers[0];
                comps_push(ans, c);
            } else {
                if (wn == wc) { wc *= 2; work = (work_t *)realloc(work, wc * sizeof(work_t)); }
                work[wn].hm = c.next_hm; work[wn].thr = c.thr + 1; wn++;
            }
        }
        free(a); or_table_free(w.hm);
    }
    free(work); free(queue);
    qsort(ans->a, ans->n, sizeof(comp_t), comp_cmp);               /* Collections.sort(ans) :144 */
    return ans;
}
void or_comps_free(or_comps *c) { if (!c) return; for (uint64_t i = 0; i < c->n; i++) free(c->a[i].kmers); free(c->a); free(c); }
uint64_t or_comps_count(const or_comps *c) { return c->n; }
int or_comps_get(const or_comps *c, uint64_t i, uint64_t *size, int64_t *weight, int *thr, const uint64_t **kmers) {
    if (i >= c->n) return fail("index");
    *size = c->a[i].size; *weight = c->a[i].weight; *thr = c->a[i].thr; *kmers = c->a[i].kmers;
    return 0;
}
/* ConnectedComponent.saveComponents :80-93; ComponentsBuilder.run :146-152 */
int or_comps_write(const or_comps *c, const char *components_bin, const char *stat_txt) {
    FILE *f = fopen(components_bin, "wb");
    if (!f) return fail("can't write %s", components_bin);
    uint8_t b[8];
    put_be(b, c->n, 4); fwrite(b, 1, 4, f);
    for (uint64_t i = 0; i < c->n; i++) {
        put_be(b, c->a[i].size, 4); fwrite(b, 1, 4, f);
        put_be(b, (uint64_t)c->a[i].weight, 8); fwrite(b, 1, 8, f);
        for (uint64_t j = 0; j < c->a[i].nk; j++) { put_be(b, c->a[i].kmers[j], 8); fwrite(b, 1, 8, f); }
    }
    fclose(f);
    if (stat_txt) {
        f = fopen(stat_txt, "w");
        if (!f) return fail("can't write %s", stat_txt);
        fprintf(f, "# component.no\tcomponent.size\tcomponent.weight\tusedFreqThreshold\n");
        for (uint64_t i = 0; i < c->n; i++)
            fprintf(f, "%llu\t%llu\t%lld\t%d\n", (unsigned long long)(i + 1), (unsigned long long)c->a[i].size,
                    (long long)c->a[i].weight, c->a[i].thr);
        fclose(f);
    }
    return 0;
}
/* ConnectedComponent.loadComponents :95-122 */
or_comps *or_comps_load(const char *components_bin) {
    size_t n; uint8_t *buf = (uint8_t *)slurp(components_bin, &n);
    if (!buf) return NULL;
    if (n < 4) { free(buf); fail("Can't load components: file corrupted or format mismatch"); return NULL; }
    or_comps *c = (or_comps *)calloc(1, sizeof *c);
    uint64_t cnt = get_be(buf, 4); size_t p = 4;
    for (uint64_t i = 0; i < cnt; i++) {
        if (p + 12 > n) { free(buf); or_comps_free(c); fail("Can't load components: file corrupted"); return NULL; }
        comp_t x; memset(&x, 0, sizeof x);
        uint64_t sz = get_be(buf + p, 4); x.weight = (int64_t)get_be(buf + p + 4, 8); p += 12;
        if (p + 8 * sz > n) { free(buf); or_comps_free(c); fail("Can't load components: file corrupted"); return NULL; }
        for (uint64_t j = 0; j < sz; j++) { comp_add(&x, get_be(buf + p, 8), 0); p += 8; }
        x.thr = 0;
        comps_push(c, x);
    }
    free(buf);
    return c;
}

/* ------------------------------------------------------------------ */
/* A12 features                                                        */
/* ------------------------------------------------------------------ */
/* FeaturesCalculatorMain.runImpl :97-103 (hm.put(kmer,0)), :137-162 (resetValues +
 * calculatePresenceForKmers -> KmersPresenceWorker :577-588), buildAndPrintVector :169-236 */
/* buildAndPrintVector :186-206, the per-component loop: with --selected (:55-57, :113-116) only the k-mers with
 * selected.getWithZero(kmer) > 0 take part -- in the sum, in kmersFound and in kmersCount (0.0 / 0.0 = NaN when none is) */
static void build_vector(const or_comps *c, const or_table *hm, int threshold, const or_table *selected, int64_t *vec, double *breadth) {
    for (uint64_t i = 0; i < c->n; i++) {
        int64_t kmers = 0, cnt = 0, found = 0;
        for (uint64_t j = 0; j < c->a[i].nk; j++) {
            if (selected == NULL || get_with_zero(selected, c->a[i].kmers[j]) > 0) {
                int64_t value = get_with_zero(hm, c->a[i].kmers[j]);
                if (value > threshold) { kmers += value; found++; }
                cnt++;
            }
        }
        vec[i] = kmers;
        if (breadth) breadth[i] = (double)found / (double)cnt;
    }
}
int or_features(const or_comps *c, const or_table *sample, int threshold, int64_t *vec, double *breadth) {
    return or_features_selected(c, sample, threshold, NULL, vec, breadth);
}
int or_features_selected(const or_comps *c, const or_table *sample, int threshold, const or_table *selected, int64_t *vec, double *breadth) {
    or_table *hm = or_table_new();
    for (uint64_t i = 0; i < c->n; i++)
        for (uint64_t j = 0; j < c->a[i].nk; j++) table_put(hm, c->a[i].kmers[j], 0);
    for (uint64_t i = 0; i < sample->cap; i++)                        /* every record of the .kmers.bin */
        if (sample->keys[i] != EMPTY_KEY && or_table_get(hm, sample->keys[i]) != -1)
            table_add_bound(hm, sample->keys[i], sample->vals[i], INT64_MAX);
    build_vector(c, hm, threshold, selected, vec, breadth);
    or_table_free(hm);
    return 0;
}

/* FeaturesCalculatorMain.runImpl reads branch (:117-131): hm (k-mer -> long, BigLong2LongHashMap: no saturation at 32767)
 * holds the component k-mers; IOUtils.calculatePresenceForReads / ReadsPresenceWorker.process (src/io/IOUtils.java:806-834)
 * adds 1 for every k-mer of every read that is a component k-mer; then buildAndPrintVector as above */
int or_features_reads(const or_comps *c, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads, int k, int threshold,
                      int64_t *vec, double *breadth) {
    return or_features_reads_selected(c, bases, offsets, n_reads, k, threshold, NULL, vec, breadth);
}
int or_features_reads_selected(const or_comps *c, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads, int k, int threshold,
                               const or_table *selected, int64_t *vec, double *breadth) {
    if (k < 1 || k > 31) return fail("k must be in [1,31]");
    or_table *hm = or_table_new();
    for (uint64_t i = 0; i < c->n; i++)
        for (uint64_t j = 0; j < c->a[i].nk; j++) table_put(hm, c->a[i].kmers[j], 0);
    for (uint64_t r = 0; r < n_reads; r++) {
        const uint8_t *s = bases + offsets[r];
        uint64_t len = offsets[r + 1] - offsets[r];
        if (len < (uint64_t)k) continue;
        uint64_t fw = 0;
        for (int i = 0; i < k; i++) {
            int cc = nuc_code(s[i]);
            if (cc < 0) { or_table_free(hm); return fail("bad base in buffer"); }
            fw = (fw << 2) | (uint64_t)cc;
        }
        skmer km = sk_make(fw, k);
        if (or_table_get(hm, sk_canon(km)) != -1) table_add_bound(hm, sk_canon(km), 1, INT64_MAX);
        for (uint64_t i = (uint64_t)k; i < len; i++) {
            int cc = nuc_code(s[i]);
            if (cc < 0) { or_table_free(hm); return fail("bad base in buffer"); }
            sk_shift_right(&km, cc, k);
            if (or_table_get(hm, sk_canon(km)) != -1) table_add_bound(hm, sk_canon(km), 1, INT64_MAX);
        }
    }
    build_vector(c, hm, threshold, selected, vec, breadth);
    or_table_free(hm);
    return 0;
}

/* ------------------------------------------------------------------ */
/* A13 Bray-Curtis                                                     */
/* ------------------------------------------------------------------ */
/* DistanceMatrixCalculatorMain.brayCurtisDistance :140-152 (vectors parsed as doubles :125-138) */
int or_bray_curtis(const int64_t *vecs, int n_samples, int n_comp, double *out) {
    for (int i = 0; i < n_samples; i++) {
        out[i * n_samples + i] = 0.0;
        for (int j = i + 1; j < n_samples; j++) {
            double sumdiff = 0, sum = 0;
            for (int p = 0; p < n_comp; p++) {
                double a = (double)vecs[(size_t)i * n_comp + p], b = (double)vecs[(size_t)j * n_comp + p];
                sumdiff += fabs(a - b);
                sum += fabs(a) + fabs(b);
            }
            out[i * n_samples + j] = out[j * n_samples + i] = sumdiff / sum;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* Multi-threaded CPU baseline of the counting loop (bench.py only)    */
/* ------------------------------------------------------------------ */
/* Structure of IOUtils.loadReads :772-803 + run :838-865 + ReadsWorker.run (src/io/ReadsWorker.java:28-41):
 * workers pull 32768-read ranges (READS_WORK_RANGE_SIZE :29) and call addAndBound per k-mer on
 * BigLong2ShortHashMap(log2(P)+4, 12): shard = murmurHash3((int)key)&mask (BigLong2ShortHashMap.java:68-71),
 * slot = murmurHash3(key)&capMask, linear probe, FREE=0 side-car, one lock per shard
 * (Long2ShortHashMap.java:119-157), grow x2 at 0.75 (:191-214).  Reads are already parsed (in memory),
 * so the reference's serial parser (ReadsDispatcher.java:34-53) is NOT charged to this baseline. */
typedef struct {
    pthread_mutex_t lock;
    int64_t *keys; int16_t *vals; uint32_t cap, size, max_fill;
    int has_free; int16_t free_val;
} bshard;
typedef struct {
    bshard *sh; uint32_t mask;
    const uint8_t *bases; const uint64_t *off; uint64_t n_reads; int k;
    uint64_t next; pthread_mutex_t disp;
    uint64_t n_occ;
} bctx;
static inline uint32_t murmur32(uint32_t h) { h ^= h >> 16; h *= 0x85ebca6bU; h ^= h >> 13; h *= 0xc2b2ae35U; h ^= h >> 16; return h; }
static void bshard_grow(bshard *s) {
    uint32_t nc = s->cap * 2;
    int64_t *nk = (int64_t *)calloc(nc, sizeof(int64_t)); int16_t *nv = (int16_t *)calloc(nc, sizeof(int16_t));
    for (uint32_t i = 0; i < s->cap; i++) if (s->keys[i]) {
        uint32_t p = (uint32_t)mix64((uint64_t)s->keys[i]) & (nc - 1);
        while (nk[p]) p = (p + 1) & (nc - 1);
        nk[p] = s->keys[i]; nv[p] = s->vals[i];
    }
    free(s->keys); free(s->vals);
    s->keys = nk; s->vals = nv; s->cap = nc; s->max_fill = (uint32_t)ceil(nc * 0.75);
}
static inline void bshard_add(bshard *s, int64_t key) {
    pthread_mutex_lock(&s->lock);
    if (key == 0) {
        if (!s->has_free) { s->has_free = 1; s->size++; s->free_val = 0; }
        if (s->free_val < SHORT_MAX) s->free_val++;
    } else {
        uint32_t p = (uint32_t)mix64((uint64_t)key) & (s->cap - 1);
        while (s->keys[p] && s->keys[p] != key) p = (p + 1) & (s->cap - 1);
        if (s->vals[p] < SHORT_MAX) s->vals[p]++;
        if (!s->keys[p]) { s->keys[p] = key; if (++s->size >= s->max_fill) bshard_grow(s); }
    }
    pthread_mutex_unlock(&s->lock);
}
static void *bworker(void *arg) {
    bctx *c = (bctx *)arg;
    int k = c->k; uint64_t occ = 0;
    for (;;) {
        pthread_mutex_lock(&c->disp);
        uint64_t lo = c->next; c->next += 32768;
        pthread_mutex_unlock(&c->disp);
        if (lo >= c->n_reads) break;
        uint64_t hi = lo + 32768 < c->n_reads ? lo + 32768 : c->n_reads;
        for (uint64_t r = lo; r < hi; r++) {
            const uint8_t *s = c->bases + c->off[r]; uint64_t len = c->off[r + 1] - c->off[r];
            if (len < (uint64_t)k) continue;
            uint64_t fw = 0;
            for (int i = 0; i < k; i++) fw = (fw << 2) | (uint64_t)nuc_code(s[i]);
            skmer km = sk_make(fw, k);
            for (uint64_t i = (uint64_t)k;; i++) {
                int64_t key = (int64_t)sk_canon(km);
                bshard_add(&c->sh[murmur32((uint32_t)key) & c->mask], key);
                occ++;
                if (i >= len) break;
                sk_shift_right(&km, nuc_code(s[i]), k);
            }
        }
    }
    __sync_fetch_and_add(&c->n_occ, occ);
    return NULL;
}
uint64_t or_cpu_baseline_count(const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads, int k, int threads, uint64_t *n_occ) {
    bctx c; memset(&c, 0, sizeof c);
    int lg = 0; while ((1 << (lg + 1)) <= threads) lg++;
    uint32_t ns = 1u << (lg + 4);
    c.sh = (bshard *)calloc(ns, sizeof(bshard)); c.mask = ns - 1;
    for (uint32_t i = 0; i < ns; i++) {
        pthread_mutex_init(&c.sh[i].lock, NULL);
        c.sh[i].cap = 4096; c.sh[i].max_fill = 3072;
        c.sh[i].keys = (int64_t *)calloc(4096, sizeof(int64_t)); c.sh[i].vals = (int16_t *)calloc(4096, sizeof(int16_t));
    }
    c.bases = bases; c.off = offsets; c.n_reads = n_reads; c.k = k;
    pthread_mutex_init(&c.disp, NULL);
    pthread_t *th = (pthread_t *)malloc((size_t)threads * sizeof(pthread_t));
    for (int i = 0; i < threads; i++) pthread_create(&th[i], NULL, bworker, &c);
    for (int i = 0; i < threads; i++) pthread_join(th[i], NULL);
    uint64_t distinct = 0;
    for (uint32_t i = 0; i < ns; i++) { distinct += c.sh[i].size; free(c.sh[i].keys); free(c.sh[i].vals); }
    free(c.sh); free(th);
    if (n_occ) *n_occ = c.n_occ;
    return distinct;
}

/* ------------------------------------------------------------------ */
/* The same baseline WITH the reference's reader and dump (bench.py)   */
/* ------------------------------------------------------------------ */
/* kmer-counter on one FASTA file as the reference runs it:
 *  - IOUtils.run (src/io/IOUtils.java:838-865) starts P ReadsWorkers; a worker's loop (src/io/ReadsWorker.java:28-41) asks
 *    the ReadsDispatcher for the next <= 32768 reads; getWorkRange is `synchronized` (src/io/ReadsDispatcher.java:34-53) and
 *    PARSES them from the file inside the monitor (FastaReader, itmo!/io/readers/FastaReader.java:53-104: multi-line
 *    records, reads with N dropped) -- the reference's serial section -- then the worker counts its batch in parallel;
 *  - IOUtils.printKmers (src/io/IOUtils.java:45-71): ONE thread walks every slot of every shard, writes the entries with
 *    count > b as 10-byte big-endian records and tallies the histogram of all counts.
 * Times are returned separately: sec[0] = loadReads (reader + counting), sec[1] = printKmers. */
typedef struct {
    bctx b;                        /* shards, k, n_occ (bases / off / next unused) */
    FILE *f; pthread_mutex_t rd;   /* the dispatcher's monitor */
    char *line; size_t line_cap;   /* getline buffer (used under the monitor) */
    char *pending; size_t pend_len, pend_cap; int have_pending_header, eof;
    uint64_t n_reads;
} fctx;
/* next record under the monitor -> appended to (buf, off); returns 0 at end of file */
static int fa_next(fctx *c, uint8_t **buf, size_t *len, size_t *cap) {
    /* header line already consumed (have_pending_header) or to be found */
    ssize_t n;
    if (!c->have_pending_header) {
        for (;;) {
            n = getline(&c->line, &c->line_cap, c->f);
            if (n < 0) return 0;
            if (c->line[0] == '>' || c->line[0] == ';') break;
        }
    }
    c->have_pending_header = 0;
    size_t start = *len; int bad = 0;
    for (;;) {
        n = getline(&c->line, &c->line_cap, c->f);
        if (n < 0) { c->eof = 1; break; }
        if (c->line[0] == '>' || c->line[0] == ';') { c->have_pending_header = 1; break; }
        while (n > 0 && (c->line[n - 1] == '\n' || c->line[n - 1] == '\r')) n--;
        if (*len + (size_t)n > *cap) { *cap = (*len + (size_t)n) * 2 + 4096; *buf = (uint8_t *)realloc(*buf, *cap); }
        for (ssize_t i = 0; i < n; i++) { char ch = c->line[i]; if (ch == 'N' || ch == 'n') bad = 1; (*buf)[(*len)++] = (uint8_t)ch; }
    }
    if (bad) *len = start;          /* FastaReaderFromXQSource: a read with N is skipped */
    return 1;
}
static void *fworker(void *arg) {
    fctx *c = (fctx *)arg;
    const int k = c->b.k; uint64_t occ = 0, nr = 0;
    uint8_t *buf = NULL; size_t len = 0, cap = 0;
    uint64_t *off = (uint64_t *)malloc(32769 * sizeof(uint64_t));
    for (;;) {
        int m = 0; len = 0; off[0] = 0;
        pthread_mutex_lock(&c->rd);                 /* ReadsDispatcher.getWorkRange: parse up to 32768 reads */
        while (m < 32768 && !(c->eof && !c->have_pending_header)) {
            size_t before = len;
            if (!fa_next(c, &buf, &len, &cap)) break;
            if (len > before) { m++; off[m] = len; }
        }
        pthread_mutex_unlock(&c->rd);
        if (m == 0) break;
        nr += (uint64_t)m;
        for (int r = 0; r < m; r++) {
            const uint8_t *s = buf + off[r]; uint64_t L = off[r + 1] - off[r];
            if (L < (uint64_t)k) continue;
            uint64_t fw = 0;
            for (int i = 0; i < k; i++) fw = (fw << 2) | (uint64_t)nuc_code(s[i]);
            skmer km = sk_make(fw, k);
            for (uint64_t i = (uint64_t)k;; i++) {
                int64_t key = (int64_t)sk_canon(km);
                bshard_add(&c->b.sh[murmur32((uint32_t)key) & c->b.mask], key);
                occ++;
                if (i >= L) break;
                sk_shift_right(&km, nuc_code(s[i]), k);
            }
        }
    }
    free(buf); free(off);
    __sync_fetch_and_add(&c->b.n_occ, occ);
    __sync_fetch_and_add(&c->n_reads, nr);
    return NULL;
}
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
int or_cpu_baseline_file(const char *fasta, int k, int threads, int bcut, const char *kmers_bin, uint64_t *res, double *sec) {
    /* res[0] = occurrences, [1] = distinct, [2] = written (count > bcut), [3] = reads */
    fctx c; memset(&c, 0, sizeof c);
    c.f = fopen(fasta, "rb");
    if (!c.f) return -1;
    setvbuf(c.f, NULL, _IOFBF, 1 << 22);
    int lg = 0; while ((1 << (lg + 1)) <= threads) lg++;
    uint32_t ns = 1u << (lg + 4);
    c.b.sh = (bshard *)calloc(ns, sizeof(bshard)); c.b.mask = ns - 1; c.b.k = k;
    for (uint32_t i = 0; i < ns; i++) {
        pthread_mutex_init(&c.b.sh[i].lock, NULL);
        c.b.sh[i].cap = 4096; c.b.sh[i].max_fill = 3072;
        c.b.sh[i].keys = (int64_t *)calloc(4096, sizeof(int64_t)); c.b.sh[i].vals = (int16_t *)calloc(4096, sizeof(int16_t));
    }
    pthread_mutex_init(&c.rd, NULL);
    const double t0 = now_s();
    pthread_t *th = (pthread_t *)malloc((size_t)threads * sizeof(pthread_t));
    for (int i = 0; i < threads; i++) pthread_create(&th[i], NULL, fworker, &c);
    for (int i = 0; i < threads; i++) pthread_join(th[i], NULL);
    const double t1 = now_s();
    /* printKmers: one thread, every slot */
    FILE *o = kmers_bin ? fopen(kmers_bin, "wb") : NULL;
    if (o) setvbuf(o, NULL, _IOFBF, 1 << 24);       /* (16 MiB buffer: IOUtils.java:52) */
    uint64_t distinct = 0, written = 0, *hist = (uint64_t *)calloc(SHORT_MAX + 1, sizeof(uint64_t));
    for (uint32_t i = 0; i < ns; i++) {
        bshard *s = &c.b.sh[i];
        for (uint32_t j = 0; j < s->cap; j++) if (s->keys[j]) {
            const int16_t v = s->vals[j]; hist[v]++; distinct++;
            if (v > bcut) {
                uint8_t rec[10]; uint64_t kk = (uint64_t)s->keys[j];
                for (int q = 0; q < 8; q++) rec[q] = (uint8_t)(kk >> (8 * (7 - q)));
                rec[8] = (uint8_t)((uint16_t)v >> 8); rec[9] = (uint8_t)v;
                if (o) fwrite(rec, 1, 10, o);
                written++;
            }
        }
        if (s->has_free) { hist[s->free_val]++; distinct++; if (s->free_val > bcut) { uint8_t rec[10] = {0}; rec[8] = (uint8_t)((uint16_t)s->free_val >> 8); rec[9] = (uint8_t)s->free_val; if (o) fwrite(rec, 1, 10, o); written++; } }
        free(s->keys); free(s->vals);
    }
    if (o) fclose(o);
    const double t2 = now_s();
    res[0] = c.b.n_occ; res[1] = distinct; res[2] = written; res[3] = c.n_reads;
    sec[0] = t1 - t0; sec[1] = t2 - t1;
    free(hist); free(c.b.sh); free(th); free(c.line); fclose(c.f);
    return 0;
}

/* ---- NO-REFERENCE EXTENSION: canonical k-mer counts for 32 <= k <= 63 (the reference rejects k > 31,
 * src/tools/KmersCounterMain.java:66-73).  The k <= 31 definitions (ShortKmer: first base most significant, A0 G1 C2 T3,
 * canonical = min(fw, rc); Long2ShortHashMap.addAndBound: saturation at 32767; loadReads: reads shorter than max(k, minLen)
 * give nothing) on 2k-bit numbers.  Checker of metafast_amd/csrc/mf_wide.hip only. ---- */
typedef unsigned __int128 or_u128;
static int or_cmp_u128(const void *a, const void *b) {
    const or_u128 x = *(const or_u128 *)a, y = *(const or_u128 *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}
/* *n_out distinct k-mers into hi[], lo[], cnt[] (ascending; arrays of capacity cap); returns 0, or -1 if cap is too small */
int or_count_wide(const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads, int k, int min_len, uint64_t *hi, uint64_t *lo,
                  int32_t *cnt, uint64_t cap, uint64_t *n_out, uint64_t *n_occ_out) {
    uint64_t n_occ = 0;
    for (uint64_t r = 0; r < n_reads; r++) {
        const uint64_t len = offsets[r + 1] - offsets[r];
        if (len >= (uint64_t)k && (int64_t)len >= (int64_t)min_len) n_occ += len - (uint64_t)k + 1;
    }
    or_u128 *all = (or_u128 *)malloc((n_occ ? n_occ : 1) * sizeof(or_u128));
    if (!all) return -2;
    const or_u128 mask = k == 64 ? ~(or_u128)0 : (((or_u128)1 << (2 * k)) - 1);
    uint64_t m = 0;
    for (uint64_t r = 0; r < n_reads; r++) {
        const uint64_t s = offsets[r], len = offsets[r + 1] - s;
        if (len < (uint64_t)k || (int64_t)len < (int64_t)min_len) continue;
        or_u128 fw = 0, rc = 0;
        for (uint64_t i = 0; i < len; i++) {
            const int c = nuc_code(bases[s + i]);
            fw = ((fw << 2) | (or_u128)c) & mask;
            rc = (rc >> 2) | ((or_u128)(3 - c) << (2 * k - 2));
            if (i + 1 >= (uint64_t)k) all[m++] = fw < rc ? fw : rc;
        }
    }
    qsort(all, m, sizeof(or_u128), or_cmp_u128);
    uint64_t nd = 0;
    for (uint64_t i = 0; i < m;) {
        uint64_t j = i;
        while (j < m && all[j] == all[i]) j++;
        if (nd >= cap) { free(all); return -1; }
        hi[nd] = (uint64_t)(all[i] >> 64); lo[nd] = (uint64_t)all[i];
        cnt[nd] = (int32_t)((j - i) > 32767 ? 32767 : (j - i));
        nd++;
        i = j;
    }
    free(all);
    *n_out = nd; *n_occ_out = m;
    return 0;
}

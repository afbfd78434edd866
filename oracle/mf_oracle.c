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

#define OKEY uint64_t
#define OKEY_BITS 64
#define OKEY_MAXK 31
#include "mf_oracle_core.inc"

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

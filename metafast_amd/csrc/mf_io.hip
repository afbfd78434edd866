// mf_io.hip -- host side of the C-ABI: readers, writers and the file-level entry points.
// Plain C++ (no kernels); formats follow SURVEY.md Appendix A, each function cites the reference it replaces.
#include "mf_common.h"
#include <zlib.h>
#include <dlfcn.h>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <chrono>
#include <climits>
#include <fcntl.h>
#include <string>
#include <sys/stat.h>
#include <thread>
#include <atomic>
#include <mutex>
#include <memory>
#include <unistd.h>
#include <vector>

int mf_count_core(mf_ctx *ctx, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n_reads, uint64_t n_bases, int k,
                  int min_len, mf_table **out, int thr, uint64_t *n_all);
int mf_table_count_hist(const mf_table *t, std::vector<uint64_t> &hist);
int mf_seqs_to_host(const mf_seqs *s, std::vector<uint8_t> &bases, std::vector<uint64_t> &off, std::vector<int32_t> &avg,
                    std::vector<int32_t> &mn, std::vector<int32_t> &mx);
int mf_comps_from_host(mf_ctx *ctx, int k, const std::vector<uint64_t> &sizes, const std::vector<int64_t> &weights,
                       const std::vector<int32_t> &thr, const std::vector<uint64_t> &offsets, const std::vector<uint64_t> &kmers,
                       mf_comps **out);

// ---------------------------------------------------------------------------------------------
// small file helpers
// ---------------------------------------------------------------------------------------------
static int read_whole_file(const char *path, std::vector<char> &buf) {
    FILE *f = fopen(path, "rb");
    if (!f) return mf_set_error("can't open '%s'", path);
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    buf.resize((size_t)sz);
    if (sz && fread(buf.data(), 1, (size_t)sz, f) != (size_t)sz) { fclose(f); return mf_set_error("short read on '%s'", path); }
    fclose(f);
    return MF_OK;
}
// ---- MF_IO_TIMING=1: one line per file-level call with the seconds of its phases (diagnostics of the drop-in path: tools/cli_rate.py)
struct io_timer {
    bool on; double t0 = 0, tl = 0; std::string msg;
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    explicit io_timer(const char *what) {
        static const bool env = getenv("MF_IO_TIMING") != nullptr;
        on = env;
        if (on) { t0 = tl = now(); msg = std::string("[mf] ") + what + ":"; }
    }
    void lap(const char *name) { if (!on) return; const double t = now(); char b[96]; snprintf(b, sizeof b, " %s %.3f s,", name, t - tl); msg += b; tl = t; }
    ~io_timer() { if (on) fprintf(stderr, "%s total %.3f s\n", msg.c_str(), now() - t0); }
};

// ---- the context's file cache (option file_cache = GB it may hold; the metafast.sh driver switches it on).  matrix-builder runs its
// steps in ONE process and every step reads what the step before has just written: seq-builder-many the .kmers.bin files of
// kmer-counter-many, features-calculator the same files again and components.bin once per sample (DistanceMatrixBuilderMain.java:
// 108-170 wires the steps through their output files).  The files are written as the reference writes them; the table / the
// components they were written FROM stay in HBM under (real path, size, mtime), and a load of that file hands out another handle on
// the same object (mf_table::refs) instead of reading, decoding, sorting and partitioning 10 bytes per k-mer again -- 1.3 of the
// 3.8 s of `metafast.sh -i` on 2 x 20 M reads.  Least recently used entries go when the budget is exceeded, all of them when the
// arena cannot serve an allocation (mf_alloc).
static bool file_identity(const char *path, std::string &real, uint64_t &size, int64_t &mtime_ns) {
    char buf[PATH_MAX];
    if (!realpath(path, buf)) return false;
    struct stat st;
    if (stat(buf, &st) != 0) return false;
    real = buf; size = (uint64_t)st.st_size; mtime_ns = (int64_t)st.st_mtim.tv_sec * 1000000000ll + st.st_mtim.tv_nsec;
    return true;
}
static void file_cache_drop(mf_ctx *ctx, size_t i) {
    mf_file_entry *e = ctx->file_cache[i];
    ctx->file_cache.erase(ctx->file_cache.begin() + (long)i);
    ctx->file_cache_bytes -= e->bytes;
    if (e->t) mf_table_destroy(e->t);
    if (e->c) mf_comps_destroy(e->c);
    delete e;
}
void mf_file_cache_clear(mf_ctx *ctx) { while (!ctx->file_cache.empty()) file_cache_drop(ctx, ctx->file_cache.size() - 1); }
static void file_cache_put(mf_ctx *ctx, const char *path, mf_table *t, int thr, mf_comps *c) {
    if (ctx->opt_file_cache_gb <= 0) return;
    std::string real; uint64_t size; int64_t mt;
    if (!file_identity(path, real, size, mt)) return;
    for (size_t i = 0; i < ctx->file_cache.size(); i++) if (ctx->file_cache[i]->path == real) { file_cache_drop(ctx, i); break; }
    const size_t bytes = t ? t->keys_bytes + t->counts_bytes + t->part_off_bytes : c->kmers_bytes + c->comp_bytes;
    const size_t budget = (size_t)ctx->opt_file_cache_gb << 30;
    if (bytes > budget) return;
    while (!ctx->file_cache.empty() && ctx->file_cache_bytes + bytes > budget) {
        size_t old = 0;
        for (size_t i = 1; i < ctx->file_cache.size(); i++) if (ctx->file_cache[i]->stamp < ctx->file_cache[old]->stamp) old = i;
        file_cache_drop(ctx, old);
    }
    mf_file_entry *e = new mf_file_entry();
    e->path = real; e->size = size; e->mtime_ns = mt; e->t = t; e->thr = thr; e->c = c; e->bytes = bytes; e->stamp = ++ctx->file_cache_clock;
    if (t) t->refs++;
    if (c) c->refs++;
    ctx->file_cache.push_back(e);
    ctx->file_cache_bytes += bytes;
}
static mf_file_entry *file_cache_get(mf_ctx *ctx, const char *path) {
    if (ctx->file_cache.empty()) return nullptr;
    std::string real; uint64_t size; int64_t mt;
    if (!file_identity(path, real, size, mt)) return nullptr;
    for (size_t i = 0; i < ctx->file_cache.size(); i++) {
        mf_file_entry *e = ctx->file_cache[i];
        if (e->path != real) continue;
        if (e->size != size || e->mtime_ns != mt) { file_cache_drop(ctx, i); return nullptr; }      // (somebody else wrote the file since)
        e->stamp = ++ctx->file_cache_clock;
        return e;
    }
    return nullptr;
}

#include "mf_parse.h"
#include <sys/mman.h>   // the host-side parsers (plain C++: also built under ASan / UBSan, tests/test_host_sanitized_cpu.py)

// ---- streaming reader for plain FASTA / FASTQ files: constant host memory, PCIe busy while the host parses --------------
// The file is cut into pieces of SR_PIECE bytes (ends moved to the next record start, at most SR_SLACK further).  Up to 64
// workers take pieces in turn: pread the piece into a private buffer, parse it with the SERIAL parser above into one of
// the worker's two PINNED chunks, hipMemcpyAsync it to the piece's slot of a device buffer and go on with the other chunk
// (a chunk is re-used when the event behind its copy has fired).  The compacting D2D copies into the final (bases,
// offsets) layout are the caller's.  Returns 1 if the file needs the whole-file reader instead (a record longer than the
// slack, a FASTQ file with empty lines, no '\n' line ends), < 0 on errors, 0 when done.
// (piece / slack sizes: options stream_piece_bytes, stream_slack_bytes -- the tests use small ones to get many cuts)
struct sr_piece { uint64_t dev_off = 0, n_bases = 0; std::vector<uint64_t> offsets; };
struct sr_file { mf_buf<uint8_t> dev; std::vector<sr_piece> pieces; };
// first record start at a buffer position >= from (the byte before it is a '\n', or it is the file's first byte), or n
static int stream_file_to_device(mf_ctx *ctx, const char *path, int fmt, sr_file &out) {
    int fd = open(path, O_RDONLY);
    if (fd < 0) return mf_set_error("can't open '%s'", path);
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return mf_set_error("can't stat '%s'", path); }
    const size_t fsize = (size_t)st.st_size;
    const size_t SR_PIECE = (size_t)std::max<int64_t>(ctx->opt_sr_piece, 4096), SR_SLACK = (size_t)std::max<int64_t>(ctx->opt_sr_slack, 1024);
    const size_t np = (fsize + SR_PIECE - 1) / SR_PIECE;
    const int W = (int)std::min<size_t>(std::min<size_t>((size_t)std::max(ctx->host_threads, 1), 64), np);
    const size_t chunk = (size_t)SR_PIECE + SR_SLACK + 64;
    // quality offset: the first 1000 records (ReadersUtils.java:63-77)
    int qoff = 64;
    if (fmt == 2) {
        std::vector<char> head(std::min<size_t>(fsize, 4u << 20));
        if (pread(fd, head.data(), head.size(), 0) != (ssize_t)head.size()) { close(fd); return mf_set_error("short read on '%s'", path); }
        read_batch tmp;
        qoff = parse_fastq_pass(head.data(), head.size(), path, 0, 0, tmp);
        if (qoff < 0) { close(fd); return head.size() < fsize ? 1 : qoff; }      // (cut mid-record: let the whole-file reader decide)
    }
    {
        io_timer tm("stream reader set-up");
        if (ctx->pin_pool_bytes < (size_t)2 * W * chunk) {
            const size_t want = (size_t)2 * std::min<size_t>((size_t)std::max(ctx->host_threads, 1), 64) * chunk;
            if (mf_ensure_pin_pool(ctx, want) != MF_OK) { close(fd); return 1; }
            tm.lap("staging pool");
        }
        if (out.dev.alloc(ctx, np * chunk) != MF_OK) { close(fd); return 1; }
        tm.lap("device buffer");
    }
    out.pieces.assign(np, sr_piece());
    std::atomic<size_t> next{0};
    std::atomic<int> state{0};                       // 0 ok, 1 = use the whole-file reader, < 0 = error
    std::string err; std::mutex err_mu;
    std::vector<std::thread> th;
    for (int w = 0; w < W; w++)
        th.emplace_back([&, w]() {
            (void)hipSetDevice(ctx->device);
            std::vector<char> raw(chunk + 1);
            uint8_t *pin[2] = {(uint8_t *)ctx->pin_pool + (size_t)(2 * w) * chunk, (uint8_t *)ctx->pin_pool + (size_t)(2 * w + 1) * chunk};
            hipEvent_t ev[2]; bool busy[2] = {false, false};
            (void)hipEventCreateWithFlags(&ev[0], hipEventDisableTiming); (void)hipEventCreateWithFlags(&ev[1], hipEventDisableTiming);
            int cur = 0;
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= np || state.load() != 0) break;
                // bytes [i*PIECE - 1, (i+1)*PIECE + SLACK) of the file; the record starts at or after i*PIECE and
                // (i+1)*PIECE bound the piece
                const size_t lo = i ? i * SR_PIECE - 1 : 0, hi = std::min<size_t>(fsize, (i + 1) * (size_t)SR_PIECE + SR_SLACK);
                size_t got = 0;
                while (got < hi - lo) { const ssize_t r = pread(fd, raw.data() + got, hi - lo - got, (off_t)(lo + got)); if (r <= 0) break; got += (size_t)r; }
                if (got != hi - lo) { std::lock_guard<std::mutex> g(err_mu); err = std::string("short read on '") + path + "'"; state = -1; break; }
                const size_t n = hi - lo;
                const size_t s0 = sr_record_start(raw.data(), n, i ? 1 : 0, i == 0, fmt);
                size_t e0 = n;
                if (hi < fsize || (i + 1) * (size_t)SR_PIECE < fsize) {
                    const size_t from = (i + 1) * (size_t)SR_PIECE - lo;
                    e0 = from < n ? sr_record_start(raw.data(), n, from, false, fmt) : n;
                    if (e0 >= n && hi < fsize) { state = 1; break; }              // no record start within the slack
                }
                if (i && s0 > (size_t)SR_SLACK + 1) { state = 1; break; }
                const char *pb = raw.data() + std::min(s0, e0); const size_t pn = e0 > s0 ? e0 - s0 : 0;
                if (fmt == 2 && pn && (memmem(pb, pn, "\n\n", 2) || memmem(pb, pn, "\n\r\n", 3))) { state = 1; break; }
                if (i == 0 && pn && !memchr(pb, '\n', std::min<size_t>(pn, 1u << 20))) { state = 1; break; }
                if (busy[cur]) { (void)hipEventSynchronize(ev[cur]); busy[cur] = false; }
                read_batch rb;
                rb.bases.use_external(pin[cur], chunk);
                const int rc = fmt == 1 ? parse_fasta(pb, pn, path, rb) : parse_fastq_pass(pb, pn, path, 1, qoff, rb);
                if (rc < 0) { std::lock_guard<std::mutex> g(err_mu); if (state.load() >= 0) { err = mf_last_error(); state = rc; } break; }
                sr_piece &P = out.pieces[i];
                P.dev_off = i * chunk; P.n_bases = rb.bases.size(); P.offsets = std::move(rb.offsets);
                if (P.n_bases) {
                    if (hipMemcpyAsync(out.dev.p + P.dev_off, pin[cur], P.n_bases, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
                        hipEventRecord(ev[cur], ctx->stream) != hipSuccess) {
                        std::lock_guard<std::mutex> g(err_mu); err = "H2D copy failed"; state = -1; break;
                    }
                    busy[cur] = true;
                    cur ^= 1;
                }
            }
            for (int j = 0; j < 2; j++) { if (busy[j]) (void)hipEventSynchronize(ev[j]); (void)hipEventDestroy(ev[j]); }
        });
    for (auto &x : th) x.join();
    close(fd);
    const int stt = state.load();
    if (stt < 0) return mf_set_error("%s", err.c_str());
    if (stt == 1) { out.dev.reset(); out.pieces.clear(); return 1; }
    return 0;
}

__global__ void k_offsets_rebase(const uint64_t *__restrict__ in, uint64_t n, uint64_t add, uint64_t *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] + add;
}
// files -> (bases, offsets) in HBM (the layout mf_count_device takes); all files of the call form ONE read set
// a plain-gzip FASTA / FASTQ file large enough for the several-thread inflater (mf_inflate.h: 32 MB), with the device parser on
static bool gz_straight_to_device(mf_ctx *ctx, const char *path) {
    std::string p(path);
    if (!ctx->opt_device_parse || !ends_with_nocase(p, ".gz")) return false;
    if (getenv("MF_FAST_INFLATE") && atoi(getenv("MF_FAST_INFLATE")) == 0) return false;
    p.resize(p.size() - 3);
    if (!(ends_with_nocase(p, ".fastq") || ends_with_nocase(p, ".fq") || ends_with_nocase(p, ".fasta") || ends_with_nocase(p, ".fa") || ends_with_nocase(p, ".fn") || ends_with_nocase(p, ".fna"))) return false;
    struct stat st;
    return stat(path, &st) == 0 && (int64_t)st.st_size >= ctx->opt_gz_device_min;
}
static int load_reads_to_device(mf_ctx *ctx, const char *const *files, int nfiles, mf_buf<uint8_t> &db, mf_buf<uint64_t> &doff,
                                uint64_t *n_reads, uint64_t *n_bases, double *t_parse, double *t_h2d) {
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    MF_HIP(hipSetDevice(ctx->device));
    // every file becomes a list of pieces: bases either already in HBM (streaming reader) or still on the host
    // (dev_offsets: a file the DEVICE parsed, mf_dparse.hip -- its offsets are in HBM too; n_reads says how many)
    struct piece { const uint8_t *dev = nullptr; const uint8_t *host = nullptr; uint64_t n_bases = 0; const std::vector<uint64_t> *offsets = nullptr;
                   const uint64_t *dev_offsets = nullptr; uint64_t n_reads = 0; };
    std::vector<piece> pieces;
    struct dp_file { mf_buf<uint8_t> b; mf_buf<uint64_t> o; };
    std::vector<std::unique_ptr<dp_file>> dparsed;
    std::vector<std::unique_ptr<sr_file>> streamed;
    std::vector<std::unique_ptr<std::vector<read_batch>>> parsed;
    // Compressed files (a paired-end library is two .fastq.gz files, a multi-lane one eight): gzip and bzip2 streams inflate on ONE thread each --
    // 0.22 GB/s of FASTA, the whole step (tools/gz_rate.py: kmer-counter 2.87 s where `gzip -dc` alone takes 3.45 s) --, so the files of a library
    // inflate side by side, up to four at a time; they join the pieces in the order of the command line, and so do their errors.
    // (round 5: what the host has inflated is FASTA / FASTQ text like any plain file's: it goes up through the staging chunks and the DEVICE parses
    // it -- mf_dparse_mem --; host parse, pageable copies and the offsets' rebase were 0.7 of a 20 M-read file's 1.85 s)
    struct pre_file { int rc = 1; std::unique_ptr<std::vector<read_batch>> parts; std::unique_ptr<raw_file> content; int fmt = 0; std::string err; };      // rc 1: not read ahead
    std::vector<pre_file> prep((size_t)std::max(nfiles, 0));
    {
        std::vector<int> comp;
        for (int i = 0; i < nfiles; i++) {
            std::string p(files[i]);
            if (gz_straight_to_device(ctx, files[i])) continue;                  // (a large .gz file takes every thread by itself, and the one upload path: in its turn below)
            if (ends_with_nocase(p, ".gz") || ends_with_nocase(p, ".bz2")) comp.push_back(i);
        }
        if (comp.size() >= 2) {
            std::atomic<size_t> next{0};
            const size_t T = std::min<size_t>(comp.size(), 4);
            const int per = std::max(1, ctx->host_threads / (int)T);
            std::vector<std::thread> th;
            for (size_t t = 0; t < T; t++)
                th.emplace_back([&]() {
                    for (;;) {
                        const size_t j = next++;
                        if (j >= comp.size()) break;
                        pre_file &P = prep[(size_t)comp[j]];
                        P.content = std::make_unique<raw_file>();
                        const int rc = read_file_content(files[comp[j]], per, *P.content, &P.fmt);
                        if (rc < 0) P.err = mf_last_error();                     // (the message lives in this thread)
                        P.rc = rc < 0 ? rc : 0;
                    }
                });
            for (auto &x : th) x.join();
        }
    }
    for (int i = 0; i < nfiles; i++) {
        std::string p(files[i]);
        int fmt = 0;
        if (ends_with_nocase(p, ".fastq") || ends_with_nocase(p, ".fq")) fmt = 2;
        else if (ends_with_nocase(p, ".fasta") || ends_with_nocase(p, ".fa") || ends_with_nocase(p, ".fn") || ends_with_nocase(p, ".fna")) fmt = 1;
        struct stat st;
        if (fmt && ctx->opt_device_parse && stat(files[i], &st) == 0 && (size_t)st.st_size >= (size_t)ctx->opt_device_parse_min) {
            auto df = std::make_unique<dp_file>();
            uint64_t r = 0, b = 0;
            const int rc = mf_dparse_file(ctx, files[i], fmt, df->b, df->o, &r, &b);
            if (rc < 0) return rc;
            if (rc == 0) ctx->n_dparse_files++; else ctx->n_dparse_stepped_back++;
            if (rc == 0) {
                if (nfiles == 1) { db.swap(df->b); doff.swap(df->o); *n_reads = r; *n_bases = b; if (t_parse) *t_parse = now() - t0; if (t_h2d) *t_h2d = 0; return MF_OK; }
                piece P; P.dev = df->b.p; P.n_bases = b; P.dev_offsets = df->o.p; P.n_reads = r;
                pieces.push_back(P);
                dparsed.push_back(std::move(df));
                continue;
            }
            // (1: a file the device parser is not sure about -- the host readers below take it, with the reference's error messages)
        }
        if (fmt && ctx->opt_stream_reader && stat(files[i], &st) == 0 && (size_t)st.st_size >= (size_t)std::max<int64_t>(ctx->opt_sr_piece, 4096) / 2) {
            auto sf = std::make_unique<sr_file>();
            const int rc = stream_file_to_device(ctx, files[i], fmt, *sf);
            if (rc < 0) return rc;
            if (rc == 0) {
                for (auto &P : sf->pieces) pieces.push_back(piece{sf->dev.p + P.dev_off, nullptr, P.n_bases, &P.offsets});
                streamed.push_back(std::move(sf));
                continue;
            }
        }
        auto parts = std::make_unique<std::vector<read_batch>>();
        {
            pre_file &P = prep[(size_t)i];
            if (P.rc < 0) return mf_set_error("%s", P.err.c_str());
            if (P.rc == 1 && gz_straight_to_device(ctx, files[i])) {
                // a large .fa.gz / .fq.gz: inflated on many threads into the staging chunks, parsed in HBM (mf_dparse_gz)
                raw_file packed;
                MF_TRY(read_file_parallel(files[i], packed, ctx->host_threads));
                std::string inner(files[i]); inner.resize(inner.size() - 3);
                const int gfmt = (ends_with_nocase(inner, ".fastq") || ends_with_nocase(inner, ".fq")) ? 2 : 1;
                auto df = std::make_unique<dp_file>();
                uint64_t r = 0, b = 0;
                const int rc = mf_dparse_gz(ctx, files[i], packed.data(), packed.size(), gfmt, df->b, df->o, &r, &b);
                if (rc < 0) return rc;
                if (rc == 0) {
                    ctx->n_dparse_files++; ctx->n_gz_device++;
                    if (nfiles == 1) { db.swap(df->b); doff.swap(df->o); *n_reads = r; *n_bases = b; if (t_parse) *t_parse = now() - t0; if (t_h2d) *t_h2d = 0; return MF_OK; }
                    piece Q; Q.dev = df->b.p; Q.n_bases = b; Q.dev_offsets = df->o.p; Q.n_reads = r;
                    pieces.push_back(Q);
                    dparsed.push_back(std::move(df));
                    continue;
                }
                // (1: not a file for that way -- the host inflates it below)
            }
            if (P.rc == 1) {                                                    // (not read ahead: now)
                P.content = std::make_unique<raw_file>();
                const double ta = now();
                MF_TRY(read_file_content(files[i], ctx->host_threads, *P.content, &P.fmt));
                if (getenv("MF_IO_TIMING")) fprintf(stderr, "[mf] %s: read %.3f s\n", files[i], now() - ta);
            }
            const bool packed = ends_with_nocase(p, ".gz") || ends_with_nocase(p, ".bz2");
            if (packed && (P.fmt == 1 || P.fmt == 2) && ctx->opt_device_parse && P.content->size() >= (size_t)ctx->opt_device_parse_min) {
                auto df = std::make_unique<dp_file>();
                uint64_t r = 0, b = 0;
                const int rc = mf_dparse_mem(ctx, files[i], P.content->data(), P.content->size(), P.fmt, df->b, df->o, &r, &b);
                if (rc < 0) return rc;
                if (rc == 0) {
                    ctx->n_dparse_files++;
                    P.content.reset();
                    if (nfiles == 1) { db.swap(df->b); doff.swap(df->o); *n_reads = r; *n_bases = b; if (t_parse) *t_parse = now() - t0; if (t_h2d) *t_h2d = 0; return MF_OK; }
                    piece Q; Q.dev = df->b.p; Q.n_bases = b; Q.dev_offsets = df->o.p; Q.n_reads = r;
                    pieces.push_back(Q);
                    dparsed.push_back(std::move(df));
                    continue;
                }
                ctx->n_dparse_stepped_back++;
            }
            MF_TRY(parse_file_content(*P.content, P.fmt, files[i], ctx->host_threads, *parts));
            P.content.reset();
        }
        for (auto &rb : *parts) pieces.push_back(piece{nullptr, rb.bases.data(), rb.bases.size(), &rb.offsets});
        parsed.push_back(std::move(parts));
    }
    const double t1 = now();
    // pieces -> one (bases, offsets) pair in HBM: bases piece by piece, offsets rebased on the host
    uint64_t nb = 0, nr = 0;
    std::vector<uint64_t> pb(pieces.size()), pr(pieces.size());
    for (size_t t = 0; t < pieces.size(); t++) { pb[t] = nb; pr[t] = nr; nb += pieces[t].n_bases; nr += pieces[t].offsets ? pieces[t].offsets->size() - 1 : pieces[t].n_reads; }
    // (not a std::vector: value-initialising 160 MB of offsets for 20 M reads on one thread cost 35 ms of a 0.14 s load; the threads below
    // touch the pages as they fill them)
    raw_file offsets_buf;
    if (!offsets_buf.alloc_bytes((nr + 1) * 8)) return mf_set_error("out of host memory (%llu reads)", (unsigned long long)nr);
    uint64_t *const offsets = reinterpret_cast<uint64_t *>(offsets_buf.data());
    offsets[0] = 0;
    {
        std::vector<std::thread> th;
        int T = (int)std::min<size_t>(pieces.size(), (size_t)std::max(ctx->host_threads, 1));
        for (int w = 0; w < T; w++)
            th.emplace_back([&, w]() {
                for (size_t t = (size_t)w; t < pieces.size(); t += (size_t)T) {
                    if (!pieces[t].offsets) continue;                             // (offsets in HBM: rebased there, below)
                    const std::vector<uint64_t> &o = *pieces[t].offsets;
                    for (size_t i = 1; i < o.size(); i++) offsets[pr[t] + i] = o[i] + pb[t];
                }
            });
        for (auto &x : th) x.join();
    }
    MF_TRY(db.alloc(ctx, nb + 64)); MF_TRY(doff.alloc(ctx, nr + 1));
    for (size_t t = 0; t < pieces.size(); t++)
        if (pieces[t].n_bases)
            MF_HIP(hipMemcpyAsync(db.p + pb[t], pieces[t].dev ? (const void *)pieces[t].dev : (const void *)pieces[t].host, pieces[t].n_bases,
                                  pieces[t].dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
    MF_HIP(hipMemcpyAsync(doff.p, offsets, (nr + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    for (size_t t = 0; t < pieces.size(); t++)
        if (pieces[t].dev_offsets && pieces[t].n_reads)
            k_offsets_rebase<<<(unsigned)((pieces[t].n_reads + 255) / 256), 256, 0, ctx->stream>>>(pieces[t].dev_offsets + 1, pieces[t].n_reads, pb[t], doff.p + pr[t] + 1);
    MF_HIP(hipStreamSynchronize(ctx->stream));
    *n_reads = nr; *n_bases = nb;
    if (t_parse) *t_parse = t1 - t0;
    if (t_h2d) *t_h2d = now() - t1;
    return MF_OK;
}

// ReadersUtils.readDnaLazy for a list of files (itmo!/io/ReadersUtils.java:81-102): the reads the readers hand on, in HBM
extern "C" int mf_reads_load(mf_ctx *ctx, const char *const *files, int nfiles, mf_reads **out) {
    mf_range rng_("mf:reads_load(files)");
    if (!ctx || !out || (nfiles && !files)) return mf_set_error("mf_reads_load: NULL argument");
    *out = nullptr;
    mf_buf<uint8_t> db; mf_buf<uint64_t> doff;
    uint64_t nr = 0, nb = 0;
    MF_TRY(load_reads_to_device(ctx, files, nfiles, db, doff, &nr, &nb, nullptr, nullptr));
    mf_reads *R = new mf_reads();
    R->ctx = ctx; R->n = nr; R->n_bases = nb;
    R->bases_bytes = db.bytes(); R->offsets_bytes = doff.bytes();
    R->d_bases = db.take(); R->d_offsets = doff.take();
    *out = R;
    return MF_OK;
}
extern "C" void mf_reads_destroy(mf_reads *r) {
    if (!r) return;
    if (r->d_bases) mf_release(r->ctx, r->d_bases, r->bases_bytes);
    if (r->d_offsets) mf_release(r->ctx, r->d_offsets, r->offsets_bytes);
    delete r;
}
extern "C" int mf_reads_stats(const mf_reads *r, uint64_t *n_reads, uint64_t *n_bases) {
    if (!r) return mf_set_error("reads handle is NULL");
    if (n_reads) *n_reads = r->n;
    if (n_bases) *n_bases = r->n_bases;
    return MF_OK;
}
extern "C" int mf_reads_device_view(const mf_reads *r, const void **d_bases, const void **d_offsets) {
    if (!r) return mf_set_error("reads handle is NULL");
    if (d_bases) *d_bases = r->d_bases;
    if (d_offsets) *d_offsets = r->d_offsets;
    return MF_OK;
}
extern "C" int mf_reads_export(const mf_reads *r, uint8_t *bases, uint64_t *offsets) {
    if (!r) return mf_set_error("reads handle is NULL");
    MF_HIP(hipSetDevice(r->ctx->device));
    if (bases && r->n_bases) MF_HIP(hipMemcpyAsync(bases, r->d_bases, r->n_bases, hipMemcpyDeviceToHost, r->ctx->stream));
    if (offsets) MF_HIP(hipMemcpyAsync(offsets, r->d_offsets, (r->n + 1) * 8, hipMemcpyDeviceToHost, r->ctx->stream));
    MF_HIP(hipStreamSynchronize(r->ctx->stream));
    return MF_OK;
}

int mf_count_files_streamed(mf_ctx *ctx, const char *const *files, int nfiles, int k, int min_read_len, int thr, uint64_t *n_all, mf_table **out);
// IOUtils.loadReads (src/io/IOUtils.java:772-803): all files into one table
static int count_reads_impl(mf_ctx *ctx, const char *const *files, int nfiles, int k, int min_read_len, int threshold, mf_table **out,
                            uint64_t *n_distinct_all, const char *who) {
    if (!ctx || !out || (nfiles && !files)) return mf_set_error("%s: NULL argument", who);
    *out = nullptr;
    if (k < 1) return mf_set_error("The size of k-mer must be at least 1.");
    if (k > 31) return mf_set_error("The size of k-mer must be no more than 31.");
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    {
        // large plain FASTA / FASTQ files are counted while they cross PCIe (mf_stream.hip); 1: not that way, nothing was produced
        const int src = mf_count_files_streamed(ctx, files, nfiles, k, min_read_len, threshold < 0 ? -1 : threshold, n_distinct_all, out);
        if (src <= 0) return src;
        if (n_distinct_all) *n_distinct_all = 0;
    }
    mf_buf<uint8_t> db; mf_buf<uint64_t> doff;
    uint64_t nr = 0, nb = 0; double tp = 0, th = 0;
    MF_TRY(load_reads_to_device(ctx, files, nfiles, db, doff, &nr, &nb, &tp, &th));
    const double t2 = now();
    int rc = mf_count_core(ctx, db.p, doff.p, nr, nb, k, min_read_len, out, threshold < 0 ? -1 : threshold, n_distinct_all);
    {
        static const bool env = getenv("MF_IO_TIMING") != nullptr;
        if (env) fprintf(stderr, "[mf] count_reads (%d file(s), %llu reads): read+parse+H2D %.3f s, offsets+compaction %.3f s, count %.3f s (hipMalloc so far: %llu calls, %.1f GB, %.3f s)\n", nfiles, (unsigned long long)nr, tp, th, now() - t2,
                         (unsigned long long)ctx->n_hipmalloc, ctx->b_hipmalloc / 1e9, ctx->t_hipmalloc);
    }
    if (ctx->opt_verbose)
        fprintf(stderr, "[mf] count_reads: read+parse %.3f s, offsets+H2D %.3f s, count %.3f s (%llu reads, %llu bases, %d host threads)\n",
                tp, th, now() - t2, (unsigned long long)nr, (unsigned long long)nb, ctx->host_threads);
    return rc;
}
extern "C" int mf_count_reads(mf_ctx *ctx, const char *const *files, int nfiles, int k, int min_read_len, mf_table **out) {
    return count_reads_impl(ctx, files, nfiles, k, min_read_len, -1, out, nullptr, "mf_count_reads");
}
// KmersCounterMain.runImpl (src/tools/KmersCounterMain.java:77-99): loadReads, then printKmers keeps value > threshold
extern "C" int mf_count_reads_above(mf_ctx *ctx, const char *const *files, int nfiles, int k, int min_read_len, int threshold, mf_table **out,
                                    uint64_t *n_distinct_all) {
    mf_range rng_("mf:count_reads(files)");
    if (n_distinct_all) *n_distinct_all = 0;
    return count_reads_impl(ctx, files, nfiles, k, min_read_len, threshold, out, n_distinct_all, "mf_count_reads_above");
}

// ---------------------------------------------------------------------------------------------
// A5 / A6
// ---------------------------------------------------------------------------------------------
// IOUtils.printKmers (src/io/IOUtils.java:45-71) + QuickQuantitativeStatistics.printToFile (:65-72)
// 10-byte big-endian records <-> (key, count) arrays, on the device: 256 records per workgroup staged through LDS so that
// HBM sees whole dwords
#define REC_PER_BLOCK 256
__global__ __launch_bounds__(REC_PER_BLOCK) void k_records_encode(const uint64_t *__restrict__ keys, const uint16_t *__restrict__ cnts, uint64_t n,
                                                                   uint32_t *__restrict__ out) {
    __shared__ __attribute__((aligned(4))) uint8_t st[REC_PER_BLOCK * 10];
    const uint64_t base = (uint64_t)blockIdx.x * REC_PER_BLOCK, i = base + threadIdx.x;
    if (i < n) {
        const uint64_t k = keys[i]; const uint32_t c = cnts[i];
        uint8_t *p = st + threadIdx.x * 10;
#pragma unroll
        for (int b = 0; b < 8; b++) p[b] = (uint8_t)(k >> (8 * (7 - b)));
        p[8] = (uint8_t)(c >> 8); p[9] = (uint8_t)c;
    }
    __syncthreads();
    const uint64_t nrec = n - base < REC_PER_BLOCK ? n - base : REC_PER_BLOCK;
    const uint32_t nbytes = (uint32_t)nrec * 10u, nw = (nbytes + 3u) / 4u;        // (the output buffer is padded to a dword)
    for (uint32_t w = threadIdx.x; w < nw; w += REC_PER_BLOCK) out[base * 10 / 4 + w] = reinterpret_cast<const uint32_t *>(st)[w];
}
// counts come out as freq + 1 for the records with freq > thr and 0 for the others (a selection on "> 0" follows)
__global__ __launch_bounds__(REC_PER_BLOCK) void k_records_decode(const uint32_t *__restrict__ raw, uint64_t n, int thr, uint64_t *__restrict__ keys,
                                                                   uint16_t *__restrict__ cnts) {
    __shared__ __attribute__((aligned(4))) uint8_t st[REC_PER_BLOCK * 10];
    const uint64_t base = (uint64_t)blockIdx.x * REC_PER_BLOCK, i = base + threadIdx.x;
    const uint64_t nrec = n - base < REC_PER_BLOCK ? n - base : REC_PER_BLOCK;
    const uint32_t nw = ((uint32_t)nrec * 10u + 3u) / 4u;
    for (uint32_t w = threadIdx.x; w < nw; w += REC_PER_BLOCK) reinterpret_cast<uint32_t *>(st)[w] = raw[base * 10 / 4 + w];
    __syncthreads();
    if (i < n) {
        const uint8_t *p = st + threadIdx.x * 10;
        uint64_t k = 0;
#pragma unroll
        for (int b = 0; b < 8; b++) k = (k << 8) | p[b];
        const int f = (int)(int16_t)(((uint32_t)p[8] << 8) | p[9]);
        keys[i] = k;
        cnts[i] = f > thr ? (uint16_t)(f + 1) : (uint16_t)0;
    }
}
__global__ void k_counts_minus_one(uint16_t *__restrict__ c, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) c[i] = (uint16_t)(c[i] - 1);
}
int mf_table_select_sorted(const mf_table *t, int threshold, mf_buf<uint64_t> &sk, mf_buf<uint16_t> &sc, uint64_t *n);
// ---- bytes in HBM / in host memory -> a file, fast.  The file-level seams write what the reference writes (10 bytes per k-mer,
// 8 per component member, FASTA text): hundreds of megabytes per sample that used to go through one pageable buffer and one
// fwrite (3.5 GB/s: 0.2 s per .kmers.bin of a 20 M-read sample).  Here: slots of the context's pinned pool, a D2H copy per slot,
// and a pwrite per slot on a thread of its own while the next slot's copy runs (the page cache takes ~2 GB/s per thread).
int mf_ensure_pin_pool(mf_ctx *ctx, size_t want) {
    if (ctx->pin_pool_bytes >= want && ctx->pin_pool_pinned == (ctx->opt_host_pinned != 0)) return MF_OK;
    if (ctx->pin_pool) { if (ctx->pin_pool_pinned) hipHostFree(ctx->pin_pool); else free(ctx->pin_pool); ctx->pin_pool = nullptr; ctx->pin_pool_bytes = 0; }
    if (ctx->opt_host_pinned) {
        if (hipHostMalloc(&ctx->pin_pool, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); ctx->pin_pool = nullptr; return mf_set_error("no pinned host memory (%zu bytes)", want); }
        ctx->pin_pool_pinned = true;
    } else {
        // (plain memory, untouched: the threads that fill it take the page faults, in parallel)
        if (posix_memalign(&ctx->pin_pool, 2 << 20, want) != 0) { ctx->pin_pool = nullptr; return mf_set_error("no host memory (%zu bytes)", want); }
        ctx->pin_pool_pinned = false;
    }
    ctx->pin_pool_bytes = want;
    return MF_OK;
}
static int pwrite_all(int fd, const void *p, size_t n, off_t off) {
    const char *q = (const char *)p;
    while (n) { const ssize_t w = pwrite(fd, q, n, off); if (w <= 0) return -1; q += w; n -= (size_t)w; off += w; }
    return 0;
}
static int device_to_file(mf_ctx *ctx, const void *d_src, size_t bytes, int fd, off_t file_off, const char *path) {
    if (!bytes) return MF_OK;
    const size_t SLOT = (size_t)16 << 20;
    MF_TRY(mf_ensure_pin_pool(ctx, 8 * SLOT));
    const size_t nslot = std::min<size_t>(ctx->pin_pool_bytes / SLOT, 24);
    std::vector<std::thread> wr(nslot);
    std::atomic<int> bad{0};
    // (copies into a shared mapping of the file instead of pwrites were measured in round 4: the same 11 GB/s -- the page cache's own pace)
    size_t i = 0;
    for (size_t at = 0; at < bytes; at += SLOT, i++) {
        const size_t s = i % nslot, m = std::min(SLOT, bytes - at);
        if (wr[s].joinable()) wr[s].join();
        char *h = (char *)ctx->pin_pool + s * SLOT;
        if (hipMemcpyAsync(h, (const char *)d_src + at, m, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) { bad = 2; break; }
        wr[s] = std::thread([=, &bad]() { if (pwrite_all(fd, h, m, file_off + (off_t)at) != 0) bad = 1; });
    }
    for (auto &t : wr) if (t.joinable()) t.join();
    if (bad == 2) return mf_set_error("device copy failed while writing '%s'", path);
    if (bad) return mf_set_error("can't write '%s'", path);
    return MF_OK;
}
// host memory -> file, the ranges written by up to 16 threads
static int host_to_file(const char *data, size_t bytes, int fd, off_t file_off, const char *path) {
    if (!bytes) return MF_OK;
    const size_t T = std::min<size_t>(16, (bytes + ((size_t)8 << 20) - 1) / ((size_t)8 << 20));
    std::vector<std::thread> th; std::atomic<int> bad{0};
    for (size_t t = 0; t < T; t++) {
        const size_t lo = bytes * t / T, hi = bytes * (t + 1) / T;
        th.emplace_back([=, &bad]() { if (pwrite_all(fd, data + lo, hi - lo, file_off + (off_t)lo) != 0) bad = 1; });
    }
    for (auto &x : th) x.join();
    if (bad) return mf_set_error("can't write '%s'", path);
    return MF_OK;
}
// n sorted (k-mer, count) pairs in HBM -> 10-byte big-endian records in `path`
static int write_records_file(mf_ctx *ctx, const uint64_t *d_keys, const uint16_t *d_cnts, uint64_t n, const char *path) {
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) return mf_set_error("can't write '%s'", path);
    int rc = MF_OK;
    if (n) {
        // records are encoded in HBM, 2^25 at a time (320 MB), and stream down through the pinned slots
        const uint64_t SLAB = 1ull << 25;
        mf_buf<uint32_t> enc;
        if (enc.alloc(ctx, (std::min(n, SLAB) * 10 + 3) / 4 + 1) != MF_OK) { close(fd); return -1; }
        for (uint64_t i = 0; i < n && rc == MF_OK; i += SLAB) {
            const uint64_t m = std::min(SLAB, n - i);
            k_records_encode<<<(unsigned)((m + REC_PER_BLOCK - 1) / REC_PER_BLOCK), REC_PER_BLOCK, 0, ctx->stream>>>(d_keys + i, d_cnts + i, m, enc.p);
            rc = device_to_file(ctx, enc.p, m * 10, fd, (off_t)(i * 10), path);
        }
    }
    if (close(fd) != 0 && rc == MF_OK) rc = mf_set_error("can't write '%s'", path);
    return rc;
}
int mf_table_select_filtered_sorted(const mf_table *t, int threshold, mf_table *filter, int filter_threshold, mf_buf<uint64_t> &sk, mf_buf<uint16_t> &sc,
                                    uint64_t *n);
extern "C" int mf_table_write_kmers_filtered(const mf_table *t, int threshold, mf_table *filter, int filter_threshold, const char *kmers_bin,
                                             uint64_t *n_good) {
    if (!t || !filter || !kmers_bin) return mf_set_error("mf_table_write_kmers_filtered: NULL argument");
    if (t->k != filter->k) return mf_set_error("mf_table_write_kmers_filtered: tables with different k (%d, %d)", t->k, filter->k);
    uint64_t n = 0;
    mf_buf<uint64_t> sk; mf_buf<uint16_t> sc;
    MF_TRY(mf_table_select_filtered_sorted(t, threshold, filter, filter_threshold, sk, sc, &n));
    MF_TRY(write_records_file(t->ctx, sk.p, sc.p, n, kmers_bin));
    if (n_good) *n_good = n;
    return MF_OK;
}
extern "C" int mf_table_write_kmers(const mf_table *t, int threshold, const char *kmers_bin, const char *stat_txt, uint64_t *n_good) {
    mf_range rng_("mf:write_kmers(file)");
    if (!t || !kmers_bin) return mf_set_error("mf_table_write_kmers: NULL argument");
    mf_ctx *ctx = t->ctx;
    io_timer tm("write_kmers");
    uint64_t n = 0;
    mf_buf<uint64_t> sk; mf_buf<uint16_t> sc;
    MF_TRY(mf_table_select_sorted(t, threshold, sk, sc, &n));
    tm.lap("select+sort");
    MF_TRY(write_records_file(ctx, sk.p, sc.p, n, kmers_bin));
    tm.lap("encode+D2H+write");
    if (ctx->opt_file_cache_gb > 0) {
        // the table the file was written from stays: itself when it holds exactly the records of the file (the cut was made while
        // counting, mf_count_reads_above), a filtered copy otherwise
        mf_table *keep = const_cast<mf_table *>(t); bool mine = false;
        if (!(t->owns_arrays && n == t->n)) { keep = nullptr; if (mf_table_filter(t, threshold, &keep) == MF_OK) mine = true; else keep = nullptr; }
        if (keep) { file_cache_put(ctx, kmers_bin, keep, threshold, nullptr); if (mine) mf_table_destroy(keep); }
        tm.lap("cache");
    }
    FILE *f = nullptr;
    if (stat_txt) {
        std::vector<uint64_t> hist;
        MF_TRY(mf_table_count_hist(t, hist));
        f = fopen(stat_txt, "w");
        if (!f) return mf_set_error("can't write '%s'", stat_txt);
        fprintf(f, "# k-mer frequency\tnumber of such k-mers\n");
        for (size_t v = 0; v < hist.size(); v++) if (hist[v]) fprintf(f, "%zu\t%llu\n", v, (unsigned long long)hist[v]);
        fprintf(f, "\n");
        fclose(f);
    }
    if (n_good) *n_good = n;
    return MF_OK;
}
// IOUtils.loadKmers (src/io/IOUtils.java:369-401), Kmers2HMWorker.processKmer (:249-257), KmersLoadWorker (src/io/KmersLoadWorker.java:16-34):
// the files go to HBM as they are and are decoded there; records with freq <= freq_threshold are dropped; a k-mer that
// occurs in several files gets the (saturating) sum
int mf_table_from_device_pairs(mf_ctx *ctx, const uint64_t *d_keys, const uint16_t *d_vals, uint64_t n, int k, mf_table **out);
extern "C" int mf_table_load_kmers(mf_ctx *ctx, const char *const *files, int nfiles, int freq_threshold, int k, mf_table **out) {
    mf_range rng_("mf:load_kmers(file)");
    if (!ctx || !out || (nfiles && !files)) return mf_set_error("mf_table_load_kmers: NULL argument");
    *out = nullptr;
    MF_HIP(hipSetDevice(ctx->device));
    io_timer tm("load_kmers");
    if (nfiles == 1) {
        mf_file_entry *e = file_cache_get(ctx, files[0]);
        if (e && e->t && e->t->k == k) {
            // (the file's records all have count > e->thr: a threshold at or below that keeps every one of them -- the same table)
            if (freq_threshold <= e->thr) { e->t->refs++; *out = e->t; tm.lap("resident"); return MF_OK; }
            // (the filter allocates, and an allocation the arena cannot serve empties the file cache -- mf_alloc --, which would hand the
            // cached table's arrays back while the filter still reads them: hold a handle of our own across the call, and do not
            // touch `e` after it)
            mf_table *src = e->t; src->refs++;
            const int rc = mf_table_filter(src, freq_threshold, out);
            mf_table_destroy(src);
            tm.lap("resident, filtered");
            return rc;
        }
    }
    // the files' bytes go to HBM as they are (staged pieces, mf_upload_file) and are decoded there: no host copy of a file is made
    std::vector<int> fds((size_t)nfiles, -1);
    std::vector<size_t> fsz((size_t)nfiles, 0);
    struct closer { std::vector<int> &f; ~closer() { for (int d : f) if (d >= 0) close(d); } } closer_{fds};
    uint64_t total = 0;
    for (int i = 0; i < nfiles; i++) {
        fds[(size_t)i] = open(files[i], O_RDONLY);
        struct stat stf;
        if (fds[(size_t)i] < 0 || fstat(fds[(size_t)i], &stf) != 0) return mf_set_error("can't open '%s'", files[i]);
        fsz[(size_t)i] = (size_t)stf.st_size;
        if (fsz[(size_t)i] % 10) return mf_set_error("Can't load k-mers file '%s': size is not a multiple of the 10-byte record", files[i]);
        total += fsz[(size_t)i] / 10;
    }
    mf_buf<uint64_t> dk; mf_buf<uint16_t> dc;
    MF_TRY(dk.alloc(ctx, total)); MF_TRY(dc.alloc(ctx, total));
    uint64_t at = 0;
    for (int i = 0; i < nfiles; i++) {
        const uint64_t m = fsz[(size_t)i] / 10;
        if (!m) continue;
        mf_buf<uint32_t> raw; MF_TRY(raw.alloc(ctx, (m * 10 + 3) / 4 + 1));
        int urc = mf_upload_file(ctx, fds[(size_t)i], m * 10, reinterpret_cast<uint8_t *>(raw.p));
        if (urc == 1) {                                   // (no staging memory: the whole file through one host buffer)
            raw_file rf;
            MF_TRY(read_file_parallel(files[i], rf, ctx->host_threads));
            if (rf.size() != m * 10) return mf_set_error("short read on '%s'", files[i]);
            MF_HIP(hipMemcpyAsync(raw.p, rf.data(), m * 10, hipMemcpyHostToDevice, ctx->stream));
            MF_HIP(hipStreamSynchronize(ctx->stream));
            urc = 0;
        }
        if (urc < 0) return urc;
        k_records_decode<<<(unsigned)((m + REC_PER_BLOCK - 1) / REC_PER_BLOCK), REC_PER_BLOCK, 0, ctx->stream>>>(raw.p, m, freq_threshold, dk.p + at, dc.p + at);
        MF_HIP(hipStreamSynchronize(ctx->stream));
        at += m;
    }
    tm.lap("H2D+decode");
    // keep the records with freq > freq_threshold (encoded as freq + 1 > 0)
    mf_table *all = nullptr;
    {
        mf_table tmp; tmp.ctx = ctx; tmp.k = k; tmp.n = total; tmp.d_keys = dk.p; tmp.d_counts = dc.p; tmp.owns_arrays = false;
        MF_TRY(mf_table_filter(&tmp, 0, &all));
    }
    k_counts_minus_one<<<(unsigned)((all->n + 255) / 256 + 1), 256, 0, ctx->stream>>>(all->d_counts, all->n);
    int rc = mf_table_from_device_pairs(ctx, all->d_keys, all->d_counts, all->n, k, out);
    const hipError_t se = hipStreamSynchronize(ctx->stream);
    tm.lap("filter+partition");
    mf_table_destroy(all);                                  // (on every path: the synchronise used to return past it)
    if (se != hipSuccess) return mf_set_error("hipStreamSynchronize failed: %s", hipGetErrorString(se));
    return rc;
}

// ---------------------------------------------------------------------------------------------
// A8 unitig files
// ---------------------------------------------------------------------------------------------
// Sequence.printSequences (src/structures/Sequence.java:26-37), FastaDedicatedWriter.writeData (:33-49),
// TextUtils.printWithLineLimit (itmo!/utils/TextUtils.java:35-45): 70 columns
extern "C" int mf_seqs_write_fasta(const mf_seqs *s, const char *path) {
    if (!s || !path) return mf_set_error("mf_seqs_write_fasta: NULL argument");
    io_timer tm("write_fasta");
    std::vector<uint8_t> b; std::vector<uint64_t> o; std::vector<int32_t> a, lo, hi;
    MF_TRY(mf_seqs_to_host(s, b, o, a, lo, hi));
    tm.lap("D2H+order");
    // the text is made by up to 32 threads, each for a range of sequences (sizes first, then the bytes), and written in one go
    const uint64_t n = s->n;
    const int T = (int)std::max<uint64_t>(1, std::min<uint64_t>(32, n / 4096 + 1));
    std::vector<uint64_t> tsize((size_t)T + 1, 0);
    auto hdr = [&](uint64_t i, char *buf) {
        return (size_t)snprintf(buf, 128, ">%llu length=%llu av_weight=%d min_weight=%d max_weight=%d\n", (unsigned long long)(i + 1), (unsigned long long)(o[i + 1] - o[i]), a[i], lo[i], hi[i]);
    };
    auto body = [&](uint64_t len) { return len + (len ? (len - 1) / 70 : 0) + 1; };      // 70 columns, the last line never empty (an empty sequence: one empty line)
    {
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t]() {
                char buf[128]; uint64_t sz = 0;
                for (uint64_t i = n * t / T; i < n * (t + 1) / T; i++) sz += hdr(i, buf) + body(o[i + 1] - o[i]);
                tsize[(size_t)t + 1] = sz;
            });
        for (auto &x : th) x.join();
    }
    for (int t = 0; t < T; t++) tsize[(size_t)t + 1] += tsize[(size_t)t];
    raw_file text;
    if (!text.alloc_bytes(tsize[(size_t)T] + 1)) return mf_set_error("out of memory writing '%s'", path);
    {
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t]() {
                char *w = text.data() + tsize[(size_t)t];
                for (uint64_t i = n * t / T; i < n * (t + 1) / T; i++) {
                    w += hdr(i, w);                                  // (snprintf's terminating 0 is overwritten by what follows; one spare byte at the end)
                    const uint64_t len = o[i + 1] - o[i];
                    const uint8_t *q = b.data() + o[i];
                    uint64_t j = 0;
                    while ((j + 1) * 70 < len) { memcpy(w, q + j * 70, 70); w += 70; *w++ = '\n'; j++; }
                    memcpy(w, q + j * 70, len - j * 70); w += len - j * 70; *w++ = '\n';
                }
            });
        for (auto &x : th) x.join();
    }
    tm.lap("text");
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) return mf_set_error("can't write '%s'", path);
    int rc = host_to_file(text.data(), tsize[(size_t)T], fd, 0, path);
    if (close(fd) != 0 && rc == MF_OK) rc = mf_set_error("can't write '%s'", path);
    tm.lap("write");
    return rc;
}
// SeqBuilderMain.runImpl (src/tools/SeqBuilderMain.java:78-160)
extern "C" int mf_build_unitigs(mf_ctx *ctx, mf_table *t, int k, int freq_threshold, int min_len, const char *seq_fasta,
                                const char *distribution, uint64_t *n_seq) {
    mf_range rng_("mf:seq_builder(files)");
    if (!ctx || !t || !seq_fasta) return mf_set_error("mf_build_unitigs: NULL argument");
    if (k != t->k) return mf_set_error("mf_build_unitigs: k=%d but the table was built with k=%d", k, t->k);
    if (distribution) {                                         // stat[min(value,1023)]++, lines "i stat[i]" for i=1..1023 (:84-98,170-176)
        std::vector<uint64_t> hist;
        MF_TRY(mf_table_count_hist(t, hist));
        uint64_t stat[1024] = {0};
        for (size_t v = 0; v < hist.size(); v++) stat[v >= 1024 ? 1023 : v] += hist[v];
        FILE *f = fopen(distribution, "w");
        if (!f) return mf_set_error("can't write '%s'", distribution);
        for (int i = 1; i < 1024; i++) fprintf(f, "%d %llu\n", i, (unsigned long long)stat[i]);
        fclose(f);
    }
    mf_seqs *s = nullptr;
    {
        io_timer tm("build_unitigs (device)");
        MF_TRY(mf_build_unitigs_device(ctx, t, freq_threshold, min_len, &s));
    }
    int rc = mf_seqs_write_fasta(s, seq_fasta);
    if (n_seq) *n_seq = s->n;
    mf_seqs_destroy(s);
    return rc;
}

// ---------------------------------------------------------------------------------------------
// A11 components files
// ---------------------------------------------------------------------------------------------
// ConnectedComponent.saveComponents (src/structures/ConnectedComponent.java:80-93); stat file ComponentsBuilder.java:146-152
// (the file image is made in HBM -- the members sorted per component as the reference's writer finds them, byte-swapped, the 12-byte
// headers spliced in -- and streams down through the pinned slots: 0.33 -> 0.05 s for the components of 2 x 20 M reads)
int mf_sort_kmers_by_comp(mf_ctx *ctx, const uint32_t *d_comp, const uint64_t *d_kmers, uint64_t n, int key_bits, uint32_t n_comps, uint64_t *d_out);
__global__ __launch_bounds__(256) void k_comps_encode(const uint64_t *__restrict__ kmers, const uint64_t *__restrict__ koff, const uint64_t *__restrict__ foff,
                                                      const long long *__restrict__ weights, uint32_t n_comp, uint32_t *__restrict__ raw) {
    if (blockIdx.x == 0 && threadIdx.x == 0) raw[0] = __builtin_bswap32(n_comp);
    for (uint32_t c = blockIdx.x; c < n_comp; c += gridDim.x) {
        const uint64_t w0 = foff[c] / 4, k0 = koff[c], sz = koff[c + 1] - k0;
        if (threadIdx.x == 0) {
            const unsigned long long w = (unsigned long long)weights[c];
            raw[w0] = __builtin_bswap32((uint32_t)sz); raw[w0 + 1] = __builtin_bswap32((uint32_t)(w >> 32)); raw[w0 + 2] = __builtin_bswap32((uint32_t)w);
        }
        for (uint64_t j = threadIdx.x; j < sz; j += blockDim.x) {
            const uint64_t x = kmers[k0 + j];
            raw[w0 + 3 + 2 * j] = __builtin_bswap32((uint32_t)(x >> 32)); raw[w0 + 4 + 2 * j] = __builtin_bswap32((uint32_t)x);
        }
    }
}
extern "C" int mf_comps_write(const mf_comps *cc, const char *components_bin, const char *stat_txt) {
    if (!cc || !components_bin) return mf_set_error("mf_comps_write: NULL argument");
    mf_comps *c = const_cast<mf_comps *>(cc);
    mf_ctx *ctx = c->ctx;
    io_timer tm("write_components");
    if (c->n >= 0xFFFFFFFFull) return mf_set_error("components: too many components for the file format");
    MF_HIP(hipSetDevice(ctx->device));
    const uint64_t n = c->n, nk = c->n_kmers;
    std::vector<uint64_t> koff(n + 1, 0), foff(n + 1, 0);
    foff[0] = 4;
    for (uint64_t i = 0; i < n; i++) { koff[i + 1] = koff[i] + c->sizes[i]; foff[i + 1] = foff[i] + 12 + 8 * c->sizes[i]; }
    if (koff[n] != nk) return mf_set_error("components: sizes do not add up to the member list");
    const uint64_t total = foff[n];
    mf_buf<uint32_t> raw; MF_TRY(raw.alloc(ctx, total / 4 + 1));
    if (n) {
        mf_buf<uint64_t> sorted, dko, dfo; mf_buf<long long> dw;
        MF_TRY(sorted.alloc(ctx, std::max<uint64_t>(nk, 1))); MF_TRY(dko.alloc(ctx, n + 1)); MF_TRY(dfo.alloc(ctx, n + 1)); MF_TRY(dw.alloc(ctx, n));
        if (nk) MF_TRY(mf_sort_kmers_by_comp(ctx, c->d_comp, c->d_kmers, nk, 2 * (c->k > 0 ? c->k : 32), (uint32_t)std::max<uint64_t>(n, 1), sorted.p));
        MF_HIP(hipMemcpyAsync(dko.p, koff.data(), (n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        MF_HIP(hipMemcpyAsync(dfo.p, foff.data(), (n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        MF_HIP(hipMemcpyAsync(dw.p, c->weights.data(), n * 8, hipMemcpyHostToDevice, ctx->stream));
        k_comps_encode<<<(unsigned)std::min<uint64_t>(n, 65535), 256, 0, ctx->stream>>>(sorted.p, dko.p, dfo.p, dw.p, (uint32_t)n, raw.p);
        MF_HIP(hipStreamSynchronize(ctx->stream));
    } else {
        const uint32_t zero = 0;
        MF_HIP(hipMemcpyAsync(raw.p, &zero, 4, hipMemcpyHostToDevice, ctx->stream));
        MF_HIP(hipStreamSynchronize(ctx->stream));
    }
    tm.lap("image");
    const int fd = open(components_bin, O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) return mf_set_error("can't write '%s'", components_bin);
    int rc = device_to_file(ctx, raw.p, total, fd, 0, components_bin);
    if (close(fd) != 0 && rc == MF_OK) rc = mf_set_error("can't write '%s'", components_bin);
    MF_TRY(rc);
    tm.lap("write");
    if (stat_txt) {
        FILE *f = fopen(stat_txt, "w");
        if (!f) return mf_set_error("can't write '%s'", stat_txt);
        fprintf(f, "# component.no\tcomponent.size\tcomponent.weight\tusedFreqThreshold\n");
        for (uint64_t i = 0; i < c->n; i++)
            fprintf(f, "%llu\t%llu\t%lld\t%d\n", (unsigned long long)(i + 1), (unsigned long long)c->sizes[i], (long long)c->weights[i], c->thr[i]);
        fclose(f);
    }
    return MF_OK;
}
// ConnectedComponent.loadComponents (:95-122).  The host only walks the component headers (size, weight); the file goes to
// HBM as it is and the k-mers are byte-swapped there, one workgroup per component.
__global__ __launch_bounds__(256) void k_comps_decode(const uint32_t *__restrict__ raw, const uint64_t *__restrict__ foff, const uint64_t *__restrict__ koff,
                                                      uint32_t n_comp, uint64_t *__restrict__ kmers, uint32_t *__restrict__ comp) {
    for (uint32_t c = blockIdx.x; c < n_comp; c += gridDim.x) {
        const uint64_t w0 = foff[c] / 4, k0 = koff[c], sz = koff[c + 1] - k0;
        for (uint64_t j = threadIdx.x; j < sz; j += blockDim.x) {
            const uint32_t hi = __builtin_bswap32(raw[w0 + 2 * j]), lo = __builtin_bswap32(raw[w0 + 2 * j + 1]);
            kmers[k0 + j] = ((uint64_t)hi << 32) | lo;
            comp[k0 + j] = c;
        }
    }
}
extern "C" int mf_comps_load(mf_ctx *ctx, const char *components_bin, mf_comps **out) {
    mf_range rng_("mf:load_components(file)");
    if (!ctx || !components_bin || !out) return mf_set_error("mf_comps_load: NULL argument");
    *out = nullptr;
    io_timer tm("load_components");
    if (mf_file_entry *e = file_cache_get(ctx, components_bin)) {
        if (e->c) { e->c->refs++; *out = e->c; tm.lap("resident"); return MF_OK; }
    }
    // The host only walks the component headers (one every 12 + 8 x size bytes): through a mapping of the file, which touches one page per
    // component; the file's bytes themselves go to HBM in staged pieces (mf_upload_file) and are decoded there.  (Reading the whole file into
    // a host buffer first cost 80 - 140 ms for 0.24 GB: first-touch page faults, the copy, and giving the pages back.)
    const int fd = open(components_bin, O_RDONLY);
    struct stat stc;
    if (fd < 0 || fstat(fd, &stc) != 0) { if (fd >= 0) close(fd); return mf_set_error("Can't load components: file not found (%s)", components_bin); }
    struct fd_closer { int f; ~fd_closer() { close(f); } } fdc{fd};
    const size_t n = (size_t)stc.st_size;
    void *map = n ? mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
    raw_file buf;                                         // (only where the file cannot be mapped)
    if (n && map == MAP_FAILED) { map = nullptr; if (read_file_parallel(components_bin, buf, ctx->host_threads) < 0) return mf_set_error("Can't load components: file not found (%s)", components_bin); }
    struct unmapper { void *m; size_t n; ~unmapper() { if (m) munmap(m, n); } } um{map, n};
    tm.lap("open");
    const uint8_t *p = map ? (const uint8_t *)map : (const uint8_t *)buf.data();
    std::unique_ptr<mf_comps, void (*)(mf_comps *)> C(new mf_comps(), [](mf_comps *c) { mf_comps_destroy(c); });
    std::vector<uint64_t> foff, koff;
    MF_TRY(comps_walk_headers(p, n, C->sizes, C->weights, foff, koff));          // (the host walks the headers only)
    tm.lap("headers");
    const uint64_t cnt = C->sizes.size();
    C->ctx = ctx; C->k = 0; C->n = cnt;
    C->thr.assign(cnt, 0);
    const uint64_t nk = koff[cnt];
    if (nk >= 0xFFFFFFFFull) return mf_set_error("components: too many k-mers");
    C->n_kmers = nk;
    MF_HIP(hipSetDevice(ctx->device));
    void *q = nullptr;
    MF_TRY(mf_alloc(ctx, (nk ? nk : 1) * 8, &q)); C->d_kmers = (uint64_t *)q; C->kmers_bytes = (nk ? nk : 1) * 8;
    MF_TRY(mf_alloc(ctx, (nk ? nk : 1) * 4, &q)); C->d_comp = (uint32_t *)q; C->comp_bytes = (nk ? nk : 1) * 4;
    if (nk) {
        mf_buf<uint32_t> raw; MF_TRY(raw.alloc(ctx, n / 4 + 2));
        mf_buf<uint64_t> dfo, dko; MF_TRY(dfo.alloc(ctx, cnt + 1)); MF_TRY(dko.alloc(ctx, cnt + 1));
        int urc = map ? mf_upload_file(ctx, fd, n, reinterpret_cast<uint8_t *>(raw.p)) : 1;
        if (urc < 0) return urc;
        if (urc == 1) {                                   // (no staging memory / no mapping: one pageable copy)
            if (!buf.p && read_file_parallel(components_bin, buf, ctx->host_threads) < 0) return mf_set_error("Can't load components: file not found (%s)", components_bin);
            MF_HIP(hipMemcpyAsync(raw.p, buf.data(), n, hipMemcpyHostToDevice, ctx->stream));
        }
        MF_HIP(hipMemcpyAsync(dfo.p, foff.data(), (cnt + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        MF_HIP(hipMemcpyAsync(dko.p, koff.data(), (cnt + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        tm.lap("alloc+H2D");
        k_comps_decode<<<(unsigned)std::min<uint64_t>(cnt, 65535), 256, 0, ctx->stream>>>(raw.p, dfo.p, dko.p, (uint32_t)cnt, C->d_kmers, C->d_comp);
        MF_HIP(hipStreamSynchronize(ctx->stream));
        tm.lap("decode");
    }
    C->host_ready = false;             // (member lists on the host: built from the device arrays when somebody asks)
    *out = C.release();
    // (features-calculator loads the same components.bin once per library, FeaturesCalculatorMain.java:82 is one load for all of them: a context
    // that did not write the file -- the driver's workers on the other devices -- keeps what it has just decoded)
    file_cache_put(ctx, components_bin, nullptr, 0, *out);
    return MF_OK;
}
// ComponentCutterMain.runImpl :92-108
extern "C" int mf_cut_components(mf_ctx *ctx, mf_table *cutter, int k, int b1, int b2, const char *components_bin,
                                 const char *stat_txt, uint64_t *n_comp) {
    mf_range rng_("mf:component_cutter(files)");
    if (!ctx || !cutter || !components_bin) return mf_set_error("mf_cut_components: NULL argument");
    if (k != cutter->k) return mf_set_error("mf_cut_components: k=%d but the table was built with k=%d", k, cutter->k);
    if (cutter->n == 0)                                                          // ComponentCutterMain.java:84-86
        return mf_set_error("No sequences were found in input files! The following steps will be useless");
    mf_comps *c = nullptr;
    {
        io_timer tm("cut_components (device)");
        MF_TRY(mf_cut_components_device(ctx, cutter, b1, b2, &c));
    }
    int rc = mf_comps_write(c, components_bin, stat_txt);
    if (rc == MF_OK) file_cache_put(ctx, components_bin, nullptr, 0, c);
    if (n_comp) *n_comp = c->n;
    mf_comps_destroy(c);
    return rc;
}

// ComponentCutterMain.runImpl :78-114 over the ranks of a communicator (mf_comm.hip): every rank reads the .seq.fasta files of ITS libraries
// (IOUtils.loadReads :81 -- the readers' part of it), the ranks cut the components together (mf_cut_components_sharded: the same object on every
// rank), rank 0 writes components.bin and the statistics file, and every rank keeps the components in its context's file cache: the
// features step of a library finds them in the HBM of the device that library lives on.
extern "C" int mf_cut_components_sharded_files(mf_comm *cm, const char *const *seq_files, int nfiles, int k, int min_len, int b1, int b2,
                                               const char *components_bin, const char *stat_txt, uint64_t *n_comp) {
    mf_range rng_("mf:component_cutter(files, sharded)");
    if (!cm || !components_bin || nfiles < 0 || (nfiles && !seq_files)) return mf_set_error("mf_cut_components_sharded_files: bad argument");
    mf_ctx *ctx = mf_comm_ctx(cm);
    // this rank's files.  A failure here must not leave the peers waiting in the exchange: every rank says how it went first
    mf_reads *mine = nullptr;
    const int rc_load = nfiles ? mf_reads_load(ctx, seq_files, nfiles, &mine) : MF_OK;
    struct rguard { mf_reads *r; ~rguard() { if (r) mf_reads_destroy(r); } } rg{mine};
    const std::string load_err = rc_load < 0 ? mf_last_error() : "";
    if (mf_comm_agree(cm, rc_load == MF_OK) < 0) { if (rc_load < 0) { mf_set_error("%s", load_err.c_str()); return MF_ERR; } return MF_ERR_TOGETHER; }
    mf_comps *c = nullptr;
    MF_TRY(mf_cut_components_sharded(cm, mine ? mine->d_bases : nullptr, mine ? mine->d_offsets : nullptr, mine ? mine->n : 0, mine ? mine->n_bases : 0, k, min_len, b1, b2, &c));
    struct cguard { mf_comps *p; ~cguard() { mf_comps_destroy(p); } } cg{c};
    int rc = MF_OK;
    if (mf_comm_rank(cm) == 0) rc = mf_comps_write(c, components_bin, stat_txt);
    if (mf_comm_agree(cm, rc == MF_OK) < 0) return rc < 0 ? rc : MF_ERR_TOGETHER;      // (also the barrier behind the write: the file exists for every rank's cache entry)
    file_cache_put(ctx, components_bin, nullptr, 0, c);
    if (n_comp) *n_comp = c->n;
    return MF_OK;
}

// ---------------------------------------------------------------------------------------------
// A12 feature files
// ---------------------------------------------------------------------------------------------
// java.lang.Double.toString: shortest digits that round-trip; plain decimal for 1e-3 <= |d| < 1e7, else d.dddE[-]n
static std::string java_double(double d) {
    if (std::isnan(d)) return "NaN";
    if (std::isinf(d)) return d > 0 ? "Infinity" : "-Infinity";
    if (d == 0) return std::signbit(d) ? "-0.0" : "0.0";
    char buf[64];
    int prec = 1;
    for (; prec <= 17; prec++) { snprintf(buf, sizeof buf, "%.*e", prec - 1, d); if (strtod(buf, nullptr) == d) break; }
    std::string s(buf);                       // [-]d.ddde[+-]xx
    bool neg = s[0] == '-';
    if (neg) s = s.substr(1);
    size_t epos = s.find('e');
    std::string mant = s.substr(0, epos);
    int ex = atoi(s.c_str() + epos + 1);
    std::string digits;
    for (char ch : mant) if (ch != '.') digits.push_back(ch);
    std::string out;
    double ad = std::fabs(d);
    if (ad >= 1e-3 && ad < 1e7) {
        if (ex >= 0) {
            std::string ip = digits.substr(0, std::min<size_t>(digits.size(), (size_t)ex + 1));
            while ((int)ip.size() < ex + 1) ip.push_back('0');
            std::string fp = digits.size() > (size_t)ex + 1 ? digits.substr(ex + 1) : "0";
            out = ip + "." + fp;
        } else out = "0." + std::string((size_t)(-ex - 1), '0') + digits;
    } else {
        out = digits.substr(0, 1) + "." + (digits.size() > 1 ? digits.substr(1) : "0") + "E" + std::to_string(ex);
    }
    return neg ? "-" + out : out;
}
static int write_features_files(const std::vector<int64_t> &vec, const std::vector<double> &br, const char *vec_path, const char *breadth_path);
// FeaturesCalculatorMain.runImpl reads branch (:117-131): the files of ONE library
extern "C" int mf_features_reads_selected(mf_ctx *ctx, const char *components_bin, const char *const *files, int nfiles, int k, int threshold,
                                          mf_table *selected, const char *vec_path, const char *breadth_path) {
    if (!ctx || !components_bin || (nfiles && !files)) return mf_set_error("mf_features_reads: NULL argument");
    mf_comps *c = nullptr;
    MF_TRY(mf_comps_load(ctx, components_bin, &c));
    if (c->n == 0) { mf_comps_destroy(c); return mf_set_error("No components were found in input files! Can't continue the calculations."); }
    mf_buf<uint8_t> db; mf_buf<uint64_t> doff;
    uint64_t nr = 0, nb = 0;
    int rc = load_reads_to_device(ctx, files, nfiles, db, doff, &nr, &nb, nullptr, nullptr);
    std::vector<int64_t> vec(c->n); std::vector<double> br(c->n);
    if (rc == MF_OK) rc = mf_features_reads_device_selected(ctx, c, db.p, doff.p, nr, nb, k, selected, threshold, vec.data(), br.data());
    if (rc == MF_OK) rc = write_features_files(vec, br, vec_path, breadth_path);
    mf_comps_destroy(c);
    return rc;
}
extern "C" int mf_features_reads(mf_ctx *ctx, const char *components_bin, const char *const *files, int nfiles, int k, int threshold,
                                 const char *vec_path, const char *breadth_path) {
    return mf_features_reads_selected(ctx, components_bin, files, nfiles, k, threshold, nullptr, vec_path, breadth_path);
}
// FeaturesCalculatorMain.runImpl kmers-file branch (:137-162) + buildAndPrintVector output (:217-230)
extern "C" int mf_features_selected(mf_ctx *ctx, const char *components_bin, const char *kmers_bin, int k, int threshold, mf_table *selected,
                                    const char *vec_path, const char *breadth_path) {
    if (!ctx || !components_bin || !kmers_bin) return mf_set_error("mf_features: NULL argument");
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    mf_comps *c = nullptr;
    MF_TRY(mf_comps_load(ctx, components_bin, &c));
    if (c->n == 0) { mf_comps_destroy(c); return mf_set_error("No components were found in input files! Can't continue the calculations."); }
    const double t1 = now();
    mf_table *t = nullptr;
    const char *files[1] = {kmers_bin};
    // calculatePresenceForKmers streams EVERY record of the file (no threshold on load): freq_threshold = -1 keeps all
    int rc = mf_table_load_kmers(ctx, files, 1, -1, k, &t);
    if (rc < 0) { mf_comps_destroy(c); return rc; }
    const double t2 = now();
    std::vector<int64_t> vec(c->n); std::vector<double> br(c->n);
    rc = mf_features_device_selected(ctx, c, t, selected, threshold, vec.data(), br.data());
    const double t3 = now();
    if (rc == MF_OK) rc = write_features_files(vec, br, vec_path, breadth_path);
    if (getenv("MF_IO_TIMING")) fprintf(stderr, "[mf] features: components %.3f s, k-mers file %.3f s, features %.3f s, output %.3f s\n", t1 - t0, t2 - t1, t3 - t2, now() - t3);
    mf_table_destroy(t);
    mf_comps_destroy(c);
    return rc;
}
extern "C" int mf_features(mf_ctx *ctx, const char *components_bin, const char *kmers_bin, int k, int threshold,
                           const char *vec_path, const char *breadth_path) {
    return mf_features_selected(ctx, components_bin, kmers_bin, k, threshold, nullptr, vec_path, breadth_path);
}
// buildAndPrintVector output (:217-230): one long per line / one Double.toString per line
static int write_features_files(const std::vector<int64_t> &vec, const std::vector<double> &br, const char *vec_path, const char *breadth_path) {
    if (vec_path) {
        FILE *f = fopen(vec_path, "w");
        if (!f) return mf_set_error("Can't write vector to file %s", vec_path);
        for (int64_t v : vec) fprintf(f, "%lld\n", (long long)v);
        fclose(f);
    }
    if (breadth_path) {
        FILE *f = fopen(breadth_path, "w");
        if (!f) return mf_set_error("Can't write vector to file %s", breadth_path);
        // (Double.toString by trial: up to 17 snprintf + strtod round trips per value -- made by up to 16 threads)
        const size_t n = br.size(), T = std::max<size_t>(1, std::min<size_t>(16, n / 1024));
        std::vector<std::string> part(T);
        std::vector<std::thread> th;
        for (size_t t = 0; t < T; t++)
            th.emplace_back([&, t]() { std::string &o = part[t]; for (size_t i = n * t / T; i < n * (t + 1) / T; i++) { o += java_double(br[i]); o.push_back('\n'); } });
        for (auto &x : th) x.join();
        for (auto &o : part) fwrite(o.data(), 1, o.size(), f);
        fclose(f);
    }
    return MF_OK;
}

// test hook (not part of the ABI; tests/test_inflate_cpu.py): the image of a gzip file -> its content through the whole-buffer decoder alone (mode 1:
// MF_ERR where it refuses), through zlib alone (0), in the product's order (2), or through the decoder with the several-thread form forced on small
// inputs (3: pieces of 48 KB; MF_ERR also where that form steps back to one thread).  *out: malloc'd, mf_debug_free
// (mode 4: the several-thread form into a byte_sink -- what mf_dparse_gz does with pinned chunks -- here into slots of 64 KB that are copied to their place)
struct debug_host_sink : mfz::byte_sink {
    std::vector<std::vector<uint8_t>> slots; std::vector<uint8_t> all; std::mutex m;
    uint8_t *acquire(int w) override { return slots[(size_t)w].data(); }
    bool commit(int w, size_t offset, size_t len) override {
        std::lock_guard<std::mutex> g(m);
        if (all.size() < offset + len) all.resize(offset + len);
        memcpy(all.data() + offset, slots[(size_t)w].data(), len);
        return true;
    }
};
extern "C" int mf_debug_gunzip(const void *in, uint64_t n, int mode, void **out, uint64_t *out_n) {
    if (!out || !out_n || (!in && n)) return mf_set_error("mf_debug_gunzip: NULL argument");
    raw_file packed, plain;
    packed.p = (char *)malloc(n + 64); packed.n = n;
    if (!packed.p) return mf_set_error("out of host memory");
    if (n) memcpy(packed.p, in, n);
    memset(packed.p + n, 0, 64);
    int rc = MF_OK;
    if (mode == 4) {
        debug_host_sink sink;
        sink.workers = 3; sink.slot_bytes = (size_t)64 << 10;
        sink.slots.assign(3, std::vector<uint8_t>(sink.slot_bytes));
        size_t total = 0;
        if (!mfz::gunzip_to_sink(reinterpret_cast<const uint8_t *>(packed.p), n, 4, &sink, &total, (size_t)48 << 10)) return mf_set_error("mf_debug_gunzip: not a file for the sink");
        if (sink.all.size() != total) return mf_set_error("mf_debug_gunzip: the sink got %zu of %zu bytes", sink.all.size(), total);
        plain.p = (char *)malloc(total ? total : 1); plain.n = total;
        if (total) memcpy(plain.p, sink.all.data(), total);
    } else
    if (mode == 1 || mode == 3) {
        char *q = nullptr; size_t m = 0;
        bool par = false;
        if (!mfz::gunzip(reinterpret_cast<const uint8_t *>(packed.p), n, 4, &q, &m, mode == 3 ? 0 : (size_t)32 << 20, mode == 3 ? (size_t)48 << 10 : (size_t)2 << 20, &par))
            return mf_set_error("mf_debug_gunzip: the whole-buffer decoder refuses this input");
        plain.p = q; plain.n = m;
        if (mode == 3 && !par) return mf_set_error("mf_debug_gunzip: the several-thread form stepped back");
    } else rc = mode == 0 ? inflate_gz_zlib(packed, plain, "(memory)") : inflate_gz(packed, plain, "(memory)", 4);
    if (rc < 0) return rc;
    *out = plain.p; *out_n = plain.n;
    plain.p = nullptr;
    return MF_OK;
}
extern "C" void mf_debug_free(void *p) { free(p); }

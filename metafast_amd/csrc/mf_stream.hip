// mf_stream.hip -- the STREAMED count of read files (round 6): upload || parse || level-1 scatter.
//
// IOUtils.loadReads (src/io/IOUtils.java:772-803) feeds its counting workers while ReadsDispatcher (src/io/ReadsDispatcher.java:34-53) is still
// reading: batches of 32 K reads go to the workers as they come off the file.  Until round 6 this build uploaded and parsed a library WHOLE and
// only then counted it: 100 M reads (15.4 GB of FASTA) took 0.30 s to cross PCIe, 0.07 s to parse and 0.07 s to count, one after the other.
// Here the files are cut into pieces at record borders (256 MB each); a producer thread uploads piece after piece on a stream of its own into
// a ring of three buffers, and the context's stream parses piece i (mf_dparse.hip), masks it and scatters its super-k-mer records into the
// level-1 digit regions (mf_skm.hip, skm_stream_*) while piece i + 1 is crossing.  What the level-1 scatter must know before it has seen the
// whole library -- the size of every digit region -- comes from a SAMPLE: 256 KB chunks spread evenly over the files (a 64th of their bytes),
// trimmed to whole records, parsed like a small file.  After the last piece the count goes on from the level-1 records exactly as a count of
// resident reads does (split, LDS tables, gather), so the table is the same table.
//
// Anything unusual -- a file the device parser is not sure about, a record longer than the border search looks, a digit region that the sample
// sized too small, a plan without a split level, a shard of a distributed count -- returns 1 with nothing produced, and the caller loads the
// files whole (the reference's error messages come from there).
#include "mf_common.h"
#include "mf_parse.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <fcntl.h>
#include <memory>
#include <mutex>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

struct mf_skm_stream;
int mf_skm_stream_begin(mf_ctx *ctx, int k, const std::vector<int> &lv, bool adaptive, int table_bits, const uint8_t *s_bases, uint64_t s_nbases,
                        const uint32_t *s_vmask, uint64_t s_nwords, double scale, mf_skm_stream **out);
int mf_skm_stream_piece(mf_skm_stream *S, const uint8_t *d_bases, uint64_t n_bases, const uint32_t *vmask, uint64_t n_words);
int mf_skm_stream_finish(mf_skm_stream *S, uint64_t n_occ, uint64_t n_bases, int thr, uint64_t *n_all, mf_table **out);
void mf_skm_stream_free(mf_skm_stream *S);
int mf_count_plan(mf_ctx *ctx, uint64_t n_occ, uint64_t n_reads, uint64_t n_bases, int k, int min_len, std::vector<int> &lv, std::vector<int> &slv, bool &assembled, int &B);
int mf_mask_reads(mf_ctx *ctx, const uint64_t *d_offsets, uint64_t n_reads, uint64_t n_bases, int k, int min_len, uint32_t *vmask, unsigned long long *d_nocc);
int mf_upload_range(mf_ctx *ctx, int fd, size_t file_off, size_t len, uint8_t *d_dst, hipStream_t up);
int mf_upload_mem(mf_ctx *ctx, const uint8_t *mem, size_t len, uint8_t *d_dst, hipStream_t up);
std::mutex &mf_upload_turn(mf_ctx *ctx);
int mf_upload_pool(mf_ctx *ctx);
int mf_dparse_device(mf_ctx *ctx, const char *path, uint8_t *d_raw, size_t room, size_t n, int fmt, int qoff, mf_buf<uint8_t> &bases, mf_buf<uint64_t> &offsets, uint64_t *n_reads, uint64_t *n_bases);

#define ST_CHUNK ((size_t)256 << 10)      // the device parser's chunk (DP_CHUNK): the buffers hold whole chunks + 64 bytes
// (where the pieces are cut and what the sample holds is host work on bytes nobody has checked: st_record_start, st_plan_pieces, st_sample live in
// mf_parse.h and run under ASan + UBSan in the CPU suite -- tests/host/parse_harness.cpp "stream", tests/test_host_sanitized_cpu.py)
struct st_slot { mf_buf<uint8_t> raw; int piece = -1; bool ready = false; };

// 0: done (*out), 1: not this way (nothing produced), < 0: error
int mf_count_files_streamed(mf_ctx *ctx, const char *const *files, int nfiles, int k, int min_read_len, int thr, uint64_t *n_all, mf_table **out) {
    if (!ctx->opt_stream_count || !ctx->opt_device_parse || !ctx->opt_skm || k < MF_SKM_MIN_K || k > 31 || ctx->own_world > 1 || ctx->opt_l1_bits >= 0 ||
        ctx->opt_union_samples > 0 || min_read_len > 0 || nfiles < 1 || ctx->opt_skm_dyn == 0)
        return 1;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    // ---- the files: plain FASTA or plain FASTQ, all of one kind
    int fmt = 0;
    std::vector<int> fds; std::vector<size_t> sizes; size_t total = 0;
    struct closer { std::vector<int> &f; ~closer() { for (int d : f) if (d >= 0) close(d); } } close_all{fds};
    for (int i = 0; i < nfiles; i++) {
        const std::string p(files[i]);
        int f = 0;
        if (ends_with_nocase(p, ".fastq") || ends_with_nocase(p, ".fq")) f = 2;
        else if (ends_with_nocase(p, ".fasta") || ends_with_nocase(p, ".fa") || ends_with_nocase(p, ".fn") || ends_with_nocase(p, ".fna")) f = 1;
        if (!f || (fmt && f != fmt)) return 1;
        fmt = f;
        const int fd = open(files[i], O_RDONLY);
        if (fd < 0) return 1;                                  // (the whole-file path says which file and why)
        fds.push_back(fd);
        struct stat sb;
        if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) return 1;
        sizes.push_back((size_t)sb.st_size); total += (size_t)sb.st_size;
        if ((size_t)sb.st_size < 2 * ST_SAMPLE) return 1;      // (every file gives sample chunks)
    }
    if (total < (size_t)std::max<int64_t>(ctx->opt_stream_count_min, 1 << 20)) return 1;
    MF_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (!ctx->up_stream) {
        hipStream_t s2 = nullptr;
        if (hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return 1; }
        ctx->up_stream = (void *)s2;
    }
    hipStream_t up = (hipStream_t)ctx->up_stream;
    const double t01 = now();
    // (a fresh process: the pinned staging chunks take 30 ms to make -- beside the border search and the sample, not in front of them)
    int pool_rc = 0;
    std::thread pool_thread([&]() { (void)hipSetDevice(ctx->device); pool_rc = mf_upload_pool(ctx); });
    struct joiner { std::thread &t; ~joiner() { if (t.joinable()) t.join(); } } join_pool{pool_thread};
    // ---- FASTQ: the quality offset of every file from its head (ReadersUtils.java:63-77), on the host
    std::vector<int> qoff((size_t)nfiles, 64);
    if (fmt == 2)
        for (int i = 0; i < nfiles; i++) {
            std::vector<char> head(std::min<size_t>(sizes[(size_t)i], (size_t)4 << 20));
            if (pread(fds[(size_t)i], head.data(), head.size(), 0) != (ssize_t)head.size()) return 1;
            // (the last record of the head may be cut: the sniffing reads whole lines only as far as it needs -- 1000 records)
            size_t cut = head.size(); while (cut && head[cut - 1] != '\n') cut--;
            const size_t lr = st_last_record_start(head.data(), cut, 2);
            if (lr == (size_t)-1) return 1;
            read_batch tmp;
            const int q = parse_fastq_pass(head.data(), lr, files[i], 0, 0, tmp);
            if (q < 0) return 1;
            qoff[(size_t)i] = q;
        }
    // ---- pieces: cut at record borders
    const size_t P = (size_t)std::max<int64_t>(ctx->opt_stream_count_piece, (int64_t)ST_WINDOW);        // (tests go down to 1 MB)
    std::vector<st_piece> pieces;
    size_t piece_max = 0;
    if (st_plan_pieces(fds, sizes, fmt, P, pieces, &piece_max) != 0) return 1;   // (records longer than the search window: assembled sequences, not reads)
    if (pieces.size() < 2) return 1;
    const double t02 = now();
    // ---- the sample: chunks spread evenly over the files, trimmed to whole records, one after the other in a host buffer
    // (a 128th of the bytes, 64 MB at most: ~5000 sampled records per digit region at 1024 regions, and the regions get four standard deviations + an eighth)
    const size_t n_chunks = std::min<size_t>(std::min<size_t>(std::max<size_t>(total / 128 / ST_SAMPLE, 64), 512), piece_max / ST_SAMPLE);      // (it goes up into a piece's buffer)
    std::unique_ptr<char, void (*)(void *)> sample_mem((char *)malloc(n_chunks * ST_SAMPLE), free);      // (not a vector: 64 MB of zeroes first cost 15 ms)
    char *const sample = sample_mem.get();
    if (!sample) return 1;
    size_t s_bytes = 0;
    if (st_sample(fds, sizes, total, fmt, n_chunks, std::max(1, std::min(ctx->host_threads, 16)), sample, &s_bytes) != 0) return 1;
    if (s_bytes < ST_SAMPLE) return 1;
    const double scale = (double)total / (double)s_bytes;
    // ---- buffers: the sample's text and a ring of three pieces
    const double t05 = now();
    const size_t room = (piece_max + ST_CHUNK - 1) / ST_CHUNK * ST_CHUNK + 64;
    const int NS = 3;
    st_slot slot[NS];
    for (int j = 0; j < NS; j++) if (slot[j].raw.alloc(ctx, room) != MF_OK) return 1;
    const double t07 = now();
    pool_thread.join();
    if (pool_rc != 0) return 1;
    const double t1 = now();
    std::mutex mu; std::condition_variable cv;
    std::atomic<int> stop{0};
    int up_rc = 0;                                       // (under mu)
    bool sample_up = false;
    // the sample goes up first, into slot 0's buffer (free until piece 0 has gone up behind it: the producer waits for the parse of the sample)
    bool sample_parsed = false;
    // sub-pieces: what one worker reads and copies at a time (the staging chunks of mf_dparse.hip's uploader: two per worker, pinned)
    const size_t SUB = (size_t)std::max<int64_t>(ctx->opt_device_parse_piece, 1 << 16);
    const int W = (int)std::min<int64_t>(std::max<int64_t>(ctx->opt_device_parse_threads, 1), std::max(ctx->host_threads, 1));
    std::vector<size_t> first_sub(pieces.size() + 1, 0);
    for (size_t i = 0; i < pieces.size(); i++) first_sub[i + 1] = first_sub[i] + (pieces[i].len + SUB - 1) / SUB;
    std::vector<size_t> done_sub(pieces.size(), 0);      // (under mu)
    size_t assigned = 0;                                 // pieces that have got their slot, in order (under mu)
    std::atomic<size_t> next_sub{0};
    std::thread producer([&]() {
        (void)hipSetDevice(ctx->device);
        std::lock_guard<std::mutex> turn(mf_upload_turn(ctx));        // (one upload at a time per device, mf_dparse.hip)
        int rc = mf_upload_mem(ctx, (const uint8_t *)sample, s_bytes, slot[0].raw.p, up);
        { std::lock_guard<std::mutex> g(mu); sample_up = true; if (rc != 0) up_rc = rc < 0 ? rc : -1; }
        cv.notify_all();
        if (rc != 0) return;
        // ONE run of workers over all the pieces (a run per piece -- threads started, events made, the last sub-pieces waited for -- kept PCIe busy for
        // 0.8 of the time): a worker takes the next sub-piece of the whole library, waits until its piece has a slot of the ring, reads it into a staging
        // chunk and sends it; a piece is ready when all its sub-pieces have arrived, which a worker notices when it reuses a chunk or has to wait
        std::vector<std::thread> th;
        for (int w = 0; w < W; w++)
            th.emplace_back([&, w]() {
                (void)hipSetDevice(ctx->device);
                uint8_t *pin[2] = {(uint8_t *)ctx->up_pool + (size_t)(2 * w) * SUB, (uint8_t *)ctx->up_pool + (size_t)(2 * w + 1) * SUB};
                hipEvent_t ev[2]; long busy[2] = {-1, -1};                         // (the piece whose sub-piece is on its way from that chunk)
                (void)hipEventCreateWithFlags(&ev[0], hipEventDisableTiming); (void)hipEventCreateWithFlags(&ev[1], hipEventDisableTiming);
                auto landed = [&](int c) {
                    if (busy[c] < 0) return;
                    const bool ok = hipEventSynchronize(ev[c]) == hipSuccess;
                    const size_t i = (size_t)busy[c]; busy[c] = -1;
                    std::lock_guard<std::mutex> g(mu);
                    if (!ok) { up_rc = -1; (void)hipGetLastError(); }
                    if (++done_sub[i] == first_sub[i + 1] - first_sub[i]) slot[(i + 1) % NS].ready = true;
                    cv.notify_all();
                };
                int cur = 0;
                for (;;) {
                    const size_t sidx = next_sub.fetch_add(1);
                    if (sidx >= first_sub.back() || stop.load()) break;
                    const size_t i = (size_t)(std::upper_bound(first_sub.begin(), first_sub.end(), sidx) - first_sub.begin()) - 1;
                    const int j = (int)((i + 1) % NS);           // (piece 0 into slot 1: slot 0 holds the sample until it has been parsed)
                    bool go = false;
                    {
                        std::unique_lock<std::mutex> g(mu);
                        auto can = [&]() { return stop.load() || up_rc != 0 || i < assigned || (i == assigned && slot[j].piece < 0 && (j != 0 || sample_parsed)); };
                        if (!can()) {
                            g.unlock(); landed(0); landed(1); g.lock();           // (what this worker has sent may be what the consumer is waiting for)
                            cv.wait(g, can);
                        }
                        if (!stop.load() && up_rc == 0) {
                            if (i == assigned) { slot[j].piece = (int)i; slot[j].ready = false; assigned++; cv.notify_all(); }
                            go = true;
                        }
                    }
                    if (!go) break;
                    landed(cur);
                    const size_t lo = (sidx - first_sub[i]) * SUB, len = std::min(SUB, pieces[i].len - lo);
                    size_t got = 0;
                    while (got < len) { const ssize_t r = pread(fds[(size_t)pieces[i].file], pin[cur] + got, len - got, (off_t)(pieces[i].off + lo + got)); if (r <= 0) break; got += (size_t)r; }
                    if (got != len || hipMemcpyAsync(slot[j].raw.p + lo, pin[cur], len, hipMemcpyHostToDevice, up) != hipSuccess || hipEventRecord(ev[cur], up) != hipSuccess) {
                        (void)hipGetLastError();
                        std::lock_guard<std::mutex> g(mu); up_rc = -1; cv.notify_all();
                        break;
                    }
                    busy[cur] = (long)i;
                    cur ^= 1;
                }
                landed(0); landed(1);
                (void)hipEventDestroy(ev[0]); (void)hipEventDestroy(ev[1]);
            });
        for (auto &x : th) x.join();
    });
    // from here on the producer must be stopped and joined on every way out
    mf_skm_stream *S = nullptr;
    int rc = MF_OK;
    uint64_t n_reads = 0, n_bases = 0;
    mf_buf<unsigned long long> d_occ;
    const char *why = "";
    auto body = [&]() -> int {
        MF_TRY(d_occ.alloc(ctx, 2));
        MF_HIP(hipMemsetAsync(d_occ.p, 0, 16, st));
        {
            std::unique_lock<std::mutex> g(mu);
            cv.wait(g, [&]() { return sample_up; });
            if (up_rc != 0) { why = " (a piece could not be read or copied)"; return 1; }
        }
        // ---- the sample: parse, mask, plan, regions
        {
            mf_buf<uint8_t> sb; mf_buf<uint64_t> so; uint64_t snr = 0, snb = 0;
            const int prc = mf_dparse_device(ctx, files[0], slot[0].raw.p, room, s_bytes, fmt, qoff[0], sb, so, &snr, &snb);
            { std::lock_guard<std::mutex> g(mu); sample_parsed = true; }
            cv.notify_all();
            if (prc != 0) { why = " (the device parser is not sure about the sample)"; return prc; }
            if (!snr || !snb) return 1;
            mf_buf<uint32_t> vm; MF_TRY(vm.alloc(ctx, (snb + 31) / 32));
            MF_TRY(mf_mask_reads(ctx, so.p, snr, snb, k, min_read_len, vm.p, d_occ.p + 1));
            unsigned long long s_occ = 0;
            MF_HIP(hipMemcpyAsync(&s_occ, d_occ.p + 1, 8, hipMemcpyDeviceToHost, st));
            MF_HIP(hipStreamSynchronize(st));
            if (!s_occ) return 1;
            const uint64_t occ_est = (uint64_t)((double)s_occ * scale), reads_est = (uint64_t)((double)snr * scale) + 1, bases_est = (uint64_t)((double)snb * scale) + 1;
            std::vector<int> lv, slv; bool assembled = false; int B = 0;
            MF_TRY(mf_count_plan(ctx, occ_est, reads_est, bases_est, k, min_read_len, lv, slv, assembled, B));
            if (assembled) { why = " (long sequences)"; return 1; }
            const int brc = mf_skm_stream_begin(ctx, k, slv, true, 0, sb.p, snb, vm.p, (snb + 31) / 32, scale, &S);
            if (brc != MF_OK) { why = " (a plan without a split level, or no room)"; return brc == MF_SKM_FALLBACK ? 1 : brc; }
        }
        // ---- the pieces
        for (size_t i = 0; i < pieces.size(); i++) {
            const int j = (int)((i + 1) % NS);
            {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&]() { return up_rc != 0 || (slot[j].piece == (int)i && slot[j].ready); });
                if (up_rc != 0) { why = " (a piece could not be read or copied)"; return 1; }
            }
            mf_buf<uint8_t> pb; mf_buf<uint64_t> po; uint64_t nr = 0, nb = 0;
            const int prc = mf_dparse_device(ctx, files[pieces[i].file], slot[j].raw.p, room, pieces[i].len, fmt, qoff[(size_t)pieces[i].file], pb, po, &nr, &nb);
            { std::lock_guard<std::mutex> g(mu); slot[j].piece = -1; slot[j].ready = false; }       // (the parser has synchronised: the text is not needed any more)
            cv.notify_all();
            if (prc != 0) { why = " (the device parser is not sure about a piece)"; return prc; }
            if (!nr || !nb) continue;
            mf_buf<uint32_t> vm; MF_TRY(vm.alloc(ctx, (nb + 31) / 32));
            MF_TRY(mf_mask_reads(ctx, po.p, nr, nb, k, min_read_len, vm.p, d_occ.p));
            const int src = mf_skm_stream_piece(S, pb.p, nb, vm.p, (nb + 31) / 32);
            if (src != MF_OK) return src == MF_SKM_FALLBACK ? 1 : src;
            n_reads += nr; n_bases += nb;
            // (pb, po, vm go back to the arena here; whatever takes their place is written by later work of this same stream)
        }
        return MF_OK;
    };
    rc = body();
    { std::lock_guard<std::mutex> g(mu); stop = 1; }
    cv.notify_all();
    producer.join();
    const double t2 = now();
    if (rc == MF_OK) {
        unsigned long long n_occ = 0;
        if (hipMemcpyAsync(&n_occ, d_occ.p, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) rc = mf_set_error("streamed count: %s", hipGetErrorString(hipGetLastError()));
        else if (!n_occ) rc = 1;
        else {
            for (int j = 0; j < NS; j++) slot[j].raw.reset();
            rc = mf_skm_stream_finish(S, n_occ, n_bases, thr, n_all, out);
            if (rc == MF_SKM_FALLBACK) { rc = 1; why = " (a digit region too small, or the count behind level 1 did not fit)"; }
        }
    } else (void)hipStreamSynchronize(st);
    if (S) mf_skm_stream_free(S);
    if (rc == MF_OK) ctx->n_streamed++; else if (rc == 1) ctx->n_stream_stepped_back++;
    static const bool env = getenv("MF_IO_TIMING") != nullptr;
    if (env || ctx->opt_verbose)
        fprintf(stderr, "[mf] streamed count (%d file(s), %.2f GB, %zu pieces, sample %.1f MB): %s%s; files + stream %.3f s, borders %.3f s, sample %.3f s (+ buffers %.3f s, pinned chunks %.3f s), upload || parse || scatter %.3f s, rest of the count %.3f s; %llu reads\n",
                nfiles, total / 1e9, pieces.size(), s_bytes / 1e6, rc == MF_OK ? "done" : (rc == 1 ? "stepped back to whole files" : "failed"), why, t01 - t0, t02 - t01, t05 - t02, t07 - t05, t1 - t07, t2 - t1, now() - t2, (unsigned long long)n_reads);
    return rc;
}

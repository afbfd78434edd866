// Sanitizer harness of the host-side parsers (metafast_amd/csrc/mf_parse.h, the code mf_io.hip runs on files it has never seen):
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all parse_harness.cpp -o parse_harness -lz -ldl -lpthread
//   parse_harness reads <file> <threads> [dump]   parse_reads_file (by extension: FASTA / FASTQ / .gz / .bz2 / .binq); dump = offsets + bases, raw
//   parse_harness cuts <file> <fmt 1|2> <piece> <slack>   the streaming reader's record cutter at every piece boundary
//   parse_harness comps <file>                     the header walk of a components.bin
//   parse_harness stream <fmt> <piece> <chunks> <sample out> <file>...   the streamed count's piece plan and sample (mf_stream.hip)
// Exit code 0 = parsed, 1 = rejected with a message (both fine); anything else is a finding (ASAN_OPTIONS=exitcode=99).
// Test infrastructure (tests/test_host_sanitized_cpu.py); CPU only -- nothing here touches a GPU.
#include "../../metafast_amd/csrc/mf_parse.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[1024] = "";
int mf_set_error(const char *fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
    return MF_ERR;
}
extern "C" const char *mf_last_error(void) { return g_err; }

static uint64_t fnv(uint64_t h, const void *p, size_t n) { const uint8_t *q = (const uint8_t *)p; for (size_t i = 0; i < n; i++) { h ^= q[i]; h *= 0x100000001B3ull; } return h; }

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: parse_harness reads|cuts|comps <file> ...\n"); return 2; }
    const std::string mode = argv[1];
    const char *path = argv[2];
    if (mode == "reads") {
        const int threads = argc > 3 ? atoi(argv[3]) : 4;
        std::vector<read_batch> parts;
        if (parse_reads_file(path, threads, parts) < 0) { printf("rejected: %s\n", mf_last_error()); return 1; }
        uint64_t nr = 0, nb = 0, h = 0xCBF29CE484222325ull;
        FILE *dump = argc > 4 ? fopen(argv[4], "wb") : nullptr;
        std::vector<uint64_t> offs(1, 0);
        for (auto &rb : parts) {
            for (size_t i = 1; i < rb.offsets.size(); i++) { offs.push_back(nb + rb.offsets[i]); const uint64_t len = rb.offsets[i] - rb.offsets[i - 1]; h = fnv(h, &len, 8); }
            h = fnv(h, rb.bases.data(), rb.bases.size());
            nr += rb.offsets.size() - 1; nb += rb.bases.size();
        }
        if (dump) {
            const uint64_t hd[2] = {nr, nb};
            fwrite(hd, 8, 2, dump); fwrite(offs.data(), 8, offs.size(), dump);
            for (auto &rb : parts) if (rb.bases.size()) fwrite(rb.bases.data(), 1, rb.bases.size(), dump);
            fclose(dump);
        }
        printf("reads %llu bases %llu fnv %016llx\n", (unsigned long long)nr, (unsigned long long)nb, (unsigned long long)h);
        return 0;
    }
    if (mode == "cuts") {
        const int fmt = argc > 3 ? atoi(argv[3]) : 1;
        const size_t piece = argc > 4 ? (size_t)atoll(argv[4]) : 4096, slack = argc > 5 ? (size_t)atoll(argv[5]) : 1024;
        raw_file buf;
        if (read_file_parallel(path, buf, 2) < 0) { printf("rejected: %s\n", mf_last_error()); return 1; }
        // the windows stream_file_to_device hands the cutter: bytes [i*piece - 1, (i+1)*piece + slack) of the file
        uint64_t h = 0xCBF29CE484222325ull, cuts = 0;
        const size_t fsize = buf.size();
        for (size_t i = 0; i * piece < fsize; i++) {
            const size_t lo = i ? i * piece - 1 : 0, hi = std::min(fsize, (i + 1) * piece + slack), n = hi - lo;
            const size_t s0 = sr_record_start(buf.data() + lo, n, i ? 1 : 0, i == 0, fmt);
            const size_t from = (i + 1) * piece - lo;
            const size_t e0 = from < n ? sr_record_start(buf.data() + lo, n, from, false, fmt) : n;
            h = fnv(h, &s0, sizeof s0); h = fnv(h, &e0, sizeof e0); cuts++;
            if (s0 > n || e0 > n) { printf("cut beyond the window\n"); return 3; }
        }
        printf("cuts %llu fnv %016llx\n", (unsigned long long)cuts, (unsigned long long)h);
        return 0;
    }
    if (mode == "stream") {
        // parse_harness stream <fmt 1|2> <piece bytes> <chunks> <sample out> <file> [<file> ...]: the piece plan and the sample of the streamed count
        if (argc < 7) { fprintf(stderr, "usage: parse_harness stream <fmt> <piece> <chunks> <sample out> <file>...\n"); return 2; }
        const int fmt = atoi(argv[2]);
        const size_t P = (size_t)atoll(argv[3]), n_chunks = (size_t)atoll(argv[4]);
        std::vector<int> fds; std::vector<size_t> sizes; size_t total = 0;
        for (int i = 6; i < argc; i++) {
            const int fd = open(argv[i], O_RDONLY);
            struct stat sb;
            if (fd < 0 || fstat(fd, &sb) != 0) { printf("rejected: can't open %s\n", argv[i]); return 1; }
            fds.push_back(fd); sizes.push_back((size_t)sb.st_size); total += (size_t)sb.st_size;
        }
        std::vector<st_piece> pieces; size_t piece_max = 0;
        if (st_plan_pieces(fds, sizes, fmt, P, pieces, &piece_max) != 0) { printf("rejected: no piece plan\n"); return 1; }
        for (auto &pc : pieces) printf("piece %d %zu %zu\n", pc.file, pc.off, pc.len);
        std::vector<char> sample(n_chunks * ST_SAMPLE + 1);
        size_t s_bytes = 0;
        if (st_sample(fds, sizes, total, fmt, n_chunks, 4, sample.data(), &s_bytes) != 0) { printf("rejected: no sample\n"); return 1; }
        FILE *f = fopen(argv[5], "wb");
        if (f) { fwrite(sample.data(), 1, s_bytes, f); fclose(f); }
        printf("sample %zu bytes, largest piece %zu\n", s_bytes, piece_max);
        for (int fd : fds) close(fd);
        return 0;
    }
    if (mode == "comps") {
        raw_file buf;
        if (read_file_parallel(path, buf, 2) < 0) { printf("rejected: %s\n", mf_last_error()); return 1; }
        std::vector<uint64_t> sizes, foff, koff; std::vector<int64_t> weights;
        if (comps_walk_headers((const uint8_t *)buf.data(), buf.size(), sizes, weights, foff, koff) < 0) { printf("rejected: %s\n", mf_last_error()); return 1; }
        uint64_t h = 0xCBF29CE484222325ull;
        for (size_t i = 0; i < sizes.size(); i++) {                       // (touch every k-mer the offsets promise)
            if (foff[i] + 8 * sizes[i] > buf.size()) { printf("offsets beyond the file\n"); return 3; }
            h = fnv(h, buf.data() + foff[i], 8 * sizes[i]); h = fnv(h, &weights[i], 8);
        }
        printf("components %zu kmers %llu fnv %016llx\n", sizes.size(), (unsigned long long)(koff.empty() ? 0 : koff.back()), (unsigned long long)h);
        return 0;
    }
    fprintf(stderr, "unknown mode %s\n", mode.c_str());
    return 2;
}

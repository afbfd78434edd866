// metafast_main.cpp -- metafast.sh-compatible command line driver for the MI355X hot path.
//
// Host-side mirror (C++; the image has no JDK) of the reference's tool shells for this path, on top of the C-ABI only
// (include/metafast_hip.h):
//   src/Runner.java:27-30 + itmo!/Runner.java:108-182          tool registry, -t/--tool, -ts/--tools, --version, default matrix-builder
//   itmo!/utils/tool/Tool.java:61-143, 212-214, 318-392        launch options (-w -p -c --force -s -f -v -h), <workDir>/<tool>/, SUCCESS
//   src/tools/KmersCounterMain.java, KmersCounterForManyFilesMain.java, SeqBuilderMain.java, SeqBuilderForManyFilesMain.java,
//   ComponentCutterMain.java, FeaturesCalculatorMain.java, DistanceMatrixCalculatorMain.java, DistanceMatrixBuilderMain.java
// Parameter names, defaults, output file names and the workDir layout are the reference's, and so is the
// step bookkeeping of the Tool framework (in.properties / out.properties / SUCCESS per step, -c/--force/-s/-f, the
// "rewrite them?" prompt, log + logs/log_<ts>, output_description.txt), so that a Java run can continue a workDir this
// program wrote and the other way round.  Errors: message on stderr, exit 1.
#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <functional>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>
#include "../../include/metafast_hip.h"

using std::string;
using std::vector;

// ------------------------------------------------------------------------------------------------ utilities
static bool g_verbose = false;
static FILE *g_logfile = nullptr, *g_logfile2 = nullptr;      // <workDir>/log and <workDir>/logs/log_<ts> (identical)
static std::mutex g_log_mutex;                                // (the per-device workers of a step log too)
static std::atomic<bool> g_workers_active{false};                         // other threads are inside library calls: die() must not run static destructors under them
static void logmsg(const char *level, const char *fmt, ...) {
    char buf[4096];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    std::lock_guard<std::mutex> lock(g_log_mutex);
    bool debug = !strcmp(level, "DEBUG");
    if (!debug || g_verbose) fprintf(stderr, "%s: %s\n", level, buf);
    if (g_logfile || g_logfile2) {                          // "%d{dd-MMM-yy  HH:mm:ss,SSS}  %-5p  %m%n" (Tool.java:700)
        struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts);
        struct tm tmv; localtime_r(&ts.tv_sec, &tmv);
        char d[64]; strftime(d, sizeof d, "%d-%b-%y  %H:%M:%S", &tmv);
        for (FILE *f : {g_logfile, g_logfile2}) if (f) { fprintf(f, "%s,%03ld  %-5s  %s\n", d, ts.tv_nsec / 1000000, level, buf); fflush(f); }
    }
}
[[noreturn]] static void die(const char *fmt, ...) {
    char buf[4096];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    logmsg("ERROR", "%s", buf);
    if (g_workers_active) { fflush(nullptr); _exit(1); }
    exit(1);                                            // Tool.java:450-463: ExecutionFailedException -> exit code 1
}
static void check(int rc) { if (rc < 0) die("%s", mf_last_error()); }
static bool exists(const string &p) { struct stat st; return stat(p.c_str(), &st) == 0; }
static bool is_dir(const string &p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode); }
static void mkdirs(const string &p) {
    string cur;
    for (size_t i = 0; i <= p.size(); i++) {
        if (i == p.size() || p[i] == '/') { if (!cur.empty() && !is_dir(cur) && mkdir(cur.c_str(), 0777) != 0 && errno != EEXIST) die("can't create directory %s", cur.c_str()); }
        if (i < p.size()) cur.push_back(p[i]);
    }
}
static string basename_of(const string &p) { size_t s = p.find_last_of('/'); return s == string::npos ? p : p.substr(s + 1); }
static bool ends_with_ci(const string &s, const string &suf) {
    if (suf.size() > s.size()) return false;
    for (size_t i = 0; i < suf.size(); i++) if (tolower((unsigned char)s[s.size() - suf.size() + i]) != tolower((unsigned char)suf[i])) return false;
    return true;
}
// FileUtils.removeExtension (itmo!/utils/FileUtils.java:199-210): first matching extension only
static string remove_ext(const string &s, std::initializer_list<const char *> exts) {
    for (const char *e : exts) { string x = e[0] == '.' ? e : string(".") + e; if (ends_with_ci(s, x)) return s.substr(0, s.size() - x.size()); }
    return s;
}
// NamedSource.name(): FastaReader.java:22 / FastqReader.java:25
static string library_name(const string &path) {
    string b = basename_of(path);
    if (ends_with_ci(b, ".gz")) b = b.substr(0, b.size() - 3);          // FastaGZReader.java:17, FastqGZReader.java:21
    else if (ends_with_ci(b, ".bz2")) b = b.substr(0, b.size() - 4);    // FastaBZ2Reader.java:20
    if (ends_with_ci(b, ".binq")) return remove_ext(b, {".binq"});         // BinqReader.java:19
    if (ends_with_ci(b, ".fastq") || ends_with_ci(b, ".fq")) return remove_ext(b, {".fastq", ".fq"});
    return remove_ext(b, {".fasta", ".fa", ".fn", ".fna"});
}
static string timestamp() {                                 // Tool.java:664 "yyyy.MM.dd_HH.mm.ss"
    time_t t = time(nullptr); struct tm tmv; localtime_r(&t, &tmv);
    char b[64]; strftime(b, sizeof b, "%Y.%m.%d_%H.%M.%S", &tmv);
    return b;
}
static void touch(const string &p) { FILE *f = fopen(p.c_str(), "w"); if (f) fclose(f); }
static string abspath(const string &p) {                    // File.getAbsolutePath: no normalisation
    if (!p.empty() && p[0] == '/') return p;
    char cwd[4096]; if (!getcwd(cwd, sizeof cwd)) return p;
    return string(cwd) + "/" + p;
}
static string group_digits(uint64_t v) {                    // NumUtils.groupDigits: 1'234'567
    string s = std::to_string(v), o;
    for (size_t i = 0; i < s.size(); i++) { o.push_back(s[i]); size_t r = s.size() - 1 - i; if (r && r % 3 == 0) o.push_back('\''); }
    return o;
}

// ------------------------------------------------------------------------------------------------ argument parsing
struct Args {
    std::map<string, vector<string>> opt;      // canonical long name -> values
    bool has(const string &k) const { return opt.count(k) > 0; }
    string get(const string &k, const string &def = "") const { auto it = opt.find(k); return it == opt.end() || it->second.empty() ? def : it->second[0]; }
    int geti(const string &k, int def) const {
        auto it = opt.find(k); if (it == opt.end() || it->second.empty()) return def;
        char *e; long v = strtol(it->second[0].c_str(), &e, 10); if (*e) die("Can't parse integer value '%s' of option --%s", it->second[0].c_str(), k.c_str());
        return (int)v;
    }
    vector<string> list(const string &k) const { auto it = opt.find(k); return it == opt.end() ? vector<string>() : it->second; }
};
struct OptDef { const char *lng; const char *sht; bool multi; bool flag; };
// every option of the tools on the path (short names as in the reference; note --maximal-bad-frequence in the counters,
// KmersCounterMain.java:40, vs --maximal-bad-frequency elsewhere)
static const OptDef OPTS[] = {
    {"tool", "t", false, false}, {"tools", "ts", false, true}, {"version", "", false, true}, {"work-dir", "w", false, false},
    {"available-processors", "p", false, false}, {"continue", "c", false, true}, {"force", "", false, true},
    {"start", "s", false, false}, {"finish", "f", false, false}, {"verbose", "v", false, true}, {"help", "h", false, true},
    {"help-all", "ha", false, true}, {"memory", "m", false, false},
    {"k", "k", false, false}, {"reads", "i", true, false}, {"k-mers", "i", true, false}, {"sequences", "i", true, false},
    {"maximal-bad-frequence", "b", false, false}, {"maximal-bad-frequency", "b", false, false},
    {"bottom-cut-percent", "bp", false, false}, {"sequence-len", "l", false, false}, {"min-seq-len", "l", false, false},
    {"output-dir", "o", false, false}, {"stats-dir", "", false, false},
    {"min-component-size", "b1", false, false}, {"max-component-size", "b2", false, false}, {"components-file", "cm", false, false},
    {"kmers", "ka", true, false}, {"selected", "", true, false}, {"threshold", "", false, false},
    {"features", "", true, false}, {"without-names", "wn", false, true}, {"matrix-file", "", false, false},
    {"output-format", "", false, false}, {"heatmap-file", "", false, false}, {"newMatrix-file", "", false, false},
    {"without-renumbering", "wr", false, true}, {"colors-file", "col", false, false}, {"invert-colors", "", false, true},
    {"kmers-file", "kf", false, false}, {"output-file", "o", false, false}, {"split", "", false, true}, {"long", "", false, true},
    {"use-reads-for-calculating-features", "", false, true}, {"device", "", false, false}, {"devices", "", false, false},
    {"positiveReads", "pos", true, false}, {"negativeReads", "neg", true, false}, {"filter-kmers", "", true, false}, {"max-thresh", "", false, false},
};
// `ctx_i` says what -i means for the selected tool
static Args parse_args(int argc, char **argv, string *tool_out) {
    // first pass: find the tool (needed to resolve the overloaded short options)
    string tool = "matrix-builder";                                         // src/Runner.java:29
    for (int i = 1; i + 1 < argc; i++) if (!strcmp(argv[i], "-t") || !strcmp(argv[i], "--tool")) tool = argv[i + 1];
    *tool_out = tool;
    auto long_of_short = [&](const string &s) -> string {
        if (s == "i") {
            if (tool == "heatmap-maker") return "matrix-file";                 // HeatMapMakerMain.java:34-36
            if (tool == "seq-builder" || tool == "seq-builder-many" || tool == "kmers-filter") return "k-mers";
            if (tool == "component-cutter") return "sequences";
            return "reads";
        }
        if (s == "b") return (tool == "kmer-counter" || tool == "kmer-counter-many" || tool == "kmers-filter") ? "maximal-bad-frequence" : "maximal-bad-frequency";
        if (s == "l") return (tool == "seq-builder" || tool == "seq-builder-many") ? "sequence-len" : "min-seq-len";
        if (s == "o") return (tool == "view" || tool == "bin2fasta") ? "output-file" : "output-dir";
        if (s == "cf") return "components-file";                           // ViewMain.java:45, BinaryToFasta.java:47
        for (auto &o : OPTS) if (o.sht[0] && s == o.sht) return o.lng;
        return "";
    };
    Args a;
    for (int i = 1; i < argc;) {
        string tok = argv[i];
        // launcher-level options handled by stub.sh in the reference (src/stub.sh:6-19): accepted and ignored
        if (tok == "-ea" || tok.rfind("-X", 0) == 0 || tok.rfind("-agentlib:", 0) == 0) { i++; continue; }
        string name;
        if (tok.rfind("--", 0) == 0) { name = tok.substr(2); if (name == "new-matrix-file") name = "newMatrix-file"; }
        else if (tok.size() > 1 && tok[0] == '-') name = long_of_short(tok.substr(1));
        else die("Unknown argument '%s'", tok.c_str());
        const OptDef *def = nullptr;
        for (auto &o : OPTS) if (name == o.lng) def = &o;
        if (!def) die("Unrecognized option: %s", tok.c_str());
        i++;
        auto &vals = a.opt[name];
        if (def->flag) {                                                    // booleans take an optional true/false (Parameter.java:51-58)
            if (i < argc && (!strcmp(argv[i], "true") || !strcmp(argv[i], "false"))) { vals.push_back(argv[i]); i++; }
            else vals.push_back("true");
            continue;
        }
        if (i >= argc) die("Missing argument for option: %s", tok.c_str());
        if (def->multi) { while (i < argc && !(argv[i][0] == '-' && strlen(argv[i]) > 1 && !isdigit((unsigned char)argv[i][1]))) vals.push_back(argv[i++]); }
        else vals.push_back(argv[i++]);
    }
    return a;
}

// ------------------------------------------------------------------------------------------------ tools
// Devices.  The reference's drivers loop over all libraries in one process (KmersCounterForManyFilesMain.java:80-108,
// SeqBuilderForManyFilesMain.java:82-94, FeaturesCalculatorMain.java:137-162): here `--devices a,b,...` (default: every device the process
// sees; `--device n` = one) names the GPUs, every entry gets a context of its own, and the three per-library steps deal their libraries
// round-robin to one worker thread per entry (library i -> entry i mod D, in every step: what entry d wrote in kmer-counter-many is
// still in ITS HBM when seq-builder-many and features-calculator ask for the file, `file_cache`).  The steps need no exchange -- the
// files are the interchange --, component-cutter joins all libraries (ComponentCutterMain.java:78-114) on entry 0 from the .seq.fasta
// files.  An entry may name a device twice (--devices 0,0: two contexts, two streams on one GPU).
struct Env {
    vector<int> devs;                           // --devices
    vector<mf_ctx *> ctxs;                      // one per entry, made by the first thread that needs it
    string work_dir;
    bool cont = false;
    string start_ts;
};
static thread_local int t_slot = 0;             // the entry of Env::devs the calling thread works for
static void parse_devices(Env &e, const Args &a) {
    if (!e.devs.empty()) return;
    if (a.has("devices")) {
        const string v = a.get("devices");
        size_t i = 0;
        while (i <= v.size()) {
            size_t j = v.find(',', i); if (j == string::npos) j = v.size();
            const string tok = v.substr(i, j - i);
            char *end = nullptr; const long d = strtol(tok.c_str(), &end, 10);
            if (tok.empty() || *end || d < 0) die("Can't parse --devices '%s' (a comma-separated list of device numbers)", v.c_str());
            e.devs.push_back((int)d);
            i = j + 1;
        }
    } else if (a.has("device")) e.devs.push_back(a.geti("device", 0));
    else {
        const int n = std::max(1, mf_device_count());
        for (int d = 0; d < n; d++) e.devs.push_back(d);
        // Two contexts per device where the libraries are small next to its memory: while one library's files are read and written (the page
        // cache takes 9 GB/s, a sample's .kmers.bin is as large as a third of its reads) the device counts the other one's.  A library needs
        // about 8 x its read file in HBM at the peak of its count (reads + two record buffers + lists + table + index); two of the largest
        // must fit in 0.8 of the device.  MF_CONTEXTS_PER_DEVICE=1 (or an explicit --devices / --device) switches it off.
        size_t largest = 0, nlib = 0;
        for (const char *opt : {"reads", "k-mers", "kmers"})
            for (auto &f : a.list(opt)) {
                struct stat st;
                if (stat(f.c_str(), &st) != 0) continue;
                const size_t packed = ends_with_ci(f, ".gz") || ends_with_ci(f, ".bz2") ? 5 : 1;      // (compressed reads: about five times their size once inflated)
                largest = std::max(largest, (size_t)st.st_size * packed); nlib++;
            }
        const char *cpd = getenv("MF_CONTEXTS_PER_DEVICE");
        uint64_t hbm = 0;
        bool two = cpd ? atoi(cpd) >= 2 : false;
        if (!cpd && nlib >= 2 * (size_t)n && mf_device_memory(0, &hbm) == MF_OK && hbm > 0 && (double)largest * 8.0 * 2.0 < 0.8 * (double)hbm) two = true;
        if (two) for (int d = 0; d < n; d++) e.devs.push_back(d);
    }
    e.ctxs.assign(e.devs.size(), nullptr);
}
static mf_ctx *ctx_of(Env &e, const Args &a) {
    parse_devices(e, a);
    mf_ctx *&ctx = e.ctxs[(size_t)t_slot];
    if (!ctx) {
        // (-p/--available-processors: the host-side parsers' threads, shared out over the entries that work side by side)
        const int procs = a.geti("available-processors", (int)sysconf(_SC_NPROCESSORS_ONLN));
        check(mf_ctx_create(e.devs[(size_t)t_slot], std::max(1, procs / (int)e.devs.size()), &ctx));
        // the steps of one run hand their results on through the reference's files; what a step has just written stays in HBM for the
        // step that loads it next (a quarter of the device's memory at most -- shared out over the entries that name the same device;
        // MF_FILE_CACHE=0 switches it off)
        const char *fc = getenv("MF_FILE_CACHE");
        int same = 0; for (int d : e.devs) same += d == e.devs[(size_t)t_slot];
        check(mf_ctx_set_option(ctx, "file_cache", fc ? atoll(fc) : -(int64_t)same));
        if (const char *mo = getenv("MF_OPTIONS")) {                      // library options for A/B runs: MF_OPTIONS=name=value,name=value
            string all(mo); size_t i = 0;
            while (i < all.size()) {
                size_t j = all.find(',', i); if (j == string::npos) j = all.size();
                const string kv = all.substr(i, j - i); const size_t q = kv.find('=');
                if (q != string::npos) check(mf_ctx_set_option(ctx, kv.substr(0, q).c_str(), atoll(kv.c_str() + q + 1)));
                i = j + 1;
            }
        }
    }
    return ctx;
}
// a per-library step: item i is done by the worker of entry i mod D (entry 0 = the calling thread)
static void for_each_library(Env &e, const Args &a, size_t n, const std::function<void(size_t)> &fn) {
    parse_devices(e, a);
    const size_t D = std::min(e.devs.size(), n);
    if (D <= 1) { for (size_t i = 0; i < n; i++) fn(i); return; }
    logmsg("DEBUG", "%zu libraries on %zu device contexts", n, D);
    g_workers_active = true;
    vector<std::thread> th;
    for (size_t d = 1; d < D; d++)
        th.emplace_back([&, d]() {
            t_slot = (int)d;
            if (e.ctxs[d]) check(mf_ctx_bind_thread(e.ctxs[d]));           // (made by the worker of an earlier step: HIP's device is per thread)
            for (size_t i = d; i < n; i += D) fn(i);
            if (e.ctxs[d]) mf_ctx_synchronize(e.ctxs[d]);
        });
    for (size_t i = 0; i < n; i += D) fn(i);
    for (auto &t : th) t.join();
    g_workers_active = false;
}
static vector<const char *> cptrs(const vector<string> &v) { vector<const char *> p; for (auto &s : v) p.push_back(s.c_str()); return p; }
static void check_k(int k) {                                                 // KmersCounterMain.java:66-73
    if (k <= 0) die("The size of k-mer must be at least 1.");
    if (k > 31) die("The size of k-mer must be no more than 31.");
}

// kmer-counter (src/tools/KmersCounterMain.java:65-137)
static string run_kmer_counter(Env &e, const Args &a, const vector<string> &files, int k, int b, const string &out_dir, const string &stats_dir) {
    check_k(k);
    if (files.empty()) die("Mandatory option --reads is not set");
    mf_ctx *ctx = ctx_of(e, a);
    for (auto &f : files) logmsg("INFO", "Loading file %s...", basename_of(f).c_str());
    mf_table *t = nullptr;
    auto fp = cptrs(files);
    // loadReads + printKmers(hm, b, ...) (KmersCounterMain.java:77-99): only the entries with value > b are handed on, so the
    // cut is made inside the counting kernels and the uncut table (> 2^32 entries for a large sample) never exists; the
    // .stat.txt histogram still covers every counted k-mer (the table remembers what the cut dropped)
    uint64_t size = 0;
    check(mf_count_reads_above(ctx, fp.data(), (int)fp.size(), k, 0, b, &t, &size));
    mkdirs(out_dir); mkdirs(stats_dir);
    string name;                                                             // getName :122-137
    if (files.size() == 2) {
        string n1 = library_name(files[0]), n2 = library_name(files[1]);
        auto ends = [](const string &s, const char *x) { return s.size() >= 3 && s.compare(s.size() - 3, 3, x) == 0; };
        if ((ends(n1, "_r1") && ends(n2, "_r2")) || (ends(n1, "_R1") && ends(n2, "_R2"))) name = n1.substr(0, n1.size() - 3);
        else name = n1 + "+";
    } else name = library_name(files[0]) + (files.size() > 1 ? "+" : "");
    string out = out_dir + "/" + name + ".kmers.bin", st = stats_dir + "/" + name + ".stat.txt";
    uint64_t good = 0;
    check(mf_table_write_kmers(t, b, out.c_str(), st.c_str(), &good));
    logmsg("INFO", "%s k-mers found, %s (%.1f%%) of them is good (not erroneous)", group_digits(size).c_str(), group_digits(good).c_str(),
           size ? good * 100.0 / size : 0.0);
    if (size == 0) logmsg("WARN", "No k-mers found in reads! Perhaps you reads file is empty or k-mer size is too big");
    else if (good == 0 || good < (uint64_t)(size * 0.03)) logmsg("WARN", "Too few good k-mers were found! Perhaps you should decrease k-mer size or --maximal-bad-frequency value");
    logmsg("INFO", "Good k-mers printed to %s", out.c_str());
    mf_table_destroy(t);
    return out;
}
// kmer-counter-many (src/tools/KmersCounterForManyFilesMain.java:66-108): sort, pair _r1/_r2, one counter per sample
static vector<string> run_kmer_counter_many(Env &e, const Args &a, vector<string> files, int k, int b, const string &wd) {
    if (files.empty()) die("Mandatory option --reads is not set");
    std::sort(files.begin(), files.end());
    string out_dir = a.get("output-dir", wd + "/kmers"), stats_dir = a.get("stats-dir", wd + "/stats");
    mkdirs(wd + "/sub-counter");
    auto ends = [](const string &s, const char *x) { return s.size() >= 3 && s.compare(s.size() - 3, 3, x) == 0; };
    vector<vector<string>> libs;                                             // :80-108, one entry per library
    for (size_t i = 0; i < files.size();) {
        string n = library_name(files[i]);
        bool pair = i + 1 < files.size() && ((ends(n, "_r1") && ends(library_name(files[i + 1]), "_r2")) || (ends(n, "_R1") && ends(library_name(files[i + 1]), "_R2")));
        libs.emplace_back(files.begin() + i, files.begin() + i + (pair ? 2 : 1));
        i += pair ? 2 : 1;
    }
    vector<string> outs(libs.size());
    for_each_library(e, a, libs.size(), [&](size_t i) { outs[i] = run_kmer_counter(e, a, libs[i], k, b, out_dir, stats_dir); });
    return outs;
}
// seq-builder (src/tools/SeqBuilderMain.java:78-160)
static string run_seq_builder(Env &e, const Args &a, const vector<string> &files, int k, int b, int bp, int l, const string &wd, const string &out_dir,
                              bool write_distribution = true) {
    if (files.empty()) die("Mandatory option --k-mers is not set");
    mf_ctx *ctx = ctx_of(e, a);
    mf_table *t = nullptr;
    auto fp = cptrs(files);
    check(mf_table_load_kmers(ctx, fp.data(), (int)fp.size(), b, k, &t));
    if (bp >= 0) {                                                           // bottom-cut-percent :103-115
        uint64_t n = 0; check(mf_table_export(t, -1, nullptr, nullptr, 0, &n));
        vector<uint64_t> keys(n); vector<uint16_t> cnts(n);
        if (n) check(mf_table_export(t, -1, keys.data(), cnts.data(), n, &n));
        vector<uint64_t> stat(1024, 0); uint64_t total = 0;
        for (uint16_t c : cnts) { total += c; stat[c >= 1024 ? 1023 : c]++; }
        uint64_t to_cut = total * (uint64_t)bp / 100, cur = 0;
        logmsg("INFO", "Using bottom cut percent = %d", bp);
        for (int i = 0; i < 1023; i++) { if (cur >= to_cut) { b = i; break; } cur += (uint64_t)i * stat[i]; }
    }
    logmsg("INFO", "Using maximal bad frequency = %d", b);
    mkdirs(wd); mkdirs(out_dir);
    string base = remove_ext(basename_of(files[0]), {".kmers.bin"});
    string fasta = out_dir + "/" + base + (files.size() > 1 ? "+" : "") + ".seq.fasta";
    string distr = wd + "/distribution";
    uint64_t nseq = 0;
    check(mf_build_unitigs(ctx, t, k, b, l, fasta.c_str(), write_distribution ? distr.c_str() : nullptr, &nseq));
    logmsg("INFO", "%s sequences found", group_digits(nseq).c_str());
    if (nseq == 0) logmsg("WARN", "No sequences were found! Perhaps you should decrease --min-seq-len or --maximal-bad-frequency values");
    logmsg("INFO", "Sequences printed to %s", fasta.c_str());
    mf_table_destroy(t);
    return fasta;
}
// seq-builder-many (src/tools/SeqBuilderForManyFilesMain.java:82-94): one seq-builder per k-mers file, all in <wd>/sub-builder.  Every one of
// them writes sub-builder/distribution (SeqBuilderMain.java:98), so the file a run leaves behind is the LAST library's; with several
// device contexts at work only that one writes it.
static vector<string> run_seq_builder_many(Env &e, const Args &a, const vector<string> &kmers, int k, int b, int bp, int l, const string &wd, const string &out_dir) {
    vector<string> fs(kmers.size());
    parse_devices(e, a);
    const bool par = std::min(e.devs.size(), kmers.size()) > 1;
    for_each_library(e, a, kmers.size(), [&](size_t i) { fs[i] = run_seq_builder(e, a, {kmers[i]}, k, b, bp, l, wd + "/sub-builder", out_dir, !par || i + 1 == kmers.size()); });
    return fs;
}
// component-cutter (src/tools/ComponentCutterMain.java:78-114)
// ... with several device contexts (round 6): every entry reads the .seq.fasta files of ITS libraries (library i -> entry i mod W, as in the
// per-library steps: the device that built them), the entries cut the components together -- every entry owns a shard of the cutter table,
// the exchanges run inside the library over a communicator of this process's threads (mf_comm_create_local: slices copied straight into the
// peers' buffers, over xGMI between devices) -- and entry 0 writes components.bin; every entry keeps the components for the features of
// its libraries.  W = the largest power of two <= the entries; MF_REPLICATED_CUTTER=1, k < 20 or a single entry: entry 0 alone, as before.
// Returns false when the entries gave up together (the caller then takes entry 0 alone).
static bool run_component_cutter_sharded(Env &e, const Args &a, const vector<string> &files, int k, int l, int b1, int b2, const string &comp_file, const string &stat, uint64_t *nc) {
    parse_devices(e, a);
    size_t W = 1;
    while (2 * W <= e.devs.size() && 2 * W <= 64) W *= 2;
    if (W < 2 || k < 20 || getenv("MF_REPLICATED_CUTTER")) return false;
    // (entries that share ONE device gain nothing from shards -- two contexts on a GPU: 0.17 s against 0.13 on the benchmark's two libraries --:
    // the cutter is sharded when the entries name at least two devices; MF_SHARDED_CUTTER=1 shards whatever they name: tests on a one-GPU box)
    { bool two = false; for (size_t d = 1; d < W; d++) two |= e.devs[d] != e.devs[0]; if (!two && !getenv("MF_SHARDED_CUTTER")) return false; }
    const int keep_slot = t_slot;
    for (size_t d = 0; d < W; d++) { t_slot = (int)d; ctx_of(e, a); }           // (a context is made by whoever needs it first; its worker binds it below)
    t_slot = keep_slot;
    vector<mf_comm *> comms(W, nullptr);
    check(mf_comm_create_local(e.ctxs.data(), (int)W, comms.data()));
    logmsg("DEBUG", "Cutting components on %zu device contexts (sharded cutter table)", W);
    vector<int> rcs(W, MF_OK); vector<string> errs(W); vector<uint64_t> ncs(W, 0);
    auto work = [&](size_t d) {
        vector<string> mine;
        for (size_t i = d; i < files.size(); i += W) mine.push_back(files[i]);
        auto fp = cptrs(mine);
        rcs[d] = mf_cut_components_sharded_files(comms[d], fp.data(), (int)fp.size(), k, l, b1, b2, comp_file.c_str(), stat.c_str(), &ncs[d]);
        if (rcs[d] < 0) errs[d] = mf_last_error();
    };
    g_workers_active = true;
    vector<std::thread> th;
    for (size_t d = 1; d < W; d++) th.emplace_back([&, d]() { t_slot = (int)d; if (mf_ctx_bind_thread(e.ctxs[d]) < 0) { rcs[d] = MF_ERR; errs[d] = mf_last_error(); } work(d); mf_ctx_synchronize(e.ctxs[d]); });
    work(0);
    for (auto &t : th) t.join();
    g_workers_active = false;
    for (mf_comm *c : comms) mf_comm_destroy(c);
    for (size_t d = 0; d < W; d++)
        if (rcs[d] < 0) {
            if (rcs[d] == MF_ERR_TOGETHER) continue;
            die("%s", errs[d].c_str());                                          // (a file that cannot be read, ...: the reference's message)
        }
    for (size_t d = 0; d < W; d++) if (rcs[d] == MF_ERR_TOGETHER) { logmsg("WARN", "%s -- cutting the components on one device", errs[d].c_str()); return false; }
    *nc = ncs[0];
    return true;
}
static string run_component_cutter(Env &e, const Args &a, const vector<string> &files, int k, int l, int b1, int b2, const string &wd, const string &comp_file) {
    if (files.empty()) die("Mandatory option --sequences is not set");
    {
        mkdirs(wd);
        const string stat = wd + "/components-stat-" + std::to_string(b1) + "-" + std::to_string(b2) + ".txt";
        uint64_t nc = 0;
        logmsg("DEBUG", "Loading sequences from files...");
        if (run_component_cutter_sharded(e, a, files, k, l, b1, b2, comp_file, stat, &nc) && nc > 0) {
            // (no component at all: the run below repeats the step on one device and says what the reference says -- "No sequences were found" or
            // "No components were extracted" --, a case of toy inputs)
            logmsg("INFO", "Searching for components...");
            logmsg("INFO", "Total %s components were found", group_digits(nc).c_str());
            logmsg("INFO", "Components saved to %s", comp_file.c_str());
            return comp_file;
        }
    }
    mf_ctx *ctx = ctx_of(e, a);
    mf_table *t = nullptr;
    auto fp = cptrs(files);
    logmsg("DEBUG", "Loading sequences from files...");
    check(mf_count_reads(ctx, fp.data(), (int)fp.size(), k, l, &t));
    uint64_t size = 0; check(mf_table_stats(t, &size, nullptr));
    if (size == 0) die("No sequences were found in input files! The following steps will be useless");
    logmsg("INFO", "Searching for components...");
    mkdirs(wd);
    string stat = wd + "/components-stat-" + std::to_string(b1) + "-" + std::to_string(b2) + ".txt";
    uint64_t nc = 0;
    check(mf_cut_components(ctx, t, k, b1, b2, comp_file.c_str(), stat.c_str(), &nc));
    logmsg("INFO", "Total %s components were found", group_digits(nc).c_str());
    if (nc == 0) logmsg("WARN", "No components were extracted! Perhaps you should decrease --min-component-size value");
    logmsg("INFO", "Components saved to %s", comp_file.c_str());
    mf_table_destroy(t);
    return comp_file;
}
// features-calculator (src/tools/FeaturesCalculatorMain.java:77-167), k-mers files branch
static vector<string> run_features(Env &e, const Args &a, const string &comp_file, const vector<string> &reads, const vector<string> &kmers,
                                   int k, int thr, const string &wd) {
    if (comp_file.empty()) die("Mandatory option --components-file is not set");
    if (kmers.empty() && reads.empty()) die("No input files: pass reads (-i) or k-mers files (-ka)");
    parse_devices(e, a);
    string out_dir = wd + "/vectors";
    mkdirs(out_dir);
    // --selected (FeaturesCalculatorMain.java:55-57, 113-116): selected = IOUtils.loadKmers(selectedKmers, 0, ...), one table per device context
    const vector<string> sel_files = a.list("selected");
    vector<mf_table *> sel(e.devs.size(), nullptr);
    auto selected = [&]() -> mf_table * {
        if (sel_files.empty()) return nullptr;
        mf_table *&t = sel[(size_t)t_slot];
        if (!t) { auto fp = cptrs(sel_files); check(mf_table_load_kmers(ctx_of(e, a), fp.data(), (int)fp.size(), 0, k, &t)); }
        return t;
    };
    // reads files first, one vector per FILE, then k-mers files (FeaturesCalculatorMain.java:117-162)
    vector<string> vecs(reads.size() + kmers.size());
    for_each_library(e, a, vecs.size(), [&](size_t i) {
        mf_ctx *ctx = ctx_of(e, a);
        if (i < reads.size()) {
            const string &rf = reads[i];
            string base = library_name(rf);
            string vec = out_dir + "/" + base + ".vec", br = out_dir + "/" + base + ".breadth";
            const char *fs[1] = {rf.c_str()};
            check(mf_features_reads_selected(ctx, comp_file.c_str(), fs, 1, k, thr, selected(), vec.c_str(), br.c_str()));
            logmsg("INFO", "Features for file %s printed to %s", basename_of(rf).c_str(), vec.c_str());
            vecs[i] = vec;
        } else {
            const string &kf = kmers[i - reads.size()];
            string base = remove_ext(basename_of(kf), {".kmers.bin"});
            string vec = out_dir + "/" + base + ".vec", br = out_dir + "/" + base + ".breadth";
            check(mf_features_selected(ctx, comp_file.c_str(), kf.c_str(), k, thr, selected(), vec.c_str(), br.c_str()));
            logmsg("INFO", "Features for file %s printed to %s", basename_of(kf).c_str(), vec.c_str());
            vecs[i] = vec;
        }
    });
    for (mf_table *t : sel) if (t) mf_table_destroy(t);
    return vecs;
}
// Double.toString (what Java's "%s" prints for a double): shortest digits that round-trip, decimal notation in [1e-3, 1e7)
static string java_double(double d) {
    if (std::isnan(d)) return "NaN";
    if (std::isinf(d)) return d > 0 ? "Infinity" : "-Infinity";
    if (d == 0) return std::signbit(d) ? "-0.0" : "0.0";
    char buf[64];
    for (int prec = 1; prec <= 17; prec++) { snprintf(buf, sizeof buf, "%.*e", prec - 1, d); if (strtod(buf, nullptr) == d) break; }
    string s(buf);
    bool neg = s[0] == '-';
    if (neg) s = s.substr(1);
    size_t epos = s.find('e');
    int ex = atoi(s.c_str() + epos + 1);
    string digits;
    for (char ch : s.substr(0, epos)) if (ch != '.') digits.push_back(ch);
    string out;
    double ad = std::fabs(d);
    if (ad >= 1e-3 && ad < 1e7) {
        if (ex >= 0) {
            string ip = digits.substr(0, std::min<size_t>(digits.size(), (size_t)ex + 1));
            while ((int)ip.size() < ex + 1) ip.push_back('0');
            out = ip + "." + (digits.size() > (size_t)ex + 1 ? digits.substr(ex + 1) : string("0"));
        } else out = "0." + string((size_t)(-ex - 1), '0') + digits;
    } else out = digits.substr(0, 1) + "." + (digits.size() > 1 ? digits.substr(1) : string("0")) + "E" + std::to_string(ex);
    return neg ? "-" + out : out;
}
// one matrix cell in a Java format string (PrintWriter.printf(format, double), DistanceMatrixCalculatorMain.java:112-116):
// "%s" -> Double.toString, "%[flags][width][.prec]{f,e,g}" -> printf; anything else is refused
static string format_cell(const string &fmt, double v) {
    if (fmt == "%s") return java_double(v);
    size_t i = 0;
    bool ok = fmt.size() >= 2 && fmt[0] == '%';
    for (i = 1; ok && i + 1 < fmt.size(); i++) if (!strchr("0123456789.+- ", fmt[i])) ok = false;
    if (!ok || !strchr("feg", fmt[fmt.size() - 1])) die("Unsupported --output-format '%s' (use %%s or %%[.N]f / e / g)", fmt.c_str());
    char buf[128]; snprintf(buf, sizeof buf, fmt.c_str(), v);
    return buf;
}
// DistanceMatrixCalculatorMain.printMatrix (:91-123): perm == nullptr prints the original order
static void print_matrix(const vector<double> &m, int n, const string &path, const vector<string> *names, const int *perm, const string &fmt) {
    size_t slash = path.find_last_of('/'); if (slash != string::npos) mkdirs(path.substr(0, slash));
    FILE *out = fopen(path.c_str(), "w");
    if (!out) die("Failed to print matrix to %s", path.c_str());
    if (names) { fprintf(out, "#"); for (int i = 0; i < n; i++) fprintf(out, "\t%s", (*names)[perm ? perm[i] : i].c_str()); fprintf(out, "\n"); }
    for (int i = 0; i < n; i++) {
        if (names) fprintf(out, "%s\t", (*names)[perm ? perm[i] : i].c_str());
        for (int j = 0; j < n; j++) {
            if (j) fprintf(out, "\t");
            fprintf(out, "%s", format_cell(fmt, perm ? m[(size_t)perm[i] * n + perm[j]] : m[(size_t)i * n + j]).c_str());
        }
        fprintf(out, "\n");
    }
    fclose(out);
}
// heatmap-maker, the numeric half (src/tools/HeatMapMakerMain.java:93-145): average-linkage clustering of the samples
// (FullHeatMap.clusterObjects :221-296: O(n^3), the FIRST closest pair in row-major order is merged, the merged node stays
// at the smaller index with the old node on the left) and the matrix renumbered in the dendrogram's leaf order
// (renumber :327-337).  The image itself is not rendered.
struct HNode { int no = -1, left = -1, right = -1; };
static void hm_group(const vector<HNode> &t, int node, vector<int> &out) {
    if (node < 0) return;
    if (t[node].no >= 0) { out.push_back(t[node].no); return; }
    hm_group(t, t[node].left, out); hm_group(t, t[node].right, out);
}
static vector<int> heatmap_order(const vector<double> &m, int n) {
    vector<HNode> t(n);
    vector<int> nodes(n);
    for (int i = 0; i < n; i++) { t[i].no = i; nodes[i] = i; }
    vector<double> dist((size_t)n * n, 0.0);
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) dist[(size_t)i * n + j] = m[(size_t)i * n + j] / 1 / 1;
    auto between = [&](const vector<int> &g1, const vector<int> &g2) {
        if (g1.empty() || g2.empty()) return -1.0;
        double sum = 0;
        for (int a : g1) for (int b : g2) sum += m[(size_t)a * n + b];
        return sum / (double)g1.size() / (double)g2.size();
    };
    int count = n, root = n > 0 ? 0 : -1;
    while (count > 1) {
        double best = 1.7976931348623157e308; int bi = -1, bj = -1;
        for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++)
            if (nodes[i] >= 0 && nodes[j] >= 0 && dist[(size_t)i * n + j] < best) { best = dist[(size_t)i * n + j]; bi = i; bj = j; }
        if (bi < 0 || best < 0) die("Internal error. Wrong minDist index.");
        HNode r; r.left = nodes[bi]; r.right = nodes[bj];
        t.push_back(r); root = (int)t.size() - 1;
        nodes[bi] = root; nodes[bj] = -1;
        vector<int> g1; hm_group(t, root, g1);
        for (int i = 0; i < n; i++) {
            dist[(size_t)i * n + bj] = dist[(size_t)bj * n + i] = -1;
            if (i != bi) { vector<int> g2; hm_group(t, nodes[i], g2); dist[(size_t)i * n + bi] = dist[(size_t)bi * n + i] = between(g1, g2); }
        }
        count--;
    }
    vector<int> perm;
    hm_group(t, root, perm);
    return perm;
}
static string run_heatmap_maker(Env &e, const Args &a, const string &matrix_path, const string &new_matrix_tpl) {
    FILE *fp = fopen(matrix_path.c_str(), "r");
    if (!fp) die("Can't read matrix file %s", matrix_path.c_str());
    vector<vector<string>> rows; char *line = nullptr; size_t cap = 0;
    while (getline(&line, &cap, fp) > 0) {
        vector<string> cells; string cur;
        for (char *q = line; *q && *q != '\n' && *q != '\r'; q++) { if (*q == '\t') { if (!cur.empty()) cells.push_back(cur); cur.clear(); } else cur.push_back(*q); }
        if (!cur.empty()) cells.push_back(cur);
        rows.push_back(cells);
    }
    free(line); fclose(fp);
    if (rows.empty()) die("No data to read in matrix file %s", matrix_path.c_str());
    const size_t fn = rows[0].size();
    if (fn > rows.size()) die("Can't parse matrix, columns' number > rows' number");
    for (size_t i = 0; i < fn; i++) if (rows[i].size() != fn) die("Can't parse matrix, columns' number is different for different rows");
    for (size_t i = fn; i < rows.size(); i++)                                                                                // HeatMapMakerMain.java:204-209 (white space is no token)
        for (auto &cell : rows[i]) if (cell.find_first_not_of(" \t\f") != string::npos) die("Can't parse matrix, too much rows");
    // (an empty first line: the reference indexes dataArray[0][0] of a 0 x 0 array there, :214 -- an exception, i.e. a failed run)
    if (fn == 0) die("Can't parse matrix, the first line of %s is empty", matrix_path.c_str());
    const bool with_names = rows[0][0] == "#";
    const int n = (int)fn - (with_names ? 1 : 0);
    vector<string> names;
    vector<double> m((size_t)n * n);
    for (int i = 0; i < n; i++) {
        if (with_names) names.push_back(rows[0][i + 1]);
        for (int j = 0; j < n; j++) {
            const string &cell = rows[i + (with_names ? 1 : 0)][j + (with_names ? 1 : 0)];
            char *end; m[(size_t)i * n + j] = strtod(cell.c_str(), &end);
            if (*end) die("Can't parse matrix, '%s' is not a number", cell.c_str());
        }
    }
    if (a.get("without-renumbering", "false") == "true") return matrix_path;
    vector<int> perm = heatmap_order(m, n);
    string path = new_matrix_tpl.empty() ? remove_ext(matrix_path, {".txt"}) + "_renumbered.txt" : new_matrix_tpl;
    size_t p = path.find("$DT"); if (p != string::npos) path.replace(p, 3, e.start_ts);
    print_matrix(m, n, path, with_names ? &names : nullptr, perm.data(), a.get("output-format", "%.4f"));
    logmsg("INFO", "Renumbered matrix saved to %s", path.c_str());
    return path;
}

// ---- view / bin2fasta: text dumps of the binary files (src/tools/ViewMain.java:64-131, src/tools/BinaryToFasta.java:74-170).
// Host-only.  The reference prints the k-mers of a .kmers.bin in its hash map's iteration order; here: file order.
static string kmer_string(uint64_t km, int k) {             // ShortKmer.toString (itmo!/dna/kmers/ShortKmer.java:153-160)
    string s((size_t)k, 'A');
    for (int i = 0; i < k; i++) s[(size_t)i] = "AGCT"[(km >> (2 * (k - 1 - i))) & 3u];
    return s;
}
static uint64_t be_read(const unsigned char *p, int n) { uint64_t v = 0; for (int i = 0; i < n; i++) v = (v << 8) | p[i]; return v; }
static vector<unsigned char> slurp(const string &path) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) die("Can't read file %s", path.c_str());
    vector<unsigned char> b; unsigned char buf[1 << 16]; size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + n);
    fclose(f);
    return b;
}
struct HostComp { int64_t weight; vector<uint64_t> kmers; };
static vector<HostComp> read_components(const string &path) {       // ConnectedComponent.loadComponents :95-122
    vector<unsigned char> b = slurp(path);
    if (b.size() < 4) die("Can't load components from %s", path.c_str());
    size_t pos = 0; uint64_t n = be_read(&b[0], 4); pos = 4;
    // (every component has its 12-byte header at least: a count the file cannot hold is a wrong file, not 2^32 empty components to
    // allocate for -- found by the sanitizer build, tests/test_host_sanitized_cpu.py)
    if (4 + 12 * n > b.size()) die("Can't load components from %s", path.c_str());
    vector<HostComp> cs((size_t)n);
    for (auto &cp : cs) {
        if (pos + 12 > b.size()) die("Can't load components from %s", path.c_str());
        uint64_t sz = be_read(&b[pos], 4); cp.weight = (int64_t)be_read(&b[pos + 4], 8); pos += 12;
        if (pos + 8 * sz > b.size()) die("Can't load components from %s", path.c_str());
        cp.kmers.resize((size_t)sz);
        for (auto &km : cp.kmers) { km = be_read(&b[pos], 8); pos += 8; }
    }
    return cs;
}
static FILE *open_out(const string &path) { if (path.empty()) return stdout; FILE *f = fopen(path.c_str(), "w"); if (!f) die("Couldn't open output file"); return f; }
static void close_out(FILE *f) { if (f != stdout) fclose(f); else fflush(f); }
static void run_view(const Args &a, int k) {
    const string kf = a.get("kmers-file"), cf = a.get("components-file");
    if (kf.empty() && cf.empty()) { logmsg("WARN", "No input file is selected  --->  no data to display!"); return; }
    FILE *out = open_out(a.get("output-file"));
    if (!kf.empty()) {
        const int rec = a.get("long", "false") == "true" ? 16 : 10;          // IOUtils.loadLongKmers / loadKmers
        vector<unsigned char> b = slurp(kf);
        fprintf(out, "Kmer\tCount\n");
        for (size_t p = 0; p + rec <= b.size(); p += rec)
            fprintf(out, "%s\t%lld\n", kmer_string(be_read(&b[p], 8), k).c_str(), rec == 10 ? (long long)(int16_t)be_read(&b[p + 8], 2) : (long long)be_read(&b[p + 8], 8));
    }
    if (!cf.empty()) {
        vector<HostComp> cs = read_components(cf);
        logmsg("INFO", "%zu components loaded from %s", cs.size(), cf.c_str());
        fprintf(out, "%zu components:\n", cs.size());
        for (size_t i = 0; i < cs.size(); i++) {
            fprintf(out, "Component %zu, size = %zu kmers, weight = %lld. Kmers:\n", i + 1, cs[i].kmers.size(), (long long)cs[i].weight);
            for (uint64_t km : cs[i].kmers) fprintf(out, "%s\n", kmer_string(km, k).c_str());
            fprintf(out, "\n");
        }
    }
    close_out(out);
}
static void run_bin2fasta(const Args &a, int k) {
    const string kf = a.get("kmers-file"), cf = a.get("components-file"), prefix = a.get("output-file");
    if (kf.empty() && cf.empty()) { logmsg("WARN", "No input file is selected  --->  no data to display!"); return; }
    if (!prefix.empty()) { size_t s = prefix.find_last_of('/'); if (s != string::npos) mkdirs(prefix.substr(0, s)); }
    if (!kf.empty()) {
        FILE *out = open_out(prefix.empty() ? "" : prefix + ".fasta");
        vector<unsigned char> b = slurp(kf);
        size_t i = 1;
        for (size_t p = 0; p + 10 <= b.size(); p += 10, i++) fprintf(out, ">%zu\n%s\n", i, kmer_string(be_read(&b[p], 8), k).c_str());
        close_out(out);
    }
    if (!cf.empty()) {
        vector<HostComp> cs = read_components(cf);
        logmsg("INFO", "%zu components loaded from %s", cs.size(), cf.c_str());
        if (a.get("split", "false") == "true") {
            for (size_t i = 0; i < cs.size(); i++) {
                FILE *out = open_out(prefix.empty() ? "" : prefix + "_" + std::to_string(i + 1) + ".fasta");
                size_t j = 1;
                for (uint64_t km : cs[i].kmers) fprintf(out, ">%zu\n%s\n", j++, kmer_string(km, k).c_str());
                close_out(out);
            }
        } else {
            FILE *out = open_out(prefix.empty() ? "" : prefix + ".fasta");
            for (size_t i = 0; i < cs.size(); i++) { size_t j = 1; for (uint64_t km : cs[i].kmers) fprintf(out, ">%zu_%zu\n%s\n", i + 1, j++, kmer_string(km, k).c_str()); }
            close_out(out);
        }
    }
}

// dist-matrix-calculator (src/tools/DistanceMatrixCalculatorMain.java:51-123)
static string run_dist_matrix(Env &e, const Args &a, const vector<string> &features, const string &matrix_path_tpl) {
    if (features.empty()) die("Mandatory option --features is not set");
    vector<vector<int64_t>> vs;
    for (auto &f : features) {
        FILE *fp = fopen(f.c_str(), "r");
        if (!fp) die("Failed to read features from %s", f.c_str());
        vector<int64_t> v; char line[256];
        while (fgets(line, sizeof line, fp)) { if (line[0] != '\n' && line[0] != 0) v.push_back((int64_t)strtod(line, nullptr)); }
        fclose(fp);
        vs.push_back(v);
    }
    size_t nc = vs[0].size();
    for (auto &v : vs) if (v.size() != nc) die("feature files have different numbers of components");
    vector<int64_t> flat; for (auto &v : vs) flat.insert(flat.end(), v.begin(), v.end());
    int ns = (int)vs.size();
    vector<double> m((size_t)ns * ns, 0.0);
    check(mf_bray_curtis(flat.data(), ns, (int)nc, m.data()));
    string path = matrix_path_tpl;
    size_t p = path.find("$DT"); if (p != string::npos) path.replace(p, 3, e.start_ts);
    vector<string> names;
    for (auto &f : features) names.push_back(remove_ext(basename_of(f), {"vec"}));
    print_matrix(m, ns, path, a.get("without-names", "false") != "true" ? &names : nullptr, nullptr, a.get("output-format", "%.4f"));
    logmsg("INFO", "Distance matrix printed to %s", path.c_str());
    return path;
}

// ------------------------------------------------------------------------------------------------ step bookkeeping
// Tool.runAsStep (itmo!/utils/tool/Tool.java:318-392) and its property files (:740-960): every tool run, top level or sub-step,
// owns a directory with in.properties (its input parameters, written before it runs), out.properties (its output
// parameters) and SUCCESS.  A finished step is re-used by --continue iff in.properties and SUCCESS exist and every stored
// input equals the current one.  PropertiesConfiguration.save with delimiter parsing disabled: "key = value", one line
// per value of a multi-valued key, null parameters left out, files as absolute paths (Tool.objectToString :950-966).
struct PV {
    string name; vector<string> vals; bool set = true;                       // set == false: null, not written
    PV(string n, vector<string> v) : name(std::move(n)), vals(std::move(v)) {}
    PV(string n, const string &v) : name(std::move(n)), vals{v} {}
    PV(string n, int v) : name(std::move(n)), vals{std::to_string(v)} {}
    static PV null(string n) { PV p(std::move(n), vector<string>()); p.set = false; return p; }
    static PV file(string n, const string &v) { return v.empty() ? null(std::move(n)) : PV(std::move(n), abspath(v)); }
    static PV files(string n, const vector<string> &v) { vector<string> o; for (auto &x : v) o.push_back(abspath(x)); return PV(std::move(n), o); }
    static PV flag(string n, bool v) { return PV(std::move(n), string(v ? "true" : "false")); }
};
typedef std::map<string, vector<string>> Props;
static void props_write(const string &path, const vector<PV> &ps) {
    FILE *f = fopen(path.c_str(), "w");
    if (!f) die("Can't dump configuration to %s", path.c_str());
    for (auto &p : ps) if (p.set) for (auto &v : p.vals) {
        string esc; for (char c : v) { if (c == '\\') esc += "\\\\"; else esc.push_back(c); }
        fprintf(f, "%s = %s\n", p.name.c_str(), esc.c_str());
    }
    fclose(f);
}
static bool props_read(const string &path, Props &out) {
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return false;
    char *line = nullptr; size_t cap = 0;
    while (getline(&line, &cap, f) > 0) {
        string l(line); while (!l.empty() && (l.back() == '\n' || l.back() == '\r')) l.pop_back();
        size_t i = 0; while (i < l.size() && isspace((unsigned char)l[i])) i++;
        if (i == l.size() || l[i] == '#' || l[i] == '!') continue;
        size_t ke = i; while (ke < l.size() && l[ke] != '=' && l[ke] != ':' && !isspace((unsigned char)l[ke])) ke++;
        string key = l.substr(i, ke - i);
        size_t v = ke; while (v < l.size() && isspace((unsigned char)l[v])) v++;
        if (v < l.size() && (l[v] == '=' || l[v] == ':')) { v++; while (v < l.size() && isspace((unsigned char)l[v])) v++; }
        string val; for (size_t j = v; j < l.size(); j++) { if (l[j] == '\\' && j + 1 < l.size()) j++; val.push_back(l[j]); }
        out[key].push_back(val);
    }
    free(line); fclose(f);
    return true;
}
static bool props_equal(const vector<PV> &cur, const Props &stored) {      // (null and an empty list both leave no line)
    for (auto &p : cur) {
        auto it = stored.find(p.name);
        const vector<string> none, &now = p.set ? p.vals : none, &then = it == stored.end() ? none : it->second;
        if (now != then) { logmsg("DEBUG", "Parameter %s changed from last run", p.name.c_str()); return false; }
    }
    return true;
}
static vector<string> props_list(const Props &p, const string &k) { auto it = p.find(k); return it == p.end() ? vector<string>() : it->second; }
// one sub-step of a composite tool (Tool.runAllSteps :485-529): `force` turns true once a step has run, so everything
// after it runs too.  run() does the work and returns the output parameters; load() takes them from out.properties.
template <class RUN, class LOAD>
static void run_as_step(const string &name, const string &dir, const vector<PV> &in, const string &start, bool &force, RUN run, LOAD load) {
    if (name == start) force = true;
    mkdirs(dir);
    const string inp = dir + "/in.properties", outp = dir + "/out.properties", succ = dir + "/SUCCESS";
    const bool f = force || !exists(inp);
    bool can = exists(inp) && exists(succ);
    if (!f && can) { Props stored; can = props_read(inp, stored) && props_equal(in, stored); }
    if (!f && can) {
        logmsg("INFO", "SUCCESS file found for tool %s - loading results...", name.c_str());
        Props out; props_read(outp, out);
        load(out);
        return;
    }
    logmsg("DEBUG", "Running tool %s", name.c_str());
    unlink(succ.c_str()); unlink(outp.c_str());
    props_write(inp, in);
    const vector<PV> out = run();
    props_write(outp, out);
    touch(succ);
    force = true;
}
// output_description.txt in the current directory and in the workDir (DistanceMatrixBuilderMain.java:81-83, 178-200;
// IOUtils.tryToAppendDescription src/io/IOUtils.java:217-231)
static vector<string> g_desc_files;
static void describe(const string &path, const char *msg) {
    for (auto &d : g_desc_files) { FILE *f = fopen(d.c_str(), "a"); if (!f) continue; fprintf(f, "\n%s\n   %s\n", path.c_str(), msg); fclose(f); }
}
static const char *TOOLS_TEXT =
    "kmer-counter\t\tCount k-mers in given reads\n"
    "kmer-counter-many\tCount k-mers in many files (one library = one output)\n"
    "seq-builder\t\tMetagenome De Bruijn graph analysis and sequences building\n"
    "seq-builder-many\tseq-builder for many k-mers files\n"
    "component-cutter\tBuild graph components from sequences\n"
    "features-calculator\tCalculate features values for input reads/k-mers files\n"
    "dist-matrix-calculator\tCalculate the distance matrix using features values\n"
    "heatmap-maker\t\tCluster the samples of a distance matrix and renumber it (no image)\n"
    "kmer-counter-posneg\tCount k-mers for files from two groups independently\n"
    "kmers-filter\t\tFilter k-mers from test set according to known samples\n"
    "view\t\t\tView different binary objects (k-mers files, components)\n"
    "bin2fasta\t\tConverts different binary objects to FASTA format\n"
    "matrix-builder\t\tBuild the distance matrix for input sequences (default tool)\n";

// input parameters of a tool as in.properties lists them: declaration order of the reference's Parameter fields, values
// = what the run will use (defaults filled in).  matrix-builder: its own, then the sub-tools' parameters it does not fix
// (Tool.addSubTool :168-207).
static vector<PV> tool_inputs(const string &tool, const Args &a, const string &wd, const string &ts) {
    auto dt = [&](string p) { size_t q = p.find("$DT"); if (q != string::npos) p.replace(q, 3, ts); return p; };
    auto opt_i = [&](const char *n) { return a.has(n) ? PV(n, a.get(n)) : PV::null(n); };
    auto opt_f = [&](const char *n) { return a.has(n) ? PV::file(n, a.get(n)) : PV::null(n); };
    auto flag = [&](const char *n) { return PV::flag(n, a.get(n, "false") == "true"); };
    vector<PV> v;
    if (tool == "kmer-counter" || tool == "kmer-counter-many") {
        v = {opt_i("k"), PV::files("reads", a.list("reads")), PV("maximal-bad-frequence", a.get("maximal-bad-frequence", "1")),
             PV::file("output-dir", a.get("output-dir", wd + "/kmers")), PV::file("stats-dir", a.get("stats-dir", wd + "/stats"))};
    } else if (tool == "seq-builder" || tool == "seq-builder-many") {
        v = {opt_i("k"), PV::files("k-mers", a.list("k-mers")),
             tool == "seq-builder" ? PV("maximal-bad-frequency", a.get("maximal-bad-frequency", "1")) : opt_i("maximal-bad-frequency"),
             opt_i("bottom-cut-percent"), opt_i("sequence-len"), PV::file("output-dir", a.get("output-dir", wd + "/sequences"))};
    } else if (tool == "component-cutter") {
        v = {opt_i("k"), PV("min-seq-len", a.get("min-seq-len", "100")), PV("min-component-size", a.get("min-component-size", "1000")),
             PV("max-component-size", a.get("max-component-size", "10000")), PV::files("sequences", a.list("sequences")),
             PV::file("components-file", a.get("components-file", wd + "/components.bin"))};
    } else if (tool == "features-calculator") {
        v = {opt_i("k"), opt_f("components-file"), PV::files("reads", a.list("reads")), PV::files("kmers", a.list("kmers")),
             a.has("selected") ? PV::files("selected", a.list("selected")) : PV::null("selected"), PV("threshold", a.get("threshold", "0"))};
    } else if (tool == "dist-matrix-calculator") {
        v = {PV::files("features", a.list("features")), flag("without-names"),
             PV::file("matrix-file", dt(a.get("matrix-file", wd + "/dist_matrix_$DT_original_order.txt"))), PV("output-format", a.get("output-format", "%.4f"))};
    } else if (tool == "heatmap-maker") {
        v = {opt_f("matrix-file"), opt_f("colors-file"), flag("without-renumbering"), opt_f("newMatrix-file"), opt_f("heatmap-file"),
             flag("invert-colors"), PV("output-format", a.get("output-format", "%.4f"))};
    } else if (tool == "matrix-builder") {
        v = {PV("k", a.get("k", "31")), PV::files("reads", a.list("reads")), PV("maximal-bad-frequency", a.get("maximal-bad-frequency", "1")),
             PV("min-seq-len", a.get("min-seq-len", "100")), flag("use-reads-for-calculating-features"),
             PV::file("matrix-file", dt(a.get("matrix-file", wd + "/matrices/dist_matrix_$DT.txt"))),
             PV::file("heatmap-file", dt(a.get("heatmap-file", wd + "/matrices/dist_matrix_$DT_heatmap.png"))),
             PV::file("stats-dir", a.get("stats-dir", wd + "/kmer-counter-many/stats")), opt_i("bottom-cut-percent"),
             PV("min-component-size", a.get("min-component-size", "1000")), PV("max-component-size", a.get("max-component-size", "10000")),
             a.has("selected") ? PV::files("selected", a.list("selected")) : PV::null("selected"), flag("without-names"),
             PV("output-format", a.get("output-format", "%.4f")), opt_f("colors-file"), flag("without-renumbering"), flag("invert-colors")};
    } else if (tool == "kmer-counter-posneg") {
        v = {opt_i("k"), PV::files("positiveReads", a.list("positiveReads")), PV::files("negativeReads", a.list("negativeReads")),
             PV("maximal-bad-frequency", a.get("maximal-bad-frequency", "1")), PV::file("output-dir", a.get("output-dir", wd + "/kmers_posneg"))};
    } else if (tool == "kmers-filter") {
        v = {opt_i("k"), PV::files("k-mers", a.list("k-mers")), PV::files("filter-kmers", a.list("filter-kmers")),
             PV("maximal-bad-frequence", a.get("maximal-bad-frequence", "1")), PV("max-thresh", a.get("max-thresh", "0")),
             PV::file("output-dir", a.get("output-dir", wd + "/kmers")), PV::file("stats-dir", a.get("stats-dir", wd + "/stats"))};
    } else if (tool == "view" || tool == "bin2fasta") {
        v = {opt_i("k"), opt_f("kmers-file"), opt_f("components-file"), opt_f("output-file")};
    }
    return v;
}

static double since_start() {
    static struct timespec t0 = [] { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t; }();
    struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t);
    return (t.tv_sec - t0.tv_sec) + (t.tv_nsec - t0.tv_nsec) * 1e-9;
}
int main(int argc, char **argv) {
    since_start();
    string tool;
    Args a = parse_args(argc, argv, &tool);
    if (a.has("version")) { printf("MetaFast (MI355X HIP hot path) %s\n", mf_version()); return 0; }
    if (a.has("tools")) { printf("Available tools:\n%s", TOOLS_TEXT); return 0; }
    if (a.has("help") || a.has("help-all")) {
        printf("Usage: metafast.sh [-t <tool>] [options]\n\nTools:\n%s\nLaunch options: -w/--work-dir <dir>  -p/--available-processors <n>  -c/--continue  --force  "
               "-s/--start <step>  -f/--finish <step>  -v/--verbose  --devices <a,b,...> (default: every visible device, twice where two libraries fit side by side; the per-library steps run one library per entry)  --device <n>\nTool options follow the reference (see SURVEY.md 8(b1)).\n", TOOLS_TEXT);
        return 0;
    }
    static const char *KNOWN[] = {"kmer-counter", "kmer-counter-many", "seq-builder", "seq-builder-many", "component-cutter", "features-calculator",
                                  "dist-matrix-calculator", "heatmap-maker", "view", "bin2fasta", "matrix-builder", "kmer-counter-posneg", "kmers-filter"};
    if (std::find_if(std::begin(KNOWN), std::end(KNOWN), [&](const char *n) { return tool == n; }) == std::end(KNOWN)) {
        fprintf(stderr, "ERROR: Tool '%s' not found !\n", tool.c_str());          // itmo!/Runner.java:136-139
        return 1;
    }
    g_verbose = a.get("verbose", "false") == "true";
    Env e;
    e.work_dir = a.get("work-dir", "workDir");
    e.start_ts = timestamp();
    const string wd = e.work_dir;
    mkdirs(wd); mkdirs(wd + "/logs");
    {   // Tool.updateFileLoggers :666-690: <workDir>/log (this run only) and <workDir>/logs/log_<ts>, same content
        time_t t = time(nullptr); struct tm tmv; localtime_r(&t, &tmv);
        char hdr[128]; strftime(hdr, sizeof hdr, "Log created at %d-%b-%Y (%a) %H:%M:%S", &tmv);
        g_logfile = fopen((wd + "/log").c_str(), "w");
        g_logfile2 = fopen((wd + "/logs/log_" + e.start_ts).c_str(), "w");
        for (FILE *f : {g_logfile, g_logfile2}) if (f) { fprintf(f, "%s\n", hdr); fflush(f); }
    }
    // ---- --continue / --force (Tool.run :400-462).  matrix-builder forces by default, unless -c or -s is given
    // (DistanceMatrixBuilderMain.java:23-26, 211-216).
    bool cont = a.get("continue", "false") == "true", force = a.get("force", "false") == "true";
    if (tool == "matrix-builder") {
        if (a.has("force")) die("Cannot parse command line: Unrecognized option: --force");       // (removed from its launch options, :25)
        force = !(a.has("continue") || a.has("start"));
    }
    const string start = a.get("start"), finish = a.get("finish");
    const string inprop = wd + "/in.properties";
    if (cont && force) die("Continue and force options can't be set simultaneously");
    if (exists(inprop)) {
        if (cont) {}
        else if (force) { if (tool != "matrix-builder") logmsg("WARN", "Force run, all data in working directory will be rewritten!"); }
        else {
            fprintf(stderr, "Working directory (%s/) contains files from previous run, rewrite them? [Yes(y)/No(n), default:No] ", wd.c_str());
            char ans[64] = ""; if (!fgets(ans, sizeof ans, stdin)) ans[0] = 0;
            string s(ans); while (!s.empty() && isspace((unsigned char)s.back())) s.pop_back();
            for (auto &c : s) c = (char)tolower((unsigned char)c);
            if (s == "y" || s == "yes") force = true; else return 1;
        }
    } else force = true;
    e.cont = cont;
    // ---- runAsStep for the tool itself (:318-392)
    bool can = exists(inprop) && exists(wd + "/SUCCESS");
    if (!force) {
        Props stored;
        if (props_read(inprop, stored)) {
            // parameters not given on the command line take last run's values (loadUnsetParametersFromProperties :757-793)
            for (auto &kv : stored) if (!a.has(kv.first)) a.opt[kv.first] = kv.second;
            can = can && props_equal(tool_inputs(tool, a, wd, e.start_ts), stored);
        } else can = false;
    }
    if (!force && start.empty() && can) {
        logmsg("INFO", "SUCCESS file found for tool %s - loading results...", tool.c_str());
        return 0;
    }
    unlink((wd + "/SUCCESS").c_str()); unlink((wd + "/out.properties").c_str());
    const int k_dflt = tool == "matrix-builder" ? 31 : -1;
    int k = a.geti("k", k_dflt);
    vector<PV> outs;

    // mandatory parameters are checked before anything is written (Tool.checkArguments :712-724)
    auto need = [&](const char *n, const char *shrt) { if (!a.has(n)) die("Mandatory argument --%s (-%s) not set", n, shrt); };
    if (tool == "kmer-counter" || tool == "kmer-counter-many") { need("k", "k"); need("reads", "i"); }
    else if (tool == "seq-builder" || tool == "seq-builder-many") { need("k", "k"); need("k-mers", "i"); need("sequence-len", "l"); }
    else if (tool == "component-cutter") { need("k", "k"); need("sequences", "i"); }
    else if (tool == "features-calculator") { need("k", "k"); need("components-file", "cm"); }
    else if (tool == "dist-matrix-calculator") { if (!a.has("features")) die("Mandatory argument --features not set"); }
    else if (tool == "heatmap-maker") need("matrix-file", "i");
    else if (tool == "matrix-builder") need("reads", "i");
    else if (tool == "kmer-counter-posneg") { need("k", "k"); need("positiveReads", "pos"); need("negativeReads", "neg"); }
    else if (tool == "kmers-filter") { need("k", "k"); need("k-mers", "i"); if (!a.has("filter-kmers")) die("Mandatory argument --filter-kmers not set"); }
    props_write(inprop, tool_inputs(tool, a, wd, e.start_ts));

    if (tool == "kmer-counter") {
        outs = {PV::file("resulting-kmers-file", run_kmer_counter(e, a, a.list("reads"), k, a.geti("maximal-bad-frequence", 1), a.get("output-dir", wd + "/kmers"),
                                                                  a.get("stats-dir", wd + "/stats")))};
    } else if (tool == "kmer-counter-many") {
        check_k(k);
        outs = {PV::files("resulting-kmers-files", run_kmer_counter_many(e, a, a.list("reads"), k, a.geti("maximal-bad-frequence", 1), wd))};
    } else if (tool == "seq-builder" || tool == "seq-builder-many") {
        if (a.has("maximal-bad-frequency") && a.has("bottom-cut-percent") && tool == "seq-builder-many") die("-b and -bp can not be set both");
        int b = a.geti("maximal-bad-frequency", 1), bp = a.has("bottom-cut-percent") ? a.geti("bottom-cut-percent", 0) : -1, l = a.geti("sequence-len", 100);
        string out_dir = a.get("output-dir", wd + "/sequences");
        if (tool == "seq-builder") outs = {PV::file("output-file", run_seq_builder(e, a, a.list("k-mers"), k, b, bp, l, wd, out_dir))};
        else outs = {PV::files("output-files", run_seq_builder_many(e, a, a.list("k-mers"), k, b, bp, l, wd, out_dir))};
    } else if (tool == "component-cutter") {
        const int b1 = a.geti("min-component-size", 1000), b2 = a.geti("max-component-size", 10000);
        string cf = run_component_cutter(e, a, a.list("sequences"), k, a.geti("min-seq-len", 100), b1, b2, wd, a.get("components-file", wd + "/components.bin"));
        outs = {PV::file("components-file", cf), PV::file("components-stat", wd + "/components-stat-" + std::to_string(b1) + "-" + std::to_string(b2) + ".txt")};
    } else if (tool == "features-calculator") {
        outs = {PV::files("features-files", run_features(e, a, a.get("components-file"), a.list("reads"), a.list("kmers"), k, a.geti("threshold", 0), wd)),
                PV::file("features-dir", wd + "/vectors")};
    } else if (tool == "dist-matrix-calculator") {
        run_dist_matrix(e, a, a.list("features"), a.get("matrix-file", wd + "/dist_matrix_$DT_original_order.txt"));
    } else if (tool == "kmer-counter-posneg") {
        // KmersCounterPositiveNegative.java:66-108: two kmer-counter-many steps with the work directories <workDir>/pos and /neg
        vector<string> pos = a.list("positiveReads"), neg = a.list("negativeReads");
        logmsg("INFO", "Found %zu samples in positive class and %zu samples in negative class to process", pos.size(), neg.size());
        if (pos.empty() || neg.empty()) die("No libraries to process!!! Can't continue the calculations.");
        check_k(k);
        const int b = a.geti("maximal-bad-frequency", 1);
        vector<string> kp, kn;
        for (int side = 0; side < 2; side++) {
            const string d = wd + (side ? "/neg" : "/pos");
            const vector<string> &files = side ? neg : pos;
            vector<string> &res = side ? kn : kp;
            run_as_step("kmer-counter-many", d, {PV("k", k), PV::files("reads", files), PV("maximal-bad-frequence", b), PV::file("output-dir", d + "/kmers"),
                                                 PV::file("stats-dir", d + "/stats")},
                        start, force,
                        [&]() { Args sub = a; sub.opt["output-dir"] = {d + "/kmers"}; sub.opt["stats-dir"] = {d + "/stats"};
                                res = run_kmer_counter_many(e, sub, files, k, b, d); return vector<PV>{PV::files("resulting-kmers-files", res)}; },
                        [&](const Props &o) { res = props_list(o, "resulting-kmers-files"); });
        }
        outs = {PV::files("resulting-pos-kmers-files", kp), PV::files("resulting-neg-kmers-files", kn)};
    } else if (tool == "kmers-filter") {
        // KmersFilter.java:80-121: every input k-mers file is written again with the records whose k-mer is frequent enough
        // in the filter files (IOUtils.filterAndPrintKmers, src/io/IOUtils.java:101-123)
        check_k(k);
        const int b = a.geti("maximal-bad-frequence", 1), mt = a.geti("max-thresh", 0);
        const string out_dir = a.get("output-dir", wd + "/kmers");
        mkdirs(out_dir);
        mf_ctx *ctx = ctx_of(e, a);
        const vector<string> ff = a.list("filter-kmers");
        auto fp = cptrs(ff);
        mf_table *filter = nullptr;
        check(mf_table_load_kmers(ctx, fp.data(), (int)fp.size(), b, k, &filter));
        vector<string> written;
        for (auto &f : a.list("k-mers")) {
            mf_table *t = nullptr;
            const char *one[1] = {f.c_str()};
            check(mf_table_load_kmers(ctx, one, 1, b, k, &t));
            string name = basename_of(f);
            for (size_t q; (q = name.find(".kmers.bin")) != string::npos;) name.erase(q, 10);        // replaceAll(".kmers.bin", "")
            const string out = out_dir + "/" + name + ".kmers.bin";
            uint64_t c = 0, size = 0;
            check(mf_table_write_kmers_filtered(t, b, filter, mt * (int)ff.size(), out.c_str(), &c));
            check(mf_table_stats(t, &size, nullptr));
            logmsg("INFO", "%s k-mers found, %s (%.1f%%) of them survived after filtering", group_digits(size).c_str(), group_digits(c).c_str(),
                   size ? c * 100.0 / size : 0.0);
            logmsg("INFO", "Filtered k-mers printed to %s", out.c_str());
            mf_table_destroy(t);
            written.push_back(out);
        }
        mf_table_destroy(filter);
        outs = {written.empty() ? PV::null("resulting-kmers-file") : PV::file("resulting-kmers-file", written.back())};
    } else if (tool == "view") {
        run_view(a, k);
    } else if (tool == "bin2fasta") {
        run_bin2fasta(a, k);
    } else if (tool == "heatmap-maker") {
        string nm = run_heatmap_maker(e, a, a.get("matrix-file"), a.get("newMatrix-file"));
        outs = {PV::null("heatmap-file"), a.get("without-renumbering", "false") == "true" ? PV::null("newMatrix-file-out") : PV::file("newMatrix-file-out", nm)};
    } else if (tool == "matrix-builder") {
        // DistanceMatrixBuilderMain.java:88-175: steps kmer-counter-many, seq-builder-many, component-cutter, features-calculator,
        // dist-matrix-calculator, heatmap-maker (its image is not rendered here)
        vector<string> reads = a.list("reads");
        logmsg("INFO", "Found %zu libraries to process", reads.size());
        if (reads.empty()) die("No libraries to process!!! Can't continue the calculations.");
        const bool use_reads = a.get("use-reads-for-calculating-features", "false") == "true";     // DistanceMatrixBuilderMain.java:162-165
        check_k(k);
        int b = a.geti("maximal-bad-frequency", a.geti("maximal-bad-frequence", 1)), l = a.geti("min-seq-len", 100);
        int b1 = a.geti("min-component-size", 1000), b2 = a.geti("max-component-size", 10000);
        static const char *STEPS[] = {"kmer-counter-many", "seq-builder-many", "component-cutter", "features-calculator", "dist-matrix-calculator", "heatmap-maker"};
        for (const string *bound : {&start, &finish})                           // Tool.checkBoundExistence :726-741
            if (!bound->empty() && std::find_if(std::begin(STEPS), std::end(STEPS), [&](const char *n) { return *bound == n; }) == std::end(STEPS))
                die("There is no substep with name '%s' in step matrix-builder!", bound->c_str());
        {   // createOutputDescFiles :178-200
            g_desc_files = {"output_description.txt", wd + "/output_description.txt"};
            time_t t = time(nullptr); struct tm tmv; localtime_r(&t, &tmv);
            char hdr[128]; strftime(hdr, sizeof hdr, "%d-%b-%Y (%a) %H:%M:%S", &tmv);
            for (auto &d : g_desc_files) {
                FILE *f = fopen(d.c_str(), "w");
                if (!f) { logmsg("WARN", "Can't create file %s, skipping", d.c_str()); continue; }
                fprintf(f, "# Output files' description for run started at %s\n\n%s\n%s\n   Identical files with run log\n\n%s\n%s\n   Identical files with output files' description\n",
                        hdr, (wd + "/log").c_str(), (wd + "/logs/log_" + e.start_ts).c_str(), g_desc_files[0].c_str(), g_desc_files[1].c_str());
                fclose(f);
            }
        }
        const string d1 = wd + "/kmer-counter-many", d2 = wd + "/seq-builder-many", d3 = wd + "/component-cutter", d4 = wd + "/features-calculator",
                     d5 = wd + "/dist-matrix-calculator", d6 = wd + "/heatmap-maker";
        const string dirs[] = {d1, d2, d3, d4, d5, d6};
        // --finish <step>: stop after it; the next step's results are outdated then (:512-527)
        auto finished = [&](int i) {
            if (finish.empty() || finish != STEPS[i]) return false;
            if (i + 1 < 6) { unlink((dirs[i + 1] + "/SUCCESS").c_str()); unlink((dirs[i + 1] + "/in.properties").c_str()); }
            return true;
        };
        auto done = [&]() { for (mf_ctx *c : e.ctxs) if (c) mf_ctx_destroy(c); return 0; };
        vector<string> kmers, seqs, vecs;
        // 1 kmer-counter-many
        const string stats_dir = a.get("stats-dir", d1 + "/stats");
        run_as_step(STEPS[0], d1, {PV("k", k), PV::files("reads", reads), PV("maximal-bad-frequence", b), PV::file("output-dir", d1 + "/kmers"), PV::file("stats-dir", stats_dir)},
                    start, force,
                    [&]() { Args sub = a; sub.opt.erase("output-dir"); sub.opt["stats-dir"] = {stats_dir};
                            kmers = run_kmer_counter_many(e, sub, reads, k, b, d1); return vector<PV>{PV::files("resulting-kmers-files", kmers)}; },
                    [&](const Props &o) { kmers = props_list(o, "resulting-kmers-files"); });
        describe(stats_dir, "Directory with kmer frequency statistics (statistics files is in text format for every input reads file)");
        if (finished(0)) return done();
        // 2 seq-builder-many
        const int bp = a.has("bottom-cut-percent") ? a.geti("bottom-cut-percent", 0) : -1;
        run_as_step(STEPS[1], d2, {PV("k", k), PV::files("k-mers", kmers), PV("maximal-bad-frequency", b), bp >= 0 ? PV("bottom-cut-percent", bp) : PV::null("bottom-cut-percent"),
                                   PV("sequence-len", l), PV::file("output-dir", d2 + "/sequences")},
                    start, force,
                    [&]() { seqs = run_seq_builder_many(e, a, kmers, k, b, bp, l, d2, d2 + "/sequences");
                            return vector<PV>{PV::files("output-files", seqs)}; },
                    [&](const Props &o) { seqs = props_list(o, "output-files"); });
        describe(d2 + "/sequences", "Directory with FASTA files - paths from reads for every library");
        if (finished(1)) return done();
        // 3 component-cutter
        const string comp = d3 + "/components.bin", cstat = d3 + "/components-stat-" + std::to_string(b1) + "-" + std::to_string(b2) + ".txt";
        run_as_step(STEPS[2], d3, {PV("k", k), PV("min-seq-len", l), PV("min-component-size", b1), PV("max-component-size", b2), PV::files("sequences", seqs), PV::file("components-file", comp)},
                    start, force,
                    [&]() { run_component_cutter(e, a, seqs, k, l, b1, b2, d3, comp); return vector<PV>{PV::file("components-file", comp), PV::file("components-stat", cstat)}; },
                    [&](const Props &) {});
        describe(cstat, "File with components' statistics (in text format)");
        describe(comp, "File with extracted components (in binary format)");
        if (finished(2)) return done();
        // 4 features-calculator
        const vector<string> none;
        run_as_step(STEPS[3], d4, {PV("k", k), PV::file("components-file", comp), PV::files("reads", use_reads ? reads : none), PV::files("kmers", use_reads ? none : kmers),
                                   a.has("selected") ? PV::files("selected", a.list("selected")) : PV::null("selected"), PV("threshold", 0)},
                    start, force,
                    [&]() { vecs = use_reads ? run_features(e, a, comp, reads, {}, k, 0, d4) : run_features(e, a, comp, {}, kmers, k, 0, d4);
                            return vector<PV>{PV::files("features-files", vecs), PV::file("features-dir", d4 + "/vectors")}; },
                    [&](const Props &o) { vecs = props_list(o, "features-files"); });
        describe(d4 + "/vectors", "Directory with features values files for every library (in text format)");
        if (finished(3)) return done();
        // 5 dist-matrix-calculator: the matrix in the original order goes to <workDir>/matrices (:132-134)
        string mpath = wd + "/matrices/dist_matrix_" + e.start_ts + "_original_order.txt";
        const bool wn = a.get("without-names", "false") == "true";
        const string fmt = a.get("output-format", "%.4f");
        run_as_step(STEPS[4], d5, {PV::files("features", vecs), PV::flag("without-names", wn), PV::file("matrix-file", mpath), PV("output-format", fmt)},
                    start, force,
                    [&]() { mkdirs(wd + "/matrices"); mpath = run_dist_matrix(e, a, vecs, mpath); return vector<PV>(); },
                    [&](const Props &) {});
        describe(mpath, "File with resulted distance matrix between samples keeping original order");
        if (finished(4)) return done();
        // 6 heatmap-maker, numeric half: dendrogram order + renumbered matrix (:137-145)
        string npath = a.get("matrix-file", wd + "/matrices/dist_matrix_$DT.txt");
        { size_t q = npath.find("$DT"); if (q != string::npos) npath.replace(q, 3, e.start_ts); }
        const bool wr = a.get("without-renumbering", "false") == "true";
        run_as_step(STEPS[5], d6, {PV::file("matrix-file", mpath), a.has("colors-file") ? PV::file("colors-file", a.get("colors-file")) : PV::null("colors-file"),
                                   PV::flag("without-renumbering", wr), PV::file("newMatrix-file", npath),
                                   PV::file("heatmap-file", tool_inputs(tool, a, wd, e.start_ts)[6].vals[0]), PV::flag("invert-colors", a.get("invert-colors", "false") == "true"),
                                   PV("output-format", fmt)},
                    start, force,
                    [&]() { run_heatmap_maker(e, a, mpath, npath); return vector<PV>{PV::null("heatmap-file"), wr ? PV::null("newMatrix-file-out") : PV::file("newMatrix-file-out", npath)}; },
                    [&](const Props &) {});
        if (!wr) describe(npath, "File with resulted distance matrix between samples with new order based on adjacency of the samples");
        logmsg("INFO", "heatmap image (dist_matrix_<date>_heatmap.png) is not rendered by the HIP path");
        if (finished(5)) return done();
    }
    const double t_work = since_start();
    for (mf_ctx *c : e.ctxs) if (c) mf_ctx_synchronize(c);
    if (finish.empty()) {                               // (a run cut short by --finish leaves no SUCCESS, :377-379)
        props_write(wd + "/out.properties", outs);
        touch(wd + "/SUCCESS");
    }
    if (g_logfile) fclose(g_logfile);
    if (g_logfile2) fclose(g_logfile2);
    if (getenv("MF_IO_TIMING")) fprintf(stderr, "[mf] driver: %.3f s until the last step was done\n", t_work);
    // every output is written and closed.  Handing hundreds of megabytes of pinned memory and the arena's regions back one by one, and
    // the HIP runtime's own shutdown after that, cost 0.2 s; the process ends here and the driver reclaims them with it
    // (MF_CLEAN_EXIT=1: the long way, for leak checkers)
    fflush(nullptr);
    if (getenv("MF_CLEAN_EXIT")) { for (mf_ctx *c : e.ctxs) if (c) mf_ctx_destroy(c); return 0; }
    _exit(0);
}

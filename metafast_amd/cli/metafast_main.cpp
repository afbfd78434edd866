// metafast_main.cpp -- metafast.sh-compatible command line driver for the MI355X hot path.
//
// Host-side mirror (C++; the image has no JDK) of the reference's tool shells for this path, on top of the C-ABI only
// (include/metafast_hip.h):
//   src/Runner.java:27-30 + itmo!/Runner.java:108-182          tool registry, -t/--tool, -ts/--tools, --version, default matrix-builder
//   itmo!/utils/tool/Tool.java:61-143, 212-214, 318-392        launch options (-w -p -c --force -s -f -v -h), <workDir>/<tool>/, SUCCESS
//   src/tools/KmersCounterMain.java, KmersCounterForManyFilesMain.java, SeqBuilderMain.java, SeqBuilderForManyFilesMain.java,
//   ComponentCutterMain.java, FeaturesCalculatorMain.java, DistanceMatrixCalculatorMain.java, DistanceMatrixBuilderMain.java
// Parameter names, defaults, output file names and the workDir layout are the reference's; the Tool
// framework itself (in/out.properties, interactive prompts, log4j) is not reproduced.  Errors: message on stderr, exit 1.
#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <set>
#include <string>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>
#include "../../include/metafast_hip.h"

using std::string;
using std::vector;

// ------------------------------------------------------------------------------------------------ utilities
static bool g_verbose = false;
static FILE *g_logfile = nullptr;
static void logmsg(const char *level, const char *fmt, ...) {
    char buf[4096];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    bool debug = !strcmp(level, "DEBUG");
    if (!debug || g_verbose) fprintf(stderr, "%s: %s\n", level, buf);
    if (g_logfile) { fprintf(g_logfile, "%s: %s\n", level, buf); fflush(g_logfile); }
}
[[noreturn]] static void die(const char *fmt, ...) {
    char buf[4096];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    logmsg("ERROR", "%s", buf);
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
    {"output-format", "", false, false}, {"heatmap-file", "", false, false}, {"new-matrix-file", "", false, false},
    {"without-renumbering", "", false, true},
    {"kmers-file", "kf", false, false}, {"output-file", "o", false, false}, {"split", "", false, true}, {"long", "", false, true},
    {"use-reads-for-calculating-features", "", false, true}, {"device", "", false, false},
};
// `ctx_i` says what -i means for the selected tool
static Args parse_args(int argc, char **argv, string *tool_out) {
    // first pass: find the tool (needed to resolve the overloaded short options)
    string tool = "matrix-builder";                                         // src/Runner.java:29
    for (int i = 1; i + 1 < argc; i++) if (!strcmp(argv[i], "-t") || !strcmp(argv[i], "--tool")) tool = argv[i + 1];
    *tool_out = tool;
    auto long_of_short = [&](const string &s) -> string {
        if (s == "i") {
            if (tool == "seq-builder" || tool == "seq-builder-many") return "k-mers";
            if (tool == "component-cutter") return "sequences";
            return "reads";
        }
        if (s == "b") return (tool == "kmer-counter" || tool == "kmer-counter-many") ? "maximal-bad-frequence" : "maximal-bad-frequency";
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
        if (tok.rfind("--", 0) == 0) name = tok.substr(2);
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
struct Env {
    mf_ctx *ctx = nullptr;
    string work_dir;
    bool cont = false;
    string start_ts;
};
static mf_ctx *ctx_of(Env &e, const Args &a) {
    if (!e.ctx) check(mf_ctx_create(a.geti("device", 0), a.geti("available-processors", (int)sysconf(_SC_NPROCESSORS_ONLN)), &e.ctx));
    return e.ctx;
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
    check(mf_count_reads(ctx, fp.data(), (int)fp.size(), k, 0, &t));
    mkdirs(out_dir); mkdirs(stats_dir);
    string name;                                                             // getName :122-137
    if (files.size() == 2) {
        string n1 = library_name(files[0]), n2 = library_name(files[1]);
        auto ends = [](const string &s, const char *x) { return s.size() >= 3 && s.compare(s.size() - 3, 3, x) == 0; };
        if ((ends(n1, "_r1") && ends(n2, "_r2")) || (ends(n1, "_R1") && ends(n2, "_R2"))) name = n1.substr(0, n1.size() - 3);
        else name = n1 + "+";
    } else name = library_name(files[0]) + (files.size() > 1 ? "+" : "");
    string out = out_dir + "/" + name + ".kmers.bin", st = stats_dir + "/" + name + ".stat.txt";
    uint64_t good = 0, size = 0;
    check(mf_table_write_kmers(t, b, out.c_str(), st.c_str(), &good));
    check(mf_table_stats(t, &size, nullptr));
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
    vector<string> outs;
    auto ends = [](const string &s, const char *x) { return s.size() >= 3 && s.compare(s.size() - 3, 3, x) == 0; };
    for (size_t i = 0; i < files.size();) {
        string n = library_name(files[i]);
        bool pair = i + 1 < files.size() && ((ends(n, "_r1") && ends(library_name(files[i + 1]), "_r2")) || (ends(n, "_R1") && ends(library_name(files[i + 1]), "_R2")));
        vector<string> fs(files.begin() + i, files.begin() + i + (pair ? 2 : 1));
        outs.push_back(run_kmer_counter(e, a, fs, k, b, out_dir, stats_dir));
        i += pair ? 2 : 1;
    }
    return outs;
}
// seq-builder (src/tools/SeqBuilderMain.java:78-160)
static string run_seq_builder(Env &e, const Args &a, const vector<string> &files, int k, int b, int bp, int l, const string &wd, const string &out_dir) {
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
    check(mf_build_unitigs(ctx, t, k, b, l, fasta.c_str(), distr.c_str(), &nseq));
    logmsg("INFO", "%s sequences found", group_digits(nseq).c_str());
    if (nseq == 0) logmsg("WARN", "No sequences were found! Perhaps you should decrease --min-seq-len or --maximal-bad-frequency values");
    logmsg("INFO", "Sequences printed to %s", fasta.c_str());
    mf_table_destroy(t);
    return fasta;
}
// component-cutter (src/tools/ComponentCutterMain.java:78-114)
static string run_component_cutter(Env &e, const Args &a, const vector<string> &files, int k, int l, int b1, int b2, const string &wd, const string &comp_file) {
    if (files.empty()) die("Mandatory option --sequences is not set");
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
    if (a.has("selected")) die("--selected is not supported by the HIP path");
    if (kmers.empty() && reads.empty()) die("No input files: pass reads (-i) or k-mers files (-ka)");
    mf_ctx *ctx = ctx_of(e, a);
    string out_dir = wd + "/vectors";
    mkdirs(out_dir);
    vector<string> vecs;
    // reads files first, one vector per FILE, then k-mers files (FeaturesCalculatorMain.java:117-162)
    for (auto &rf : reads) {
        string base = library_name(rf);
        string vec = out_dir + "/" + base + ".vec", br = out_dir + "/" + base + ".breadth";
        const char *fs[1] = {rf.c_str()};
        check(mf_features_reads(ctx, comp_file.c_str(), fs, 1, k, thr, vec.c_str(), br.c_str()));
        logmsg("INFO", "Features for file %s printed to %s", basename_of(rf).c_str(), vec.c_str());
        vecs.push_back(vec);
    }
    for (auto &kf : kmers) {
        string base = remove_ext(basename_of(kf), {".kmers.bin"});
        string vec = out_dir + "/" + base + ".vec", br = out_dir + "/" + base + ".breadth";
        check(mf_features(ctx, comp_file.c_str(), kf.c_str(), k, thr, vec.c_str(), br.c_str()));
        logmsg("INFO", "Features for file %s printed to %s", basename_of(kf).c_str(), vec.c_str());
        vecs.push_back(vec);
    }
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

// one step of a composite tool: <workDir>/<name>/ + SUCCESS marker (Tool.java:212-214, 318-392); -c skips finished steps
struct Step { string name; string dir; };
static void step_finish(const Step &s) { touch(s.dir + "/SUCCESS"); }
static vector<string> list_files(const string &dir, const string &suffix) {
    vector<string> out; string cmd = "ls -1 '" + dir + "' 2>/dev/null"; FILE *p = popen(cmd.c_str(), "r");
    if (!p) return out;
    char line[4096];
    while (fgets(line, sizeof line, p)) { string s(line); while (!s.empty() && (s.back() == '\n' || s.back() == '\r')) s.pop_back(); if (ends_with_ci(s, suffix)) out.push_back(dir + "/" + s); }
    pclose(p); std::sort(out.begin(), out.end());
    return out;
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
    "view\t\t\tView different binary objects (k-mers files, components)\n"
    "bin2fasta\t\tConverts different binary objects to FASTA format\n"
    "matrix-builder\t\tBuild the distance matrix for input sequences (default tool)\n";

int main(int argc, char **argv) {
    string tool;
    Args a = parse_args(argc, argv, &tool);
    if (a.has("version")) { printf("MetaFast (MI355X HIP hot path) %s\n", mf_version()); return 0; }
    if (a.has("tools")) { printf("Available tools:\n%s", TOOLS_TEXT); return 0; }
    if (a.has("help") || a.has("help-all")) {
        printf("Usage: metafast.sh [-t <tool>] [options]\n\nTools:\n%s\nLaunch options: -w/--work-dir <dir>  -p/--available-processors <n>  -c/--continue  --force  "
               "-s/--start <step>  -f/--finish <step>  -v/--verbose  --device <n>\nTool options follow the reference (see SURVEY.md 8(b1)).\n", TOOLS_TEXT);
        return 0;
    }
    g_verbose = a.get("verbose", "false") == "true";
    Env e;
    e.work_dir = a.get("work-dir", "workDir");
    e.cont = a.has("continue");
    e.start_ts = timestamp();
    mkdirs(e.work_dir);
    g_logfile = fopen((e.work_dir + "/log").c_str(), "a");
    const string wd = e.work_dir;
    const int k_dflt = tool == "matrix-builder" ? 31 : -1;
    int k = a.geti("k", k_dflt);

    if (tool == "kmer-counter") {
        if (!a.has("k")) die("Mandatory option -k is not set");
        run_kmer_counter(e, a, a.list("reads"), k, a.geti("maximal-bad-frequence", 1), a.get("output-dir", wd + "/kmers"), a.get("stats-dir", wd + "/stats"));
    } else if (tool == "kmer-counter-many") {
        if (!a.has("k")) die("Mandatory option -k is not set");
        check_k(k);
        run_kmer_counter_many(e, a, a.list("reads"), k, a.geti("maximal-bad-frequence", 1), wd);
    } else if (tool == "seq-builder" || tool == "seq-builder-many") {
        if (!a.has("k")) die("Mandatory option -k is not set");
        if (!a.has("sequence-len")) die("Mandatory option --sequence-len is not set");
        if (a.has("maximal-bad-frequency") && a.has("bottom-cut-percent") && tool == "seq-builder-many") die("-b and -bp can not be set both");
        int b = a.geti("maximal-bad-frequency", 1), bp = a.has("bottom-cut-percent") ? a.geti("bottom-cut-percent", 0) : -1, l = a.geti("sequence-len", 100);
        string out_dir = a.get("output-dir", wd + "/sequences");
        if (tool == "seq-builder") run_seq_builder(e, a, a.list("k-mers"), k, b, bp, l, wd, out_dir);
        else for (auto &f : a.list("k-mers")) run_seq_builder(e, a, {f}, k, b, bp, l, wd + "/sub-builder", out_dir);
    } else if (tool == "component-cutter") {
        if (!a.has("k")) die("Mandatory option -k is not set");
        run_component_cutter(e, a, a.list("sequences"), k, a.geti("min-seq-len", 100), a.geti("min-component-size", 1000), a.geti("max-component-size", 10000), wd,
                             a.get("components-file", wd + "/components.bin"));
    } else if (tool == "features-calculator") {
        if (!a.has("k")) die("Mandatory option -k is not set");
        run_features(e, a, a.get("components-file"), a.list("reads"), a.list("kmers"), k, a.geti("threshold", 0), wd);
    } else if (tool == "dist-matrix-calculator") {
        run_dist_matrix(e, a, a.list("features"), a.get("matrix-file", wd + "/dist_matrix_$DT_original_order.txt"));
    } else if (tool == "view") {
        run_view(a, k);
    } else if (tool == "bin2fasta") {
        run_bin2fasta(a, k);
    } else if (tool == "heatmap-maker") {
        if (!a.has("matrix-file")) die("Mandatory option --matrix-file is not set");
        run_heatmap_maker(e, a, a.get("matrix-file"), a.get("new-matrix-file"));
    } else if (tool == "matrix-builder") {
        // DistanceMatrixBuilderMain.java:88-175: steps kmer-counter-many, seq-builder-many, component-cutter, features-calculator,
        // dist-matrix-calculator (+ heatmap-maker: rendering, out of scope)
        vector<string> reads = a.list("reads");
        if (reads.empty()) die("No libraries to process!!! Can't continue the calculations.");
        const bool use_reads = a.get("use-reads-for-calculating-features", "false") == "true";     // DistanceMatrixBuilderMain.java:162-165
        logmsg("INFO", "Found %zu libraries to process", reads.size());
        check_k(k);
        int b = a.geti("maximal-bad-frequency", a.geti("maximal-bad-frequence", 1)), l = a.geti("min-seq-len", 100);
        int b1 = a.geti("min-component-size", 1000), b2 = a.geti("max-component-size", 10000);
        string start = a.get("start"), finish = a.get("finish");
        Step s1{"kmer-counter-many", wd + "/kmer-counter-many"}, s2{"seq-builder-many", wd + "/seq-builder-many"}, s3{"component-cutter", wd + "/component-cutter"},
             s4{"features-calculator", wd + "/features-calculator"}, s5{"dist-matrix-calculator", wd + "/matrices"};
        // -s/--start <step>: reuse everything before it; -c/--continue: reuse leading steps that have a SUCCESS marker;
        // once one step runs, all later ones run (Tool.java:485-529)
        bool running = false;
        auto should_run = [&](const Step &s) {
            if (running) return true;
            if (!start.empty()) { if (s.name == start) running = true; return running; }
            if (e.cont && exists(s.dir + "/SUCCESS")) return false;
            running = true;
            return true;
        };
        auto stop_after = [&](const Step &s) { return !finish.empty() && s.name == finish; };
        vector<string> kmers, seqs, vecs;
        // 1
        if (should_run(s1)) {
            Args sub = a; sub.opt.erase("output-dir");
            kmers = run_kmer_counter_many(e, sub, reads, k, b, s1.dir); step_finish(s1);
        } else { logmsg("INFO", "Step %s: reusing results", s1.name.c_str()); kmers = list_files(s1.dir + "/kmers", ".kmers.bin"); }
        if (stop_after(s1)) return 0;
        // 2
        if (should_run(s2)) {
            for (auto &f : kmers) seqs.push_back(run_seq_builder(e, a, {f}, k, b, -1, l, s2.dir + "/sub-builder", s2.dir + "/sequences"));
            step_finish(s2);
        } else { logmsg("INFO", "Step %s: reusing results", s2.name.c_str()); seqs = list_files(s2.dir + "/sequences", ".seq.fasta"); }
        if (stop_after(s2)) return 0;
        // 3
        string comp = s3.dir + "/components.bin";
        if (should_run(s3)) { run_component_cutter(e, a, seqs, k, l, b1, b2, s3.dir, comp); step_finish(s3); }
        else logmsg("INFO", "Step %s: reusing results", s3.name.c_str());
        if (stop_after(s3)) return 0;
        // 4
        if (should_run(s4)) { vecs = use_reads ? run_features(e, a, comp, reads, {}, k, 0, s4.dir) : run_features(e, a, comp, {}, kmers, k, 0, s4.dir); step_finish(s4); }
        else { logmsg("INFO", "Step %s: reusing results", s4.name.c_str()); vecs = list_files(s4.dir + "/vectors", ".vec"); }
        if (stop_after(s4)) return 0;
        // 5
        string mpath = run_dist_matrix(e, a, vecs, wd + "/matrices/dist_matrix_$DT_original_order.txt");
        // 6: heatmap-maker, numeric half: dendrogram order + renumbered matrix (DistanceMatrixBuilderMain.java:137-145)
        run_heatmap_maker(e, a, mpath, a.get("matrix-file", wd + "/matrices/dist_matrix_$DT.txt"));
        logmsg("INFO", "heatmap image (dist_matrix_<date>_heatmap.png) is not rendered by the HIP path");
    } else {
        die("Unknown tool '%s' (use -ts to list the tools of the HIP hot path)", tool.c_str());
    }
    touch(wd + "/SUCCESS");
    if (e.ctx) mf_ctx_destroy(e.ctx);
    if (g_logfile) fclose(g_logfile);
    return 0;
}

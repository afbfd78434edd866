// mf_ctx.hip -- context, workspace cache, HIP-event kernel timers, options.
#include "mf_common.h"
#include <time.h>
#include <stdlib.h>

static thread_local char g_err[1024] = "";

int mf_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return MF_ERR;
}

extern "C" const char *mf_last_error(void) { return g_err; }

// ---- roctx ranges (mf_range) ----
#include <dlfcn.h>
#include <mutex>
static int (*g_roctx_push)(const char *) = nullptr;
static int (*g_roctx_pop)(void) = nullptr;
static bool roctx_ready() {
    static std::once_flag once;
    std::call_once(once, [] {
        const char *want = getenv("MF_ROCTX");
        if (want ? atoi(want) == 0 : getenv("ROCP_TOOL_LIBRARIES") == nullptr) return;
        void *h = nullptr;
        for (const char *n : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!h) return;
        g_roctx_push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
        g_roctx_pop = (int (*)(void))dlsym(h, "roctxRangePop");
        if (!g_roctx_push || !g_roctx_pop) g_roctx_push = nullptr;
    });
    return g_roctx_push != nullptr;
}
mf_range::mf_range(const char *name) : on(roctx_ready()) { if (on) g_roctx_push(name); }
mf_range::~mf_range() { if (on) g_roctx_pop(); }

// devices this process can see (what `--devices` of the driver defaults to); no context is made
extern "C" int mf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}
// bytes of device memory (no context is made): what a host sizes its per-device concurrency by
extern "C" int mf_device_memory(int device, uint64_t *total_bytes) {
    if (!total_bytes) return mf_set_error("mf_device_memory: NULL argument");
    hipDeviceProp_t prop;
    MF_HIP(hipGetDeviceProperties(&prop, device));
    *total_bytes = (uint64_t)prop.totalGlobalMem;
    return MF_OK;
}
// HIP's current device is a property of the calling THREAD: a context made on one thread and then used from another (the driver's
// per-device workers take turns over the steps of a run) is bound to the new thread first.  One calling thread at a time per context.
extern "C" int mf_ctx_bind_thread(mf_ctx *ctx) {
    if (!ctx) return mf_set_error("ctx is NULL");
    MF_HIP(hipSetDevice(ctx->device));
    return MF_OK;
}
extern "C" int mf_ctx_device(const mf_ctx *ctx) { return ctx ? ctx->device : mf_set_error("ctx is NULL"); }
extern "C" const char *mf_version(void) { return "metafast_hip 0.1 (gfx950)"; }

extern "C" int mf_ctx_create(int device, int host_threads, mf_ctx **out) {
    if (!out) return mf_set_error("mf_ctx_create: out is NULL");
    *out = nullptr;
    struct timespec ts0; clock_gettime(CLOCK_MONOTONIC, &ts0);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return mf_set_error("mf_ctx_create: no HIP device available (%s); this library has no CPU fallback",
                            e != hipSuccess ? hipGetErrorString(e) : "0 devices");
    if (device < 0 || device >= ndev) return mf_set_error("mf_ctx_create: device %d out of range [0,%d)", device, ndev);
    MF_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    MF_HIP(hipGetDeviceProperties(&prop, device));
    mf_ctx *c = new mf_ctx();
    c->device = device;
    c->host_threads = host_threads > 0 ? host_threads : 1;
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    hipError_t se = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (se != hipSuccess) { delete c; return mf_set_error("hipStreamCreate failed: %s", hipGetErrorString(se)); }
    c->own_stream = true;
    *out = c;
    if (getenv("MF_IO_TIMING")) {
        struct timespec ts1; clock_gettime(CLOCK_MONOTONIC, &ts1);
        fprintf(stderr, "[mf] ctx_create: %.3f s (HIP runtime start, device %d, %d CUs)\n", (ts1.tv_sec - ts0.tv_sec) + (ts1.tv_nsec - ts0.tv_nsec) * 1e-9, device, c->n_cu);
    }
    return MF_OK;
}

// give every completely free region back to the driver
extern "C" int mf_ctx_trim(mf_ctx *ctx) {
    if (!ctx) return MF_OK;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    for (size_t i = 0; i < ctx->regions.size();) {
        auto &r = ctx->regions[i];
        if (r.free_spans.size() == 1 && r.free_spans[0].off == 0 && r.free_spans[0].sz == r.size) {
            hipFree(r.base);
            ctx->arena_bytes -= r.size;
            ctx->regions.erase(ctx->regions.begin() + i);
        } else i++;
    }
    return MF_OK;
}

// ... only as many as it takes to hand `want` bytes back, smallest regions first: a region costs its next user 35 ms per GiB
// of hipMalloc (150 GiB: 5 s), so a caller that is short of a gigabyte should not make the arena drop its 100 GB record
// buffers.  *freed (may be NULL) = bytes given back (less than `want` when there were not enough idle regions).
extern "C" int mf_ctx_trim_bytes(mf_ctx *ctx, uint64_t want, uint64_t *freed) {
    if (freed) *freed = 0;
    if (!ctx) return MF_OK;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    uint64_t got = 0;
    while (got < want) {
        int best = -1;
        for (size_t i = 0; i < ctx->regions.size(); i++) {
            auto &r = ctx->regions[i];
            if (r.free_spans.size() == 1 && r.free_spans[0].off == 0 && r.free_spans[0].sz == r.size && (best < 0 || r.size < ctx->regions[best].size)) best = (int)i;
        }
        if (best < 0) break;
        hipFree(ctx->regions[best].base);
        got += ctx->regions[best].size;
        ctx->arena_bytes -= ctx->regions[best].size;
        ctx->regions.erase(ctx->regions.begin() + best);
    }
    if (freed) *freed = got;
    return MF_OK;
}

extern "C" void mf_ctx_destroy(mf_ctx *ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    mf_file_cache_clear(ctx);
    for (auto &r : ctx->regions) hipFree(r.base);
    ctx->regions.clear();
    for (auto &r : ctx->pending) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    for (auto ev : ctx->event_pool) hipEventDestroy(ev);
    if (ctx->pin_pool) { if (ctx->pin_pool_pinned) hipHostFree(ctx->pin_pool); else free(ctx->pin_pool); }
    if (ctx->up_pool) { if (ctx->up_pool_pinned) hipHostFree(ctx->up_pool); else free(ctx->up_pool); }
    if (ctx->up_stream) (void)hipStreamDestroy((hipStream_t)ctx->up_stream);
    if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int mf_ctx_set_stream(mf_ctx *ctx, void *hip_stream) {
    if (!ctx) return mf_set_error("ctx is NULL");
    if (ctx->own_stream && ctx->stream) { hipStreamSynchronize(ctx->stream); hipStreamDestroy(ctx->stream); }
    // NULL = the device's default (null) stream, which is what torch uses unless told otherwise: work submitted by the
    // caller on that stream (allocations' fill kernels, copies) is then ordered before the library's kernels
    ctx->stream = (hipStream_t)hip_stream;
    ctx->own_stream = false;
    return MF_OK;
}

extern "C" int mf_ctx_set_option(mf_ctx *ctx, const char *name, int64_t v) {
    if (!ctx || !name) return mf_set_error("mf_ctx_set_option: NULL argument");
    std::string s(name);
    if (s == "l1_bits") { if (v > MF_MAX_DIGIT_BITS) return mf_set_error("l1_bits > %d", MF_MAX_DIGIT_BITS); ctx->opt_l1_bits = v; }
    else if (s == "l2_bits") { if (v > MF_MAX_DIGIT_BITS) return mf_set_error("l2_bits > %d", MF_MAX_DIGIT_BITS); ctx->opt_l2_bits = v; }
    else if (s == "part_target") { if (v < 1 || v > 16384) return mf_set_error("part_target out of [1,16384]"); ctx->opt_part_target = v; }
    else if (s == "part_target_long") { if (v < 1 || v > 16384) return mf_set_error("part_target_long out of [1,16384]"); ctx->opt_part_target_long = v; }
    else if (s == "scatter_staged") ctx->opt_scatter_staged = v;
    else if (s == "profile") ctx->opt_profile = v;
    else if (s == "l1_blocks") ctx->opt_l1_blocks = v;
    else if (s == "verbose") ctx->opt_verbose = v;
    else if (s == "ablate") ctx->opt_ablate = v;
    else if (s == "scatter_fast") ctx->opt_scatter_fast = v;
    else if (s == "skm") ctx->opt_skm = v;
    else if (s == "skm_batches") ctx->opt_skm_batches = v;
    else if (s == "skm_slices") { if (v < 0 || v > 64 || (v & (v - 1))) return mf_set_error("skm_slices must be 0 or a power of two <= 64"); ctx->opt_skm_slices = v; }
    else if (s == "arena_cap_gb") ctx->opt_arena_cap_gb = v;
    else if (s == "skm_shared") ctx->opt_skm_shared = v;
    else if (s == "skm_dedupe") { if (v != 0 && v != 1 && v != 5) return mf_set_error("skm_dedupe must be 0, 1 or 5"); ctx->opt_skm_dedupe = v; }
    else if (s == "stream_reader") ctx->opt_stream_reader = v;
    else if (s == "stream_piece_bytes") ctx->opt_sr_piece = v;
    else if (s == "stream_slack_bytes") ctx->opt_sr_slack = v;
    else if (s == "skm_dyn") ctx->opt_skm_dyn = v;
    else if (s == "nbr_global") ctx->opt_nbr_global = v;
    else if (s == "ut_plain_rounds") ctx->opt_ut_plain_rounds = v;
    else if (s == "cc_compress") ctx->opt_cc_compress = v;
    else if (s == "stream_count") ctx->opt_stream_count = v;
    else if (s == "stream_count_min_bytes") ctx->opt_stream_count_min = v;
    else if (s == "stream_count_piece_bytes") ctx->opt_stream_count_piece = v;
    else if (s == "stream_count_test_pct") ctx->opt_stream_count_test_pct = v;
    else if (s == "wide_skm") ctx->opt_wide_skm = v;
    else if (s == "wide_skm_min") ctx->opt_wide_skm_min = v;
    else if (s == "wide_skm_unit") ctx->opt_wide_skm_unit = v;
    else if (s == "wide_skm_lead") ctx->opt_wide_skm_lead = v;
    else if (s == "wide_skm_merge") ctx->opt_wide_skm_merge = v;
    else if (s == "wide_skm_pack") ctx->opt_wide_skm_pack = v;
    else if (s == "wide_skm_fine") ctx->opt_wide_skm_fine = v;
    else if (s == "wide_skm_lazy_order") ctx->opt_wide_skm_lazy_order = v;
    else if (s == "dcc_sparse") ctx->opt_dcc_sparse = v;
    else if (s == "dcc_test_fail") { ctx->opt_dcc_test_fail = v; ctx->dcc_test_calls[1] = ctx->dcc_test_calls[2] = 0; }
    else if (s == "cc_sparse") ctx->opt_cc_sparse = v;
    else if (s == "gz_device_min_bytes") ctx->opt_gz_device_min = v;
    else if (s == "gz_piece_bytes") { if (v < 4096) return mf_set_error("gz_piece_bytes must be at least 4096"); ctx->opt_gz_piece = v; }
    else if (s == "ut_double_after") { if (v < 1 || v > 64) return mf_set_error("ut_double_after must be in [1, 64]"); ctx->opt_ut_double_after = v; }
    else if (s == "wide_finish") ctx->opt_wide_finish = v ? 1 : 0;
    else if (s == "wide_big_bucket") { if (v < 1 || v > 256) return mf_set_error("wide_big_bucket must be in [1, 256]"); ctx->opt_wide_big_bucket = v; }
    else if (s == "wide_ablate") ctx->opt_wide_ablate = v;
    else if (s == "wide_distinct") { if (v < 1 || v > 1280) return mf_set_error("wide_distinct must be in [1, 1280]"); ctx->opt_wide_distinct = v; }
    else if (s == "wide_passes") { if (v < 0 || v > 65536) return mf_set_error("wide_passes must be in [0, 65536]"); ctx->opt_wide_passes = v; }
    else if (s == "file_cache") {                       // GB; -1: a quarter of the device's memory, -n: an n-th of that (n contexts share the device)
        if (v < 0) { size_t fr = 0, tot = 0; MF_HIP(hipSetDevice(ctx->device)); MF_HIP(hipMemGetInfo(&fr, &tot)); v = std::max<int64_t>(1, (int64_t)(tot >> 32) / -v); }
        ctx->opt_file_cache_gb = v;
        if (!v) mf_file_cache_clear(ctx);
    }
    else if (s == "skm_pilot") ctx->opt_skm_pilot = v;
    else if (s == "device_parse") ctx->opt_device_parse = v;
    else if (s == "device_parse_min_bytes") ctx->opt_device_parse_min = v;
    else if (s == "device_parse_piece_bytes") ctx->opt_device_parse_piece = v;
    else if (s == "device_parse_threads") ctx->opt_device_parse_threads = v;
    else if (s == "host_pinned") ctx->opt_host_pinned = v;
    else if (s == "skm_dynq") ctx->opt_skm_dynq = v;
    else if (s == "part_good") { if (v < 16 || v > 4096) return mf_set_error("part_good out of [16,4096]"); ctx->opt_part_good = v; }
    else if (s == "unit_parts_long") { if (v < 0 || v > 4) return mf_set_error("unit_parts_long out of [0,4]"); ctx->opt_unit_parts_long = v; }
    else if (s == "skm_unit_records") { if (v != 0 && (v < 64 || v > (1 << 20))) return mf_set_error("skm_unit_records out of [64,2^20] (0: by k)"); ctx->opt_skm_unit_records = v; }
    else if (s == "skm_unit_distinct") { if (v < 64 || v > 3400) return mf_set_error("skm_unit_distinct out of [64,3400]"); ctx->opt_skm_unit_distinct = v; }
    else if (s == "union_samples") ctx->opt_union_samples = v;
    else return mf_set_error("unknown option '%s'", name);
    return MF_OK;
}

// counters and gauges of the context, by name (-1: unknown name)
extern "C" int64_t mf_ctx_stat(mf_ctx *ctx, const char *name) {
    if (!ctx || !name) return mf_set_error("mf_ctx_stat: NULL argument");
    const std::string s(name);
    if (s == "slice_restarts") return (int64_t)ctx->n_slice_restarts;
    if (s == "gz_files_inflated_into_hbm") return (int64_t)ctx->n_gz_device;
    if (s == "pilot_runs") return (int64_t)ctx->n_pilots;
    if (s == "unitig_doublings") return (int64_t)ctx->n_ut_doubled;
    if (s == "wide_big_entries") return (int64_t)ctx->n_wide_big;
    if (s == "wide_hashed_entries") return (int64_t)ctx->n_wide_hashed;
    if (s == "streamed_counts") return (int64_t)ctx->n_streamed;
    if (s == "streamed_counts_stepped_back") return (int64_t)ctx->n_stream_stepped_back;
    if (s == "device_parsed_files") return (int64_t)ctx->n_dparse_files;
    if (s == "device_parser_stepped_back") return (int64_t)ctx->n_dparse_stepped_back;
    if (s == "hipmalloc_calls") return (int64_t)ctx->n_hipmalloc;
    if (s == "hipmalloc_bytes") return (int64_t)ctx->b_hipmalloc;
    if (s == "hipmalloc_us") return (int64_t)(ctx->t_hipmalloc * 1e6);
    if (s == "arena_bytes") return (int64_t)ctx->arena_bytes;
    if (s == "arena_idle_bytes") return (int64_t)mf_arena_idle(ctx);
    return mf_set_error("mf_ctx_stat: unknown name '%s'", name);
}
extern "C" int mf_ctx_synchronize(mf_ctx *ctx) {
    if (!ctx) return mf_set_error("ctx is NULL");
    MF_HIP(hipStreamSynchronize(ctx->stream));
    return MF_OK;
}

// ---- workspace arena ----
static const size_t MF_ALIGN = 256;
// spare_big: leave the idle record buffers of huge samples (a free span of >= 48 GiB, four times the request or more)
// alone.  Something small and long-lived carved out of one -- a table, an index -- pins it: the next sample's level-1
// buffer then needs a new region, the device is full, everything idle goes back to the driver, and the 126 GB region is
// allocated again at 35 ms per GiB (8 samples x 200 M reads at k = 21 on one GPU spent 3 s of an 8 s step there).
static bool arena_take(mf_ctx *ctx, size_t bytes, void **out, bool spare_big) {
    // best fit over all regions
    int br = -1, bs = -1; size_t best = ~(size_t)0;
    for (size_t r = 0; r < ctx->regions.size(); r++)
        for (size_t s = 0; s < ctx->regions[r].free_spans.size(); s++) {
            size_t sz = ctx->regions[r].free_spans[s].sz;
            if (spare_big && sz >= ((size_t)48 << 30) && sz / 4 >= bytes) continue;
            if (sz >= bytes && sz < best) { best = sz; br = (int)r; bs = (int)s; }
        }
    if (br < 0) return false;
    auto &R = ctx->regions[br];
    mf_ctx::span sp = R.free_spans[bs];
    *out = R.base + sp.off;
    if (sp.sz == bytes) R.free_spans.erase(R.free_spans.begin() + bs);
    else { R.free_spans[bs].off += bytes; R.free_spans[bs].sz -= bytes; }
    return true;
}
int mf_alloc(mf_ctx *ctx, size_t bytes, void **out) {
    bytes = (bytes + MF_ALIGN - 1) & ~(MF_ALIGN - 1);
    if (arena_take(ctx, bytes, out, true)) return MF_OK;
    // new region: small requests share 256 MiB regions, large ones get their own
    size_t rsz = bytes < ((size_t)256 << 20) ? ((size_t)256 << 20) : bytes;
    if (bytes >= ((size_t)1 << 30)) {
        // ... with up to 3 % of slack: the next sample's buffer of this kind is a few megabytes larger or smaller, and a region
        // that is a hair too small is a region allocated twice
        const size_t gran = (size_t)1 << (63 - __builtin_clzll((unsigned long long)bytes) - 5);
        rsz = (bytes + gran - 1) & ~(gran - 1);
    }
    void *p = nullptr;
    struct timespec ta; clock_gettime(CLOCK_MONOTONIC, &ta);
    hipError_t e = hipMalloc(&p, rsz);
    if (e != hipSuccess && rsz != bytes) { (void)hipGetLastError(); rsz = bytes; e = hipMalloc(&p, rsz); }
    { struct timespec tb; clock_gettime(CLOCK_MONOTONIC, &tb); ctx->t_hipmalloc += (tb.tv_sec - ta.tv_sec) + (tb.tv_nsec - ta.tv_nsec) * 1e-9; ctx->n_hipmalloc++; ctx->b_hipmalloc += rsz; }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (arena_take(ctx, bytes, out, false)) return MF_OK;       // (rather a pinned big region than a dropped one)
        if (!ctx->file_cache.empty()) {                             // (objects kept for files that may be loaded again: a convenience, gone first)
            mf_file_cache_clear(ctx);
            if (arena_take(ctx, bytes, out, true)) return MF_OK;
        }
        // idle regions go back to the driver, the smallest first and only as many as it takes
        if (ctx->opt_verbose) fprintf(stderr, "[mf] arena: hipMalloc(%.1f GB) failed with %.1f GB idle: giving idle regions back\n", rsz / 1e9, mf_arena_idle(ctx) / 1e9);
        for (;;) {
            uint64_t got = 0;
            mf_ctx_trim_bytes(ctx, 1, &got);
            if (!got) break;
            e = hipMalloc(&p, rsz);
            if (e == hipSuccess) break;
            (void)hipGetLastError();
        }
        if (e != hipSuccess) return mf_set_error("hipMalloc(%zu bytes) failed: %s", rsz, hipGetErrorString(e));
    }
    if (ctx->opt_verbose && rsz >= ((size_t)1 << 30))
        fprintf(stderr, "[mf] arena: new region of %.1f GB (request %.1f GB; arena %.1f GB, idle %.1f GB)\n", rsz / 1e9, bytes / 1e9, ctx->arena_bytes / 1e9, mf_arena_idle(ctx) / 1e9);
    mf_ctx::region R; R.base = (char *)p; R.size = rsz; R.free_spans.push_back({0, rsz});
    ctx->regions.push_back(R);
    ctx->arena_bytes += rsz;
    if (!arena_take(ctx, bytes, out, false)) return mf_set_error("arena: internal error");
    return MF_OK;
}
// bytes the arena holds but nobody uses (they can serve the next request without a hipMalloc)
size_t mf_arena_idle(const mf_ctx *ctx) {
    size_t t = 0;
    for (auto &R : ctx->regions) for (auto &f : R.free_spans) t += f.sz;
    return t;
}
void mf_release(mf_ctx *ctx, void *p, size_t bytes) {
    if (!p) return;
    bytes = (bytes + MF_ALIGN - 1) & ~(MF_ALIGN - 1);
    for (auto &R : ctx->regions) {
        if ((char *)p < R.base || (char *)p >= R.base + R.size) continue;
        size_t off = (size_t)((char *)p - R.base);
        auto &fs = R.free_spans;
        size_t i = 0;
        while (i < fs.size() && fs[i].off < off) i++;
        fs.insert(fs.begin() + i, {off, bytes});
        if (i + 1 < fs.size() && fs[i].off + fs[i].sz == fs[i + 1].off) { fs[i].sz += fs[i + 1].sz; fs.erase(fs.begin() + i + 1); }
        if (i > 0 && fs[i - 1].off + fs[i - 1].sz == fs[i].off) { fs[i - 1].sz += fs[i].sz; fs.erase(fs.begin() + i); }
        return;
    }
    fprintf(stderr, "[mf] warning: mf_release of a pointer that is not in the arena\n");
}

// ---- timers ----
mf_ktimer::mf_ktimer(mf_ctx *c, const char *name) : ctx(c), idx(-1) {
    if (!c->opt_profile) return;
    mf_timer_rec r;
    r.name = name;
    auto get = [&](hipEvent_t *ev) {
        if (!c->event_pool.empty()) { *ev = c->event_pool.back(); c->event_pool.pop_back(); }
        else hipEventCreate(ev);
    };
    get(&r.a); get(&r.b);
    hipEventRecord(r.a, c->stream);
    c->pending.push_back(r);
    idx = (int)c->pending.size() - 1;
}
mf_ktimer::~mf_ktimer() {
    if (idx >= 0) hipEventRecord(ctx->pending[idx].b, ctx->stream);
}
int mf_collect_timers(mf_ctx *ctx) {
    if (ctx->pending.empty()) return MF_OK;
    MF_HIP(hipStreamSynchronize(ctx->stream));
    for (auto &r : ctx->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            auto &t = ctx->timings[r.name];
            t.n += 1; t.total_ms += ms; if (ms > t.max_ms) t.max_ms = ms;
        }
        ctx->event_pool.push_back(r.a);
        ctx->event_pool.push_back(r.b);
    }
    ctx->pending.clear();
    return MF_OK;
}
extern "C" int64_t mf_ctx_kernel_time(mf_ctx *ctx, const char *kernel, double *total_ms) {
    if (!ctx || !kernel) return mf_set_error("NULL argument");
    if (mf_collect_timers(ctx) < 0) return MF_ERR;
    auto it = ctx->timings.find(kernel);
    if (it == ctx->timings.end()) { if (total_ms) *total_ms = 0; return 0; }
    if (total_ms) *total_ms = it->second.total_ms;
    return it->second.n;
}
extern "C" int mf_ctx_kernel_report(mf_ctx *ctx, char *buf, uint64_t cap) {
    if (!ctx || !buf || !cap) return mf_set_error("NULL argument");
    if (mf_collect_timers(ctx) < 0) return MF_ERR;
    std::string s;
    char line[256];
    for (auto &kv : ctx->timings) {
        snprintf(line, sizeof line, "%s\t%lld\t%.6f\t%.6f\n", kv.first.c_str(), (long long)kv.second.n, kv.second.total_ms, kv.second.max_ms);
        s += line;
    }
    if (s.size() + 1 > cap) return mf_set_error("report buffer too small (%zu needed)", s.size() + 1);
    memcpy(buf, s.c_str(), s.size() + 1);
    return MF_OK;
}
extern "C" int mf_ctx_reset_timers(mf_ctx *ctx) {
    if (!ctx) return mf_set_error("ctx is NULL");
    if (mf_collect_timers(ctx) < 0) return MF_ERR;
    ctx->timings.clear();
    return MF_OK;
}

int mf_debug_sync(mf_ctx *ctx, const char *what) {
    fprintf(stderr, "[mf] launched %s ...", what); fflush(stderr);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipGetLastError();
    fprintf(stderr, " %s\n", e == hipSuccess ? "ok" : hipGetErrorString(e)); fflush(stderr);
    if (e != hipSuccess) return mf_set_error("%s failed: %s", what, hipGetErrorString(e));
    return MF_OK;
}

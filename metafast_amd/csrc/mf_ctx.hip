// mf_ctx.hip -- context, workspace cache, HIP-event kernel timers, options.
#include "mf_common.h"

static thread_local char g_err[1024] = "";

int mf_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return MF_ERR;
}

extern "C" const char *mf_last_error(void) { return g_err; }
extern "C" const char *mf_version(void) { return "metafast_hip 0.1 (gfx950)"; }

extern "C" int mf_ctx_create(int device, int host_threads, mf_ctx **out) {
    if (!out) return mf_set_error("mf_ctx_create: out is NULL");
    *out = nullptr;
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
    return MF_OK;
}

extern "C" int mf_ctx_trim(mf_ctx *ctx) {
    if (!ctx) return MF_OK;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    for (auto &b : ctx->free_list) hipFree(b.p);
    ctx->free_list.clear();
    ctx->cached_bytes = 0;
    return MF_OK;
}

extern "C" void mf_ctx_destroy(mf_ctx *ctx) {
    if (!ctx) return;
    mf_ctx_trim(ctx);
    for (auto &r : ctx->pending) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    for (auto ev : ctx->event_pool) hipEventDestroy(ev);
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
    else if (s == "part_target") { if (v < 16 || v > 4096) return mf_set_error("part_target out of [16,4096]"); ctx->opt_part_target = v; }
    else if (s == "scatter_staged") ctx->opt_scatter_staged = v;
    else if (s == "profile") ctx->opt_profile = v;
    else if (s == "l1_blocks") ctx->opt_l1_blocks = v;
    else if (s == "verbose") ctx->opt_verbose = v;
    else return mf_set_error("unknown option '%s'", name);
    return MF_OK;
}

extern "C" int mf_ctx_synchronize(mf_ctx *ctx) {
    if (!ctx) return mf_set_error("ctx is NULL");
    MF_HIP(hipStreamSynchronize(ctx->stream));
    return MF_OK;
}

// ---- workspace cache: best-fit reuse of freed blocks (hipMalloc/hipFree stay out of timed loops) ----
int mf_alloc(mf_ctx *ctx, size_t bytes, void **out) {
    bytes = (bytes + 255) & ~(size_t)255;
    int best = -1;
    for (size_t i = 0; i < ctx->free_list.size(); i++) {
        size_t sz = ctx->free_list[i].sz;
        if (sz >= bytes && sz <= bytes + bytes / 4 + 4096 && (best < 0 || sz < ctx->free_list[best].sz)) best = (int)i;
    }
    if (best >= 0) {
        *out = ctx->free_list[best].p;
        ctx->cached_bytes -= ctx->free_list[best].sz;
        ctx->free_list.erase(ctx->free_list.begin() + best);
        return MF_OK;
    }
    hipError_t e = hipMalloc(out, bytes);
    if (e != hipSuccess) {
        // give cached blocks back and retry once
        (void)hipGetLastError();
        mf_ctx_trim(ctx);
        e = hipMalloc(out, bytes);
        if (e != hipSuccess) { (void)hipGetLastError(); return mf_set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e)); }
    }
    return MF_OK;
}
void mf_release(mf_ctx *ctx, void *p, size_t bytes) {
    if (!p) return;
    bytes = (bytes + 255) & ~(size_t)255;
    ctx->free_list.push_back({p, bytes});
    ctx->cached_bytes += bytes;
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
            t.first += 1; t.second += ms;
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
    if (total_ms) *total_ms = it->second.second;
    return it->second.first;
}
extern "C" int mf_ctx_kernel_report(mf_ctx *ctx, char *buf, uint64_t cap) {
    if (!ctx || !buf || !cap) return mf_set_error("NULL argument");
    if (mf_collect_timers(ctx) < 0) return MF_ERR;
    std::string s;
    char line[256];
    for (auto &kv : ctx->timings) {
        snprintf(line, sizeof line, "%s\t%lld\t%.6f\n", kv.first.c_str(), (long long)kv.second.first, kv.second.second);
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

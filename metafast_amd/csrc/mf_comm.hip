// mf_comm.hip -- the exchanges of the multi-GPU path, behind the C-ABI (round 6).
//
// The reference is ONE JVM: ComponentCutterMain.runImpl (src/tools/ComponentCutterMain.java:78-114) joins the sequences of all libraries in one
// map and IOUtils.run (src/io/IOUtils.java:846-862) waits for its worker threads on a latch.  With one library per GPU the join becomes an
// exchange; until round 5 it was driven from Python over torch.distributed (metafast_amd/pipeline.py), so a Java host got no collective
// from this library and the shipped metafast.sh cut components on one device.  Here:
//
//   mf_comm                 a communicator: rank, world, three primitives on buffers in HBM -- a gather of a few host integers, an
//                           all-gather and an all-to-all of slices of known sizes -- and three transports:
//     local                 the ranks are THREADS of one process, one context (= device, stream) each: metafast.sh --devices a,b,...  A rank
//                           copies its slice STRAIGHT into every peer's receive buffer (hipMemcpyPeerAsync over xGMI with peer access
//                           enabled, a device-to-device copy when two ranks share a GPU): the direct all-gather SURVEY.md 5 / 8(e) asks
//                           for on a fully connected xGMI node -- no ring, 7 links busy at once;
//     rccl                  one process per GPU: RCCL looked up at run time (dlopen, as roctx and libbz2 are), slices travel as grouped
//                           ncclSend / ncclRecv pairs -- RCCL's point-to-point path, again every link at once instead of a ring;
//     external              the host brings the three primitives (function pointers): MPI under a Java host, torch.distributed in the tests
//                           (gloo on a box with one GPU);
//   mf_comm_gather_sequences, mf_cut_components_sharded, mf_features_allgather
//                           the path's exchange steps themselves: the protocol of the sharded cutter (mf_dcc_* in mf_cc.hip) -- shard
//                           sizes, neighbour queries and answers, per threshold level the half pairs / completed pairs / per-component
//                           records, the members -- runs here, in C++, on the context's stream and arena.
//
// Failures are agreed on: a rank whose library call fails keeps the exchanges going with zeroed buffers of the agreed sizes until the next
// integer gather, which carries every rank's status; all ranks then return MF_ERR_TOGETHER at the same point, and the caller can take the
// replicated path on all of them.  A rank that cannot even do that (no memory for an exchange buffer) aborts the communicator: the local
// transport wakes its peers with an error instead of leaving them in a barrier.
#include <dlfcn.h>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <numeric>
#include "mf_common.h"

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct mf_comm {
    mf_ctx *ctx = nullptr;
    int rank = 0, world = 1;
    uint64_t n_coll = 0, bytes_in = 0; double seconds = 0;
    virtual ~mf_comm() {}
    virtual const char *kind() const = 0;
    virtual int gather_ints_impl(const int64_t *vals, int len, int64_t *out) = 0;
    virtual int all_gather_impl(const void *d_send, void *d_recv, const uint64_t *bytes) = 0;
    virtual int all_to_all_impl(const void *d_send, const uint64_t *sb, void *d_recv, const uint64_t *rb) = 0;
    virtual void abort() {}
    // (with the accounting bench.py reports: collectives, bytes received, seconds inside)
    int gather_ints(const int64_t *vals, int len, int64_t *out) {
        const double t0 = now_s();
        const int rc = world == 1 && !force() ? (memcpy(out, vals, (size_t)len * 8), MF_OK) : gather_ints_impl(vals, len, out);
        if (world > 1 || force()) { n_coll++; bytes_in += (uint64_t)len * 8 * world; seconds += now_s() - t0; }
        if (rc < 0) abort();
        return rc;
    }
    int all_gather(const void *d_send, void *d_recv, const uint64_t *bytes) {
        const double t0 = now_s();
        uint64_t tot = 0; for (int r = 0; r < world; r++) tot += bytes[r];
        int rc = MF_OK;
        if (world == 1 && !force()) { if (bytes[0] && d_send != d_recv) MF_HIP(hipMemcpyAsync(d_recv, d_send, bytes[0], hipMemcpyDeviceToDevice, ctx->stream)); }
        else { rc = all_gather_impl(d_send, d_recv, bytes); n_coll++; bytes_in += tot; seconds += now_s() - t0; }
        if (rc < 0) abort();
        return rc;
    }
    int all_to_all(const void *d_send, const uint64_t *sb, void *d_recv, const uint64_t *rb) {
        const double t0 = now_s();
        uint64_t tot = 0; for (int r = 0; r < world; r++) tot += rb[r];
        int rc = MF_OK;
        if (world == 1 && !force()) { if (sb[0] && d_send != d_recv) MF_HIP(hipMemcpyAsync(d_recv, d_send, sb[0], hipMemcpyDeviceToDevice, ctx->stream)); }
        else { rc = all_to_all_impl(d_send, sb, d_recv, rb); n_coll++; bytes_in += tot; seconds += now_s() - t0; }
        if (rc < 0) abort();
        return rc;
    }
    // (a communicator of ONE rank skips its transport -- unless it is told not to: MF_FORCE_DIST=1 lets a 1-GPU box run RCCL's code path)
    bool force() const { static const bool f = getenv("MF_FORCE_DIST") != nullptr; return f; }
};

// =============================================================================================
// local: the ranks are threads of one process
// =============================================================================================
struct local_group {
    int world = 0, alive = 0;
    std::mutex m; std::condition_variable cv; int arrived = 0; uint64_t gen = 0; bool aborted = false;
    std::vector<void *> recv; std::vector<const uint64_t *> cnt; std::vector<const int64_t *> ints; std::vector<int> ilen, device;
    int barrier() {
        std::unique_lock<std::mutex> lk(m);
        if (aborted) return -1;
        const uint64_t g = gen;
        if (++arrived == world) { arrived = 0; gen++; cv.notify_all(); return 0; }
        // (ten minutes: a peer that has died without a word must not hold the others for ever)
        if (!cv.wait_for(lk, std::chrono::seconds(600), [&] { return gen != g || aborted; })) { aborted = true; cv.notify_all(); return -1; }
        return aborted ? -1 : 0;
    }
    void abort() { std::lock_guard<std::mutex> lk(m); aborted = true; cv.notify_all(); }
};
struct mf_comm_local : mf_comm {
    std::shared_ptr<local_group> g;
    const char *kind() const override { return "local"; }
    void abort() override { g->abort(); }
    int bar() { return g->barrier() < 0 ? mf_set_error("mf_comm (local): a peer rank has aborted the exchange") : MF_OK; }
    // n bytes from this rank's src to peer p's dst, on this rank's stream
    int push(void *dst, int p, const void *src, uint64_t n) {
        if (!n) return MF_OK;
        if (g->device[p] == g->device[rank]) MF_HIP(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, ctx->stream));
        else MF_HIP(hipMemcpyPeerAsync(dst, g->device[p], src, g->device[rank], n, ctx->stream));
        return MF_OK;
    }
    int gather_ints_impl(const int64_t *vals, int len, int64_t *out) override {
        g->ints[rank] = vals; g->ilen[rank] = len;
        MF_TRY(bar());
        int rc = MF_OK;
        for (int r = 0; r < world; r++) {
            if (g->ilen[r] != len) { rc = mf_set_error("mf_comm (local): rank %d gathers %d integers, rank %d %d", r, g->ilen[r], rank, len); break; }
            memcpy(out + (size_t)r * len, g->ints[r], (size_t)len * 8);
        }
        MF_TRY(bar());
        return rc;
    }
    int all_gather_impl(const void *d_send, void *d_recv, const uint64_t *bytes) override {
        MF_HIP(hipStreamSynchronize(ctx->stream));                               // my slice is complete, my receive buffer is nobody's any more
        g->recv[rank] = d_recv;
        MF_TRY(bar());
        uint64_t off = 0; for (int r = 0; r < rank; r++) off += bytes[r];
        int rc = MF_OK;
        for (int q = 0; q < world && rc == MF_OK; q++) { const int p = (rank + q) % world; rc = push((char *)g->recv[p] + off, p, d_send, bytes[rank]); }      // (every rank starts with another peer)
        if (rc == MF_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = mf_set_error("mf_comm (local): all-gather copies failed: %s", hipGetErrorString(hipGetLastError()));
        if (rc < 0) { g->abort(); return rc; }
        MF_TRY(bar());                                                           // every peer's slice has landed here
        return MF_OK;
    }
    int all_to_all_impl(const void *d_send, const uint64_t *sb, void *d_recv, const uint64_t *rb) override {
        MF_HIP(hipStreamSynchronize(ctx->stream));
        g->recv[rank] = d_recv; g->cnt[rank] = rb;
        MF_TRY(bar());
        int rc = MF_OK;
        std::vector<uint64_t> soff((size_t)world + 1, 0);
        for (int d = 0; d < world; d++) soff[d + 1] = soff[d] + sb[d];
        for (int q = 0; q < world && rc == MF_OK; q++) {
            const int d = (rank + q) % world;
            const uint64_t *prb = g->cnt[d];
            if (prb[rank] != sb[d]) { rc = mf_set_error("mf_comm (local): rank %d sends %llu bytes to rank %d, which expects %llu", rank, (unsigned long long)sb[d], d, (unsigned long long)prb[rank]); break; }
            uint64_t roff = 0; for (int r = 0; r < rank; r++) roff += prb[r];
            rc = push((char *)g->recv[d] + roff, d, (const char *)d_send + soff[d], sb[d]);
        }
        if (rc == MF_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = mf_set_error("mf_comm (local): all-to-all copies failed: %s", hipGetErrorString(hipGetLastError()));
        if (rc < 0) { g->abort(); return rc; }
        MF_TRY(bar());
        return MF_OK;
    }
};
extern "C" int mf_comm_create_local(mf_ctx *const *ctxs, int n, mf_comm **out) {
    if (!ctxs || !out || n < 1) return mf_set_error("mf_comm_create_local: bad argument");
    for (int r = 0; r < n; r++) if (!ctxs[r]) return mf_set_error("mf_comm_create_local: context %d is NULL", r);
    auto g = std::make_shared<local_group>();
    g->world = g->alive = n;
    g->recv.assign(n, nullptr); g->cnt.assign(n, nullptr); g->ints.assign(n, nullptr); g->ilen.assign(n, 0); g->device.resize(n);
    for (int r = 0; r < n; r++) g->device[r] = ctxs[r]->device;
    // peer access between every pair of distinct devices: the copies then cross xGMI directly (without it the runtime stages them through the host)
    int cur = 0; (void)hipGetDevice(&cur);
    for (int a = 0; a < n; a++)
        for (int b = 0; b < n; b++) {
            if (g->device[a] == g->device[b]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, g->device[a], g->device[b]) == hipSuccess && can && hipSetDevice(g->device[a]) == hipSuccess)
                if (hipDeviceEnablePeerAccess(g->device[b], 0) != hipSuccess) (void)hipGetLastError();      // (already enabled, or refused: the copies then go the runtime's way)
        }
    (void)hipSetDevice(cur);
    for (int r = 0; r < n; r++) {
        mf_comm_local *c = new mf_comm_local();
        c->ctx = ctxs[r]; c->rank = r; c->world = n; c->g = g;
        out[r] = c;
    }
    return MF_OK;
}

// =============================================================================================
// rccl: one process per GPU; librccl looked up at run time
// =============================================================================================
struct rccl_uid { char b[128]; };
struct rccl_api {
    void *h = nullptr;
    int (*GetUniqueId)(rccl_uid *) = nullptr;
    int (*CommInitRank)(void **, int, rccl_uid, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*CommAbort)(void *) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
static rccl_api *rccl() {
    static rccl_api api; static std::once_flag once; static bool ok = false;
    std::call_once(once, [] {
        // (the copy a host such as PyTorch has loaded already, else the system's)
        const char *names[] = {getenv("MF_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *nm : names) {
            if (!nm || !*nm) continue;
            void *h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
            if (!h) h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
            if (h) { api.h = h; break; }
        }
        if (!api.h) return;
#define MF_SYM(field, name) *(void **)(&api.field) = dlsym(api.h, name)
        MF_SYM(GetUniqueId, "ncclGetUniqueId"); MF_SYM(CommInitRank, "ncclCommInitRank"); MF_SYM(CommDestroy, "ncclCommDestroy"); MF_SYM(CommAbort, "ncclCommAbort");
        MF_SYM(GroupStart, "ncclGroupStart"); MF_SYM(GroupEnd, "ncclGroupEnd"); MF_SYM(Send, "ncclSend"); MF_SYM(Recv, "ncclRecv"); MF_SYM(AllGather, "ncclAllGather");
        MF_SYM(GetErrorString, "ncclGetErrorString");
#undef MF_SYM
        ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.GroupStart && api.GroupEnd && api.Send && api.Recv && api.AllGather;
    });
    return ok ? &api : nullptr;
}
#define MF_NCCL(call) do { const int e__ = (call); if (e__ != 0) return mf_set_error("%s failed: %s", #call, R->GetErrorString ? R->GetErrorString(e__) : "RCCL error"); } while (0)
struct mf_comm_rccl : mf_comm {
    rccl_api *R = nullptr; void *comm = nullptr;
    mf_buf<int64_t> istage; int64_t *hstage = nullptr; size_t icap = 0;       // integer gathers: device staging (in | out), pinned host mirror
    const char *kind() const override { return "rccl"; }
    ~mf_comm_rccl() override { if (comm && R) R->CommDestroy(comm); if (hstage) (void)hipHostFree(hstage); }
    void abort() override { if (comm && R && R->CommAbort) { R->CommAbort(comm); comm = nullptr; } }
    int gather_ints_impl(const int64_t *vals, int len, int64_t *out) override {
        const size_t need = (size_t)len * ((size_t)world + 1);
        if (need > icap) {
            istage.reset(); if (hstage) { (void)hipHostFree(hstage); hstage = nullptr; }
            icap = std::max<size_t>(need, 4096);
            MF_TRY(istage.alloc(ctx, icap));
            MF_HIP(hipHostMalloc((void **)&hstage, icap * 8));
        }
        memcpy(hstage, vals, (size_t)len * 8);
        MF_HIP(hipMemcpyAsync(istage.p, hstage, (size_t)len * 8, hipMemcpyHostToDevice, ctx->stream));
        MF_NCCL(R->AllGather(istage.p, istage.p + len, (size_t)len, 4 /* ncclInt64 */, comm, ctx->stream));
        MF_HIP(hipMemcpyAsync(hstage + len, istage.p + len, (size_t)len * 8 * world, hipMemcpyDeviceToHost, ctx->stream));
        MF_HIP(hipStreamSynchronize(ctx->stream));
        memcpy(out, hstage + len, (size_t)len * 8 * world);
        return MF_OK;
    }
    // slices as point-to-point pairs inside one group: every peer's link carries its slice at the same time (a ring all-gather on xGMI's full
    // mesh would put all of it through one link per step)
    int all_gather_impl(const void *d_send, void *d_recv, const uint64_t *bytes) override {
        std::vector<uint64_t> off((size_t)world + 1, 0);
        for (int r = 0; r < world; r++) off[r + 1] = off[r] + bytes[r];
        if (bytes[rank] && (const char *)d_send != (char *)d_recv + off[rank]) MF_HIP(hipMemcpyAsync((char *)d_recv + off[rank], d_send, bytes[rank], hipMemcpyDeviceToDevice, ctx->stream));
        if (world == 1) return MF_OK;
        MF_NCCL(R->GroupStart());
        for (int q = 1; q < world; q++) {
            const int to = (rank + q) % world, from = (rank - q + world) % world;
            if (bytes[rank]) MF_NCCL(R->Send(d_send, bytes[rank], 0 /* ncclInt8 */, to, comm, ctx->stream));
            if (bytes[from]) MF_NCCL(R->Recv((char *)d_recv + off[from], bytes[from], 0, from, comm, ctx->stream));
        }
        MF_NCCL(R->GroupEnd());
        return MF_OK;
    }
    int all_to_all_impl(const void *d_send, const uint64_t *sb, void *d_recv, const uint64_t *rb) override {
        std::vector<uint64_t> so((size_t)world + 1, 0), ro((size_t)world + 1, 0);
        for (int r = 0; r < world; r++) { so[r + 1] = so[r] + sb[r]; ro[r + 1] = ro[r] + rb[r]; }
        if (sb[rank] != rb[rank]) return mf_set_error("mf_comm (rccl): rank %d sends itself %llu bytes and expects %llu", rank, (unsigned long long)sb[rank], (unsigned long long)rb[rank]);
        if (sb[rank]) MF_HIP(hipMemcpyAsync((char *)d_recv + ro[rank], (const char *)d_send + so[rank], sb[rank], hipMemcpyDeviceToDevice, ctx->stream));
        if (world == 1) return MF_OK;
        MF_NCCL(R->GroupStart());
        for (int q = 1; q < world; q++) {
            const int to = (rank + q) % world, from = (rank - q + world) % world;
            if (sb[to]) MF_NCCL(R->Send((const char *)d_send + so[to], sb[to], 0, to, comm, ctx->stream));
            if (rb[from]) MF_NCCL(R->Recv((char *)d_recv + ro[from], rb[from], 0, from, comm, ctx->stream));
        }
        MF_NCCL(R->GroupEnd());
        return MF_OK;
    }
};
extern "C" int mf_comm_rccl_id(void *id128) {
    if (!id128) return mf_set_error("mf_comm_rccl_id: NULL argument");
    rccl_api *R = rccl();
    if (!R) return mf_set_error("mf_comm_rccl_id: librccl not found (tried librccl.so.1, librccl.so, /opt/rocm/lib/librccl.so.1; MF_RCCL_LIB names another)");
    MF_NCCL(R->GetUniqueId((rccl_uid *)id128));
    return MF_OK;
}
extern "C" int mf_comm_create_rccl(mf_ctx *ctx, const void *id128, int rank, int world, mf_comm **out) {
    if (!ctx || !id128 || !out || world < 1 || rank < 0 || rank >= world) return mf_set_error("mf_comm_create_rccl: bad argument");
    *out = nullptr;
    rccl_api *R = rccl();
    if (!R) return mf_set_error("mf_comm_create_rccl: librccl not found (tried librccl.so.1, librccl.so, /opt/rocm/lib/librccl.so.1; MF_RCCL_LIB names another)");
    MF_HIP(hipSetDevice(ctx->device));
    auto c = std::make_unique<mf_comm_rccl>();
    c->ctx = ctx; c->rank = rank; c->world = world; c->R = R;
    rccl_uid id; memcpy(&id, id128, sizeof id);
    MF_NCCL(R->CommInitRank(&c->comm, world, id, rank));
    *out = c.release();
    return MF_OK;
}

// =============================================================================================
// external: the host's own primitives
// =============================================================================================
struct mf_comm_external : mf_comm {
    mf_comm_ops ops{}; void *user = nullptr;
    const char *kind() const override { return "external"; }
    int chk(int rc, const char *what) { return rc < 0 ? mf_set_error("mf_comm (external): the host's %s failed (%d)", what, rc) : MF_OK; }
    int gather_ints_impl(const int64_t *vals, int len, int64_t *out) override { return chk(ops.gather_ints(user, vals, len, out), "gather_ints"); }
    int all_gather_impl(const void *d_send, void *d_recv, const uint64_t *bytes) override {
        MF_HIP(hipStreamSynchronize(ctx->stream));                               // (the host's transport runs on a stream of its own)
        return chk(ops.all_gather(user, d_send, d_recv, bytes), "all_gather");
    }
    int all_to_all_impl(const void *d_send, const uint64_t *sb, void *d_recv, const uint64_t *rb) override {
        MF_HIP(hipStreamSynchronize(ctx->stream));
        return chk(ops.all_to_all(user, d_send, sb, d_recv, rb), "all_to_all");
    }
};
extern "C" int mf_comm_create_external(mf_ctx *ctx, int rank, int world, const mf_comm_ops *ops, void *user, mf_comm **out) {
    if (!ctx || !ops || !out || world < 1 || rank < 0 || rank >= world || !ops->gather_ints || !ops->all_gather || !ops->all_to_all)
        return mf_set_error("mf_comm_create_external: bad argument");
    mf_comm_external *c = new mf_comm_external();
    c->ctx = ctx; c->rank = rank; c->world = world; c->ops = *ops; c->user = user;
    *out = c;
    return MF_OK;
}

extern "C" void mf_comm_destroy(mf_comm *c) { delete c; }
mf_ctx *mf_comm_ctx(mf_comm *c) { return c->ctx; }
// every rank says whether it is fine: 0 when all are, else < 0 on every rank (a barrier that carries a status)
int mf_comm_agree(mf_comm *c, int ok) {
    std::vector<int64_t> all((size_t)c->world);
    const int64_t v = ok ? 1 : 0;
    if (c->gather_ints(&v, 1, all.data()) < 0) return MF_ERR;
    std::string bad;
    for (int r = 0; r < c->world; r++) if (!all[r]) bad += (bad.empty() ? "" : ", ") + std::to_string(r);
    if (!bad.empty()) { mf_set_error("rank(s) %s failed", bad.c_str()); return MF_ERR_TOGETHER; }
    return MF_OK;
}
extern "C" int mf_comm_rank(const mf_comm *c) { return c ? c->rank : mf_set_error("mf_comm_rank: NULL handle"); }
extern "C" int mf_comm_world(const mf_comm *c) { return c ? c->world : mf_set_error("mf_comm_world: NULL handle"); }
extern "C" const char *mf_comm_kind(const mf_comm *c) { return c ? c->kind() : ""; }
extern "C" int64_t mf_comm_stat(const mf_comm *c, const char *name) {
    if (!c || !name) return -1;
    if (!strcmp(name, "collectives")) return (int64_t)c->n_coll;
    if (!strcmp(name, "bytes_in")) return (int64_t)c->bytes_in;
    if (!strcmp(name, "us")) return (int64_t)(c->seconds * 1e6);
    return -1;
}
extern "C" int mf_comm_reset_stats(mf_comm *c) { if (!c) return mf_set_error("mf_comm_reset_stats: NULL handle"); c->n_coll = c->bytes_in = 0; c->seconds = 0; return MF_OK; }
extern "C" int mf_comm_gather_ints(mf_comm *c, const int64_t *vals, int n, int64_t *out) {
    if (!c || !vals || !out || n < 1) return mf_set_error("mf_comm_gather_ints: bad argument");
    return c->gather_ints(vals, n, out);
}
extern "C" int mf_comm_all_gather(mf_comm *c, const void *d_send, void *d_recv, const uint64_t *bytes_per_rank) {
    if (!c || !bytes_per_rank) return mf_set_error("mf_comm_all_gather: bad argument");
    MF_HIP(hipSetDevice(c->ctx->device));
    return c->all_gather(d_send, d_recv, bytes_per_rank);
}
extern "C" int mf_comm_all_to_all(mf_comm *c, const void *d_send, const uint64_t *send_bytes, void *d_recv, const uint64_t *recv_bytes) {
    if (!c || !send_bytes || !recv_bytes) return mf_set_error("mf_comm_all_to_all: bad argument");
    MF_HIP(hipSetDevice(c->ctx->device));
    return c->all_to_all(d_send, send_bytes, d_recv, recv_bytes);
}

// =============================================================================================
// the path's exchange steps
// =============================================================================================
// A rank's status rides in front of every integer gather: all ranks learn of a failure at the same point of the protocol.
struct agree {
    mf_comm *cm; bool err = false; std::string msg;
    void fail() { if (!err) { err = true; msg = mf_last_error(); } }
    // vals[n] of every rank -> out[world][n]; MF_ERR_TOGETHER when some rank has failed (every rank returns it here)
    int gather(const int64_t *vals, int n, std::vector<int64_t> &out) {
        const int W = cm->world;
        std::vector<int64_t> in((size_t)n + 1, 0), all(((size_t)n + 1) * W);
        in[0] = err ? 0 : 1;
        if (!err) for (int i = 0; i < n; i++) in[i + 1] = vals[i];
        if (cm->gather_ints(in.data(), n + 1, all.data()) < 0) { cm->abort(); return MF_ERR; }
        out.assign((size_t)n * W, 0);
        std::string bad;
        for (int r = 0; r < W; r++) {
            if (!all[(size_t)r * (n + 1)]) bad += (bad.empty() ? "" : ", ") + std::to_string(r);
            for (int i = 0; i < n; i++) out[(size_t)r * n + i] = all[(size_t)r * (n + 1) + 1 + i];
        }
        if (!bad.empty()) { mf_set_error("rank(s) %s failed%s%s", bad.c_str(), err ? ": " : "", err ? msg.c_str() : ""); return MF_ERR_TOGETHER; }
        return MF_OK;
    }
};
// an exchange buffer of n 8-byte words; zeroed on a rank that has failed (what its healthy peers read from it until the next status gather
// must be in-range indices and ranks); no memory: the communicator is aborted -- nothing can be agreed on without buffers
static int xbuf(agree &A, mf_buf<int64_t> &b, uint64_t n) {
    mf_ctx *ctx = A.cm->ctx;
    if (b.alloc(ctx, n ? n : 1) < 0) { A.cm->abort(); return MF_ERR; }
    if (A.err && hipMemsetAsync(b.p, 0, (n ? n : 1) * 8, ctx->stream) != hipSuccess) { A.cm->abort(); return mf_set_error("mf_comm: memset failed"); }
    return MF_OK;
}
#define MF_CALL(A, expr) do { if (!(A).err && (expr) < 0) (A).fail(); } while (0)
#define MF_X(expr) do { const int r__ = (expr); if (r__ < 0) { if (r__ != MF_ERR_TOGETHER) A.cm->abort(); return r__; } } while (0)

__global__ void k_comm_rebase(uint64_t *__restrict__ off, uint64_t n, uint64_t add) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) off[i] += add;
}
static inline unsigned xgrid(uint64_t n) { return (unsigned)std::min<uint64_t>((n + 255) / 256, 0x7FFFFFFFull); }

// every rank's sequences (unitigs) on every rank, rank after rank: IOUtils.loadReads over the .seq.fasta files of ALL libraries
// (src/tools/ComponentCutterMain.java:81) where the libraries live on different GPUs
static int gather_sequences(agree &A, const void *d_bases, const void *d_offsets, uint64_t n_seqs, uint64_t n_bases, mf_reads **out) {
    mf_comm *cm = A.cm; mf_ctx *ctx = cm->ctx; const int W = cm->world; hipStream_t st = ctx->stream;
    const int64_t mine[2] = {(int64_t)n_seqs, (int64_t)n_bases};
    std::vector<int64_t> all;
    MF_X(A.gather(mine, 2, all));
    std::vector<uint64_t> bb(W), ob(W); uint64_t ns = 0, nb = 0;
    for (int r = 0; r < W; r++) { ob[r] = (uint64_t)all[2 * r] * 8; bb[r] = (uint64_t)all[2 * r + 1]; ns += (uint64_t)all[2 * r]; nb += bb[r]; }
    auto R = std::make_unique<mf_reads>();
    R->ctx = ctx; R->n = ns; R->n_bases = nb;
    void *p = nullptr;
    R->bases_bytes = nb + 64; R->offsets_bytes = (ns + 1) * 8;
    if (mf_alloc(ctx, R->bases_bytes, &p) < 0) { cm->abort(); return MF_ERR; }
    R->d_bases = (uint8_t *)p;
    if (mf_alloc(ctx, R->offsets_bytes, &p) < 0) { mf_release(ctx, R->d_bases, R->bases_bytes); R->d_bases = nullptr; cm->abort(); return MF_ERR; }
    R->d_offsets = (uint64_t *)p;
    auto fatal = [&](int rc) { mf_reads_destroy(R.release()); cm->abort(); return rc; };
    if (hipMemsetAsync(R->d_bases + nb, 0, 64, st) != hipSuccess) return fatal(mf_set_error("mf_comm: memset failed"));
    if (cm->all_gather(d_bases, R->d_bases, bb.data()) < 0) return fatal(MF_ERR);
    if (cm->all_gather(d_offsets, R->d_offsets, ob.data()) < 0) return fatal(MF_ERR);      // (every rank's first n offsets: its last one is the next rank's base)
    uint64_t so = 0, sb = 0;
    for (int r = 0; r < W; r++) {
        const uint64_t n = (uint64_t)all[2 * r];
        if (n && sb) k_comm_rebase<<<xgrid(n), 256, 0, st>>>(R->d_offsets + so, n, sb);
        so += n; sb += bb[r];
    }
    if (hipMemcpyAsync(R->d_offsets + ns, &nb, 8, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return fatal(mf_set_error("mf_comm: gather of the sequences failed: %s", hipGetErrorString(hipGetLastError())));
    *out = R.release();
    return MF_OK;
}
extern "C" int mf_comm_gather_sequences(mf_comm *cm, const void *d_bases, const void *d_offsets, uint64_t n_seqs, uint64_t n_bases, mf_reads **out) {
    if (!cm || !out || (n_seqs && (!d_bases || !d_offsets))) return mf_set_error("mf_comm_gather_sequences: bad argument");
    *out = nullptr;
    MF_HIP(hipSetDevice(cm->ctx->device));
    agree A{cm};
    return gather_sequences(A, d_bases, d_offsets, n_seqs, n_bases, out);
}

__global__ void k_comm_min_u64(const unsigned long long *__restrict__ all, uint64_t n, int world, unsigned long long *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long m = all[i];
    for (int r = 1; r < world; r++) { const unsigned long long v = all[(uint64_t)r * n + i]; m = v < m ? v : m; }
    out[i] = m;
}

// ComponentsBuilder.splitStrategy (src/algo/ComponentsBuilder.java:24-32, 58-270) with every rank owning a shard of the cutter table:
// the exchange protocol around mf_dcc_* (mf_cc.hip; rounds 2-5 ran it from metafast_amd/pipeline.py).  Collectives: 1 (sequences' sizes) + 2
// (sequences) + 1 + 3 (shard sizes; query counts, queries, answers), per threshold level 2 integer gathers (half pairs per destination + the
// level before's oversize count; records per rank) + 1 all-to-all (half pairs) + 2 all-gathers (completed pairs; per-component records) --
// from the gathered records EVERY rank derives all kept components and the number of oversize ones itself --, and 1 + 2 + 1 + 1 at the end
// (members' counts, their k-mers and runs, the components' smallest k-mers, the last status).
static int cut_sharded(agree &A, mf_table *shard, int k, int b1, int b2, mf_comps **out, uint64_t *info) {
    mf_comm *cm = A.cm; mf_ctx *ctx = cm->ctx; const int W = cm->world, me = cm->rank; hipStream_t st = ctx->stream;
    std::vector<int64_t> g;
    // ---- global vertex ids
    { const int64_t v = shard ? (int64_t)shard->n : 0; MF_X(A.gather(&v, 1, g)); }
    std::vector<uint32_t> base((size_t)W + 1, 0);
    { uint64_t acc = 0; for (int r = 0; r < W; r++) { acc += (uint64_t)g[r]; if (acc >= 0xFFFFFFFFull) return mf_set_error("components: more than 2^32 vertices over all ranks is not supported"); base[r + 1] = (uint32_t)acc; } }
    mf_dcc *D = nullptr;
    MF_CALL(A, mf_dcc_create(ctx, shard, me, W, base.data(), &D));
    struct dguard { mf_dcc *&d; ~dguard() { if (d) mf_dcc_destroy(d); } } dg{D};
    std::vector<uint64_t> cnt(W), sbytes(W), rbytes(W);
    // ---- neighbours in other shards: queries to their owners, answers back
    {
        std::fill(cnt.begin(), cnt.end(), 0);
        MF_CALL(A, mf_dcc_queries(D, cnt.data()));
        std::vector<int64_t> v(cnt.begin(), cnt.end());
        MF_X(A.gather(v.data(), W, g));                                           // g[src][dst] = queries
        uint64_t nq = 0, na = 0;
        for (int r = 0; r < W; r++) { sbytes[r] = 16ull * (uint64_t)g[(size_t)me * W + r]; rbytes[r] = 16ull * (uint64_t)g[(size_t)r * W + me]; nq += sbytes[r] / 16; na += rbytes[r] / 16; }
        mf_buf<int64_t> q, rq, a, ra;
        MF_X(xbuf(A, q, 2 * nq)); MF_X(xbuf(A, rq, 2 * na)); MF_X(xbuf(A, a, 2 * na)); MF_X(xbuf(A, ra, 2 * nq));
        MF_CALL(A, mf_dcc_queries_fill(D, q.p));
        MF_X(cm->all_to_all(q.p, sbytes.data(), rq.p, rbytes.data()));
        MF_CALL(A, mf_dcc_answer(D, rq.p, na, a.p));
        MF_X(cm->all_to_all(a.p, rbytes.data(), ra.p, sbytes.data()));
        MF_CALL(A, mf_dcc_set_answers(D, ra.p, nq));
        if (info) info[1] = nq;
    }
    // ---- threshold levels (ComponentsBuilder.java:86-150)
    struct kept_t { std::vector<uint32_t> root, size; std::vector<int64_t> weight; std::vector<int32_t> thr; } K;
    int64_t n_big = -1;
    int levels = 0;
    for (int thr = 1; thr < (1 << 16); thr++) {
        // (the gather that opens a level also closes the one before: it carries that level's oversize count -- the same on every rank -- and the
        // status of the calls since the last gather, so all ranks leave the loop, or give up, together)
        std::vector<int64_t> v((size_t)W + 1, 0);
        v[0] = n_big;
        if (n_big != 0) { std::fill(cnt.begin(), cnt.end(), 0); MF_CALL(A, mf_dcc_level_local(D, cnt.data())); for (int r = 0; r < W; r++) v[r + 1] = (int64_t)cnt[r]; }
        MF_X(A.gather(v.data(), W + 1, g));
        for (int r = 1; r < W; r++) if (g[(size_t)r * (W + 1)] != g[0]) { mf_set_error("sharded component cutter: the ranks disagree on a level's oversize components"); return MF_ERR_TOGETHER; }
        if (g[0] == 0) break;
        auto pm = [&](int src, int dst) { return (uint64_t)g[(size_t)src * (W + 1) + 1 + dst]; };
        uint64_t nsend = 0, nrecv = 0, ntot = 0;
        std::vector<uint64_t> col(W, 0);
        for (int r = 0; r < W; r++) { sbytes[r] = 8 * pm(me, r); rbytes[r] = 8 * pm(r, me); nsend += pm(me, r); nrecv += pm(r, me); for (int s = 0; s < W; s++) col[r] += 8 * pm(s, r); ntot += col[r] / 8; }
        mf_buf<int64_t> hp, rp, allp;
        MF_X(xbuf(A, hp, nsend)); MF_X(xbuf(A, rp, nrecv)); MF_X(xbuf(A, allp, ntot));
        MF_CALL(A, mf_dcc_pairs_fill(D, hp.p));
        MF_X(cm->all_to_all(hp.p, sbytes.data(), rp.p, rbytes.data()));
        if (nrecv) MF_CALL(A, mf_dcc_pairs_complete(D, rp.p, nrecv));
        MF_X(cm->all_gather(rp.p, allp.p, col.data()));
        uint64_t n_stats = 0;
        MF_CALL(A, mf_dcc_merge(D, allp.p, ntot, &n_stats));
        hp.reset(); rp.reset(); allp.reset();
        { const int64_t v1 = A.err ? 0 : (int64_t)n_stats; MF_X(A.gather(&v1, 1, g)); }
        std::vector<uint64_t> seg((size_t)W + 1, 0), sb16(W);
        for (int r = 0; r < W; r++) { seg[r + 1] = seg[r] + (uint64_t)g[r]; sb16[r] = 16ull * (uint64_t)g[r]; }
        mf_buf<int64_t> stb, alls;
        MF_X(xbuf(A, stb, 2 * (uint64_t)g[me])); MF_X(xbuf(A, alls, 2 * seg[W]));
        MF_CALL(A, mf_dcc_stats_fill(D, stb.p));
        MF_X(cm->all_gather(stb.p, alls.p, sb16.data()));
        uint64_t n_kept = 0, nb_ = 0;
        MF_CALL(A, mf_dcc_classify(D, alls.p, seg[W], seg.data(), seg[me], (uint64_t)g[me], b1, b2, thr, &n_kept, &nb_));
        if (A.err) { n_kept = 0; nb_ = 1; }                                         // (goes on to the next gather, which tells everybody)
        n_big = (int64_t)nb_;
        if (n_kept) {
            mf_buf<int64_t> kb;
            std::vector<int64_t> h(2 * n_kept);
            if (kb.alloc(ctx, 2 * n_kept) < 0) A.fail();
            MF_CALL(A, mf_dcc_kept_fill(D, kb.p));
            if (!A.err && (hipMemcpyAsync(h.data(), kb.p, 16 * n_kept, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)) { mf_set_error("sharded component cutter: copy failed"); A.fail(); }
            if (!A.err) {
                std::vector<uint32_t> order(n_kept);
                std::iota(order.begin(), order.end(), 0u);
                std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return (uint32_t)h[2 * x] < (uint32_t)h[2 * y]; });      // by root: the order every rank agrees on
                for (uint32_t i : order) { K.root.push_back((uint32_t)h[2 * i]); K.size.push_back((uint32_t)((uint64_t)h[2 * i] >> 32)); K.weight.push_back(h[2 * i + 1]); K.thr.push_back(thr); }
            }
        }
        levels = thr;
    }
    // ---- members of the kept components, everywhere: 8 bytes per member (the k-mers, sorted by component on the rank) + one (root, count) record per
    // component and rank
    uint64_t nm = 0, nr = 0;
    MF_CALL(A, mf_dcc_members_grouped(D, &nm, &nr));
    { const int64_t v2[2] = {(int64_t)nm, (int64_t)nr}; MF_X(A.gather(v2, 2, g)); }
    std::vector<uint64_t> mb(W), rb8(W); uint64_t tm = 0, tr = 0;
    for (int r = 0; r < W; r++) { mb[r] = 8ull * (uint64_t)g[2 * r]; rb8[r] = 8ull * (uint64_t)g[2 * r + 1]; tm += (uint64_t)g[2 * r]; tr += (uint64_t)g[2 * r + 1]; }
    mf_buf<int64_t> mk, mr, allmk, allmr, mn, allmn;
    const uint64_t nkept = K.root.size();
    MF_X(xbuf(A, mk, (uint64_t)g[2 * me])); MF_X(xbuf(A, mr, (uint64_t)g[2 * me + 1])); MF_X(xbuf(A, allmk, tm)); MF_X(xbuf(A, allmr, tr));
    MF_CALL(A, mf_dcc_members_grouped_fill(D, mk.p, mr.p));
    MF_X(cm->all_gather(mk.p, allmk.p, mb.data()));
    MF_X(cm->all_gather(mr.p, allmr.p, rb8.data()));
    MF_X(xbuf(A, mn, nkept)); MF_X(xbuf(A, allmn, nkept * (uint64_t)W));
    MF_CALL(A, mf_dcc_minkeys(D, K.root.data(), nkept, mn.p));
    { std::vector<uint64_t> nb8(W, 8 * nkept); MF_X(cm->all_gather(mn.p, allmn.p, nb8.data())); }
    std::vector<uint64_t> minkey(nkept ? nkept : 1);
    if (nkept) {
        k_comm_min_u64<<<xgrid(nkept), 256, 0, st>>>((const unsigned long long *)allmn.p, nkept, W, (unsigned long long *)mn.p);
        if (hipMemcpyAsync(minkey.data(), mn.p, 8 * nkept, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { mf_set_error("sharded component cutter: copy failed"); A.fail(); }
    }
    mf_comps *C = nullptr;
    MF_CALL(A, mf_dcc_finish_grouped(D, allmk.p, tm, allmr.p, tr, K.root.data(), K.size.data(), K.weight.data(), K.thr.data(), minkey.data(), nkept, &C));
    { const int64_t v3 = C ? (int64_t)C->n : 0; const int rc = A.gather(&v3, 1, g); if (rc < 0) { if (C) mf_comps_destroy(C); if (rc != MF_ERR_TOGETHER) cm->abort(); return rc; } }      // (the last status: every rank has its components, or none keeps them)
    if (info) { info[0] = (uint64_t)levels; info[2] = tm; info[3] = base[W]; }
    *out = C;
    return MF_OK;
}
extern "C" int mf_cut_components_sharded(mf_comm *cm, const void *d_bases, const void *d_offsets, uint64_t n_seqs, uint64_t n_bases, int k, int min_len, int b1, int b2,
                                         mf_comps **out) {
    mf_range rng_("mf:components_sharded");
    if (!cm || !out || (n_seqs && (!d_bases || !d_offsets))) return mf_set_error("mf_cut_components_sharded: bad argument");
    *out = nullptr;
    const int W = cm->world;
    if (W & (W - 1) || W > 64) return mf_set_error("mf_cut_components_sharded: the number of ranks must be a power of two <= 64 (%d)", W);
    if (k < 20 || k > 31) return mf_set_error("mf_cut_components_sharded: 20 <= k <= 31 (the shards are minimizer partitions)");
    mf_ctx *ctx = cm->ctx;
    MF_HIP(hipSetDevice(ctx->device));
    agree A{cm};
    mf_reads *all = nullptr;
    { const int rc = gather_sequences(A, d_bases, d_offsets, n_seqs, n_bases, &all); if (rc < 0) return rc; }
    struct rguard { mf_reads *r; ~rguard() { mf_reads_destroy(r); } } rg{all};
    mf_table *shard = nullptr;
    MF_CALL(A, mf_count_device_shard(ctx, all->d_bases, all->d_offsets, all->n, all->n_bases, k, min_len, cm->rank, W, &shard));
    struct tguard { mf_table *&t; ~tguard() { if (t) mf_table_destroy(t); } } tg{shard};
    return cut_sharded(A, shard, k, b1, b2, out, nullptr);
}
// the same on a shard the caller has counted (mf_count_device_shard): tests drive failures and options through it; info (may be NULL):
// [0] threshold levels, [1] this rank's queries, [2] members over all ranks, [3] vertices over all ranks
extern "C" int mf_cut_components_of_shard(mf_comm *cm, mf_table *shard, int k, int b1, int b2, mf_comps **out, uint64_t *info) {
    mf_range rng_("mf:components_sharded");
    if (!cm || !out) return mf_set_error("mf_cut_components_of_shard: bad argument");
    *out = nullptr;
    MF_HIP(hipSetDevice(cm->ctx->device));
    agree A{cm};
    if (!shard) { mf_set_error("no shard"); A.fail(); }
    return cut_sharded(A, shard, k, b1, b2, out, info);
}

// The feature vectors of all ranks' samples, rank after rank, on every rank: the rows DistanceMatrixCalculatorMain reads back from the
// .vec files (src/tools/DistanceMatrixCalculatorMain.java:125-138) -- north_star's "final all-gather of per-sample component feature vectors"
extern "C" int mf_features_allgather(mf_comm *cm, const int64_t *rows, uint64_t n_rows, uint64_t n_comp, int64_t *all_rows, uint64_t capacity_rows, uint64_t *n_all_rows) {
    if (!cm || !n_all_rows || (n_rows && n_comp && !rows)) return mf_set_error("mf_features_allgather: bad argument");
    mf_ctx *ctx = cm->ctx;
    MF_HIP(hipSetDevice(ctx->device));
    agree A{cm};
    std::vector<int64_t> g;
    const int64_t mine[2] = {(int64_t)n_rows, (int64_t)n_comp};
    MF_X(A.gather(mine, 2, g));
    const int W = cm->world;
    uint64_t tot = 0; std::vector<uint64_t> bytes(W);
    for (int r = 0; r < W; r++) {
        if ((uint64_t)g[2 * r + 1] != n_comp) { mf_set_error("mf_features_allgather: rank %d has vectors of %lld components, rank %d of %llu", r, (long long)g[2 * r + 1], cm->rank, (unsigned long long)n_comp); return MF_ERR_TOGETHER; }
        tot += (uint64_t)g[2 * r]; bytes[r] = (uint64_t)g[2 * r] * n_comp * 8;
    }
    *n_all_rows = tot;
    if (!all_rows) return MF_OK;
    if (capacity_rows < tot) return mf_set_error("mf_features_allgather: room for %llu rows, %llu over all ranks", (unsigned long long)capacity_rows, (unsigned long long)tot);
    mf_buf<int64_t> mineb, allb;
    MF_X(xbuf(A, mineb, n_rows * n_comp)); MF_X(xbuf(A, allb, tot * n_comp));
    if (n_rows * n_comp) MF_HIP(hipMemcpyAsync(mineb.p, rows, n_rows * n_comp * 8, hipMemcpyHostToDevice, ctx->stream));
    MF_X(cm->all_gather(mineb.p, allb.p, bytes.data()));
    if (tot * n_comp) MF_HIP(hipMemcpyAsync(all_rows, allb.p, tot * n_comp * 8, hipMemcpyDeviceToHost, ctx->stream));
    MF_HIP(hipStreamSynchronize(ctx->stream));
    return MF_OK;
}

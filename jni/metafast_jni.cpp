// metafast_jni.cpp -- JNI natives of io.HipBackend (jni/HipBackend.java): every one forwards 1:1 to the C-ABI in
// include/metafast_hip.h.  Build where a JDK exists:
//   g++ -O2 -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../include metafast_jni.cpp \
//       -L../metafast_amd/lib -lmetafast_hip -Wl,-rpath,'$ORIGIN' -o libmetafast_jni.so
// The build image of this repository has no JDK (no jni.h): without it this file compiles to nothing, and what it calls is
// tested through the same C-ABI from Python (tests/) and from the C++ driver (metafast_amd/cli).
#if __has_include(<jni.h>)
#include <jni.h>
#include <cstdint>
#include <vector>
#include "metafast_hip.h"

namespace {
// rc < 0 -> ru.ifmo.genetics.utils.tool.ExecutionFailedException(mf_last_error()), which Tool.run turns into "log the
// message, exit code 1" (itmo-assembler Tool.java:450-463)
void raise(JNIEnv *e) {
    jclass c = e->FindClass("ru/ifmo/genetics/utils/tool/ExecutionFailedException");
    if (!c) { e->ExceptionClear(); c = e->FindClass("java/lang/RuntimeException"); }
    e->ThrowNew(c, mf_last_error());
}
struct utf {                       // a Java string as UTF-8 for the duration of a call (null stays null)
    JNIEnv *e; jstring s; const char *p;
    utf(JNIEnv *env, jstring str) : e(env), s(str), p(str ? env->GetStringUTFChars(str, nullptr) : nullptr) {}
    ~utf() { if (p) e->ReleaseStringUTFChars(s, p); }
    utf(const utf &) = delete;
};
struct utf_array {
    JNIEnv *e; std::vector<jstring> js; std::vector<const char *> p;
    utf_array(JNIEnv *env, jobjectArray a) : e(env) {
        const jsize n = a ? env->GetArrayLength(a) : 0;
        js.resize(n); p.resize(n);
        for (jsize i = 0; i < n; i++) { js[i] = (jstring)env->GetObjectArrayElement(a, i); p[i] = env->GetStringUTFChars(js[i], nullptr); }
    }
    ~utf_array() { for (size_t i = 0; i < p.size(); i++) e->ReleaseStringUTFChars(js[i], p[i]); }
    utf_array(const utf_array &) = delete;
};
}

extern "C" {
JNIEXPORT jlong JNICALL Java_io_HipBackend_ctxCreate(JNIEnv *e, jclass, jint device, jint threads) {
    mf_ctx *c = nullptr;
    if (mf_ctx_create(device, threads, &c) < 0) { raise(e); return 0; }
    return (jlong)(intptr_t)c;
}
JNIEXPORT void JNICALL Java_io_HipBackend_ctxDestroy(JNIEnv *, jclass, jlong ctx) { mf_ctx_destroy((mf_ctx *)(intptr_t)ctx); }
JNIEXPORT jlong JNICALL Java_io_HipBackend_countReads(JNIEnv *e, jclass, jlong ctx, jobjectArray files, jint k, jint minLen) {
    utf_array f(e, files);
    mf_table *t = nullptr;
    if (mf_count_reads((mf_ctx *)(intptr_t)ctx, f.p.data(), (int)f.p.size(), k, minLen, &t) < 0) { raise(e); return 0; }
    return (jlong)(intptr_t)t;
}
// long[2] = { table handle, distinct k-mers before the cut }
JNIEXPORT jlongArray JNICALL Java_io_HipBackend_countReadsAbove(JNIEnv *e, jclass, jlong ctx, jobjectArray files, jint k, jint minLen, jint threshold) {
    utf_array f(e, files);
    mf_table *t = nullptr; uint64_t n_all = 0;
    if (mf_count_reads_above((mf_ctx *)(intptr_t)ctx, f.p.data(), (int)f.p.size(), k, minLen, threshold, &t, &n_all) < 0) { raise(e); return nullptr; }
    jlongArray r = e->NewLongArray(2);
    if (!r) { mf_table_destroy(t); return nullptr; }
    const jlong v[2] = {(jlong)(intptr_t)t, (jlong)n_all};
    e->SetLongArrayRegion(r, 0, 2, v);
    return r;
}
JNIEXPORT jlong JNICALL Java_io_HipBackend_ctxTrimBytes(JNIEnv *e, jclass, jlong ctx, jlong want) {
    uint64_t freed = 0;
    if (want < 0) { e->ThrowNew(e->FindClass("java/lang/IllegalArgumentException"), "ctxTrimBytes: negative size"); return 0; }
    if (mf_ctx_trim_bytes((mf_ctx *)(intptr_t)ctx, (uint64_t)want, &freed) < 0) { raise(e); return 0; }
    return (jlong)freed;
}
JNIEXPORT void JNICALL Java_io_HipBackend_tableDropIndex(JNIEnv *e, jclass, jlong table) {
    if (mf_table_drop_index((mf_table *)(intptr_t)table) < 0) raise(e);
}
JNIEXPORT jlong JNICALL Java_io_HipBackend_tableSize(JNIEnv *e, jclass, jlong table) {
    uint64_t n = 0, total = 0;
    if (mf_table_stats((mf_table *)(intptr_t)table, &n, &total) < 0) { raise(e); return 0; }
    return (jlong)n;
}
JNIEXPORT jlong JNICALL Java_io_HipBackend_writeKmers(JNIEnv *e, jclass, jlong table, jint threshold, jstring kmersBin, jstring statTxt) {
    utf b(e, kmersBin), s(e, statTxt);
    uint64_t good = 0;
    if (mf_table_write_kmers((mf_table *)(intptr_t)table, threshold, b.p, s.p, &good) < 0) { raise(e); return 0; }
    return (jlong)good;
}
JNIEXPORT jlong JNICALL Java_io_HipBackend_writeKmersFiltered(JNIEnv *e, jclass, jlong table, jint threshold, jlong filterTable, jint filterThreshold,
                                                              jstring kmersBin) {
    utf b(e, kmersBin);
    uint64_t good = 0;
    if (mf_table_write_kmers_filtered((mf_table *)(intptr_t)table, threshold, (mf_table *)(intptr_t)filterTable, filterThreshold, b.p, &good) < 0) { raise(e); return 0; }
    return (jlong)good;
}
JNIEXPORT jlong JNICALL Java_io_HipBackend_loadKmers(JNIEnv *e, jclass, jlong ctx, jobjectArray files, jint freqThreshold, jint k) {
    utf_array f(e, files);
    mf_table *t = nullptr;
    if (mf_table_load_kmers((mf_ctx *)(intptr_t)ctx, f.p.data(), (int)f.p.size(), freqThreshold, k, &t) < 0) { raise(e); return 0; }
    return (jlong)(intptr_t)t;
}
JNIEXPORT jlong JNICALL Java_io_HipBackend_buildUnitigs(JNIEnv *e, jclass, jlong ctx, jlong table, jint k, jint freqThreshold, jint lenThreshold,
                                                        jstring seqFasta, jstring distribution) {
    utf s(e, seqFasta), d(e, distribution);
    uint64_t n = 0;
    if (mf_build_unitigs((mf_ctx *)(intptr_t)ctx, (mf_table *)(intptr_t)table, k, freqThreshold, lenThreshold, s.p, d.p, &n) < 0) { raise(e); return 0; }
    return (jlong)n;
}
JNIEXPORT jlong JNICALL Java_io_HipBackend_cutComponents(JNIEnv *e, jclass, jlong ctx, jlong table, jint k, jint b1, jint b2, jstring componentsBin,
                                                         jstring statTxt) {
    utf c(e, componentsBin), s(e, statTxt);
    uint64_t n = 0;
    if (mf_cut_components((mf_ctx *)(intptr_t)ctx, (mf_table *)(intptr_t)table, k, b1, b2, c.p, s.p, &n) < 0) { raise(e); return 0; }
    return (jlong)n;
}
JNIEXPORT void JNICALL Java_io_HipBackend_features(JNIEnv *e, jclass, jlong ctx, jstring componentsBin, jstring kmersBin, jint k, jint threshold,
                                                   jstring vec, jstring breadth) {
    utf c(e, componentsBin), kb(e, kmersBin), v(e, vec), b(e, breadth);
    if (mf_features((mf_ctx *)(intptr_t)ctx, c.p, kb.p, k, threshold, v.p, b.p) < 0) raise(e);
}
JNIEXPORT void JNICALL Java_io_HipBackend_featuresReads(JNIEnv *e, jclass, jlong ctx, jstring componentsBin, jobjectArray files, jint k,
                                                        jint threshold, jstring vec, jstring breadth) {
    utf c(e, componentsBin), v(e, vec), b(e, breadth);
    utf_array f(e, files);
    if (mf_features_reads((mf_ctx *)(intptr_t)ctx, c.p, f.p.data(), (int)f.p.size(), k, threshold, v.p, b.p) < 0) raise(e);
}
// --selected (FeaturesCalculatorMain.java:55-57, 113-116, 193): `selected` = the handle loadKmers(selectedKmers, 0, k) returned, 0 = none
JNIEXPORT void JNICALL Java_io_HipBackend_featuresSelected(JNIEnv *e, jclass, jlong ctx, jstring componentsBin, jstring kmersBin, jint k, jint threshold,
                                                           jlong selected, jstring vec, jstring breadth) {
    utf c(e, componentsBin), kb(e, kmersBin), v(e, vec), b(e, breadth);
    if (mf_features_selected((mf_ctx *)(intptr_t)ctx, c.p, kb.p, k, threshold, (mf_table *)(intptr_t)selected, v.p, b.p) < 0) raise(e);
}
JNIEXPORT void JNICALL Java_io_HipBackend_featuresReadsSelected(JNIEnv *e, jclass, jlong ctx, jstring componentsBin, jobjectArray files, jint k,
                                                                jint threshold, jlong selected, jstring vec, jstring breadth) {
    utf c(e, componentsBin), v(e, vec), b(e, breadth);
    utf_array f(e, files);
    if (mf_features_reads_selected((mf_ctx *)(intptr_t)ctx, c.p, f.p.data(), (int)f.p.size(), k, threshold, (mf_table *)(intptr_t)selected, v.p, b.p) < 0) raise(e);
}
// devices this process sees: a host that wants one library per GPU makes one context per device and drives each from a thread of its own
JNIEXPORT jint JNICALL Java_io_HipBackend_deviceCount(JNIEnv *, jclass) { return (jint)mf_device_count(); }
JNIEXPORT void JNICALL Java_io_HipBackend_ctxBindThread(JNIEnv *e, jclass, jlong ctx) { if (mf_ctx_bind_thread((mf_ctx *)(intptr_t)ctx) < 0) raise(e); }
JNIEXPORT jdoubleArray JNICALL Java_io_HipBackend_brayCurtis(JNIEnv *e, jclass, jlongArray vecs, jint nSamples, jint nComp) {
    jlong *v = e->GetLongArrayElements(vecs, nullptr);
    std::vector<double> m((size_t)nSamples * (size_t)nSamples);
    const int rc = mf_bray_curtis((const int64_t *)v, nSamples, nComp, m.data());
    e->ReleaseLongArrayElements(vecs, v, JNI_ABORT);
    if (rc < 0) { raise(e); return nullptr; }
    jdoubleArray out = e->NewDoubleArray((jsize)m.size());
    if (out) e->SetDoubleArrayRegion(out, 0, (jsize)m.size(), m.data());
    return out;
}
JNIEXPORT jlongArray JNICALL Java_io_HipBackend_tableHist(JNIEnv *e, jclass, jlong table) {
    std::vector<uint64_t> h((size_t)MF_MAX_COUNT + 1);
    if (mf_table_hist((const mf_table *)(intptr_t)table, h.data()) < 0) { raise(e); return nullptr; }
    jlongArray out = e->NewLongArray((jsize)h.size());
    if (out) e->SetLongArrayRegion(out, 0, (jsize)h.size(), (const jlong *)h.data());
    return out;
}
JNIEXPORT void JNICALL Java_io_HipBackend_tableDestroy(JNIEnv *, jclass, jlong table) { mf_table_destroy((mf_table *)(intptr_t)table); }

// ---- component-cutter on several GPUs: plain forwarding, device pointers travel as jlong
#define CTX(x) ((mf_ctx *)(intptr_t)(x))
#define DCC(x) ((mf_dcc *)(intptr_t)(x))
#define DEV(x) ((void *)(intptr_t)(x))
static jlongArray longs(JNIEnv *e, const uint64_t *v, int n) {
    jlongArray out = e->NewLongArray(n);
    if (out) e->SetLongArrayRegion(out, 0, n, (const jlong *)v);
    return out;
}
JNIEXPORT jlong JNICALL Java_io_HipBackend_countShard(JNIEnv *e, jclass, jlong ctx, jlong dBases, jlong dOffsets, jlong nSeqs, jlong nBases, jint k,
                                                      jint minLen, jint rank, jint world) {
    mf_table *t = nullptr;
    if (mf_count_device_shard(CTX(ctx), DEV(dBases), DEV(dOffsets), (uint64_t)nSeqs, (uint64_t)nBases, k, minLen, rank, world, &t) < 0) { raise(e); return 0; }
    return (jlong)(intptr_t)t;
}
static bool bad_length(JNIEnv *e, const char *what) {
    e->ThrowNew(e->FindClass("java/lang/IllegalArgumentException"), what);
    return true;
}
JNIEXPORT jlong JNICALL Java_io_HipBackend_dccCreate(JNIEnv *e, jclass, jlong ctx, jlong shard, jint rank, jint world, jintArray base) {
    if (world < 1 || e->GetArrayLength(base) < world + 1) { bad_length(e, "dccCreate: base must hold world + 1 entries"); return 0; }
    jint *b = e->GetIntArrayElements(base, nullptr);
    mf_dcc *d = nullptr;
    const int rc = mf_dcc_create(CTX(ctx), (mf_table *)(intptr_t)shard, rank, world, (const uint32_t *)b, &d);
    e->ReleaseIntArrayElements(base, b, JNI_ABORT);
    if (rc < 0) { raise(e); return 0; }
    return (jlong)(intptr_t)d;
}
JNIEXPORT void JNICALL Java_io_HipBackend_dccDestroy(JNIEnv *, jclass, jlong dcc) { mf_dcc_destroy(DCC(dcc)); }
// (the per-rank count arrays are sized from the handle's own world size, not from a caller-supplied one)
JNIEXPORT jlongArray JNICALL Java_io_HipBackend_dccQueries(JNIEnv *e, jclass, jlong dcc) {
    const int world = mf_dcc_world(DCC(dcc));
    if (world < 1) { raise(e); return nullptr; }
    std::vector<uint64_t> c((size_t)world);
    if (mf_dcc_queries(DCC(dcc), c.data()) < 0) { raise(e); return nullptr; }
    return longs(e, c.data(), world);
}
JNIEXPORT void JNICALL Java_io_HipBackend_dccQueriesFill(JNIEnv *e, jclass, jlong dcc, jlong dQueries) { if (mf_dcc_queries_fill(DCC(dcc), DEV(dQueries)) < 0) raise(e); }
JNIEXPORT void JNICALL Java_io_HipBackend_dccAnswer(JNIEnv *e, jclass, jlong dcc, jlong dQueries, jlong n, jlong dAnswers) {
    if (mf_dcc_answer(DCC(dcc), DEV(dQueries), (uint64_t)n, DEV(dAnswers)) < 0) raise(e);
}
JNIEXPORT void JNICALL Java_io_HipBackend_dccSetAnswers(JNIEnv *e, jclass, jlong dcc, jlong dAnswers, jlong n) { if (mf_dcc_set_answers(DCC(dcc), DEV(dAnswers), (uint64_t)n) < 0) raise(e); }
JNIEXPORT jlongArray JNICALL Java_io_HipBackend_dccLevelLocal(JNIEnv *e, jclass, jlong dcc) {
    const int world = mf_dcc_world(DCC(dcc));
    if (world < 1) { raise(e); return nullptr; }
    std::vector<uint64_t> c((size_t)world);
    if (mf_dcc_level_local(DCC(dcc), c.data()) < 0) { raise(e); return nullptr; }
    return longs(e, c.data(), world);
}
JNIEXPORT void JNICALL Java_io_HipBackend_dccPairsFill(JNIEnv *e, jclass, jlong dcc, jlong dPairs) { if (mf_dcc_pairs_fill(DCC(dcc), DEV(dPairs)) < 0) raise(e); }
JNIEXPORT void JNICALL Java_io_HipBackend_dccPairsComplete(JNIEnv *e, jclass, jlong dcc, jlong dPairs, jlong n) { if (mf_dcc_pairs_complete(DCC(dcc), DEV(dPairs), (uint64_t)n) < 0) raise(e); }
JNIEXPORT jlong JNICALL Java_io_HipBackend_dccMerge(JNIEnv *e, jclass, jlong dcc, jlong dPairs, jlong n) {
    uint64_t ns = 0;
    if (mf_dcc_merge(DCC(dcc), DEV(dPairs), (uint64_t)n, &ns) < 0) { raise(e); return 0; }
    return (jlong)ns;
}
JNIEXPORT void JNICALL Java_io_HipBackend_dccStatsFill(JNIEnv *e, jclass, jlong dcc, jlong dStats) { if (mf_dcc_stats_fill(DCC(dcc), DEV(dStats)) < 0) raise(e); }
JNIEXPORT jlongArray JNICALL Java_io_HipBackend_dccClassify(JNIEnv *e, jclass, jlong dcc, jlong dStats, jlong n, jlongArray segFirst, jlong ownFirst, jlong ownN, jint b1,
                                                            jint b2, jint thr) {
    const int world = mf_dcc_world(DCC(dcc));
    if (world < 1) { raise(e); return nullptr; }
    if (e->GetArrayLength(segFirst) < world + 1) { bad_length(e, "dccClassify: segFirst must hold world + 1 entries"); return nullptr; }
    jlong *sf = e->GetLongArrayElements(segFirst, nullptr);
    uint64_t r[2] = {0, 0};
    const int rc = mf_dcc_classify(DCC(dcc), DEV(dStats), (uint64_t)n, (const uint64_t *)sf, (uint64_t)ownFirst, (uint64_t)ownN, b1, b2, thr, &r[0], &r[1]);
    e->ReleaseLongArrayElements(segFirst, sf, JNI_ABORT);
    if (rc < 0) { raise(e); return nullptr; }
    return longs(e, r, 2);
}
JNIEXPORT void JNICALL Java_io_HipBackend_dccKeptFill(JNIEnv *e, jclass, jlong dcc, jlong dKept) { if (mf_dcc_kept_fill(DCC(dcc), DEV(dKept)) < 0) raise(e); }
JNIEXPORT jlong JNICALL Java_io_HipBackend_dccMembers(JNIEnv *e, jclass, jlong dcc) {
    uint64_t n = 0;
    if (mf_dcc_members(DCC(dcc), &n) < 0) { raise(e); return 0; }
    return (jlong)n;
}
JNIEXPORT void JNICALL Java_io_HipBackend_dccMembersFill(JNIEnv *e, jclass, jlong dcc, jlong dKmers, jlong dRoots) { if (mf_dcc_members_fill(DCC(dcc), DEV(dKmers), DEV(dRoots)) < 0) raise(e); }
// { members, runs }: this rank's members sorted by component (8 bytes each on the wire) + one (root, count) record per component
JNIEXPORT jlongArray JNICALL Java_io_HipBackend_dccMembersGrouped(JNIEnv *e, jclass, jlong dcc) {
    uint64_t r[2] = {0, 0};
    if (mf_dcc_members_grouped(DCC(dcc), &r[0], &r[1]) < 0) { raise(e); return nullptr; }
    return longs(e, r, 2);
}
JNIEXPORT void JNICALL Java_io_HipBackend_dccMembersGroupedFill(JNIEnv *e, jclass, jlong dcc, jlong dKmers, jlong dRuns) {
    if (mf_dcc_members_grouped_fill(DCC(dcc), DEV(dKmers), DEV(dRuns)) < 0) raise(e);
}
JNIEXPORT void JNICALL Java_io_HipBackend_dccMinkeys(JNIEnv *e, jclass, jlong dcc, jintArray keptRoot, jlong dMin) {
    const jsize n = e->GetArrayLength(keptRoot);
    jint *g = e->GetIntArrayElements(keptRoot, nullptr);
    const int rc = mf_dcc_minkeys(DCC(dcc), (const uint32_t *)g, (uint64_t)n, DEV(dMin));
    e->ReleaseIntArrayElements(keptRoot, g, JNI_ABORT);
    if (rc < 0) raise(e);
}
JNIEXPORT jlong JNICALL Java_io_HipBackend_dccFinishGrouped(JNIEnv *e, jclass, jlong dcc, jlong dKmers, jlong nMembers, jlong dRuns, jlong nRuns, jintArray keptRoot,
                                                            jintArray keptSize, jlongArray keptWeight, jintArray keptThr, jlongArray keptMinkey) {
    const jsize n = e->GetArrayLength(keptRoot);
    if (e->GetArrayLength(keptSize) != n || e->GetArrayLength(keptWeight) != n || e->GetArrayLength(keptThr) != n || e->GetArrayLength(keptMinkey) != n) {
        bad_length(e, "dccFinishGrouped: the kept arrays must have one entry per component");
        return 0;
    }
    jint *g = e->GetIntArrayElements(keptRoot, nullptr), *sz = e->GetIntArrayElements(keptSize, nullptr), *th = e->GetIntArrayElements(keptThr, nullptr);
    jlong *w = e->GetLongArrayElements(keptWeight, nullptr), *mk = e->GetLongArrayElements(keptMinkey, nullptr);
    mf_comps *c = nullptr;
    const int rc = mf_dcc_finish_grouped(DCC(dcc), DEV(dKmers), (uint64_t)nMembers, DEV(dRuns), (uint64_t)nRuns, (const uint32_t *)g, (const uint32_t *)sz, (const int64_t *)w,
                                         (const int32_t *)th, (const uint64_t *)mk, (uint64_t)n, &c);
    e->ReleaseIntArrayElements(keptRoot, g, JNI_ABORT); e->ReleaseIntArrayElements(keptSize, sz, JNI_ABORT); e->ReleaseIntArrayElements(keptThr, th, JNI_ABORT);
    e->ReleaseLongArrayElements(keptWeight, w, JNI_ABORT); e->ReleaseLongArrayElements(keptMinkey, mk, JNI_ABORT);
    if (rc < 0) { raise(e); return 0; }
    return (jlong)(intptr_t)c;
}
JNIEXPORT jlong JNICALL Java_io_HipBackend_dccFinish(JNIEnv *e, jclass, jlong dcc, jlong dKmers, jlong dRoots, jlong nMembers, jintArray keptRoot, jintArray keptSize,
                                                     jlongArray keptWeight, jintArray keptThr, jlongArray keptMinkey) {
    const jsize n = e->GetArrayLength(keptRoot);
    if (e->GetArrayLength(keptSize) != n || e->GetArrayLength(keptWeight) != n || e->GetArrayLength(keptThr) != n || e->GetArrayLength(keptMinkey) != n) {
        bad_length(e, "dccFinish: the kept arrays must have one entry per component");
        return 0;
    }
    jint *g = e->GetIntArrayElements(keptRoot, nullptr), *sz = e->GetIntArrayElements(keptSize, nullptr), *th = e->GetIntArrayElements(keptThr, nullptr);
    jlong *w = e->GetLongArrayElements(keptWeight, nullptr), *mk = e->GetLongArrayElements(keptMinkey, nullptr);
    mf_comps *c = nullptr;
    const int rc = mf_dcc_finish(DCC(dcc), DEV(dKmers), DEV(dRoots), (uint64_t)nMembers, (const uint32_t *)g, (const uint32_t *)sz, (const int64_t *)w, (const int32_t *)th,
                                 (const uint64_t *)mk, (uint64_t)n, &c);
    e->ReleaseIntArrayElements(keptRoot, g, JNI_ABORT); e->ReleaseIntArrayElements(keptSize, sz, JNI_ABORT); e->ReleaseIntArrayElements(keptThr, th, JNI_ABORT);
    e->ReleaseLongArrayElements(keptWeight, w, JNI_ABORT); e->ReleaseLongArrayElements(keptMinkey, mk, JNI_ABORT);
    if (rc < 0) { raise(e); return 0; }
    return (jlong)(intptr_t)c;
}

// ---- the exchanges behind the boundary (round 6, mf_comm): a JVM with one worker thread per GPU makes local communicators; one JVM per GPU the
// rccl ones (the id crosses by the host's own means -- a file, a socket).  MF_ERR_TOGETHER surfaces as the same ExecutionFailedException on
// every rank: ComponentCutterMain can then cut on one device.
#define COMM(x) ((mf_comm *)(intptr_t)(x))
JNIEXPORT jlongArray JNICALL Java_io_HipBackend_commCreateLocal(JNIEnv *e, jclass, jlongArray ctxs) {
    const jsize n = e->GetArrayLength(ctxs);
    std::vector<jlong> h(n);
    e->GetLongArrayRegion(ctxs, 0, n, h.data());
    std::vector<mf_ctx *> cs(n); std::vector<mf_comm *> out(n, nullptr);
    for (jsize i = 0; i < n; i++) cs[i] = (mf_ctx *)(intptr_t)h[i];
    if (mf_comm_create_local(cs.data(), (int)n, out.data()) < 0) { raise(e); return nullptr; }
    jlongArray r = e->NewLongArray(n);
    if (!r) { for (mf_comm *c : out) mf_comm_destroy(c); return nullptr; }
    for (jsize i = 0; i < n; i++) h[i] = (jlong)(intptr_t)out[i];
    e->SetLongArrayRegion(r, 0, n, h.data());
    return r;
}
JNIEXPORT jbyteArray JNICALL Java_io_HipBackend_commRcclId(JNIEnv *e, jclass) {
    jbyte id[128];
    if (mf_comm_rccl_id(id) < 0) { raise(e); return nullptr; }
    jbyteArray r = e->NewByteArray(128);
    if (r) e->SetByteArrayRegion(r, 0, 128, id);
    return r;
}
JNIEXPORT jlong JNICALL Java_io_HipBackend_commCreateRccl(JNIEnv *e, jclass, jlong ctx, jbyteArray id, jint rank, jint world) {
    if (e->GetArrayLength(id) != 128) { bad_length(e, "commCreateRccl: the id has 128 bytes"); return 0; }
    jbyte b[128];
    e->GetByteArrayRegion(id, 0, 128, b);
    mf_comm *c = nullptr;
    if (mf_comm_create_rccl((mf_ctx *)(intptr_t)ctx, b, rank, world, &c) < 0) { raise(e); return 0; }
    return (jlong)(intptr_t)c;
}
JNIEXPORT void JNICALL Java_io_HipBackend_commDestroy(JNIEnv *, jclass, jlong comm) { mf_comm_destroy(COMM(comm)); }
JNIEXPORT jlong JNICALL Java_io_HipBackend_cutComponentsSharded(JNIEnv *e, jclass, jlong comm, jobjectArray seqFiles, jint k, jint minLen, jint b1, jint b2, jstring componentsBin,
                                                                jstring statTxt) {
    utf_array f(e, seqFiles);
    utf cb(e, componentsBin), st(e, statTxt);
    uint64_t nc = 0;
    if (mf_cut_components_sharded_files(COMM(comm), f.p.data(), (int)f.p.size(), k, minLen, b1, b2, cb.p, st.p, &nc) < 0) { raise(e); return 0; }
    return (jlong)nc;
}
// rows: this rank's vectors, row after row; -> all ranks' rows, rank after rank
JNIEXPORT jlongArray JNICALL Java_io_HipBackend_featuresAllgather(JNIEnv *e, jclass, jlong comm, jlongArray rows, jint nRows, jint nComp) {
    if (nRows < 0 || nComp < 0 || e->GetArrayLength(rows) != (jsize)nRows * nComp) { bad_length(e, "featuresAllgather: rows must hold nRows x nComp values"); return nullptr; }
    jlong *v = e->GetLongArrayElements(rows, nullptr);
    uint64_t n_all = 0;
    int rc = mf_features_allgather(COMM(comm), (const int64_t *)v, (uint64_t)nRows, (uint64_t)nComp, nullptr, 0, &n_all);
    std::vector<int64_t> all(n_all * (uint64_t)nComp + 1);
    if (rc == 0) rc = mf_features_allgather(COMM(comm), (const int64_t *)v, (uint64_t)nRows, (uint64_t)nComp, all.data(), n_all, &n_all);
    e->ReleaseLongArrayElements(rows, v, JNI_ABORT);
    if (rc < 0) { raise(e); return nullptr; }
    jlongArray r = e->NewLongArray((jsize)(n_all * (uint64_t)nComp));
    if (r) e->SetLongArrayRegion(r, 0, (jsize)(n_all * (uint64_t)nComp), (const jlong *)all.data());
    return r;
}
}
#endif

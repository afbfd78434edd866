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
JNIEXPORT void JNICALL Java_io_HipBackend_tableDestroy(JNIEnv *, jclass, jlong table) { mf_table_destroy((mf_table *)(intptr_t)table); }
}
#endif

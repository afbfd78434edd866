package io;                                   // lives next to src/io/IOUtils.java of ctlab/metafast

/**
 * Native counterparts of the static seams of the matrix-builder path (see INTEGRATION.md, section 1).  Handles are the
 * addresses of the C-ABI's opaque objects (include/metafast_hip.h); the caller closes them explicitly.
 */
public final class HipBackend {
    static { System.loadLibrary("metafast_jni"); }        // libmetafast_jni.so -> libmetafast_hip.so

    public static native long ctxCreate(int device, int hostThreads);
    public static native void ctxDestroy(long ctx);
    /** IOUtils.loadReads (src/io/IOUtils.java:772) */
    public static native long countReads(long ctx, String[] files, int k, int minSeqLen);
    /** hm.size() */
    public static native long tableSize(long table);
    /** IOUtils.printKmers (src/io/IOUtils.java:45): returns the number of good k-mers written */
    public static native long writeKmers(long table, int threshold, String kmersBin, String statTxt);
    /** IOUtils.filterAndPrintKmers (src/io/IOUtils.java:101): returns the number of records written */
    public static native long writeKmersFiltered(long table, int threshold, long filterTable, int filterThreshold, String kmersBin);
    /** IOUtils.loadKmers (src/io/IOUtils.java:369) */
    public static native long loadKmers(long ctx, String[] files, int freqThreshold, int k);
    /** SequencesFinders.thresholdStrategy + Sequence.printSequences (SeqBuilderMain.java:147,160): returns the number of sequences */
    public static native long buildUnitigs(long ctx, long table, int k, int freqThreshold, int lenThreshold, String seqFasta,
                                           String distribution);
    /** ComponentsBuilder.splitStrategy + ConnectedComponent.saveComponents (ComponentCutterMain.java:94,108): returns the number of components */
    public static native long cutComponents(long ctx, long table, int k, int b1, int b2, String componentsBin, String statTxt);
    /** FeaturesCalculatorMain.buildAndPrintVector (:169), k-mers files branch */
    public static native void features(long ctx, String componentsBin, String kmersBin, int k, int threshold, String vec, String breadth);
    /** FeaturesCalculatorMain.runImpl reads branch (:117-131, --use-reads-for-calculating-features) */
    public static native void featuresReads(long ctx, String componentsBin, String[] files, int k, int threshold, String vec, String breadth);
    /** DistanceMatrixCalculatorMain.brayCurtisDistance (:140) for all pairs: row-major nSamples x nSamples */
    public static native double[] brayCurtis(long[] vecs, int nSamples, int nComp);
    public static native void tableDestroy(long table);

    private HipBackend() {}
}

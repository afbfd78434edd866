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
    /** loadReads + the cut of printKmers (value > threshold) inside the counting kernels (KmersCounterMain.java:77-99):
     *  { table of the k-mers handed on, hm.size() before the cut } */
    public static native long[] countReadsAbove(long ctx, String[] files, int k, int minSeqLen, int threshold);
    /** releases the table's lookup index until it is needed again (several libraries per GPU) */
    public static native void tableDropIndex(long table);
    /** mf_ctx_trim_bytes: idle workspace back to the driver, smallest regions first, until `want` bytes are free again; the bytes given back
     *  (a JVM that keeps device buffers of its own beside the library asks for what it is short of, not for everything) */
    public static native long ctxTrimBytes(long ctx, long want);
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
    /** the same two with --selected (FeaturesCalculatorMain.java:55-57, 113-116, 193): selected = loadKmers(selectedKmers, 0, k), 0 = none */
    public static native void featuresSelected(long ctx, String componentsBin, String kmersBin, int k, int threshold, long selected, String vec, String breadth);
    public static native void featuresReadsSelected(long ctx, String componentsBin, String[] files, int k, int threshold, long selected, String vec, String breadth);
    /** GPUs this JVM sees; one context per device, each used by one thread at a time (ctxBindThread: a thread other than the one that made the context) */
    public static native int deviceCount();
    public static native void ctxBindThread(long ctx);
    /** DistanceMatrixCalculatorMain.brayCurtisDistance (:140) for all pairs: row-major nSamples x nSamples */
    public static native double[] brayCurtis(long[] vecs, int nSamples, int nComp);
    /** the .stat.txt rows of IOUtils.printKmers (src/io/IOUtils.java:45-71) of a table counted with the cut inside the counting pass: hist[count] */
    public static native long[] tableHist(long table);
    public static native void tableDestroy(long table);

    // ---- component-cutter on several GPUs (ComponentCutterMain.java:78-114 with one process per GPU): device pointers are plain longs;
    //      the host moves the buffers between the ranks (all-to-all / all-gather) between these calls, see INTEGRATION.md section 5
    /** this rank's shard of the cutter table of all samples' unitigs (IOUtils.loadReads of ComponentCutterMain.java:81, the k-mers this rank owns) */
    public static native long countShard(long ctx, long dBases, long dOffsets, long nSeqs, long nBases, int k, int minLen, int rank, int world);
    public static native long dccCreate(long ctx, long shard, int rank, int world, int[] base);
    public static native void dccDestroy(long dcc);
    /** queries for the owners of neighbours in other shards: counts per rank; dccQueriesFill writes the 16-byte queries */
    public static native long[] dccQueries(long dcc);
    public static native void dccQueriesFill(long dcc, long dQueries);
    public static native void dccAnswer(long dcc, long dQueries, long n, long dAnswers);
    public static native void dccSetAnswers(long dcc, long dAnswers, long n);
    /** one threshold level of ComponentsBuilder.run (src/algo/ComponentsBuilder.java:86-150) */
    public static native long[] dccLevelLocal(long dcc);
    public static native void dccPairsFill(long dcc, long dPairs);
    public static native void dccPairsComplete(long dcc, long dPairs, long n);
    public static native long dccMerge(long dcc, long dPairs, long n);
    public static native void dccStatsFill(long dcc, long dStats);
    /** segFirst[world + 1]: first record of every rank in dStats; returns {kept, oversize} components of the level over ALL ranks
     *  (every rank holds every component's records: nothing more is exchanged); dccKeptFill: their (root, size, weight) records */
    public static native long[] dccClassify(long dcc, long dStats, long n, long[] segFirst, long ownFirst, long ownN, int b1, int b2, int thr);
    public static native void dccKeptFill(long dcc, long dKept);
    public static native long dccMembers(long dcc);
    public static native void dccMembersFill(long dcc, long dKmers, long dRoots);
    /** {members, runs}: the members sorted by component -- 8 bytes each on the wire + one (root, count) record per component */
    public static native long[] dccMembersGrouped(long dcc);
    public static native void dccMembersGroupedFill(long dcc, long dKmers, long dRuns);
    public static native long dccFinishGrouped(long dcc, long dKmers, long nMembers, long dRuns, long nRuns, int[] keptRoot, int[] keptSize,
                                               long[] keptWeight, int[] keptThr, long[] keptMinkey);
    public static native void dccMinkeys(long dcc, int[] keptRoot, long dMin);
    /** -> components handle (List&lt;ConnectedComponent&gt;), the same on every rank */
    public static native long dccFinish(long dcc, long dKmers, long dRoots, long nMembers, int[] keptRoot, int[] keptSize, long[] keptWeight,
                                        int[] keptThr, long[] keptMinkey);

    // ---- the exchanges behind the boundary (round 6): ComponentCutterMain.runImpl :78-114 over several GPUs in ONE call per rank
    /** one communicator per context, for one worker thread per GPU of this JVM (slices are copied straight into the peers' buffers) */
    public static native long[] commCreateLocal(long[] ctxs);
    /** one JVM per GPU: rank 0 makes the 128-byte id, hands it to the others, all create together (RCCL) */
    public static native byte[] commRcclId();
    public static native long commCreateRccl(long ctx, byte[] id, int rank, int world);
    public static native void commDestroy(long comm);
    /** every rank: the .seq.fasta files of ITS libraries; rank 0 writes components.bin + the stat file; -> number of components */
    public static native long cutComponentsSharded(long comm, String[] seqFiles, int k, int minSeqLen, int b1, int b2, String componentsBin, String statTxt);
    /** the feature vectors of all ranks' libraries, rank after rank (the rows DistanceMatrixCalculatorMain reads from the .vec files) */
    public static native long[] featuresAllgather(long comm, long[] rows, int nRows, int nComp);

    private HipBackend() {}
}

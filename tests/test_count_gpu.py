"""GPU parity: HIP counting path (through the C-ABI) vs the CPU oracle, bit-exact sorted (k-mer, count) lists."""
import os

import numpy as np
import pytest

from util import gpu_count, genome_reads, pack_reads, random_reads, to_device

pytestmark = pytest.mark.gpu


def _check(ctx, oracle, bases, off, k, min_len=0, thr=-1):
    t = gpu_count(ctx, bases, off, k, min_len)
    gk, gc = t.export(thr)
    ok, ov = oracle.Table().count_buffer(bases, off, k, min_len).export(thr)
    assert len(gk) == len(ok), (len(gk), len(ok))
    assert np.array_equal(gk, ok)
    assert np.array_equal(gc.astype(np.int32), ov)
    lens = np.diff(off.astype(np.int64))
    occ = int(np.maximum(lens - k + 1, 0)[lens >= max(min_len, k)].sum())
    assert t.occurrences() == occ
    n, total = t.stats()
    assert n == len(oracle.Table().count_buffer(bases, off, k, min_len))
    return t


def _reset(ctx):
    for name, v in (("l1_bits", -1), ("l2_bits", -1), ("part_target", 3072), ("scatter_staged", 1), ("l1_blocks", 0), ("skm", 1), ("skm_batches", 0), ("skm_dyn", 1), ("skm_slices", 0), ("skm_shared", 1), ("skm_dedupe", 1), ("arena_cap_gb", 0), ("skm_pilot", 1), ("skm_unit_distinct", 2200)):
        ctx.set_option(name, v)


def test_reference_fasta(gpu_ctx, oracle, ref_files):
    _reset(gpu_ctx)
    for f, (nd, ng) in zip(ref_files, [(17063, 16918), (8176, 7321), (14042, 11351)]):
        b, o = oracle.read_file(f)
        t = _check(gpu_ctx, oracle, b, o, 31)
        assert len(t) == nd
        assert len(t.export(1)[0]) == ng


@pytest.mark.parametrize("k", [1, 2, 5, 15, 16, 21, 30, 31])
def test_ragged_reads_all_k(gpu_ctx, oracle, k):
    _reset(gpu_ctx)
    rng = np.random.default_rng(100 + k)
    b, o = random_reads(rng, 3000, 0, 90)        # includes empty reads and reads shorter than k
    _check(gpu_ctx, oracle, b, o, k)
    _check(gpu_ctx, oracle, b, o, k, min_len=40)


def test_lowercase_and_boundaries(gpu_ctx, oracle):
    _reset(gpu_ctx)
    reads = ["acgtACGTacgtACGTacgtACGTacgtACGTa", "A" * 31, "C" * 30, "G", "", "T" * 64, "ACGT" * 16 + "A"]
    b, o = pack_reads(reads)
    _check(gpu_ctx, oracle, b, o, 31)
    _check(gpu_ctx, oracle, b, o, 4)


def test_polya_saturation(gpu_ctx, oracle):
    _reset(gpu_ctx)
    b, o = pack_reads(["A" * 40000, "T" * 100, "ACGTTGCA" * 20])
    t = _check(gpu_ctx, oracle, b, o, 31)
    k, c = t.export()
    assert k[0] == 0 and c[0] == 32767           # key 0 legal, count saturates (NumUtils.java:21-26)


def test_empty_inputs(gpu_ctx, oracle):
    _reset(gpu_ctx)
    b, o = pack_reads([])
    t = gpu_count(gpu_ctx, b, o, 31)
    assert len(t) == 0 and t.occurrences() == 0
    b, o = pack_reads(["ACGT", "AC"])
    t = gpu_count(gpu_ctx, b, o, 31)
    assert len(t) == 0


@pytest.mark.parametrize("l1,l2,staged,blocks", [(0, -1, 1, 0), (3, -1, 1, 0), (3, 4, 1, 2), (5, 5, 0, 3), (11, -1, 1, 0),
                                                 (6, 6, 1, 0), (2, 11, 1, 5)])
def test_partition_plans(gpu_ctx, oracle, l1, l2, staged, blocks):
    """every partition plan / scatter flavour must give the same table"""
    _reset(gpu_ctx)
    rng = np.random.default_rng(7)
    b, o = genome_reads(rng, 20000, 4000, 100, err=0.01)
    gpu_ctx.set_option("l1_bits", l1)
    gpu_ctx.set_option("l2_bits", l2)
    gpu_ctx.set_option("scatter_staged", staged)
    gpu_ctx.set_option("l1_blocks", blocks)
    try:
        _check(gpu_ctx, oracle, b, o, 31)
        _check(gpu_ctx, oracle, b, o, 11)
    finally:
        _reset(gpu_ctx)


def test_auto_plan_medium(gpu_ctx, oracle):
    """2.4e7 occurrences: auto plan uses two partition levels; compare everything with the oracle"""
    _reset(gpu_ctx)
    rng = np.random.default_rng(11)
    b, o = genome_reads(rng, 2_000_000, 200_000, 150, err=0.005)
    gpu_ctx.set_option("part_target", 256)       # forces 17 partition bits on this size
    try:
        _check(gpu_ctx, oracle, b, o, 31)
    finally:
        _reset(gpu_ctx)
    _check(gpu_ctx, oracle, b, o, 31)


def test_lookup_and_filter(gpu_ctx, oracle, ref_files):
    _reset(gpu_ctx)
    b, o = oracle.read_file(ref_files[2])
    t = gpu_count(gpu_ctx, b, o, 31)
    keys, cnts = t.export()
    probe = np.concatenate([keys[:100], keys[-100:], np.array([1, 2, 3, 2 ** 61 + 5], dtype=np.uint64)])
    got = t.lookup(probe)
    ot = oracle.Table().count_buffer(b, o, 31)
    want = np.array([ot.get(int(x)) for x in probe], dtype=np.int32)
    assert np.array_equal(got, want)
    f = t.filter(1)
    fk, fc = f.export()
    m = cnts > 1
    assert np.array_equal(fk, keys[m]) and np.array_equal(fc, cnts[m])


def test_from_host_insert_or_add(gpu_ctx):
    keys = np.array([5, 9, 5, 0, 9, 5, 77], dtype=np.uint64)
    cnts = np.array([30000, 1, 30000, 4, 2, 7, 65535 // 2], dtype=np.uint16)
    t = gpu_ctx.table_from_host(keys, cnts, 31)
    k, c = t.export()
    assert k.tolist() == [0, 5, 9, 77] and c.tolist() == [4, 32767, 3, 32767]


def test_synth_generator_matches_host(gpu_ctx):
    import torch
    from metafast_amd import lib as L
    n, rl = 5000, 150
    tb = torch.zeros(n * rl + 64, dtype=torch.uint8, device="cuda")
    to = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    gpu_ctx.synth_reads_device(0x4D45544146415354, 3, 1000, n, rl, 20000, tb.data_ptr(), to.data_ptr())
    hb, ho = L.synth_reads_host(0x4D45544146415354, 3, 1000, n, rl, 20000)
    assert np.array_equal(tb[: n * rl].cpu().numpy(), hb)
    assert np.array_equal(to.cpu().numpy().astype(np.uint64), ho)


# ---- super-k-mer path (default for k >= 20) against the one-record-per-k-mer path and the oracle ----
@pytest.mark.parametrize("k", list(range(20, 32)))
def test_skm_every_k(gpu_ctx, oracle, k):
    """every k the super-k-mer path is compiled for: same table from both paths, equal to the oracle"""
    _reset(gpu_ctx)
    rng = np.random.default_rng(500 + k)
    b1, o1 = genome_reads(rng, 30000, 3000, 150, err=0.01)
    b2, o2 = random_reads(rng, 1500, 0, 120)                 # ragged: empty reads, reads shorter than k
    b = np.concatenate([b1, b2]); o = np.concatenate([o1, o2[1:] + o1[-1]])
    try:
        for target in (3072, 64):                            # one and two partition levels
            gpu_ctx.set_option("part_target", target)
            gpu_ctx.set_option("skm", 1)
            gpu_ctx.set_option("skm_batches", 0 if target == 3072 else 5)     # count + gather in one / in five batches
            gpu_ctx.set_option("skm_dyn", 2 if k % 2 else 1)                  # odd k: one-pass level 1 (sampled regions)
            t = _check(gpu_ctx, oracle, b, o, k)
            assert t.records()[1] == 16
            gpu_ctx.set_option("skm", 0)
            t = _check(gpu_ctx, oracle, b, o, k)
            assert t.records()[1] == 8
    finally:
        _reset(gpu_ctx)


@pytest.mark.parametrize("k", [20, 27, 31])
def test_skm_low_complexity(gpu_ctx, oracle, k):
    """runs longer than a record holds (homopolymers, short tandem repeats: ONE minimizer for the whole read) are cut"""
    _reset(gpu_ctx)
    reads = ["A" * 500, "AC" * 200, "ACG" * 150, "T" * 33, "ACGTTGCA" * 40, "G" * 64 + "ACGTACGTTTGACCA" * 9, "C" * (k - 1), "C" * k]
    b, o = pack_reads(reads)
    t = _check(gpu_ctx, oracle, b, o, k)
    _check(gpu_ctx, oracle, b, o, k, min_len=100)
    assert t.records()[1] == 16


@pytest.mark.parametrize("n_reads,skm_path", [(20000, True), (400000, False)])
def test_skm_homopolymer_flanks(gpu_ctx, oracle, n_reads, skm_path):
    """many reads with a poly-A / poly-T stretch between random flanks: tens of thousands of distinct k-mers share the few
    M-mers made of the stretch and a base or two of flank, i.e. ONE minimizer partition holds far more distinct k-mers than
    the LDS table.  Such a partition is counted in several passes (4, 16, 64) instead of sending the whole call down the
    slower k-mer path; beyond 64 passes the k-mer path takes over.  The index gets an HBM-built region for it."""
    _reset(gpu_ctx)
    rng = np.random.default_rng(21)
    fl = rng.integers(0, 4, size=(n_reads, 90))
    lut = np.frombuffer(b"AGCT", dtype=np.uint8)
    mid = np.where((np.arange(n_reads) & 1)[:, None] == 1, ord("A"), ord("T")).astype(np.uint8).repeat(22, axis=1)
    arr = np.concatenate([lut[fl[:, :45]], mid, lut[fl[:, 45:]]], axis=1)
    b = np.concatenate([arr.reshape(-1), np.zeros(64, dtype=np.uint8)])
    o = (np.arange(n_reads + 1, dtype=np.uint64) * 112)
    t = _check(gpu_ctx, oracle, b, o, 31)
    assert (t.records()[1] == 16) == skm_path          # 16-byte records: counted by the super-k-mer path
    ok, ov = oracle.Table().count_buffer(b, o, 31).export()
    pick = rng.choice(len(ok), size=5000, replace=False)
    absent = rng.integers(0, 1 << 62, size=500, dtype=np.uint64)
    got = t.lookup(np.concatenate([ok[pick], absent]))
    assert np.array_equal(got[:5000], ov[pick]) and np.all(got[5000:] == -1)


def test_skm_lookup_filter_two_levels(gpu_ctx, oracle):
    """index over minimizer partitions (per-partition regions): present / absent keys, before and after a filter"""
    _reset(gpu_ctx)
    rng = np.random.default_rng(77)
    b, o = genome_reads(rng, 200_000, 40_000, 150, err=0.004)
    gpu_ctx.set_option("part_target", 128)
    try:
        t = gpu_count(gpu_ctx, b, o, 31)
        keys, cnts = t.export()
        absent = rng.integers(0, 2 ** 62, size=2000, dtype=np.uint64)
        probe = np.concatenate([keys[::7], absent])
        ot = oracle.Table().count_buffer(b, o, 31)
        want = np.array([ot.get(int(x)) for x in probe], dtype=np.int32)
        assert np.array_equal(t.lookup(probe), want)
        f = t.filter(1)
        fk, fc = f.export()
        m = cnts > 1
        assert np.array_equal(fk, keys[m]) and np.array_equal(fc, cnts[m])
        want_f = np.where(want > 1, want, -1)
        assert np.array_equal(f.lookup(probe), want_f)
    finally:
        _reset(gpu_ctx)


@pytest.mark.parametrize("shared", [2, 0])
@pytest.mark.parametrize("slices,dyn", [(2, 1), (4, 2), (8, 0)])
def test_skm_digit_range_slices(gpu_ctx, oracle, slices, dyn, shared):
    """a run cut into digit-range slices (what a sample with more records than HBM takes: BASELINE config 5): every slice
    counts its share of the level-1 digits, behind one level 1 over all digits (shared = 2) or scanning the reads for its own
    digits (0: what is left when the whole level-1 buffer does not fit); same table, same order, same tallies"""
    _reset(gpu_ctx)
    gpu_ctx.set_option("skm_shared", shared)
    rng = np.random.default_rng(79)
    b, o = genome_reads(rng, 300_000, 60_000, 150, err=0.004)
    try:
        gpu_ctx.set_option("part_target", 256)
        gpu_ctx.set_option("skm_dyn", dyn)
        ref = gpu_count(gpu_ctx, b, o, 31)
        rk, rc, rn = ref.device_view()
        gpu_ctx.set_option("skm_slices", slices)
        t = _check(gpu_ctx, oracle, b, o, 31)
        db, do = to_device(b, o)
        cut, n_all = gpu_ctx.count_device_above(db.data_ptr(), do.data_ptr(), len(o) - 1, len(b), 31, 1)
        keys, cnts = t.export()
        ck, cc = cut.export()
        assert n_all == len(keys) and np.array_equal(ck, keys[cnts > 1]) and np.array_equal(cc, cnts[cnts > 1])
        h = cut.hist()
        assert np.array_equal(h[:64], np.bincount(cnts, minlength=64)[:64])
        assert t.records()[0] == ref.records()[0]
    finally:
        _reset(gpu_ctx)


def test_skm_three_levels(gpu_ctx, oracle):
    """a plan with three radix levels (what 200 M reads at k = 21 need): the third level writes into the buffer the first
    level's records came from; one-pass and exact level 1"""
    _reset(gpu_ctx)
    rng = np.random.default_rng(78)
    b, o = genome_reads(rng, 500_000, 160_000, 150, err=0.004)
    try:
        gpu_ctx.set_option("part_target", 1)               # 1.9e7 occurrences -> 25 bits = 11 + 11 + 3
        for dyn in (2, 0):
            gpu_ctx.set_option("skm_dyn", dyn)
            t = _check(gpu_ctx, oracle, b, o, 31)
            assert t.records()[1] == 16
            t2 = _check(gpu_ctx, oracle, b, o, 21)             # again on the same context: the arena's idle buffers are re-used
            assert t2.records()[1] == 16
        for shared, slices in ((2, 4), (0, 2), (1, 8)):        # ... in slices: behind one level 1, scanning per slice, as it fits
            gpu_ctx.set_option("skm_slices", slices)
            gpu_ctx.set_option("skm_shared", shared)
            assert _check(gpu_ctx, oracle, b, o, 31).records()[0] == t.records()[0]
            _check(gpu_ctx, oracle, b, o, 21)
    finally:
        _reset(gpu_ctx)


def test_skm_one_pass_level1(gpu_ctx, oracle):
    """level 1 in one pass (regions from a sampled histogram, chunk-wise allocation), also when the sample misleads:
    reads sorted by genome make every third tile unrepresentative, the overflow must be detected and repaired"""
    _reset(gpu_ctx)
    rng = np.random.default_rng(2024)
    b, o = genome_reads(rng, 400_000, 120_000, 150, err=0.003)
    srt, so = [], [0]
    rl = 150
    reads = b.reshape(-1, rl)
    order = np.lexsort(reads.T[::-1][:8])                     # sorted reads: neighbouring tiles look alike, distant ones do not
    bs = reads[order].reshape(-1).copy()
    try:
        gpu_ctx.set_option("part_target", 96)                 # two levels
        for data in (b, bs):
            for mode in (2, 0):
                gpu_ctx.set_option("skm_dyn", mode)
                _check(gpu_ctx, oracle, data, o, 31)
    finally:
        _reset(gpu_ctx)


@pytest.mark.parametrize("skm", [1, 0])
def test_count_above_threshold(gpu_ctx, oracle, skm):
    """mf_count_device_above = count + IOUtils.printKmers' cut (count > threshold), done inside the counting kernels on the
    super-k-mer path: same entries as count_device().filter(), plus the number of distinct k-mers before the cut"""
    from util import to_device
    _reset(gpu_ctx)
    rng = np.random.default_rng(91)
    b, o = genome_reads(rng, 60_000, 20_000, 150, err=0.01)
    tb, to = to_device(b, o)
    ok, ov = oracle.Table().count_buffer(b, o, 31).export()
    try:
        gpu_ctx.set_option("skm", skm)
        for target, batches in ((3072, 0), (96, 3)):
            gpu_ctx.set_option("part_target", target)
            gpu_ctx.set_option("skm_batches", batches)
            for thr in (0, 1, 3, 40):
                t, n_all = gpu_ctx.count_device_above(tb.data_ptr(), to.data_ptr(), len(o) - 1, int(o[-1]), 31, thr)
                gk, gc = t.export()
                m = ov > thr
                assert n_all == len(ok)
                assert np.array_equal(gk, ok[m]) and np.array_equal(gc.astype(np.int32), ov[m])
                assert t.occurrences() == int(np.maximum(np.diff(o.astype(np.int64)) - 30, 0).sum())
                probe = np.concatenate([ok[:50], ok[-50:]])
                want = np.array([int(v) if v > thr else -1 for v in np.concatenate([ov[:50], ov[-50:]])], dtype=np.int32)
                assert np.array_equal(t.lookup(probe), want)
    finally:
        _reset(gpu_ctx)


def test_long_sequences(gpu_ctx, oracle):
    """assembled sequences (the cutter's input: mean length >= 8k selects the small-partition plan): one long sequence
    among short ones, with and without the length filter, on both paths"""
    _reset(gpu_ctx)
    rng = np.random.default_rng(13)
    al = np.frombuffer(b"ACGT", dtype=np.uint8)
    parts = [al[rng.integers(0, 4, size=n)] for n in (70, 1_500_000, 31, 30, 400, 99, 100)]
    b = np.concatenate(parts).astype(np.uint8)
    o = np.concatenate([[0], np.cumsum([len(x) for x in parts])]).astype(np.uint64)
    try:
        for skm in (1, 0):
            gpu_ctx.set_option("skm", skm)
            _check(gpu_ctx, oracle, b, o, 31)
            _check(gpu_ctx, oracle, b, o, 31, min_len=100)
            _check(gpu_ctx, oracle, b, o, 21, min_len=100)
    finally:
        _reset(gpu_ctx)


@pytest.mark.parametrize("world,slices,shared", [(2, 0, 1), (8, 0, 1), (2, 4, 2), (4, 2, 0)])
def test_shard_tables_partition_the_table(gpu_ctx, oracle, world, slices, shared):
    """mf_count_device_shard: the ranks' shards are disjoint and together the oracle's table, counts included; also when a
    rank counts its digits in slices (behind one level 1 over its digits / every slice scanning the reads)"""
    _reset(gpu_ctx)
    rng = np.random.default_rng(77)
    bases, offsets = genome_reads(rng, 30000, 4000, 150, err=0.01)
    db, do = to_device(bases, offsets)
    want_k, want_c = oracle.Table().count_buffer(bases, offsets, 31).export()
    keys, cnts = [], []
    try:
        if slices:
            gpu_ctx.set_option("part_target", 64); gpu_ctx.set_option("skm_slices", slices); gpu_ctx.set_option("skm_shared", shared)
        for r in range(world):
            t = gpu_ctx.count_device_shard(db.data_ptr(), do.data_ptr(), len(offsets) - 1, len(bases), 31, 0, r, world)
            kk, cc = t.export()
            assert len(kk) > 0
            keys.append(kk); cnts.append(cc)
        allk = np.concatenate(keys); order = np.argsort(allk, kind="stable")
        assert len(np.unique(allk)) == len(allk)
        assert np.array_equal(allk[order], want_k) and np.array_equal(np.concatenate(cnts)[order].astype(np.int32), want_c)
        with pytest.raises(Exception):
            gpu_ctx.count_device_shard(db.data_ptr(), do.data_ptr(), len(offsets) - 1, len(bases), 31, 0, 0, 3)
    finally:
        _reset(gpu_ctx)


def test_trim_bytes_gives_back_the_smallest_idle_regions(gpu_ctx, oracle):
    """mf_ctx_trim_bytes: a host short of memory gets idle workspace back without the arena dropping everything; counting
    afterwards still works (the regions are allocated again)"""
    import torch
    _reset(gpu_ctx)
    rng = np.random.default_rng(5)
    b, o = genome_reads(rng, 100_000, 20_000, 150, err=0.004)
    _check(gpu_ctx, oracle, b, o, 31)
    gpu_ctx.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    got = gpu_ctx.trim(1)                                   # one region is enough
    assert got >= 1
    assert torch.cuda.mem_get_info()[0] >= free0 + got - (64 << 20)
    got_all = gpu_ctx.trim(1 << 60)                         # more than there is: everything idle, no error
    assert got_all >= 0 and gpu_ctx.trim(1 << 20) == 0
    _check(gpu_ctx, oracle, b, o, 31)
    # the pipeline's retry: an allocation that fails with the device full of idle regions succeeds after a partial trim
    from metafast_amd import pipeline as P
    calls = []
    def alloc():
        calls.append(1)
        if len(calls) == 1:
            raise torch.OutOfMemoryError("HIP out of memory (injected)")
        return torch.zeros(16, device="cuda")
    assert P._with_room(gpu_ctx, alloc).numel() == 16 and len(calls) == 2


@pytest.mark.parametrize("dedupe", [5, 1, 0])
def test_skm_identical_records_counted_once(gpu_ctx, oracle, dedupe):
    """k_skm_count tells identical records of a unit apart and inserts their k-mers once, with the multiplicity (5: always,
    1: unless a unit shows no repeats, 0: never): very deep reads (each record ~300 times), reads without repeats, counts
    that saturate, units of more than 2048 records and units counted in several passes -- the same tables"""
    _reset(gpu_ctx)
    rng = np.random.default_rng(404)
    try:
        gpu_ctx.set_option("skm_dedupe", dedupe)
        deep_b, deep_o = genome_reads(rng, 6_000, 120_000, 150, err=0.002)           # 3000-fold
        flat_b, flat_o = genome_reads(rng, 3_000_000, 20_000, 150, err=0.0)          # 1-fold: nothing repeats
        for pt in (3072, 256, 16384):                                                 # units of ~1750, ~150 and ~9000 records
            gpu_ctx.set_option("part_target", pt)
            for k in (31, 21):
                _check(gpu_ctx, oracle, deep_b, deep_o, k)
            _check(gpu_ctx, oracle, flat_b, flat_o, 31)
        # one read 40 000 times: counts saturate at 32767 (NumUtils.addAndBound), weights beyond 16 bits never form
        one = genome_reads(rng, 400, 1, 150, err=0.0)[0]
        b = np.tile(one, 40_000); o = np.arange(40_001, dtype=np.uint64) * 150
        gpu_ctx.set_option("part_target", 3072)
        t = _check(gpu_ctx, oracle, b, o, 31)
        assert t.export()[1].max() == 32767
    finally:
        _reset(gpu_ctx)


@pytest.mark.parametrize("scale,sub16k", [(16_000_000, 82), (1_000_000, 82), (40_000, 164)])
def test_depth_independent_plan_against_the_oracle(gpu_ctx, oracle, scale, sub16k):
    """1 M reads through the AUTOMATIC plan at three sequencing depths -- 0.05-fold (pool scale 16 M: every
    k-mer nearly distinct, 8 x the distinct k-mers per occurrence the occurrence-based plan has in mind), 0.8-fold, and 20-fold with
    1 % substitutions -- bit-identical to the oracle, with the pilot in the loop (its kernel shows up in the timers) and the
    cut inside the kernels equal to filtering afterwards."""
    from metafast_amd import lib as L
    _reset(gpu_ctx)
    gpu_ctx.set_option("profile", 1)
    gpu_ctx.reset_timers()
    n = int(os.environ.get("MF_DEPTH_TEST_READS", "500000"))           # (1 M reads: 90 s of CPU checking for the three cases; the plan's branches are the same)
    b, o = L.synth_reads_host(0x4D45544146415354, 3, 0, n, 150, scale, sub16k)
    tb, to = to_device(b, o)
    t = gpu_ctx.count_device(tb.data_ptr(), to.data_ptr(), n, int(o[-1]), 31, 0)
    assert "k_skm_pilot" in gpu_ctx.kernel_report()
    gk, gc = t.export()
    ot = oracle.Table().count_buffer(b, o, 31)
    ok, ov = ot.export()
    assert np.array_equal(gk, ok) and np.array_equal(gc.astype(np.int32), ov)
    assert t.occurrences() == n * 120
    t2, n_all = gpu_ctx.count_device_above(tb.data_ptr(), to.data_ptr(), n, int(o[-1]), 31, 1)
    assert n_all == len(gk)
    k2, c2 = t2.export()
    keep = gc > 1
    assert np.array_equal(k2, gk[keep]) and np.array_equal(c2, gc[keep])
    h = t2.hist()
    assert int(h.sum()) == len(gk) and int(h[1]) == int((gc == 1).sum())
    gpu_ctx.set_option("skm_pilot", 0)                      # the plan from the occurrences alone: the same table
    t3 = gpu_ctx.count_device(tb.data_ptr(), to.data_ptr(), n, int(o[-1]), 31, 0)
    gpu_ctx.set_option("skm_pilot", 1)
    k3, c3 = t3.export()
    assert np.array_equal(k3, gk) and np.array_equal(c3, gc)
    gpu_ctx.set_option("profile", 0)

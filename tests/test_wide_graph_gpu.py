"""NO-REFERENCE EXTENSION: "counted + graphed" for 32 <= k <= 63 (metafast_amd/csrc/mf_wide.hip + mf_wgraph.hip) -- BASELINE.json's config 4
has a k = 63 leg; the reference rejects k > 31 (src/tools/KmersCounterMain.java:66-73).  No parity claim against the reference: the checker
is oracle/mf_oracle_wide.c = the pinned oracle's own text compiled for 128-bit keys (tied to the pinned build at k <= 31, the golden matrix
included: tests/test_oracle_wide_cpu.py).  Through the C-ABI, bit-exact: cut tables, unitigs (strand-normalised multisets with weights and
the 0 / 1 / 2-emission census), cutter tables, components at several threshold levels, features, the distance matrix."""
import numpy as np
import pytest

from util import branchy_reads, canon_seq, emission_census, genome_reads, pack_reads, to_device

pytestmark = pytest.mark.gpu


def _norm_seqs(seqs):
    return sorted((canon_seq(s), a, mn, mx) for s, a, mn, mx in seqs)


def _same_table(got, keys, vals):
    hi, lo, cnt = got
    return np.array_equal(hi, keys["hi"]) and np.array_equal(lo, keys["lo"]) and np.array_equal(cnt.astype(np.int32), vals)


def _gpu_pipeline(ctx, samples, k, b, l, b1, b2):
    from metafast_amd import pipeline as P
    import torch
    dev = [to_device(bs, off) + (len(off) - 1, len(bs)) for bs, off in samples]
    return P.run_samples_wide(ctx, [(d[0], d[1], d[2], d[3]) for d in dev], k=k, b=b, l=l, b1=b1, b2=b2)


def _check_pipeline(ctx, O, samples, k, b, l, b1, b2, min_levels=1):
    want = O.run_pipeline_wide(samples, k, b=b, l=l, b1=b1, b2=b2)
    got = _gpu_pipeline(ctx, samples, k, b, l, b1, b2)
    for i, ws in enumerate(want["samples"]):
        assert got["n_distinct"][i] == ws["n_distinct"]
        assert _same_table(got["goods"][i].export(), *ws["good"].export())
        gs = got["seqss"][i].export()
        assert _norm_seqs(gs) == _norm_seqs(ws["seqs"].all())
    assert _same_table(got["cutter"].export(), *want["cutter"].export())
    wc = want["comps"].all()
    gc = got["comps"].export()
    assert [(int(a), int(w), int(t)) for a, w, t in zip(gc["sizes"], gc["weights"], gc["thr"])] == [(a, w, t) for a, w, t, _ in wc]
    at = 0
    for n, _, _, km in wc:
        assert np.array_equal(gc["hi"][at:at + n], km["hi"]) and np.array_equal(gc["lo"][at:at + n], km["lo"])
        at += n
    assert at == len(gc["hi"])
    assert len({t for _, _, t, _ in wc}) >= min_levels, "the case must reach that many threshold levels"
    assert np.array_equal(got["vecs"], want["vecs"])
    for i, ws in enumerate(want["samples"]):
        v, br = want["comps"].features(ws["good"], 0)
        assert np.array_equal(got["breadths"][i], br)
    if want["matrix"] is not None:
        assert np.array_equal(got["matrix"], want["matrix"])          # the same sums, the same IEEE division
    return got, want


@pytest.mark.parametrize("k", [32, 33, 40, 47, 62, 63])
def test_wide_pipeline_against_the_128bit_oracle(gpu_ctx, oracle, k):
    """three samples of one genome with repeats and errors (branches, tips, bubbles): every stage, two threshold levels"""
    samples = [branchy_reads(20 + i, genome_seed=300 + k, n=9000) for i in range(3)]
    got, want = _check_pipeline(gpu_ctx, oracle, samples, k, b=1, l=100, b1=50, b2=3000, min_levels=2)
    assert len(want["comps"]) >= 3


@pytest.mark.parametrize("k,seed", [(33, 7), (47, 8), (63, 9)])
def test_wide_unitigs_emission_rule(gpu_ctx, oracle, k, seed):
    """the 0 / 1 / 2-emission rule (AddSequencesShiftingRightTask.processSequence :101-121) must be exercised at 128 bits too"""
    b, o = branchy_reads(seed)
    db, do = to_device(b, o)
    t, n_all = gpu_ctx.count_wide_above(db.data_ptr(), do.data_ptr(), len(o) - 1, len(b), k, 1)
    got = gpu_ctx.build_unitigs_wide(t, 1, 100).export()
    wt = oracle.WTable().count_buffer(b, o, k)
    assert n_all == len(wt)
    want = oracle.wide_build_unitigs(wt.good(1), k, 1, 100).all()
    started, long_enough, emitted = oracle.wide_unitig_census()
    assert _norm_seqs(got) == _norm_seqs(want)
    once, twice = emission_census(got)
    assert emitted == once + 2 * twice == len(got) and long_enough % 2 == 0
    never = long_enough // 2 - (once + twice)
    assert once > 20 and (twice > 0 or never > 0), (once, twice, never)


def test_wide_unitigs_cycles_palindromes_and_thresholds(gpu_ctx, oracle):
    cyc = "ACGTTGCATGCCGATAGGCTTAACCGGATATCCGGTTAAGCTTGACCATGCAGGTCAATGCCGTAAGCTAGGATCC"
    reads = [cyc * 3, "A" * 120, "ACGT" * 40, "AATT" * 40, "AT" * 70, cyc[5:] + cyc[:40]]
    b, o = pack_reads(reads)
    db, do = to_device(b, o)
    for k in (32, 33, 36, 41):
        for thr, l in ((0, k), (0, 2 * k)):
            t, _ = gpu_ctx.count_wide_above(db.data_ptr(), do.data_ptr(), len(o) - 1, len(b), k, thr)
            got = gpu_ctx.build_unitigs_wide(t, thr, l).export()
            want = oracle.wide_build_unitigs(oracle.WTable().count_buffer(b, o, k).good(thr), k, thr, l).all()
            assert _norm_seqs(got) == _norm_seqs(want), (k, thr, l)
    # an uncut table handed to the unitig builder with a threshold: the filter happens inside
    rng = np.random.default_rng(5)
    bs, off = genome_reads(rng, 5000, 2500, 150, err=0.004)
    db, do = to_device(bs, off)
    full = gpu_ctx.count_wide_table(db.data_ptr(), do.data_ptr(), len(off) - 1, len(bs), 35)
    for thr in (0, 2, 4):
        got = gpu_ctx.build_unitigs_wide(full, thr, 60).export()
        want = oracle.wide_build_unitigs(oracle.WTable().count_buffer(bs, off, 35).good(thr), 35, thr, 60).all()
        assert _norm_seqs(got) == _norm_seqs(want) and len(want) > 0
        assert _same_table(full.filter(thr).export(), *oracle.WTable().count_buffer(bs, off, 35).export(thr))


def test_wide_long_path_doubles_the_jump_words(gpu_ctx, oracle):
    """an error-free genome: one unitig of 1e5 63-mers -- a wide table has no minimizer partitions (one hop per jump word), so the
    chunked walks give way to the doubled jump words (mf_unitig.hip U3b) as for k <= 31"""
    rng = np.random.default_rng(63)
    bs, off = genome_reads(rng, 100_000, 20_000, 150, err=0.0)
    db, do = to_device(bs, off)
    before = gpu_ctx.stat("unitig_doublings")
    t, _ = gpu_ctx.count_wide_above(db.data_ptr(), do.data_ptr(), len(off) - 1, len(bs), 63, 1)
    got = gpu_ctx.build_unitigs_wide(t, 1, 100).export()
    want = oracle.wide_build_unitigs(oracle.WTable().count_buffer(bs, off, 63).good(1), 63, 1, 100).all()
    assert _norm_seqs(got) == _norm_seqs(want) and max(len(s[0]) for s in got) > 20_000
    assert gpu_ctx.stat("unitig_doublings") == before + 1


def test_wide_components_split_to_higher_thresholds(gpu_ctx, oracle):
    """deep samples of one genome, a small window: oversize components are re-split on value >= thr + 1 several times
    (ComponentsBuilder.java:86-150 on 2k-bit k-mers)"""
    samples = [branchy_reads(40 + i, genome_seed=41, n=14000) for i in range(4)]
    _check_pipeline(gpu_ctx, oracle, samples, 35, b=1, l=80, b1=30, b2=900, min_levels=3)


def test_wide_rejects_bad_arguments(gpu_ctx):
    from metafast_amd import lib as L
    rng = np.random.default_rng(1)
    bs, off = genome_reads(rng, 2000, 300, 100)
    db, do = to_device(bs, off)
    with pytest.raises(L.MetafastError):
        gpu_ctx.count_wide_above(db.data_ptr(), do.data_ptr(), len(off) - 1, len(bs), 31, 1)          # k <= 31: mf_count_device_above
    with pytest.raises(L.MetafastError):
        gpu_ctx.count_wide_above(db.data_ptr(), do.data_ptr(), len(off) - 1, len(bs), 64, 1)
    t33, _ = gpu_ctx.count_wide_above(db.data_ptr(), do.data_ptr(), len(off) - 1, len(bs), 33, 0)
    t35, _ = gpu_ctx.count_wide_above(db.data_ptr(), do.data_ptr(), len(off) - 1, len(bs), 35, 0)
    comps = gpu_ctx.cut_components_wide(t33, 1, 100000)
    with pytest.raises(L.MetafastError):
        gpu_ctx.features_wide(comps, t35, 0)                          # components of 33-mers, a sample of 35-mers
    # nothing to do: empty inputs give empty results
    e, n_all = gpu_ctx.count_wide_above(db.data_ptr(), do.data_ptr(), 0, 0, 40, 1)
    assert n_all == 0 and e.stats()[0] == 0
    assert len(gpu_ctx.build_unitigs_wide(e, 1, 100).export()) == 0
    assert len(gpu_ctx.cut_components_wide(e, 1, 100)) == 0

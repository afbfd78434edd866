"""The wide oracle (oracle/mf_oracle_wide.c = mf_oracle_core.inc compiled for 128-bit keys; NO-REFERENCE EXTENSION for k = 32..63,
the reference rejects k > 31: src/tools/KmersCounterMain.java:66-73).  It is the text the reference's golden matrix pins at 64 bits
(tests/test_oracle_golden.py); here it is (1) tied to that pinned build on every output at k <= 31, the golden matrix included, and
(2) checked for the properties that do not depend on the key width at k = 33 .. 63."""
import numpy as np
import pytest

from util import branchy_reads, canon_seq, genome_reads


def _norm_seqs(seqs):
    return sorted((canon_seq(s), a, mn, mx) for s, a, mn, mx in seqs)


@pytest.mark.parametrize("k", [15, 21, 30, 31])
def test_wide_oracle_equals_pinned_oracle_up_to_31(oracle, k):
    O = oracle
    samples = [branchy_reads(5 + i, genome_seed=77, n=2500) for i in range(2)]
    narrow_goods, narrow_seqs, cutter = [], [], O.Table()
    for bases, off in samples:
        t = O.Table().count_buffer(bases, off, k)
        keys, vals = t.export()
        good = O.Table()
        for kk, vv in zip(keys[vals > 1].tolist(), vals[vals > 1].tolist()):
            good.add(kk, vv)
        seqs = O.build_unitigs(good, k, 1, 60)
        cutter.count_seqs(seqs, k, 60)
        narrow_goods.append(good); narrow_seqs.append(seqs)
    ncomps = O.cut_components(cutter, k, 20, 400)
    w = O.run_pipeline_wide(samples, k, b=1, l=60, b1=20, b2=400)
    for i, s in enumerate(w["samples"]):
        nk, nv = narrow_goods[i].export()
        wk, wv = s["good"].export()
        assert np.array_equal(wk["lo"], nk) and not wk["hi"].any() and np.array_equal(wv, nv)
        assert _norm_seqs(s["seqs"].all()) == _norm_seqs(narrow_seqs[i].all())
    ck, cv = cutter.export()
    wk, wv = w["cutter"].export()
    assert np.array_equal(wk["lo"], ck) and np.array_equal(wv, cv)
    assert len(w["comps"]) == len(ncomps) and len(ncomps) > 3
    assert len({c[2] for c in ncomps.all()}) >= 2, "the case must reach a second threshold level"
    for a, b_ in zip(w["comps"].all(), ncomps.all()):
        assert a[:3] == b_[:3] and np.array_equal(a[3]["lo"], b_[3]) and not a[3]["hi"].any()
    for i, s in enumerate(w["samples"]):
        nv, nb = ncomps.features(narrow_goods[i], 0)
        wv_, wb = w["comps"].features(s["good"], 0)
        assert np.array_equal(nv, wv_) and np.array_equal(nb, wb)


def test_wide_oracle_reproduces_the_golden_matrix(oracle, ref_files):
    """the reference's only golden vector for the path (test_data/meta_test_matrix.txt), through the 128-bit build"""
    O = oracle
    samples = [O.read_file(f) for f in sorted(ref_files)]
    r = O.run_pipeline_wide(samples, 31)
    m = r["matrix"]
    assert m is not None
    assert (m[0][1], m[0][2], m[1][2]) == (0.5691162409506898, 0.2981399448537721, 0.8448331091037222)


@pytest.mark.parametrize("k", [32, 33, 47, 63])
def test_wide_oracle_properties_beyond_31(oracle, k):
    O = oracle
    rng = np.random.default_rng(k)
    bases, off = genome_reads(rng, 6000, 1500, 150, err=0.003)
    t = O.WTable().count_buffer(bases, off, k)
    keys, vals = t.export()
    hi, lo, cnt, n_occ = O.count_wide(bases, off, k) if k >= 32 else (None,) * 4
    assert np.array_equal(keys["hi"], hi) and np.array_equal(keys["lo"], lo) and np.array_equal(vals, cnt)   # the sort-based counter agrees
    assert int(vals.sum()) == n_occ == 1500 * (150 - k + 1)
    ints = O.w128_to_ints(keys)
    assert ints == sorted(ints) and all(x < (1 << (2 * k)) for x in ints)
    for x in ints[:50]:
        assert O.wide_revcomp(O.wide_revcomp(x, k), k) == x and x <= O.wide_revcomp(x, k)
    good = t.good(1)
    seqs = O.wide_build_unitigs(good, k, 1, k)
    allseq = seqs.all()
    assert len(allseq) > 0
    # every k-mer of every unitig is a good k-mer, weights are its counts
    for s, a, mn, mx in allseq[:40]:
        vs = []
        for i in range(len(s) - k + 1):
            x = 0
            for ch in s[i:i + k]:
                x = (x << 2) | "AGCT".index(ch)
            x = min(x, O.wide_revcomp(x, k))
            v = good.get(x)
            assert v > 1
            vs.append(v)
        assert (min(vs), max(vs), sum(vs) // len(vs)) == (mn, mx, a)
    # strand symmetry: the reverse-complemented reads give the same tables and unitigs
    comp = np.zeros(256, dtype=np.uint8)
    for a_, b_ in zip(b"ACGT", b"TGCA"):
        comp[a_] = b_
    rb = np.concatenate([comp[bases[int(off[i]):int(off[i + 1])]][::-1] for i in range(len(off) - 1)])
    t2 = O.WTable().count_buffer(rb, off, k)
    k2, v2 = t2.export()
    assert np.array_equal(k2, keys) and np.array_equal(v2, vals)
    cutter = O.WTable().count_seqs(seqs, k, k)
    comps = O.wide_cut_components(cutter, k, 5, 200)
    members = set()
    for size, weight, thr, km in comps.all():
        assert size == len(km) and 5 <= size <= 200
        ks = O.w128_to_ints(km)
        assert ks == sorted(ks) and not (members & set(ks))
        members |= set(ks)
        assert weight == sum(cutter.get(x) for x in ks)
    vec, br = comps.features(good, 0)
    assert len(vec) == len(comps) and (vec >= 0).all()

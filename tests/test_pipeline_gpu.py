"""GPU parity for unitigs, component cutter, features and the distance matrix (through the C-ABI) vs the CPU oracle,
including the reference's golden matrix on its own test_data."""
import os

import numpy as np
import pytest

from conftest import REF_DATA
from util import branchy_reads, canon_seq, emission_census, genome_reads, gpu_count, pack_reads

pytestmark = pytest.mark.gpu


def _oracle_good(oracle, bases, off, k, b):
    t = oracle.Table().count_buffer(bases, off, k)
    keys, vals = t.export(b)
    g = oracle.Table()
    for kk, vv in zip(keys.tolist(), vals.tolist()):
        g.add(kk, vv)
    return g


def _norm_seqs(seqs):
    return sorted((canon_seq(s), a, mn, mx) for s, a, mn, mx in seqs)


def _check_unitigs(ctx, oracle, bases, off, k, b, l):
    t = gpu_count(ctx, bases, off, k)
    gs = ctx.build_unitigs(t, b, l)
    got = gs.export()
    want = oracle.build_unitigs(_oracle_good(oracle, bases, off, k, b), k, b, l).all()
    assert len(got) == len(want), (len(got), len(want))
    assert _norm_seqs(got) == _norm_seqs(want)
    n, total = gs.stats()
    assert n == len(want) and total == sum(len(s[0]) for s in want)
    return gs, got


def test_unitigs_reference_data(gpu_ctx, oracle, ref_files):
    for f, (ns, nt) in zip(ref_files, [(15, 17322), (29, 7910), (25, 11123)]):
        b, o = oracle.read_file(f)
        gs, got = _check_unitigs(gpu_ctx, oracle, b, o, 31, 1, 100)
        assert (len(got), sum(len(s[0]) for s in got)) == (ns, nt)


# (printed once, printed twice, long enough but never printed) per seed: the reference's emission rule `canon(start) <=
# canon(the k-mer the walk stopped AT)` (AddSequencesShiftingRightTask.processSequence :101-121) -- the walk may stop one
# k-mer beyond the last appended one, so the two walks of one path do not always decide complementarily.  Numbers from
# the oracle's census (tests/golden/make_golden.py); UNPINNED by any reference vector: oracle = my reading of the Java.
BRANCHY_CENSUS = {7: (48, 6, 4), 8: (48, 6, 3), 9: (49, 8, 9)}


@pytest.mark.parametrize("seed", [7, 8, 9])
def test_unitigs_branchy_emission_rule(gpu_ctx, oracle, seed):
    b, o = branchy_reads(seed)
    gs, got = _check_unitigs(gpu_ctx, oracle, b, o, 31, 1, 100)
    started, long_enough, emitted = oracle.unitig_census()        # (of the oracle run inside _check_unitigs)
    once, twice = emission_census(got)
    never = long_enough // 2 - (once + twice)
    assert long_enough % 2 == 0 and emitted == once + 2 * twice == len(got)
    assert (once, twice, never) == BRANCHY_CENSUS[seed]           # all three outcomes occur, in exactly these numbers


@pytest.mark.parametrize("k,b,l", [(5, 0, 5), (11, 0, 20), (16, 1, 40), (21, 2, 30), (31, 0, 31)])
def test_unitigs_small_k_and_thresholds(gpu_ctx, oracle, k, b, l):
    rng = np.random.default_rng(50 + k)
    bs, o = genome_reads(rng, 3000 if k > 8 else 300, 1500, 80, err=0.01)
    _check_unitigs(gpu_ctx, oracle, bs, o, k, b, l)


def test_unitigs_cycles_and_homopolymers(gpu_ctx, oracle):
    # an isolated cycle (never emitted), poly-A (self loop) and a palindrome-rich even-k case
    cyc = "ACGTTGCATGCCGATAGGCTTAACCGGATATCCGGTTAAGC"
    reads = [cyc * 3, "A" * 80, "ACGT" * 30, "AATT" * 30]
    b, o = pack_reads(reads)
    for k in (7, 8, 12):
        _check_unitigs(gpu_ctx, oracle, b, o, k, 0, k)


@pytest.mark.parametrize("k,b,l", [(31, 1, 100), (23, 5, 1200), (21, 0, 21)])
def test_unitigs_long_paths_double_the_jump_words(gpu_ctx, oracle, k, b, l):
    """Round 5: a walk is a chain of dependent loads; a unitig of 1e5 .. 1e6 k-mers (what -k 23 -b 5 -l 1200, the reference's example
    parameters, leave of an abundant genome) kept one lane busy for the whole kernel.  Walks still under way after the chunked rounds double the
    jump words over the entry nodes (mf_unitig.hip U3b) and the segment cuts walk the 32-jump words.  An error-free genome of 200 kb at depth
    22: one path of 2e5 k-mers -- the doubling runs by itself (stat unitig_doublings), gives the oracle's sequences, and so does the walk with the
    doubling put off (ut_double_after = 64) and brought forward to the first round (1); the small branchy / cyclic cases with it brought forward."""
    rng = np.random.default_rng(900 + k)
    bs, o = genome_reads(rng, 200_000, 30_000, 150, err=0.0)
    before = gpu_ctx.stat("unitig_doublings")
    gs, got = _check_unitigs(gpu_ctx, oracle, bs, o, k, b, l)
    assert gpu_ctx.stat("unitig_doublings") == before + 1 and max(len(s[0]) for s in got) > 50_000
    ref = _norm_seqs(got)
    for after in (64, 1):
        try:
            gpu_ctx.set_option("ut_double_after", after)
            t = gpu_count(gpu_ctx, bs, o, k)
            assert _norm_seqs(gpu_ctx.build_unitigs(t, b, l).export()) == ref
            if after == 1 and k == 31:
                for seed in (7, 9):
                    bb, oo = branchy_reads(seed)
                    _check_unitigs(gpu_ctx, oracle, bb, oo, 31, 1, 100)
                cyc = "ACGTTGCATGCCGATAGGCTTAACCGGATATCCGGTTAAGC"
                bb, oo = pack_reads([cyc * 3, "A" * 80, "ACGT" * 30, "AATT" * 30, cyc[3:] + cyc[:20]])
                for kk in (7, 8, 12):
                    _check_unitigs(gpu_ctx, oracle, bb, oo, kk, 0, kk)
        finally:
            gpu_ctx.set_option("ut_double_after", 4)
    assert gpu_ctx.stat("unitig_doublings") >= before + 2


def _check_components(ctx, oracle, cutter_gpu, cutter_or, k, b1, b2):
    gc = ctx.cut_components(cutter_gpu, b1, b2)
    oc = oracle.cut_components(cutter_or, k, b1, b2)
    got, want = gc.export(), oc.all()
    assert [(a, b, c) for a, b, c, _ in got] == [(a, b, c) for a, b, c, _ in want]
    for (_, _, _, gk), (_, _, _, ok) in zip(got, want):
        assert np.array_equal(gk, ok)
    return gc, oc


def _pipeline(ctx, oracle, inputs, k, b, l, b1, b2):
    """inputs: list of (bases, offsets); mirrors DistanceMatrixBuilderMain's step wiring on the device"""
    import torch
    tables, goods, seqs, o_goods, o_seqs = [], [], [], [], []
    for bases, off in inputs:
        t = gpu_count(ctx, bases, off, k)
        tables.append(t)
        goods.append(t.filter(b))
        seqs.append(ctx.build_unitigs(t, b, l))
        og = _oracle_good(oracle, bases, off, k, b)
        o_goods.append(og)
        o_seqs.append(oracle.build_unitigs(og, k, b, l))
    # cutter table: k-mers of all samples' unitigs (ComponentCutterMain.java:81)
    views = [s.device_view() for s in seqs]
    nb = sum(v["n_bases"] for v in views)
    ns = sum(v["n"] for v in views)
    allb = torch.zeros(nb + 64, dtype=torch.uint8, device="cuda")
    allo = torch.zeros(ns + 1, dtype=torch.int64, device="cuda")
    pb = po = 0
    for v in views:
        if v["n"] == 0:
            continue
        tb = torch.empty(0)  # noqa: F841
        import ctypes
        # copy through torch by wrapping the device pointers
        src_b = _wrap(v["bases"], v["n_bases"], torch.uint8)
        src_o = _wrap(v["offsets"], (v["n"] + 1) * 8, torch.uint8).view(torch.int64)
        allb[pb:pb + v["n_bases"]] = src_b
        allo[po:po + v["n"]] = src_o[:-1] + pb
        pb += v["n_bases"]
        po += v["n"]
    allo[ns] = nb
    ctx.synchronize()
    cutter = ctx.count_device(allb.data_ptr(), allo.data_ptr(), ns, nb, k, l)
    o_cutter = oracle.Table()
    for s in o_seqs:
        o_cutter.count_seqs(s, k, l)
    ck, cc = cutter.export()
    ok, ov = o_cutter.export()
    assert np.array_equal(ck, ok) and np.array_equal(cc.astype(np.int32), ov)
    gc, oc = _check_components(ctx, oracle, cutter, o_cutter, k, b1, b2)
    vecs, ovecs = [], []
    for g, og in zip(goods, o_goods):
        v, br = ctx.features(gc, g, 0)
        wv, wbr = oc.features(og, 0)
        assert np.array_equal(v, wv)
        assert np.array_equal(br, wbr)
        vecs.append(v)
        ovecs.append(wv)
    return np.array(vecs).reshape(len(inputs), -1), gc


def _wrap(ptr, nbytes, dtype):
    """torch view of a device buffer owned by the library (for tests only)"""
    import torch

    class _Holder:
        pass

    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 3}
    return torch.as_tensor(h, device="cuda")


def test_threshold_levels_on_gpu(gpu_ctx, oracle):
    """four samples of one branchy genome: cutter values up to 4 + doubled unitigs, so that the split of the components
    larger than b2 runs through threshold levels 2, 3, 4 ... (ComponentsBuilder.findAllComponents / bfs :198-270)"""
    inputs = [branchy_reads(rs, genome_seed=7, n=6000) for rs in (107, 117, 127, 137)]
    vecs, gc = _pipeline(gpu_ctx, oracle, inputs, 31, 1, 100, 100, 1000)
    comps = gc.export()
    thr = sorted({c[2] for c in comps})
    assert len(comps) == 13 and thr == [1, 3, 4, 5, 6]            # (known answer of the oracle; unpinned by the reference)


def test_golden_matrix_on_gpu(gpu_ctx, oracle, ref_files):
    """the reference's golden vector: test_data/meta_test_matrix.txt (k=31 b=1 l=100 b1=1000 b2=10000)"""
    from metafast_amd import lib as L
    inputs = [oracle.read_file(f) for f in ref_files]
    vecs, gc = _pipeline(gpu_ctx, oracle, inputs, 31, 1, 100, 1000, 10000)
    assert vecs.tolist() == [[41935, 38354, 20375, 14211], [20208, 0, 0, 11337], [6517, 34484, 20359, 749]]
    m = L.bray_curtis(vecs)
    assert m[0, 1] == 0.5691162409506898 and m[0, 2] == 0.2981399448537721 and m[1, 2] == 0.8448331091037222
    assert abs(m - oracle.bray_curtis(vecs)).max() <= 1e-6     # north_star tolerance (expected exact)


def test_threshold_split_on_gpu(gpu_ctx, oracle, ref_files):
    """b1=50 b2=500 forces oversize components to be re-split at threshold 2 (ComponentsBuilder.java:157-180)"""
    from metafast_amd import lib as L
    inputs = [oracle.read_file(f) for f in ref_files]
    vecs, gc = _pipeline(gpu_ctx, oracle, inputs, 31, 1, 100, 50, 500)
    comps = gc.export()
    assert len(comps) == 37 and [(a, b, c) for a, b, c, _ in comps[:3]] == [(448, 1076, 2), (456, 1066, 2), (426, 939, 2)]
    m = L.bray_curtis(vecs)
    assert m[0, 1] == 0.40246783273019704


def test_pipeline_synthetic_samples(gpu_ctx, oracle):
    """three synthetic samples from the bench generator (shared genomes -> cross-sample components)"""
    from metafast_amd import lib as L
    inputs = [L.synth_reads_host(0x4D45544146415354, s, 0, 30000, 150, 3000) for s in range(3)]
    vecs, gc = _pipeline(gpu_ctx, oracle, inputs, 31, 1, 100, 100, 5000)
    assert len(gc) > 0
    m = L.bray_curtis(vecs)
    assert abs(m - oracle.bray_curtis(vecs)).max() <= 1e-6


def test_run_sample_single_gpu(gpu_ctx, oracle, ref_files):
    """the device-resident driver bench.py times (metafast_amd.pipeline.run_sample) on one sample = the oracle's pipeline"""
    import torch
    from metafast_amd import pipeline as P
    from util import to_device
    f = ref_files[0]
    b, o = oracle.read_file(f)
    tb, to = to_device(b, o)
    r = P.run_sample(gpu_ctx, tb, to, len(o) - 1, int(o[-1]), k=31, b=1, l=100, b1=1000, b2=10000, device="cuda")
    want = oracle.run_pipeline([f])
    s = want["samples"][0]
    assert r["n_distinct"] == s["n_distinct"] and len(r["good"]) == s["n_good"]
    assert len(r["seqs"]) == len(s["seqs"]) and len(r["comps"]) == len(want["comps"])
    assert r["vec"].tolist() == want["vecs"][0].tolist()
    assert r["matrix"].shape == (1, 1) and r["matrix"][0, 0] == 0.0
    for key in ("good", "seqs", "cutter", "comps"):
        r[key].close()


def test_features_from_reads(gpu_ctx, oracle, ref_files, tmp_path):
    """--use-reads-for-calculating-features: occurrences of the component k-mers in the READS (64-bit, not the saturated
    table counts), device form, file form and the driver's matrix-builder switch, against the oracle's restatement"""
    import os, subprocess
    from conftest import ROOT
    from util import to_device, pack_reads
    want = oracle.run_pipeline(ref_files)
    comps_file = tmp_path / "components.bin"
    want["comps"].write(str(comps_file), None)
    comps = gpu_ctx.load_components(str(comps_file))
    for f in ref_files:
        b, o = oracle.read_file(f)
        tb, to = to_device(b, o)
        for thr in (0, 2):
            ev, eb = oracle.features_from_reads(want["comps"], b, o, 31, thr)
            gv, gb = gpu_ctx.features_reads(comps, tb.data_ptr(), to.data_ptr(), len(o) - 1, int(o[-1]), 31, thr)
            assert gv.tolist() == ev.tolist() and np.array_equal(gb, eb)
    # file form
    vec, br = tmp_path / "x.vec", tmp_path / "x.breadth"
    gpu_ctx.features_reads_files(str(comps_file), [ref_files[1]], 31, 0, str(vec), str(br))
    b, o = oracle.read_file(ref_files[1])
    ev, eb = oracle.features_from_reads(want["comps"], b, o, 31, 0)
    assert [int(x) for x in vec.read_text().split()] == ev.tolist()
    assert [float(x) for x in br.read_text().split()] == eb.tolist()
    # a k-mer that occurs more often than a short count holds: poly-A component k-mers are counted in full
    reads = ["A" * 40] * 4000 + ["ACGTTGCAAGGCTTAACGGATTACAGGCATCGATCGGCTAAGCT"] * 3
    pb, po = pack_reads(reads)
    t = oracle.Table().count_buffer(pb, po, 31)
    oc = oracle.cut_components(t, 31, 1, 1000)
    cf = tmp_path / "c2.bin"
    oc.write(str(cf), None)
    gc = gpu_ctx.load_components(str(cf))
    tb, to = to_device(pb, po)
    ev, eb = oracle.features_from_reads(oc, pb, po, 31, 0)
    gv, gb = gpu_ctx.features_reads(gc, tb.data_ptr(), to.data_ptr(), len(po) - 1, int(po[-1]), 31, 0)
    assert gv.tolist() == ev.tolist() and max(ev.tolist()) == 40000          # 4000 reads x 10 k-mers, beyond 32767
    # driver: matrix-builder --use-reads-for-calculating-features (one vector per reads FILE)
    wd = tmp_path / "w"
    r = subprocess.run([os.path.join(ROOT, "metafast.sh"), "-k", "31", "-i", *ref_files, "-w", str(wd), "--use-reads-for-calculating-features"],
                       capture_output=True, text=True, timeout=300, cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    for f in ref_files:
        name = os.path.basename(f)[:-3]
        b, o = oracle.read_file(f)
        ev, _ = oracle.features_from_reads(want["comps"], b, o, 31, 0)
        got = [int(x) for x in (wd / "features-calculator" / "vectors" / (name + ".vec")).read_text().split()]
        assert got == ev.tolist()

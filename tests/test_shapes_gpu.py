"""BASELINE.json's other shapes on one MI355X.

* config 4's k = 21 leg at full size (200 M x 150 bp reads; MF_SHAPES_K21_READS overrides): the size-independent properties of
  tests/test_fullsize_gpu.py -- occurrence conservation, an independent exact recount of a key sample in plain torch ops,
  the cut inside the counting pass = filter afterwards, additivity of the two halves (saturating), the all-counts histogram.
  The graph stages of the same shape (unitigs -> cutter -> components -> features) go through the property set of
  test_fullsize_gpu.py::test_pipeline_properties at k = 21.
* the whole pipeline at k = 21 and at k = 20 (the lower edge of the super-k-mer path) on 2 x 1 M synthetic reads against the
  ORACLE's pipeline, with a component window that forces threshold >= 2 re-splits.
* config 3's semantics (8 samples -> 8 x 8 matrix) with the 8 samples one after the other on this GPU through
  pipeline.run_samples, against the ORACLE's pipeline on the same reads (scaled down to MF_SHAPES_READS = 5 M reads per
  sample; the oracle's kmer-counter and seq-builder run as 8 CPU child processes and leave the reference's own files,
  .kmers.bin / .seq.fasta, from which the oracle's cutter, features and matrix are computed here): components, vectors and
  matrix identical."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

RL = 150
SEED = 0x4D45544146415354
CHUNK = 1_000_000


def _canon_chunk(torch, lut, chunk, k):
    codes = lut[chunk.long()]
    m = RL - k + 1
    fw = torch.zeros((chunk.shape[0], m), dtype=torch.int64, device=chunk.device)
    rc = torch.zeros_like(fw)
    for i in range(k):
        c = codes[:, i:i + m]
        fw = (fw << 2) | c
        rc = rc | ((3 - c) << (2 * i))
    return torch.minimum(fw, rc)


@pytest.fixture(autouse=True)
def _give_back(gpu_ctx):
    yield
    import torch
    gpu_ctx.trim()
    torch.cuda.empty_cache()


def test_200M_reads_k21_properties(gpu_ctx):
    import torch
    k = 21
    n = int(os.environ.get("MF_SHAPES_K21_READS", "200000000"))
    import gc
    gc.collect()                            # (contexts of earlier tests that are only waiting for the collector)
    gpu_ctx.trim()                          # (what earlier tests left in the context's arena and torch's cache)
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < n * 900:                      # reads + table + ONE slice of records (the run slices itself when two full buffers do not fit)
        pytest.skip("needs %.0f GB of free HBM, %.0f GB are free" % (n * 900 / 1e9, free / 1e9))
    dev = "cuda"
    bases = torch.zeros(n * RL + 64, dtype=torch.uint8, device=dev)
    offsets = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    gpu_ctx.synth_reads_device(SEED, 0, 0, n, RL, 1_000_000, bases.data_ptr(), offsets.data_ptr())
    gpu_ctx.synchronize()
    table = gpu_ctx.count_device(bases.data_ptr(), offsets.data_ptr(), n, n * RL, k, 0)
    n_distinct, n_occ = table.stats()
    assert n_occ == n * (RL - k + 1)
    lut = torch.zeros(256, dtype=torch.int64, device=dev)
    for ch, c in (("A", 0), ("G", 1), ("C", 2), ("T", 3)):
        lut[ord(ch)] = c
    b2d = bases[: n * RL].view(n, RL)
    g = torch.Generator(device="cpu").manual_seed(11)
    m = RL - k + 1
    pres = torch.cat([_canon_chunk(torch, lut, b2d[:4096], k).reshape(-1)[torch.randperm(4096 * m, generator=g)[:5000].to(dev)],
                      _canon_chunk(torch, lut, b2d[-4096:], k).reshape(-1)[torch.randperm(4096 * m, generator=g)[:5000].to(dev)]])
    absent = torch.randint(0, 1 << 42, (3000,), generator=g, dtype=torch.int64).to(dev)       # (4^21 keys: some of these do occur)
    sample = torch.unique(torch.cat([pres, absent]))
    acc = torch.zeros(sample.numel(), dtype=torch.int64, device=dev)
    acc1 = torch.zeros_like(acc)
    h = n // 2
    last = sample.numel() - 1
    for lo in range(0, n, CHUNK):
        canon = _canon_chunk(torch, lut, b2d[lo:lo + CHUNK], k).reshape(-1)
        idx = torch.searchsorted(sample, canon).clamp_(max=last)
        hit = sample[idx] == canon
        bc = torch.bincount(idx[hit], minlength=sample.numel())
        acc += bc
        if lo + CHUNK <= h:
            acc1 += bc
        del canon, idx, hit, bc
    exact, exact1 = acc.cpu().numpy(), acc1.cpu().numpy()
    keys = sample.cpu().numpy().astype(np.uint64)
    assert (exact > 0).sum() >= 9000
    got = table.lookup(keys).astype(np.int64)
    assert np.array_equal(got, np.where(exact > 0, np.minimum(exact, 32767), -1))
    assert n_distinct < n_occ
    hist = table.hist()
    assert int(hist.sum()) == n_distinct and int(hist[0]) == 0
    # the cut inside the counting pass + its histogram of everything
    above, n_all = gpu_ctx.count_device_above(bases.data_ptr(), offsets.data_ptr(), n, n * RL, k, 1)
    assert n_all == n_distinct and np.array_equal(above.hist(), hist)
    assert len(above) == n_distinct - int(hist[1])
    assert np.array_equal(above.lookup(keys).astype(np.int64), np.where(got > 1, got, -1))
    above.close()
    # additivity: first half (a whole number of chunks) (+) second half, saturating
    h = (h // CHUNK) * CHUNK
    if 0 < h < n:
        off2 = (offsets[h:] - offsets[h]).contiguous()
        t1 = gpu_ctx.count_device(bases.data_ptr(), offsets.data_ptr(), h, h * RL, k, 0)
        a = t1.lookup(keys).astype(np.int64)
        assert np.array_equal(a, np.where(exact1 > 0, np.minimum(exact1, 32767), -1))
        t1.close()
        t2 = gpu_ctx.count_device(bases.data_ptr() + h * RL, off2.data_ptr(), n - h, (n - h) * RL, k, 0)
        b = t2.lookup(keys).astype(np.int64)
        e2 = exact - exact1
        assert np.array_equal(b, np.where(e2 > 0, np.minimum(e2, 32767), -1))
        t2.close()
    table.close()


def test_200M_reads_k21_graph_properties(gpu_ctx):
    """config 4's k = 21 leg beyond counting: unitigs, cutter, components, features of one 200 M-read sample"""
    import torch
    from test_fullsize_gpu import check_pipeline_properties
    k = 21
    n = int(os.environ.get("MF_SHAPES_K21_READS", "200000000"))
    import gc
    gc.collect()
    gpu_ctx.trim()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < n * 1100:
        pytest.skip("needs %.0f GB of free HBM, %.0f GB are free" % (n * 1100 / 1e9, free / 1e9))
    bases = torch.zeros(n * RL + 64, dtype=torch.uint8, device="cuda")
    offsets = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    gpu_ctx.synth_reads_device(SEED, 0, 0, n, RL, 1_000_000, bases.data_ptr(), offsets.data_ptr())
    gpu_ctx.synchronize()
    info = check_pipeline_properties(gpu_ctx, bases, offsets, n, k)
    assert info["n_unitigs"] > 1000 and info["n_components"] > 100


def test_config4_union_of_8_samples_k21(gpu_ctx):
    """BASELINE config 4's k = 21 leg as it is specified: EIGHT samples x 200 M reads (MF_SHAPES_UNION_READS overrides), one after the
    other on this GPU through pipeline.run_samples, the cutter table and the components over the union of all eight samples' unitigs
    (ComponentCutterMain.java:78-114), one feature vector per sample, the 8 x 8 matrix.  No oracle at 2.08e11 k-mers: the
    size-independent properties -- occurrences, components inside the window, pairwise disjoint, closed under the 8-neighbour
    relation inside the cutter table at their threshold, weight = sum of the cutter's values, every sample's vec[c] = the sum of ITS
    counts over the component, the matrix = Bray-Curtis of the vectors (symmetric, zero diagonal, samples that share genomes closer
    than those that do not)."""
    import gc
    import torch
    from metafast_amd import lib as L, pipeline as P
    from test_fullsize_gpu import _neighbours
    k, S = 21, 8
    n = int(os.environ.get("MF_SHAPES_UNION_READS", "200000000"))
    gc.collect(); gpu_ctx.trim(); torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < n * 1250:
        pytest.skip("needs %.0f GB of free HBM, %.0f GB are free" % (n * 1250 / 1e9, free / 1e9))
    bases = torch.zeros(n * RL + 64, dtype=torch.uint8, device="cuda")
    offsets = torch.zeros(n + 1, dtype=torch.int64, device="cuda")

    def samples():
        for j in range(S):
            torch.cuda.synchronize()
            gpu_ctx.synth_reads_device(SEED, j, 0, n, RL, 1_000_000, bases.data_ptr(), offsets.data_ptr())
            torch.cuda.synchronize()
            yield bases, offsets, n, n * RL

    restarts0 = gpu_ctx.stat("slice_restarts")
    r = P.run_samples(gpu_ctx, samples(), k=k, b=1, l=100, b1=1000, b2=10000)
    del bases, offsets
    # (250 of the 288 GB are in use here: no counting run may have thrown its slices away and started over for want of a PLACE in the arena)
    assert gpu_ctx.stat("slice_restarts") == restarts0
    assert r["n_occ"] == S * n * (RL - k + 1) and len(r["goods"]) == S and r["vecs"].shape[0] == S
    comps, cutter = r["comps"], r["cutter"]
    cs = comps.export()
    assert len(cs) > 1000 and r["vecs"].shape[1] == len(cs)
    sizes = np.array([c[0] for c in cs])
    assert sizes.min() >= 1000 and sizes.max() <= 10000 and all(len(c[3]) == c[0] for c in cs)
    thrs = [c[2] for c in cs]
    assert thrs == sorted(thrs) and max(thrs) >= 2                          # ConnectedComponent.compareTo: threshold first; oversize components were re-split
    allk = np.concatenate([c[3] for c in cs])
    owner = np.repeat(np.arange(len(cs)), sizes)
    order = np.argsort(allk, kind="stable")
    allk, owner = allk[order], owner[order]
    assert np.all(allk[1:] != allk[:-1])
    rng = np.random.default_rng(4)
    picks = [int(x) for x in rng.choice(len(cs), size=24, replace=False)] + [len(cs) - 1]      # (the last one: the highest threshold)
    for ci in picks:
        size, weight, thr, km = cs[ci]
        val = cutter.lookup(km).astype(np.int64)
        assert val.min() >= thr and weight == val.sum()
        nb = _neighbours(km[:200], k)
        present = cutter.lookup(nb.reshape(-1)).astype(np.int64).reshape(nb.shape) >= thr
        pos = np.searchsorted(allk, nb).clip(max=len(allk) - 1)
        assert np.array_equal(present, (allk[pos] == nb) & (owner[pos] == ci))
    cutter.drop_index()
    for si in (0, 3, 7):                                                     # a sample's vector = ITS counts over the components' k-mers
        good = r["goods"][si]
        for ci in picks[:8]:
            sv = good.lookup(cs[ci][3]).astype(np.int64)
            assert r["vecs"][si][ci] == sv[sv > 0].sum()
        good.drop_index()
    m = r["matrix"]
    assert m.shape == (S, S) and np.allclose(m, m.T) and np.all(np.diag(m) == 0) and np.abs(m - L.bray_curtis(r["vecs"])).max() <= 1e-12
    # (the generator: sample s draws from genomes 32 s .. 32 s + 63 of 128 -- neighbours share half their genomes, samples two apart none)
    assert m[0, 1] < m[0, 2] and m[3, 4] < m[3, 5] and 0 < m[0, 1] < 1
    for x in r["goods"] + r["seqss"] + [cutter, comps]:
        x.close()


@pytest.mark.parametrize("k", [21, 20])
def test_pipeline_against_the_oracle_k21_k20(gpu_ctx, oracle, tmp_path, k):
    _samples_against_the_oracle(gpu_ctx, oracle, tmp_path, S=2, k=k, b=1, l=100, b1=500, b2=5000, n=1_000_000, min_thr=3)


def test_eight_samples_one_gpu_against_the_oracle(gpu_ctx, oracle, tmp_path):
    _samples_against_the_oracle(gpu_ctx, oracle, tmp_path, S=8, k=31, b=1, l=100, b1=1000, b2=10000,
                                n=int(os.environ.get("MF_SHAPES_READS", "2000000")), min_thr=2)      # (5 M reads per sample: 155 of the suite's 640 s, the CPU side's; MF_SHAPES_READS=5000000 for that size)


def _samples_against_the_oracle(gpu_ctx, oracle, tmp_path, S, k, b, l, b1, b2, n, min_thr):
    import torch
    from metafast_amd import pipeline as P
    scale = max(n // 100, 1000)
    worker = os.path.join(ROOT, "tests", "oracle_sample_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(s), str(n), str(scale), str(k), str(b), str(l), str(tmp_path)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for s in range(S)]

    def samples():
        for s in range(S):
            bases = torch.zeros(n * RL + 64, dtype=torch.uint8, device="cuda")
            offsets = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            gpu_ctx.synth_reads_device(SEED, s, 0, n, RL, scale, bases.data_ptr(), offsets.data_ptr())
            gpu_ctx.synchronize()
            yield bases, offsets, n, n * RL

    r = P.run_samples(gpu_ctx, samples(), k=k, b=b, l=l, b1=b1, b2=b2)
    n_distinct = 0
    for p in procs:
        out, err = p.communicate(timeout=1500)
        assert p.returncode == 0, err[-2000:]
        n_distinct += int(out.strip().splitlines()[-1])
    assert r["n_distinct"] == n_distinct
    goods = [oracle.Table().load_kmers([str(tmp_path / f"s{s:02d}.kmers.bin")]) for s in range(S)]
    for g, og in zip(r["goods"], goods):
        assert len(g) == len(og)
    cutter = oracle.Table().count_files([str(tmp_path / f"s{s:02d}.seq.fasta") for s in range(S)], k, l)
    assert len(r["cutter"]) == len(cutter)
    oc = oracle.cut_components(cutter, k, b1, b2)
    got, want = r["comps"].export(), oc.all()
    assert len(want) > 50 and max(c[2] for c in want) >= min_thr
    assert [(a, w, t) for a, w, t, _ in got] == [(a, w, t) for a, w, t, _ in want]
    for (_, _, _, gk), (_, _, _, ok) in zip(got, want):
        assert np.array_equal(gk, ok)
    ovecs = np.array([oc.features(og, 0)[0] for og in goods], dtype=np.int64).reshape(S, len(want))
    assert np.array_equal(r["vecs"], ovecs)
    om = oracle.bray_curtis(ovecs)
    assert r["matrix"].shape == (S, S) and np.abs(r["matrix"] - om).max() <= 1e-6      # north_star tolerance (expected exact)
    assert np.array_equal(r["matrix"], om)

"""NO-REFERENCE EXTENSION at BASELINE config 4's per-GPU size: ONE sample of 200 M x 150 bp reads at k = 63, counted + graphed on one MI355X
(mf_wide.hip + mf_wgraph.hip), checked through properties that need no CPU pass over 1.8e10 63-mers -- the k = 63 twin of
tests/test_fullsize_gpu.py::test_pipeline_properties:

* occurrence conservation (N_occ = reads x 88) and the cut inside the pass (every kept count > b, fewer kept than distinct),
* unitigs: every k-mer of a sampled unitig is a good k-mer and (avg, min, max) are its k-mers' counts; lengths >= l,
* components: sizes within [b1, b2], pairwise disjoint, weight = the cutter values' sum, closed under the 8-neighbour relation inside the
  cutter table at the component's threshold,
* features: vec[c] = sum of the sample's counts over the component's k-mers; Bray-Curtis of a sample with itself is 0.
128-bit k-mers are Python ints on this side.  MF_WIDE_FULLSIZE_READS overrides the number of reads (the default needs ~250 GB of HBM)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K, RL = 63, 150
SEED = 0x4D45544146415354
N_READS = int(os.environ.get("MF_WIDE_FULLSIZE_READS", "200000000"))
CODE = {"A": 0, "G": 1, "C": 2, "T": 3}


def _rc(x, k=K):
    r = 0
    for _ in range(k):
        r = (r << 2) | (3 - (x & 3))
        x >>= 2
    return r


def _kmers_of(seq, k=K):
    out, fw, mask = [], 0, (1 << (2 * k)) - 1
    for i, ch in enumerate(seq):
        fw = ((fw << 2) | CODE[ch]) & mask
        if i + 1 >= k:
            out.append(min(fw, _rc(fw, k)))
    return out


def _neighbours(x, k=K):
    mask = (1 << (2 * k)) - 1
    out = []
    for n in range(4):
        for y in (((x << 2) | n) & mask, (x >> 2) | (n << (2 * k - 2))):
            out.append(min(y, _rc(y, k)))
    return out


def test_config4_k63_sample_counted_and_graphed(gpu_ctx):
    import torch
    from metafast_amd import lib as L, pipeline as P
    free, total = torch.cuda.mem_get_info()
    if free < N_READS * 1250:
        pytest.skip("needs %.0f GB of free HBM" % (N_READS * 1250 / 1e9))
    b, l, b1, b2 = 1, 100, 1000, 10000
    bases = torch.zeros(N_READS * RL + 64, dtype=torch.uint8, device="cuda")
    offsets = torch.zeros(N_READS + 1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    gpu_ctx.synth_reads_device(SEED, 0, 0, N_READS, RL, 1_000_000, bases.data_ptr(), offsets.data_ptr())
    gpu_ctx.synchronize()
    r = P.run_samples_wide(gpu_ctx, [(bases, offsets, N_READS, N_READS * RL)], k=K, b=b, l=l, b1=b1, b2=b2)
    del bases, offsets
    good, seqs, comps, cutter = r["goods"][0], r["seqss"][0], r["comps"], r["cutter"]
    n_good, n_occ, kk = good.stats()
    assert kk == K and n_occ == N_READS * (RL - K + 1) == r["n_occ"]
    assert 0 < n_good < r["n_distinct"][0] < n_occ
    rng = np.random.default_rng(3)
    # unitigs: a sample of them, k-mer by k-mer
    sq = seqs.export()
    assert len(sq) > 1000 and all(len(s[0]) >= l for s in sq[:10000])
    for i in rng.choice(len(sq), size=120, replace=False):
        s, avg, mn, mx = sq[int(i)]
        if len(s) > 4000:
            s = s[:4000]; avg = None
        cnt = good.lookup(_kmers_of(s)).astype(np.int64)
        assert cnt.min() > b
        if avg is not None:
            assert (mn, mx) == (cnt.min(), cnt.max()) and avg == int(cnt.sum() // len(cnt))
    del sq
    # components
    cs = comps.export()
    nc = len(cs["sizes"])
    assert nc > 100 and cs["sizes"].min() >= b1 and cs["sizes"].max() <= b2 and int(cs["sizes"].sum()) == len(cs["hi"])
    # pairwise disjoint: the (hi, lo) pairs are all different
    order = np.lexsort((cs["lo"], cs["hi"]))
    sh, sl = cs["hi"][order], cs["lo"][order]
    assert not np.any((sh[1:] == sh[:-1]) & (sl[1:] == sl[:-1]))
    owner_sorted = np.repeat(np.arange(nc), cs["sizes"].astype(np.int64))[order]

    def owner_of(x):
        hi, lo = x >> 64, x & 0xFFFFFFFFFFFFFFFF
        a, bnd = np.searchsorted(sh, np.uint64(hi), "left"), np.searchsorted(sh, np.uint64(hi), "right")
        j = a + np.searchsorted(sl[a:bnd], np.uint64(lo), "left")
        return int(owner_sorted[j]) if j < bnd and int(sl[j]) == lo else -1
    off = cs["offsets"].astype(np.int64)
    for ci in rng.choice(nc, size=25, replace=False):
        ci = int(ci)
        km = [(int(h) << 64) | int(lo_) for h, lo_ in zip(cs["hi"][off[ci]:off[ci + 1]], cs["lo"][off[ci]:off[ci + 1]])]
        assert km == sorted(km)
        thr, weight = int(cs["thr"][ci]), int(cs["weights"][ci])
        val = cutter.lookup(km).astype(np.int64)
        assert val.min() >= thr and weight == val.sum()
        for x in km[:60]:                                                  # closed: a neighbour in the graph is in the same component
            nb = _neighbours(x)
            present = cutter.lookup(nb).astype(np.int64) >= thr
            same = np.array([owner_of(y) == ci for y in nb])
            assert np.array_equal(present, same)
        sv = good.lookup(km).astype(np.int64)
        assert int(r["vecs"][0][ci]) == int(sv[sv > 0].sum())
        assert r["breadths"][0][ci] == (sv > 0).sum() / len(km)
    m = L.bray_curtis(np.stack([r["vecs"][0], r["vecs"][0]]))
    assert m[0, 1] == 0.0 and m[1, 0] == 0.0
    for x in (good, seqs, comps, cutter):
        x.close()
    gpu_ctx.trim()
    torch.cuda.empty_cache()

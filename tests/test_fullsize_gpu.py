"""BASELINE.json's full size (100 M x 150 bp reads, k = 31) on one MI355X, checked through properties that do not need
a CPU pass over 1.2e10 k-mers:

* occurrence conservation (N_occ = reads x 120),
* an INDEPENDENT exact recount, in plain torch tensor ops, of a sample of keys over the whole input
  (rolling fw / rc k-mers as IOUtils / ShortKmer define them: first base most significant, canonical = min(fw, rc),
  counts saturate at 32767) against mf_table_lookup,
* strand symmetry (the reverse-complemented reads give the same table),
* additivity (count(first half) (+) count(second half) = count(all), saturating),
* the threshold cut inside the counting kernels = filter afterwards,
* unitigs: every k-mer of an emitted unitig is a good k-mer and (avg, min, max) are its k-mers' counts,
* components: sizes within [b1, b2], pairwise disjoint, closed under the 8-neighbour relation inside the cutter table,
* features: vec[c] = sum of the sample's counts over the component's k-mers; Bray-Curtis of a sample with itself is 0.

MF_FULLSIZE_READS overrides the number of reads (the default needs ~230 GB of HBM at its peak)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K = 31
RL = 150
SEED = 0x4D45544146415354
N_READS = int(os.environ.get("MF_FULLSIZE_READS", "100000000"))
CHUNK = 1_000_000


def _codes_lut(torch, dev):
    lut = torch.zeros(256, dtype=torch.int64, device=dev)
    for ch, c in (("A", 0), ("G", 1), ("C", 2), ("T", 3)):
        lut[ord(ch)] = c
    return lut


def _canon_chunk(torch, lut, chunk):
    """chunk: uint8[n, RL] ASCII -> int64[n, RL-K+1] canonical k-mers (reference coding A0 G1 C2 T3)"""
    codes = lut[chunk.long()]
    m = RL - K + 1
    fw = torch.zeros((chunk.shape[0], m), dtype=torch.int64, device=chunk.device)
    rc = torch.zeros_like(fw)
    for i in range(K):
        c = codes[:, i:i + m]
        fw = (fw << 2) | c
        rc = rc | ((3 - c) << (2 * i))
    return torch.minimum(fw, rc)


def _recount(torch, lut, bases2d, sample_sorted):
    """exact occurrences of every key of sample_sorted (int64, ascending, unique) in all reads"""
    acc = torch.zeros(sample_sorted.numel(), dtype=torch.int64, device=bases2d.device)
    last = sample_sorted.numel() - 1
    for lo in range(0, bases2d.shape[0], CHUNK):
        canon = _canon_chunk(torch, lut, bases2d[lo:lo + CHUNK]).reshape(-1)
        idx = torch.searchsorted(sample_sorted, canon).clamp_(max=last)
        hit = sample_sorted[idx] == canon
        acc += torch.bincount(idx[hit], minlength=sample_sorted.numel())
        del canon, idx, hit
    return acc


@pytest.fixture(scope="module")
def full(gpu_ctx):
    import torch
    free, total = torch.cuda.mem_get_info()
    need = N_READS * 2300                   # reads, two tables and their lookup indexes at once
    if free < need:
        pytest.skip("needs %.0f GB of free HBM" % (need / 1e9))
    dev = "cuda"
    bases = torch.zeros(N_READS * RL + 64, dtype=torch.uint8, device=dev)
    offsets = torch.zeros(N_READS + 1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    gpu_ctx.synth_reads_device(SEED, 0, 0, N_READS, RL, 1_000_000, bases.data_ptr(), offsets.data_ptr())
    gpu_ctx.synchronize()
    table = gpu_ctx.count_device(bases.data_ptr(), offsets.data_ptr(), N_READS, N_READS * RL, K, 0)
    lut = _codes_lut(torch, dev)
    b2d = bases[: N_READS * RL].view(N_READS, RL)
    # sample: k-mers of reads from both ends of the input (present), plus random 62-bit keys (absent, but for luck)
    g = torch.Generator(device="cpu").manual_seed(7)
    pres = torch.cat([_canon_chunk(torch, lut, b2d[:4096]).reshape(-1)[torch.randperm(4096 * 120, generator=g)[:6000].to(dev)],
                      _canon_chunk(torch, lut, b2d[-4096:]).reshape(-1)[torch.randperm(4096 * 120, generator=g)[:6000].to(dev)]])
    absent = torch.randint(0, 1 << 62, (2000,), generator=g, dtype=torch.int64).to(dev)
    sample = torch.unique(torch.cat([pres, absent]))
    exact = _recount(torch, lut, b2d, sample)
    yield dict(torch=torch, bases=bases, offsets=offsets, b2d=b2d, table=table, lut=lut, sample=sample, exact=exact)
    table.close()


@pytest.fixture(autouse=True)
def _give_back(gpu_ctx):
    yield
    import torch
    gpu_ctx.trim()
    torch.cuda.empty_cache()


def _lookup(table, sample):
    return table.lookup(sample.cpu().numpy().astype(np.uint64)).astype(np.int64)


def test_occurrences_and_exact_recount(full):
    n_distinct, n_occ = full["table"].stats()
    assert n_occ == N_READS * (RL - K + 1)
    exact = full["exact"].cpu().numpy()
    assert (exact > 0).sum() >= 12000 - 100 and (exact == 0).sum() >= 1990
    want = np.where(exact > 0, np.minimum(exact, 32767), -1)
    assert np.array_equal(_lookup(full["table"], full["sample"]), want)
    assert n_distinct < n_occ and n_distinct >= (exact > 0).sum()


def test_strand_symmetry(full, gpu_ctx):
    torch = full["torch"]
    comp = torch.zeros(256, dtype=torch.uint8, device="cuda")
    for a, b in ("AT", "TA", "GC", "CG"):
        comp[ord(a)] = ord(b)
    rcb = torch.empty_like(full["bases"])
    r2d = rcb[: N_READS * RL].view(N_READS, RL)
    for lo in range(0, N_READS, 4 * CHUNK):
        r2d[lo:lo + 4 * CHUNK] = comp[full["b2d"][lo:lo + 4 * CHUNK].long()].flip(1)
    rcb[N_READS * RL:] = 0
    torch.cuda.synchronize()
    t = gpu_ctx.count_device(rcb.data_ptr(), full["offsets"].data_ptr(), N_READS, N_READS * RL, K, 0)
    assert t.stats() == full["table"].stats()
    assert np.array_equal(_lookup(t, full["sample"]), _lookup(full["table"], full["sample"]))
    t.close()


def test_additivity_of_halves(full, gpu_ctx):
    torch = full["torch"]
    h = N_READS // 2
    off2 = (full["offsets"][h:] - full["offsets"][h]).contiguous()
    t1 = gpu_ctx.count_device(full["bases"].data_ptr(), full["offsets"].data_ptr(), h, h * RL, K, 0)
    t2 = gpu_ctx.count_device(full["bases"].data_ptr() + h * RL, off2.data_ptr(), N_READS - h, (N_READS - h) * RL, K, 0)
    assert t1.stats()[1] + t2.stats()[1] == full["table"].stats()[1]
    a, b = _lookup(t1, full["sample"]), _lookup(t2, full["sample"])
    both = np.maximum(a, 0) + np.maximum(b, 0)
    want = np.where((a < 0) & (b < 0), -1, np.minimum(both, 32767))
    assert np.array_equal(_lookup(full["table"], full["sample"]), want)
    n, n1, n2 = full["table"].stats()[0], t1.stats()[0], t2.stats()[0]
    assert max(n1, n2) <= n <= n1 + n2
    t1.close(); t2.close()


def test_cut_inside_the_counting_pass(full, gpu_ctx):
    above, n_all = gpu_ctx.count_device_above(full["bases"].data_ptr(), full["offsets"].data_ptr(), N_READS, N_READS * RL, K, 1)
    assert n_all == full["table"].stats()[0]
    flt = full["table"].filter(1)
    assert len(above) == len(flt)
    got = _lookup(above, full["sample"])
    ref = _lookup(full["table"], full["sample"])
    assert np.array_equal(got, np.where(ref > 1, ref, -1))
    flt.close(); above.close()


def _kmers_of(seq, k=K):
    code = np.zeros(256, dtype=np.uint64)
    for ch, c in (("A", 0), ("G", 1), ("C", 2), ("T", 3)):
        code[ord(ch)] = c
    c = code[np.frombuffer(seq.encode(), dtype=np.uint8)]
    m = len(c) - k + 1
    fw = np.zeros(m, dtype=np.uint64)
    rc = np.zeros(m, dtype=np.uint64)
    for i in range(k):
        fw = (fw << np.uint64(2)) | c[i:i + m]
        rc |= (np.uint64(3) - c[i:i + m]) << np.uint64(2 * i)
    return np.minimum(fw, rc)


def _neighbours(x, k=K):
    mask = np.uint64((1 << (2 * k)) - 1)
    out = []
    for n in range(4):
        for y in (((x << np.uint64(2)) | np.uint64(n)) & mask, (x >> np.uint64(2)) | (np.uint64(n) << np.uint64(2 * k - 2))):
            r = np.zeros_like(y)
            t = y.copy()
            for _ in range(k):
                r = (r << np.uint64(2)) | (np.uint64(3) - (t & np.uint64(3)))
                t >>= np.uint64(2)
            out.append(np.minimum(y, r))
    return np.stack(out, axis=1)


def check_pipeline_properties(gpu_ctx, bases, offsets, n_reads, k, n_distinct=None, b=1, l=100, b1=1000, b2=10000):
    """the size-independent properties of unitigs, components and features of ONE sample (also used at 200 M reads, k = 21:
    tests/test_shapes_gpu.py)"""
    from metafast_amd import lib as L
    from metafast_amd import pipeline as P
    r = P.run_sample(gpu_ctx, bases, offsets, n_reads, n_reads * RL, k=k, b=b, l=l, b1=b1, b2=b2)
    good, seqs, comps, cutter = r["good"], r["seqs"], r["comps"], r["cutter"]
    if n_distinct is not None:
        assert r["n_distinct"] == n_distinct
    # unitigs: a sample of them, k-mer by k-mer
    sq = seqs.export()
    assert len(sq) > 0 and all(len(s[0]) >= l for s in sq[:10000])
    rng = np.random.default_rng(3)
    for i in rng.choice(len(sq), size=min(300, len(sq)), replace=False):
        s, avg, mn, mx = sq[int(i)]
        cnt = good.lookup(_kmers_of(s, k)).astype(np.int64)
        assert cnt.min() > b and (mn, mx) == (cnt.min(), cnt.max()) and avg == int(cnt.sum() // len(cnt))
    # components
    cs = comps.export()
    assert len(cs) > 0
    sizes = np.array([c[0] for c in cs])
    assert sizes.min() >= b1 and sizes.max() <= b2 and all(len(c[3]) == c[0] for c in cs)
    allk = np.concatenate([c[3] for c in cs])
    owner = np.repeat(np.arange(len(cs)), sizes)
    order = np.argsort(allk, kind="stable")
    allk, owner = allk[order], owner[order]
    assert np.all(allk[1:] != allk[:-1])                                   # pairwise disjoint
    for ci in rng.choice(len(cs), size=min(40, len(cs)), replace=False):
        size, weight, thr, km = cs[int(ci)]
        val = cutter.lookup(km).astype(np.int64)
        assert val.min() >= thr and weight == val.sum()
        nb = _neighbours(km[:200], k)                                       # closed: a neighbour in the graph is in the same component
        present = cutter.lookup(nb.reshape(-1)).astype(np.int64).reshape(nb.shape) >= thr
        pos = np.searchsorted(allk, nb).clip(max=len(allk) - 1)
        same = (allk[pos] == nb) & (owner[pos] == ci)
        assert np.array_equal(present, same)
        # features: the sample's counts (> b only) over the component's k-mers
        sv = good.lookup(km).astype(np.int64)
        assert r["vec"][int(ci)] == sv[sv > 0].sum()
    m = L.bray_curtis(np.stack([r["vec"], r["vec"]]))
    assert m[0, 1] == 0.0 and m[1, 0] == 0.0
    return dict(n_unitigs=len(sq), n_components=len(cs), max_thr=max(c[2] for c in cs))


def test_pipeline_properties(full, gpu_ctx):
    check_pipeline_properties(gpu_ctx, full["bases"], full["offsets"], N_READS, K, n_distinct=full["table"].stats()[0])

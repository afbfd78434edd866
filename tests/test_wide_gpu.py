"""NO-REFERENCE EXTENSION: canonical k-mer counts for 32 <= k <= 63 (metafast_amd/csrc/mf_wide.hip).  The reference rejects
k > 31 (src/tools/KmersCounterMain.java:66-73), so this is no parity test against the reference: the checker is the oracle's
own 128-bit restatement of the k <= 31 definitions (oracle/mf_oracle.c: or_count_wide), itself checked here against the
reference-pinned 64-bit oracle through an identity at k = 31 -> 32 (a 32-mer's two 31-mers)."""
import numpy as np
import pytest


@pytest.fixture(autouse=True)
def _record_path_back_on(request):
    """tests of the sort path switch the record path (round 6, option wide_skm) off: back on behind every test"""
    yield
    if "gpu_ctx" in request.fixturenames:
        request.getfixturevalue("gpu_ctx").set_option("wide_skm", 1)


def _reads(rng, n, lo, hi, genome=4000, err=0.01, polya=80):
    g = rng.integers(0, 4, genome)
    g[100:100 + polya] = 0                                            # poly-A: key 0, saturation
    g[600:640] = np.tile([0, 3], 20)                          # (AT)n: reverse-complement palindromes at even k
    al = np.frombuffer(b"AGCT", dtype=np.uint8)
    seqs = []
    for _ in range(n):
        L = int(rng.integers(lo, hi + 1))
        s = int(rng.integers(0, genome - L))
        r = g[s:s + L].copy()
        flip = rng.random(L) < err
        r[flip] = rng.integers(0, 4, int(flip.sum()))
        if rng.random() < 0.5:
            r = 3 - r[::-1]
        seqs.append(al[r])
    bases = np.concatenate(seqs)
    off = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.uint64)
    return bases, off


@pytest.mark.gpu
@pytest.mark.parametrize("k", [32, 33, 40, 47, 62, 63])
def test_wide_counts_match_the_128bit_oracle(gpu_ctx, oracle, k):
    from util import to_device
    rng = np.random.default_rng(k)
    bases, off = _reads(rng, 40000, 20, 160)                  # ragged: many reads shorter than k
    db, do = to_device(bases, off)
    for min_len in (0, 100):
        got = gpu_ctx.count_wide_device(db.data_ptr(), do.data_ptr(), len(off) - 1, len(bases), k, min_len)
        hi, lo, cnt, n_occ = oracle.count_wide(bases, off, k, min_len)
        assert got["n_occ"] == n_occ > 0 and got["k"] == k
        assert np.array_equal(got["hi"], hi) and np.array_equal(got["lo"], lo)
        assert np.array_equal(got["counts"].astype(np.int32), cnt)
        assert cnt.max() > 1000
    # saturation at 32767 and key 0 (poly-A), the all-T reads fold onto it
    pa = np.concatenate([np.full(700 * 120, ord("A"), dtype=np.uint8), np.full(300 * 120, ord("T"), dtype=np.uint8)])
    po = (np.arange(1001) * 120).astype(np.uint64)
    db, do = to_device(pa, po)
    got = gpu_ctx.count_wide_device(db.data_ptr(), do.data_ptr(), 1000, len(pa), k, 0)
    hi, lo, cnt, n_occ = oracle.count_wide(pa, po, k, 0)
    assert got["n_occ"] == n_occ == 1000 * (120 - k + 1) and len(cnt) == 1 and cnt[0] == 32767
    assert got["hi"].tolist() == [0] and got["lo"].tolist() == [0] and got["counts"].tolist() == [32767]


@pytest.mark.gpu
@pytest.mark.parametrize("k,passes", [(32, 4), (33, 16), (47, 64), (63, 8)])
def test_wide_counts_in_passes(gpu_ctx, oracle, k, passes):
    """a 200 M-read sample has 1.8e10 63-mers, more than one sort takes (2^32 entries, 32 bytes each twice over): the
    reads are counted in passes, one per prefix class of the canonical k-mers, and the ascending passes append to the same table.
    Forced pass counts (option wide_passes) on ragged reads: identical to the one-pass table and to the 128-bit oracle."""
    gpu_ctx.set_option("wide_skm", 0)          # (the sort path of mf_wide.hip: the record path of round 6 -- mf_wskm.hip -- has tests of its own below)

    from util import to_device
    rng = np.random.default_rng(100 + k)
    bases, off = _reads(rng, 30000, 20, 160)
    db, do = to_device(bases, off)
    hi, lo, cnt, n_occ = oracle.count_wide(bases, off, k, 0)
    try:
        gpu_ctx.set_option("wide_passes", passes)
        got = gpu_ctx.count_wide_device(db.data_ptr(), do.data_ptr(), len(off) - 1, len(bases), k, 0)
    finally:
        gpu_ctx.set_option("wide_passes", 0)
    assert got["n_occ"] == n_occ
    assert np.array_equal(got["hi"], hi) and np.array_equal(got["lo"], lo) and np.array_equal(got["counts"].astype(np.int32), cnt)


@pytest.mark.gpu
def test_wide_counts_at_20M_reads_k63(gpu_ctx):
    """k = 63 beyond toy size (no oracle at this size: properties): 20 M synthetic reads = 1.76e9 63-mers, counted in the automatic
    number of passes and in 32 forced ones -- the same table; occurrences = reads x 88; ascending distinct keys; the counts add up
    (no key saturates at this depth except through the generator's repeats: sum(min(c, 32767)) <= occurrences, = where none does)"""
    import torch
    n, rl, k = 20_000_000, 150, 63
    bases = torch.zeros(n * rl + 64, dtype=torch.uint8, device="cuda")
    offsets = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    gpu_ctx.synth_reads_device(0x4D45544146415354, 0, 0, n, rl, 1_000_000, bases.data_ptr(), offsets.data_ptr())
    torch.cuda.synchronize()
    a = gpu_ctx.count_wide_device(bases.data_ptr(), offsets.data_ptr(), n, n * rl, k, 0)
    assert a["n_occ"] == n * (rl - k + 1)
    hi, lo, c = a["hi"], a["lo"], a["counts"].astype(np.int64)
    assert ((hi[1:] > hi[:-1]) | ((hi[1:] == hi[:-1]) & (lo[1:] > lo[:-1]))).all()          # strictly ascending 126-bit keys
    assert (hi < (1 << 62)).all() and int(c.min()) >= 1
    sat = int((c == 32767).sum())
    assert int(c.sum()) <= a["n_occ"] and (sat > 0 or int(c.sum()) == a["n_occ"])
    try:
        gpu_ctx.set_option("wide_passes", 32)
        b = gpu_ctx.count_wide_device(bases.data_ptr(), offsets.data_ptr(), n, n * rl, k, 0)
    finally:
        gpu_ctx.set_option("wide_passes", 0)
    assert np.array_equal(b["hi"], hi) and np.array_equal(b["lo"], lo) and np.array_equal(b["counts"].astype(np.int64), c)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [32, 41, 47, 48, 63])
def test_wide_buckets_small_and_large(gpu_ctx, oracle, k):
    """Round 5: radix passes over the leading 32 bits only, the order inside those buckets made in LDS (k_wide_finish): buckets of <= 256 entries by
    walking them, larger ones through a hash table in LDS (distinct k-mers + counts, written back as runs); a bucket of more than 1280 distinct
    k-mers is gathered, sorted with the full radix sort and put back (or, when such buckets hold more than a quarter of a pass, the whole pass is).
    A genome at depth ~10 with a long poly-A stretch: mostly small buckets + a few large ones; the limits lowered by option drive every route;
    wide_finish = 0 is the 16-pass sort of round 4.  One table, the oracle's."""
    gpu_ctx.set_option("wide_skm", 0)          # (the sort path of mf_wide.hip: the record path of round 6 -- mf_wskm.hip -- has tests of its own below)

    from util import to_device
    rng = np.random.default_rng(500 + k)
    bases, off = _reads(rng, 30000, 60, 160, genome=300000, err=0.005, polya=400)
    db, do = to_device(bases, off)
    hi, lo, cnt, n_occ = oracle.count_wide(bases, off, k, 0)
    assert 2 < cnt.mean() < 50 and cnt.max() > 256
    seen = []
    for opts in ({}, {"wide_distinct": 4}, {"wide_big_bucket": 2}, {"wide_big_bucket": 2, "wide_distinct": 1}, {"wide_big_bucket": 1, "wide_distinct": 8, "wide_passes": 3},
                 {"wide_finish": 0}, {"wide_passes": 5}):
        before = gpu_ctx.stat("wide_big_entries"), gpu_ctx.stat("wide_hashed_entries")
        try:
            for o, v in opts.items():
                gpu_ctx.set_option(o, v)
            got = gpu_ctx.count_wide_device(db.data_ptr(), do.data_ptr(), len(off) - 1, len(bases), k, 0)
        finally:
            for o, v in (("wide_big_bucket", 256), ("wide_distinct", 1280), ("wide_finish", 1), ("wide_passes", 0)):
                gpu_ctx.set_option(o, v)
        assert got["n_occ"] == n_occ
        assert np.array_equal(got["hi"], hi) and np.array_equal(got["lo"], lo) and np.array_equal(got["counts"].astype(np.int32), cnt), opts
        seen.append((gpu_ctx.stat("wide_big_entries") - before[0], gpu_ctx.stat("wide_hashed_entries") - before[1]))
    assert seen[0][0] == 0 and 256 < seen[0][1] < n_occ // 4, seen          # the poly-A bucket(s) through the hash table, nothing sorted aside
    assert 0 < seen[1][0] < n_occ // 4, seen                               # ... sorted aside
    assert seen[2][0] == 0 and seen[2][1] > n_occ // 2, seen               # nearly every bucket through the hash table
    assert seen[3][0] > n_occ // 4, seen                                   # nearly every bucket refused by the table: the whole-pass sort
    assert seen[5] == (0, 0), seen


@pytest.mark.gpu
def test_wide_counts_config4_sample_k63(gpu_ctx):
    """BASELINE config 4's k = 63 leg at the size it names (one sample of 200 M reads; VERDICT r4 configs_untested: "tested at <= 20 M reads"):
    1.76e10 63-mers, 4.1e9 distinct -- no oracle and no host copy at this size (74 GB of table): the properties are checked where the table
    lies, piece by piece: occurrences = reads x 88; strictly ascending 126-bit keys inside every piece and across the seams; counts >= 1 that add
    up to the occurrences (where none saturates); the same number of distinct k-mers from the round-4 order (all 2k bits through the radix sort)
    on the first 20 M reads."""
    import torch
    from metafast_amd.pipeline import device_tensor
    gpu_ctx.trim(); torch.cuda.empty_cache()                                     # (what earlier tests left idle in the arena and in torch's cache)
    free, _ = torch.cuda.mem_get_info()
    if free < 250e9:
        pytest.skip("needs a whole MI355X")
    n, rl, k = 200_000_000, 150, 63
    bases = torch.zeros(n * rl + 64, dtype=torch.uint8, device="cuda")
    offsets = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    gpu_ctx.synth_reads_device(0x4D45544146415354, 0, 0, n, rl, 1_000_000, bases.data_ptr(), offsets.data_ptr())
    torch.cuda.synchronize()

    def check(t, reads):
        nd, occ, kk = t.stats()
        assert kk == k and occ == reads * (rl - k + 1)
        gpu_ctx.trim()                                                            # (the pass buffers go back to the driver: torch needs room for its temporaries)
        total, seen, last, sat = 0, 0, None, 0
        step = 1 << 27
        for d_hi, d_lo, d_cnt, m in t.pieces():
            assert m > 0
            hi_all = device_tensor(d_hi, m * 8, "cuda").view(torch.int64); lo_all = device_tensor(d_lo, m * 8, "cuda").view(torch.int64)
            cnt_all = device_tensor(d_cnt, m * 2, "cuda").view(torch.int16)
            for c0 in range(0, m, step):                                          # (a chunk and its left neighbour at a time)
                a0 = max(c0 - 1, 0)
                hi = hi_all[a0:c0 + step]; lo_u = lo_all[a0:c0 + step] ^ torch.iinfo(torch.int64).min      # unsigned order of the low word through a signed compare
                assert int(hi.min()) >= 0 and int(hi.max()) < (1 << 62)
                if hi.numel() > 1:
                    assert bool(((hi[1:] > hi[:-1]) | ((hi[1:] == hi[:-1]) & (lo_u[1:] > lo_u[:-1]))).all())
                cnt = cnt_all[c0:c0 + step]
                assert int(cnt.min()) >= 1
                total += int(cnt.to(torch.int64).sum()); sat += int((cnt == 32767).sum())
                del hi, lo_u, cnt
            first = (int(hi_all[0]), int(lo_all[0]) ^ (-1 << 63))
            assert last is None or first > last
            last = (int(hi_all[-1]), int(lo_all[-1]) ^ (-1 << 63))
            seen += m
        assert seen == nd and total <= occ and (sat > 0 or total == occ)
        return nd

    t = gpu_ctx.count_wide_table(bases.data_ptr(), offsets.data_ptr(), n, n * rl, k, 0)
    try:
        nd = check(t, n)
        assert len(t.pieces()) >= 2 and nd > 3_000_000_000
    finally:
        t.close()
    m = 20_000_000
    a = gpu_ctx.count_wide_table(bases.data_ptr(), offsets.data_ptr(), m, m * rl, k, 0)
    try:
        nd_a = check(a, m)
    finally:
        a.close()
    try:
        gpu_ctx.set_option("wide_finish", 0)
        b = gpu_ctx.count_wide_table(bases.data_ptr(), offsets.data_ptr(), m, m * rl, k, 0)
        try:
            assert check(b, m) == nd_a
        finally:
            b.close()
    finally:
        gpu_ctx.set_option("wide_finish", 1)


def test_wide_oracle_agrees_with_the_pinned_oracle_at_the_seam(oracle):
    """the 128-bit restatement is tied to the reference-pinned 64-bit oracle: every 32-mer occurrence contributes its first
    31-mer, so per read (len >= 32) the multiset of canonical 31-mers of starts 0 .. len-32 is determined by the 32-mers"""
    rng = np.random.default_rng(5)
    bases, off = _reads(rng, 300, 32, 90, genome=2000, err=0.0)
    hi, lo, cnt, n_occ = oracle.count_wide(bases, off, 32, 0)
    assert (hi == 0).all() and n_occ == int(sum(int(off[i + 1] - off[i]) - 31 for i in range(len(off) - 1)))
    # expand the 32-mer table into the canonical 31-mers of (prefix of fw) -- needs the orientation, so recount per occurrence
    code = np.zeros(256, dtype=np.uint64)
    for ch, c in zip(b"AGCT", range(4)):
        code[ch] = c
    want = {}
    for i in range(len(off) - 1):
        c = code[bases[int(off[i]):int(off[i + 1])]]
        for s in range(len(c) - 31):
            fw = 0
            for b in c[s:s + 32]:
                fw = (fw << 2) | int(b)
            rc = 0
            for b in c[s:s + 32][::-1]:
                rc = (rc << 2) | (3 - int(b))
            x = min(fw, rc)
            want[x] = want.get(x, 0) + 1
    keys = np.array(sorted(want), dtype=np.uint64)
    assert np.array_equal(lo, keys) and np.array_equal(cnt, np.array([min(want[int(x)], 32767) for x in keys], dtype=np.int32))
    # ... and the plain-Python recount's 31-mer half against the pinned oracle's table
    t = oracle.Table().count_buffer(bases, off, 31)
    k31 = {}
    for i in range(len(off) - 1):
        c = code[bases[int(off[i]):int(off[i + 1])]]
        for s in range(len(c) - 30):
            fw = 0
            for b in c[s:s + 31]:
                fw = (fw << 2) | int(b)
            rc = 0
            for b in c[s:s + 31][::-1]:
                rc = (rc << 2) | (3 - int(b))
            x = min(fw, rc)
            k31[x] = k31.get(x, 0) + 1
    ok, ov = t.export()
    assert np.array_equal(ok, np.array(sorted(k31), dtype=np.uint64)) and np.array_equal(ov, np.array([k31[int(x)] for x in ok]))


@pytest.mark.gpu
@pytest.mark.parametrize("k", [32, 33, 40, 47, 62, 63])
def test_wide_record_path_equals_sort_path(gpu_ctx, oracle, k):
    """round 6: the count on the RECORD path (mf_wskm.hip: super-k-mer records of 32 bytes, LDS tables with two words per key) against the
    sort path (mf_wide.hip) and the 128-bit oracle -- ragged reads with a poly-A stretch (key 0, saturation) and (AT)n palindromes, with and
    without the cut, units small enough that some overflow the LDS table and are counted in passes, reads cut at tile borders"""
    from util import to_device
    rng = np.random.default_rng(700 + k)
    bases, off = _reads(rng, 60000, 20, 220)
    db, do = to_device(bases, off)
    n = len(off) - 1
    hi, lo, cnt, n_occ = oracle.count_wide(bases, off, k, 0)
    try:
        for unit in (4000, 300, 64):
            gpu_ctx.set_option("wide_skm", 1); gpu_ctx.set_option("wide_skm_min", 1); gpu_ctx.set_option("wide_skm_unit", unit)
            got = gpu_ctx.count_wide_device(db.data_ptr(), do.data_ptr(), n, len(bases), k, 0)
            assert got["n_occ"] == n_occ
            assert np.array_equal(got["hi"], hi) and np.array_equal(got["lo"], lo) and np.array_equal(got["counts"].astype(np.int32), cnt), unit
        gpu_ctx.set_option("wide_skm_unit", 2400)
        for thr in (1, 3):
            t, n_all = gpu_ctx.count_wide_above(db.data_ptr(), do.data_ptr(), n, len(bases), k, thr)
            ghi, glo, gc = t.export()
            keep = cnt > thr
            assert n_all == len(cnt) and np.array_equal(ghi, hi[keep]) and np.array_equal(glo, lo[keep]) and np.array_equal(gc.astype(np.int32), cnt[keep])
        # the kept entries' order: by their leading 32 bits + a look at the runs of equal ones (the default), by all bits (wide_skm_lead = 0), and
        # -- 300 k-mers that share their first 20 bases: a run too long for the look -- by all bits after all
        gpu_ctx.set_option("wide_skm_lead", 0)
        got = gpu_ctx.count_wide_device(db.data_ptr(), do.data_ptr(), n, len(bases), k, 0)
        assert np.array_equal(got["hi"], hi) and np.array_equal(got["lo"], lo) and np.array_equal(got["counts"].astype(np.int32), cnt)
        gpu_ctx.set_option("wide_skm_lead", 1)
        pre = np.frombuffer(b"AAAAAAAAAACCCCCCCCCC", dtype=np.uint8)
        tails = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (300, k - 20))]
        b2 = np.concatenate([np.concatenate([pre, t]) for t in tails] + [bases]); o2 = np.concatenate([np.arange(300, dtype=np.uint64) * np.uint64(k), off + np.uint64(300 * k)])
        db2, do2 = to_device(b2, o2)
        h2, l2, c2, n2 = oracle.count_wide(b2, o2, k, 0)
        for lead in (1, 2):                         # (1: the run goes aside; 2: no room aside -- all bits of everything)
            gpu_ctx.set_option("wide_skm_lead", lead)
            got = gpu_ctx.count_wide_device(db2.data_ptr(), do2.data_ptr(), len(o2) - 1, len(b2), k, 0)
            assert got["n_occ"] == n2 and np.array_equal(got["hi"], h2) and np.array_equal(got["lo"], l2) and np.array_equal(got["counts"].astype(np.int32), c2), lead
        gpu_ctx.set_option("wide_skm_lead", 1)
        gpu_ctx.set_option("wide_skm", 0)
        old = gpu_ctx.count_wide_device(db.data_ptr(), do.data_ptr(), n, len(bases), k, 0)
        assert np.array_equal(old["hi"], hi) and np.array_equal(old["counts"].astype(np.int32), cnt)
        # min_read_len, and nothing at all
        gpu_ctx.set_option("wide_skm", 2)          # (2: the record path for inputs filtered by length too -- by default those are taken for assembled sequences)
        got = gpu_ctx.count_wide_device(db.data_ptr(), do.data_ptr(), n, len(bases), k, 150)
        h2, l2, c2, o2 = oracle.count_wide(bases, off, k, 150)
        assert got["n_occ"] == o2 and np.array_equal(got["hi"], h2) and np.array_equal(got["lo"], l2) and np.array_equal(got["counts"].astype(np.int32), c2)
    finally:
        gpu_ctx.set_option("wide_skm", 1); gpu_ctx.set_option("wide_skm_min", 1 << 20); gpu_ctx.set_option("wide_skm_unit", 2400); gpu_ctx.set_option("wide_skm_lead", 1)


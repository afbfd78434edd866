"""Round 6: the STREAMED count of read files (metafast_amd/csrc/mf_stream.hip) -- upload || parse || level-1 scatter, piece by piece --
against the count of the same files loaded whole (stream_count = 0) and against the oracle (src/io/IOUtils.java:756-803).  The limits are
lowered so that files of tens of MB go that way in 8 MB pieces."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _reads(rng, n, rl=150, genome=300_000, sub=0.004, n_rate=0.001):
    g = rng.integers(0, 4, genome, dtype=np.uint8)
    pos = rng.integers(0, genome - rl, n)
    idx = pos[:, None] + np.arange(rl)[None, :]
    r = g[idx]
    flip = rng.random((n, rl)) < sub
    r = np.where(flip, (r + rng.integers(1, 4, (n, rl), dtype=np.uint8)) & 3, r).astype(np.uint8)
    rc = rng.random(n) < 0.5
    r[rc] = (3 - r[rc])[:, ::-1]
    a = np.frombuffer(b"ACGT", dtype=np.uint8)[r]
    a[rng.random((n, rl)) < n_rate] = ord("N")
    return a


def _write_fasta(path, a, wrap=0):
    n, rl = a.shape
    head = np.frombuffer(b">r\n", dtype=np.uint8)
    rows = np.empty((n, 3 + rl + 1), dtype=np.uint8)
    rows[:, :3] = head; rows[:, 3:3 + rl] = a; rows[:, -1] = 10
    if wrap:
        rows = np.concatenate([rows[:, :3 + wrap], np.full((n, 1), 10, np.uint8), rows[:, 3 + wrap:]], axis=1)
    rows.tofile(path)


def _write_fastq(path, a, rng, qoff=33):
    n, rl = a.shape
    q = rng.integers(qoff + 2, qoff + 41, (n, rl)).astype(np.uint8)
    q[rng.random((n, rl)) < 0.0005] = qoff                       # phred 0: the base counts as N (FastaReaderFromXQSource.java:66-70)
    if qoff == 33:
        q[::3, 0] = ord("@"); q[1::7, 0] = ord("+")              # quality lines that look like header lines
    rows = np.empty((n, 3 + rl + 3 + rl + 1), dtype=np.uint8)
    rows[:, :3] = np.frombuffer(b"@r\n", dtype=np.uint8); rows[:, 3:3 + rl] = a
    rows[:, 3 + rl:6 + rl] = np.frombuffer(b"\n+\n", dtype=np.uint8); rows[:, 6 + rl:6 + 2 * rl] = q; rows[:, -1] = 10
    rows.tofile(path)


def _sorted_table(t):
    k, c = t.export()
    o = np.argsort(k, kind="stable")
    return k[o], c[o]


@pytest.fixture
def stream_ctx(gpu_ctx):
    gpu_ctx.set_option("stream_count_min_bytes", 1); gpu_ctx.set_option("stream_count_piece_bytes", 8 << 20)
    yield gpu_ctx
    for name, v in (("stream_count", 1), ("stream_count_min_bytes", 512 << 20), ("stream_count_piece_bytes", 256 << 20), ("stream_count_test_pct", 100), ("skm_slices", 0)):
        gpu_ctx.set_option(name, v)


def _both_ways(ctx, files, k, thr, expect_streamed=True):
    ctx.set_option("stream_count", 0)
    t0, all0 = ctx.count_reads_above(files, k, thr)
    ctx.set_option("stream_count", 1)
    before = ctx.stat("streamed_counts"), ctx.stat("streamed_counts_stepped_back")
    t1, all1 = ctx.count_reads_above(files, k, thr)
    after = ctx.stat("streamed_counts"), ctx.stat("streamed_counts_stepped_back")
    if expect_streamed:
        assert after == (before[0] + 1, before[1]), (before, after)
    else:
        assert after[0] == before[0], (before, after)
    k0, c0 = _sorted_table(t0); k1, c1 = _sorted_table(t1)
    assert all0 == all1 and t0.occurrences() == t1.occurrences()
    assert np.array_equal(k0, k1) and np.array_equal(c0, c1)
    assert np.array_equal(t0.hist(), t1.hist())
    t0.close()
    return t1, after


def test_streamed_count_equals_whole_file_count(stream_ctx, oracle, tmp_path):
    ctx = stream_ctx
    rng = np.random.default_rng(61)
    a = _reads(rng, 420_000)
    fa = str(tmp_path / "a.fa"); _write_fasta(fa, a)
    fw = str(tmp_path / "w.fasta"); _write_fasta(fw, a[:300_000], wrap=70)
    fq = str(tmp_path / "a.fq"); _write_fastq(fq, a, rng)
    fq64 = str(tmp_path / "b.fastq"); _write_fastq(fq64, a[20_000:], rng, qoff=64)
    for files, k, thr in (([fa], 31, 1), ([fa], 25, 0), ([fw], 21, 3), ([fq], 31, 1), ([fq64], 27, 0), ([fa, fw], 31, 1), ([fq, fq64], 23, 2), ([fa], 31, -1), ([fa], 31, 40)):
        t, _ = _both_ways(ctx, files, k, thr)
        t.close()
    # ... and against the oracle (reads of the file through the oracle's reader, IOUtils.loadReads, the cut of printKmers)
    ok, oc = oracle.Table().count_files([fq], 31).export(1)
    t, _ = ctx.count_reads_above([fq], 31, 1)
    k1, c1 = _sorted_table(t)
    assert np.array_equal(k1, ok) and np.array_equal(c1.astype(np.int64), oc.astype(np.int64))
    t.close()


def test_streamed_count_in_slices(stream_ctx, tmp_path):
    """the count behind a streamed level 1 in several slices (the level-1 buffer is borrowed by every slice)"""
    ctx = stream_ctx
    rng = np.random.default_rng(62)
    fa = str(tmp_path / "a.fa"); _write_fasta(fa, _reads(rng, 300_000))
    ctx.set_option("skm_slices", 4)
    t, _ = _both_ways(ctx, [fa], 31, 1)
    t.close()


def test_streamed_count_steps_back(stream_ctx, oracle, tmp_path):
    """what the streamed count is not sure about is done again from the whole files: regions the sample sized too small; a record the device parser
    does not take in a late piece (the host reader's reads, or the reference's error message); sequences longer than the border search; other formats"""
    from metafast_amd import lib as L
    ctx = stream_ctx
    rng = np.random.default_rng(63)
    a = _reads(rng, 300_000)
    fa = str(tmp_path / "a.fa"); _write_fasta(fa, a)
    ctx.set_option("stream_count_test_pct", 40)
    t, st = _both_ways(ctx, [fa], 31, 1, expect_streamed=False)
    t.close()
    sb = ctx.stat("streamed_counts_stepped_back")
    assert sb >= 1
    ctx.set_option("stream_count_test_pct", 100)
    # a lone CR in the last piece: the device parser steps back, the host reader takes the file
    text = open(fa, "rb").read()
    at = text.rindex(b"\n>r\n", 0, len(text) - 1000)
    while b"N" in text[at + 4:at + 4 + 150]:                    # (a read with an N is dropped before its other characters are looked at)
        at = text.rindex(b"\n>r\n", 0, at)
    odd = str(tmp_path / "odd.fa"); open(odd, "wb").write(text[:at] + b"\r>r\n" + text[at + 4:])
    t, _ = _both_ways(ctx, [odd], 31, 1, expect_streamed=False)
    t.close()
    assert ctx.stat("streamed_counts_stepped_back") == sb + 1
    # a character no reader takes: the reference's message, whichever way the count started
    bad = str(tmp_path / "bad.fa"); open(bad, "wb").write(text[:at + 4] + b"J" + text[at + 5:])
    msgs = []
    for sc in (0, 1):
        ctx.set_option("stream_count", sc)
        with pytest.raises(L.MetafastError) as e:
            ctx.count_reads_above([bad], 31, 1)
        msgs.append(str(e.value))
    assert msgs[0] == msgs[1], msgs
    # one long sequence per megabyte: no record border in the search window
    lng = str(tmp_path / "long.fa")
    with open(lng, "wb") as f:
        for i in range(12):
            f.write(b">contig\n" + np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 3_000_000)].tobytes() + b"\n")
    t, _ = _both_ways(ctx, [lng], 31, 0, expect_streamed=False)
    t.close()
    # a compressed file beside a plain one: not this way
    import gzip
    gz = str(tmp_path / "c.fa.gz"); open(gz, "wb").write(gzip.compress(text[: 8 << 20], 1))
    t, _ = _both_ways(ctx, [fa, gz], 31, 1, expect_streamed=False)
    t.close()


def test_streamed_count_many_pieces_reuses_the_ring(stream_ctx, tmp_path):
    """more pieces than ring slots, FASTQ, pieces of the smallest size; the sample's text shares the first slot"""
    ctx = stream_ctx
    rng = np.random.default_rng(64)
    a = _reads(rng, 500_000, genome=2_000_000)
    fq = str(tmp_path / "m.fq"); _write_fastq(fq, a, rng)                  # 155 MB: ~19 pieces
    t, _ = _both_ways(ctx, [fq], 31, 0)
    assert t.occurrences() > 0
    t.close()


def test_cli_streams_its_libraries(gpu_ctx, tmp_path):
    """metafast.sh with the streamed count on (limits lowered through MF_OPTIONS, two contexts on the GPU: each streams its library in its turn):
    the workDir's result files are the files of a run with stream_count = 0, byte for byte"""
    import subprocess
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from metafast_amd import lib as L
    files = []
    for s in range(2):
        n, rl = 120_000, 100
        bases, _ = L.synth_reads_host(0x4D45544146415354, s, 0, n, rl, 40000)          # (SURVEY 8(d)'s generator: shared and private pool pieces -> components)
        p = str(tmp_path / ("lib%d.fa" % s)); _write_fasta(p, np.frombuffer(bases, dtype=np.uint8)[: n * rl].reshape(n, rl)); files.append(p)

    def run(wd, opts):
        env = dict(os.environ, MF_OPTIONS=opts)
        cmd = [os.path.join(ROOT, "metafast.sh"), "-k", "21", "-b", "1", "-l", "60", "-b1", "40", "-b2", "2000", "-i", *files, "-w", str(wd), "--devices", "0,0", "-v"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=tmp_path, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        out = {}
        for sub in ("kmer-counter-many/kmers", "kmer-counter-many/stats", "seq-builder-many/sequences", "component-cutter", "features-calculator/vectors"):
            for q in sorted((wd / sub).iterdir()):
                if q.is_file() and q.name not in ("in.properties", "out.properties", "SUCCESS"):
                    out[sub + "/" + q.name] = q.read_bytes()
        out["matrix"] = sorted((wd / "matrices").glob("dist_matrix_*_original_order.txt"))[-1].read_bytes()
        return out, r.stderr

    whole, _ = run(tmp_path / "wd0", "stream_count=0")
    streamed, err = run(tmp_path / "wd1", "stream_count_min_bytes=1,stream_count_piece_bytes=2097152,part_target=256,verbose=1")
    assert err.count("streamed count (1 file(s)") == 2 and "stepped back" not in err, err[-3000:]
    assert whole.keys() == streamed.keys()
    for name in whole:
        assert whole[name] == streamed[name], name


def test_streamed_count_with_the_defaults_16M_reads(gpu_ctx, tmp_path):
    """no option lowered: 16 M reads of the benchmark sample as a 2.46 GB FASTA file in 256 MB pieces (the bench's end_to_end line) -- the table of the
    reads counted where they lie in HBM, entry for entry; 0 slice restarts, nothing stepped back"""
    import sys, torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    ctx = gpu_ctx
    n, rl, k = 16_000_000, 150, 31
    free, _ = torch.cuda.mem_get_info()
    if free < 40e9:
        pytest.skip("needs 40 GB of free HBM")
    bases = torch.zeros(n * rl + 64, dtype=torch.uint8, device="cuda"); offs = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    ctx.synth_reads_device(bench.SEED, 0, 0, n, rl, 1_000_000, bases.data_ptr(), offs.data_ptr(), 82)
    torch.cuda.synchronize()
    fa = str(tmp_path / "s.fa")
    bench._write_fasta(bases, n, rl, fa)
    t0, all0 = ctx.count_device_above(bases.data_ptr(), offs.data_ptr(), n, n * rl, k, 1) if hasattr(ctx, "count_device_above") else (None, None)
    before = ctx.stat("streamed_counts"), ctx.stat("streamed_counts_stepped_back"), ctx.stat("slice_restarts")
    t1, all1 = ctx.count_reads_above([fa], k, 1)
    assert (ctx.stat("streamed_counts"), ctx.stat("streamed_counts_stepped_back"), ctx.stat("slice_restarts")) == (before[0] + 1, before[1], before[2])
    if t0 is None:
        t = ctx.count_device(bases.data_ptr(), offs.data_ptr(), n, n * rl, k, 0)
        t0 = t.filter(1); all0 = len(t); t.close()
    assert all0 == all1 and len(t0) == len(t1) and t0.occurrences() == t1.occurrences() == n * (rl - k + 1)
    k0, c0 = _sorted_table(t0); k1, c1 = _sorted_table(t1)
    assert np.array_equal(k0, k1) and np.array_equal(c0, c1)
    t0.close(); t1.close()
    os.remove(fa)

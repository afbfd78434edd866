"""Round 5: the drop-in driver on several device contexts (metafast.sh --devices a,b,...: one library per context in the three
per-library steps, KmersCounterForManyFilesMain.java:80-108 / SeqBuilderForManyFilesMain.java:82-94 / FeaturesCalculatorMain.java:
137-162), features-calculator --selected (FeaturesCalculatorMain.java:55-57, 113-116, 193-203), the k-specialised neighbour kernels
and the CAMI example's parameters (Example.md:18-21: -k 23 -b 5 -l 1200)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
SEED = 0x4D45544146415354


@pytest.fixture(scope="module", autouse=True)
def _cli_built():
    if not os.path.exists(os.path.join(ROOT, "metafast_amd", "cli", "metafast")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "metafast_amd", "cli")])


def _write_fasta(path, sample, n, read_len=100, scale=3000, first=0):
    from metafast_amd import lib as L
    bases, _ = L.synth_reads_host(SEED, sample, first, n, read_len, scale)
    with open(path, "w") as f:
        for i, row in enumerate(np.frombuffer(bases, dtype=np.uint8).reshape(n, read_len)):
            f.write(">r%d\n%s\n" % (i, row.tobytes().decode()))


# the files of a workDir that hold results (not logs, time stamps or absolute paths)
def _result_files(wd):
    out = {}
    for sub in ("kmer-counter-many/kmers", "kmer-counter-many/stats", "seq-builder-many/sequences", "seq-builder-many/sub-builder", "component-cutter",
                "features-calculator/vectors"):
        d = wd / sub
        for p in sorted(d.iterdir()):
            if p.is_file() and p.name not in ("in.properties", "out.properties", "SUCCESS"):
                out[sub + "/" + p.name] = p.read_bytes()
    mats = sorted((wd / "matrices").glob("dist_matrix_*_original_order.txt"))       # (one per run of the step, named by its start time)
    assert len(mats) >= 1
    out["matrix"] = mats[-1].read_bytes()
    return out


def _run_cli(tmp_path, wd, files, extra, k=21, b=1, l=60, b1=40, b2=2000):
    cmd = [os.path.join(ROOT, "metafast.sh"), "-k", str(k), "-b", str(b), "-l", str(l), "-b1", str(b1), "-b2", str(b2), "-i", *files, "-w", str(wd), *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=tmp_path)
    assert r.returncode == 0, r.stderr[-3000:]
    return r


def _devices_case(oracle, tmp_path, devices):
    k, b, l, b1, b2 = 21, 1, 60, 40, 2000
    files = []
    for s, n in enumerate((9000, 7000, 11000, 8000, 6000)):              # five libraries: the contexts get 3 + 2
        p = tmp_path / ("lib%d.fa" % s)
        _write_fasta(p, s, n)
        files.append(str(p))
    one = tmp_path / "wd_one"
    many = tmp_path / "wd_many"
    _run_cli(tmp_path, one, files, ["--device", "0"])
    os.environ["MF_SHARDED_CUTTER"] = "1"            # (entries on ONE device: the driver would not shard the cutter by itself)
    try:
        r = _run_cli(tmp_path, many, files, ["--devices", devices, "-v"])
    finally:
        del os.environ["MF_SHARDED_CUTTER"]
    assert "5 libraries on 2 device contexts" in r.stderr
    # round 6: the component cutter is sharded over the contexts too (mf_cut_components_sharded_files over a local communicator) -- and replicated
    # on request; all three workDirs are the same, byte for byte
    assert "Cutting components on 2 device contexts (sharded cutter table)" in r.stderr
    repl = tmp_path / "wd_repl"
    r2 = subprocess.run([os.path.join(ROOT, "metafast.sh"), "-k", str(k), "-b", str(b), "-l", str(l), "-b1", str(b1), "-b2", str(b2), "-i", *files, "-w", str(repl),
                         "--devices", devices, "-v"], capture_output=True, text=True, timeout=900, cwd=tmp_path, env=dict(os.environ, MF_REPLICATED_CUTTER="1", MF_SHARDED_CUTTER="1"))
    assert r2.returncode == 0 and "sharded cutter table" not in r2.stderr, r2.stderr[-2000:]
    assert _result_files(repl)["component-cutter/components.bin"] == _result_files(many)["component-cutter/components.bin"]
    a, m = _result_files(one), _result_files(many)
    assert sorted(a) == sorted(m)
    for name in a:
        assert a[name] == m[name], name
    assert "seq-builder-many/sub-builder/distribution" in a
    # ... and both are the oracle's matrix
    res = oracle.run_pipeline(files, k=k, b=b, l=l, b1=b1, b2=b2)
    assert len(res["comps"]) >= 3
    lines = m["matrix"].decode().splitlines()
    assert lines[0] == "#\t" + "\t".join("lib%d" % s for s in range(5))
    for i in range(5):
        assert lines[1 + i] == ("lib%d\t" % i) + "\t".join("%.4f" % res["matrix"][i, j] for j in range(5))
    for i in range(5):
        got = np.array(m["features-calculator/vectors/lib%d.vec" % i].decode().split(), dtype=np.int64)
        assert np.array_equal(got, res["vecs"][i])
    # the step bookkeeping of the many-context run is the reference's (Tool.java:318-392): per-step SUCCESS + out.properties in library order
    outp = (many / "kmer-counter-many" / "out.properties").read_text().splitlines()
    assert outp == ["resulting-kmers-files = %s" % (many / "kmer-counter-many" / "kmers" / ("lib%d.kmers.bin" % i)) for i in range(5)]
    # --continue from features-calculator on the many-context run: the workers meet contexts made for nothing before (fresh process)
    r = subprocess.run([os.path.join(ROOT, "metafast.sh"), "-k", str(k), "-b", str(b), "-l", str(l), "-b1", str(b1), "-b2", str(b2), "-i", *files, "-w", str(many),
                        "--devices", devices, "-c", "-s", "features-calculator"], capture_output=True, text=True, timeout=900, cwd=tmp_path)

    assert r.returncode == 0, r.stderr[-3000:]
    m2 = _result_files(many)
    for name in a:
        if name != "matrix":
            assert a[name] == m2[name], name
    assert m2["matrix"] == a["matrix"]


def test_cli_two_contexts_on_one_gpu(gpu_ctx, oracle, tmp_path):
    """metafast.sh --devices 0,0: two contexts (two host threads, two streams) on the one GPU of this pool"""
    _devices_case(oracle, tmp_path, "0,0")


def test_cli_sharded_cutter_more_contexts_than_libraries(gpu_ctx, tmp_path):
    """round 6: four contexts, two libraries -- the ranks without a library bring no sequences but own their shard of the cutter table;
    three contexts: the cutter runs on the largest power of two of them (2); an unreadable .seq.fasta: the reference's message, no hang"""
    files = []
    for s in range(2):
        p = tmp_path / ("m%d.fa" % s)
        _write_fasta(p, s, 7000)
        files.append(str(p))
    one = tmp_path / "wd1"
    _run_cli(tmp_path, one, files, ["--device", "0"])
    want = (one / "component-cutter" / "components.bin").read_bytes()
    assert len(want) > 16
    os.environ["MF_SHARDED_CUTTER"] = "1"
    try:
        for devs, w in (("0,0,0,0", 4), ("0,0,0", 2)):
            wd = tmp_path / ("wd_" + devs.replace(",", ""))
            r = _run_cli(tmp_path, wd, files, ["--devices", devs, "-v"])
            assert "Cutting components on %d device contexts (sharded cutter table)" % w in r.stderr
            assert (wd / "component-cutter" / "components.bin").read_bytes() == want
            assert _result_files(wd)["matrix"] == _result_files(one)["matrix"]
        # the component-cutter alone on sequence files one of which does not exist
        seqs = sorted(str(p) for p in (one / "seq-builder-many" / "sequences").iterdir())
        r = subprocess.run([os.path.join(ROOT, "metafast.sh"), "-t", "component-cutter", "-k", "21", "-i", seqs[0], str(tmp_path / "nothing.seq.fasta"), "-w", str(tmp_path / "wdx"),
                            "--devices", "0,0"], capture_output=True, text=True, timeout=300, cwd=tmp_path)
        assert r.returncode == 1 and "nothing.seq.fasta" in r.stderr, r.stderr[-1500:]
    finally:
        del os.environ["MF_SHARDED_CUTTER"]


def test_cli_two_gpus(gpu_ctx, oracle, tmp_path):
    # (gpu_ctx first, in every test of this file: torch.cuda.device_count() before torch and the library have initialised their HIP runtimes in
    # the usual order -- torch's, then the library's -- left the library without a device on the GPU box)
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU on this box")
    _devices_case(oracle, tmp_path, "0,1")


def test_cli_default_is_every_visible_device(gpu_ctx, tmp_path):
    """without --device(s) the driver takes every device the process sees (mf_device_count)"""
    # (gpu_ctx first: torch must have initialised ITS HIP runtime before the library initialises the system's -- the second one to start finds no device)
    import torch
    from metafast_amd import lib as L
    assert L.lib().mf_device_count() == torch.cuda.device_count()
    files = []
    for s in range(2):
        p = tmp_path / ("d%d.fa" % s)
        _write_fasta(p, s, 5000)
        files.append(str(p))
    # two small libraries: every visible device, and two contexts on each where two libraries fit side by side (one GPU: 2 contexts on it)
    r = _run_cli(tmp_path, tmp_path / "wd", files, ["-v"])
    assert "2 libraries on 2 device contexts" in r.stderr
    env = dict(os.environ, MF_CONTEXTS_PER_DEVICE="1")
    r = subprocess.run([os.path.join(ROOT, "metafast.sh"), "-k", "21", "-l", "60", "-b1", "40", "-b2", "2000", "-i", *files, "-w", str(tmp_path / "wd1"), "-v"],
                       capture_output=True, text=True, timeout=300, cwd=tmp_path, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert ("2 libraries on 2 device contexts" in r.stderr) == (torch.cuda.device_count() >= 2)
    a, b = _result_files(tmp_path / "wd"), _result_files(tmp_path / "wd1")
    assert a == b
    r = subprocess.run([os.path.join(ROOT, "metafast.sh"), "-i", *files, "-w", str(tmp_path / "wd2"), "--devices", "0,99"], capture_output=True, text=True, timeout=300, cwd=tmp_path)
    assert r.returncode == 1 and "device 99 out of range" in r.stderr


def _selected_setup(gpu_ctx, oracle, tmp_path):
    """three libraries through count / unitigs / cutter / components on the GPU (files as the steps write them) + the oracle's twins"""
    k, b, l, b1, b2 = 21, 1, 60, 40, 2000
    kmers_files, seq_files, reads_files = [], [], []
    for s, n in enumerate((9000, 8000, 10000)):
        p = tmp_path / ("s%d.fa" % s)
        _write_fasta(p, s, n)
        reads_files.append(str(p))
        t = gpu_ctx.count_reads([str(p)], k)
        kb = tmp_path / ("s%d.kmers.bin" % s)
        t.write_kmers(b, str(kb))
        kmers_files.append(str(kb))
        sf = tmp_path / ("s%d.seq.fasta" % s)
        gpu_ctx.build_unitigs(t, b, l).write_fasta(str(sf))
        seq_files.append(str(sf))
    cutter = gpu_ctx.count_reads(seq_files, k, l)
    comps = gpu_ctx.cut_components(cutter, b1, b2)
    comps.write(str(tmp_path / "components.bin"))
    ocomps = oracle.cut_components(oracle.Table().count_files(seq_files, k, l), k, b1, b2)
    assert len(ocomps) >= 6
    return k, kmers_files, reads_files, comps, ocomps


def _selection_files(oracle, tmp_path, ocomps, rng):
    """two .kmers.bin files of selected k-mers (IOUtils.loadKmers sums them, :369-401): most k-mers of some components, none of
    component 1, a few k-mers that are in no component; one k-mer in both files"""
    allc = ocomps.all()
    pick = []
    for ci, (size, weight, thr, kmers) in enumerate(allc):
        if ci == 1:
            continue                                                       # a component without any selected k-mer: 0.0 / 0.0 = NaN
        kk = np.asarray(kmers, dtype=np.uint64)
        pick.append(kk[rng.random(len(kk)) < (0.6 if ci % 2 == 0 else 0.15)])
    pick = np.unique(np.concatenate(pick + [np.array([5, 77777, 123456789], dtype=np.uint64)]))
    rng.shuffle(pick)
    half = len(pick) // 2
    parts = [np.sort(pick[:half + 1]), np.sort(pick[half:])]              # pick[half] is in both
    paths = []
    for i, part in enumerate(parts):
        t = oracle.Table()
        for key in part.tolist():
            t.add(int(key), int(rng.integers(1, 9)))
        p = tmp_path / ("sel%d.kmers.bin" % i)
        t.write_kmers(0, str(p))
        paths.append(str(p))
    return paths


def test_features_selected_against_the_oracle(gpu_ctx, oracle, tmp_path):
    """--selected on the k-mers-file branch and the reads branch, handles and files: vec, found AND the breadth's denominator count the
    selected k-mers only (FeaturesCalculatorMain.java:193-203)"""
    from util import to_device
    from metafast_amd import lib as L
    rng = np.random.default_rng(11)
    k, kmers_files, reads_files, comps, ocomps = _selected_setup(gpu_ctx, oracle, tmp_path)
    sel_paths = _selection_files(oracle, tmp_path, ocomps, rng)
    osel = oracle.Table().load_kmers(sel_paths, 0)
    gsel = gpu_ctx.load_kmers(sel_paths, 0, k)
    assert [c[:3] for c in comps.export()] == [c[:3] for c in ocomps.all()]
    for thr in (0, 3):
        for s, kf in enumerate(kmers_files):
            og = oracle.Table().load_kmers([kf], -1)
            wv, wb = ocomps.features(og, thr, selected=osel)
            gt = gpu_ctx.load_kmers([kf], -1, k)
            gv, gb = gpu_ctx.features(comps, gt, thr, selected=gsel)
            assert np.array_equal(gv, wv), (thr, s)
            assert np.array_equal(np.isnan(gb), np.isnan(wb)) and np.isnan(wb[1]) and np.array_equal(gb[~np.isnan(gb)], wb[~np.isnan(wb)])
            nv, nb = gpu_ctx.features(comps, gt, thr)                       # (no selection: the plain call, and it differs)
            pv, pb = ocomps.features(og, thr)
            assert np.array_equal(nv, pv) and np.array_equal(nb, pb) and not np.array_equal(nv, gv)
            # the file form
            gpu_ctx.features_files(str(tmp_path / "components.bin"), kf, k, thr, str(tmp_path / "f.vec"), str(tmp_path / "f.breadth"), selected=gsel)
            assert [int(x) for x in (tmp_path / "f.vec").read_text().split()] == wv.tolist()
            br = (tmp_path / "f.breadth").read_text().split()
            assert br[1] == "NaN" and [float(x) for x in br if x != "NaN"] == wb[~np.isnan(wb)].tolist()
    # reads branch (--use-reads-for-calculating-features + --selected)
    for s, rf in enumerate(reads_files[:2]):
        bases = np.concatenate([np.frombuffer(line.encode(), dtype=np.uint8) for line in open(rf).read().split("\n")[1::2]])
        n = len(bases) // 100
        off = np.arange(n + 1, dtype=np.uint64) * np.uint64(100)
        wv, wb = oracle.features_from_reads(ocomps, bases, off, k, 0, selected=osel)
        tb, to = to_device(bases, off)
        gv, gb = gpu_ctx.features_reads(comps, tb.data_ptr(), to.data_ptr(), n, int(off[-1]), k, 0, selected=gsel)
        assert np.array_equal(gv, wv) and np.array_equal(gb[~np.isnan(gb)], wb[~np.isnan(wb)]) and np.isnan(gb[1])


def test_cli_selected(gpu_ctx, oracle, tmp_path):
    """metafast.sh -t features-calculator --selected a b (two device contexts: each loads the selection for itself), and matrix-builder
    --selected, against the oracle"""
    rng = np.random.default_rng(12)
    k, kmers_files, reads_files, comps, ocomps = _selected_setup(gpu_ctx, oracle, tmp_path)
    sel_paths = _selection_files(oracle, tmp_path, ocomps, rng)
    osel = oracle.Table().load_kmers(sel_paths, 0)
    wd = tmp_path / "wd"
    cmd = [os.path.join(ROOT, "metafast.sh"), "-t", "features-calculator", "-k", str(k), "-cm", str(tmp_path / "components.bin"), "-ka", *kmers_files,
           "--selected", *sel_paths, "-w", str(wd), "--devices", "0,0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=tmp_path)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "selected = %s" % sel_paths[1] in (wd / "in.properties").read_text()
    for s, kf in enumerate(kmers_files):
        wv, wb = ocomps.features(oracle.Table().load_kmers([kf], -1), 0, selected=osel)
        assert [int(x) for x in (wd / "vectors" / ("s%d.vec" % s)).read_text().split()] == wv.tolist()
        br = (wd / "vectors" / ("s%d.breadth" % s)).read_text().split()
        assert br[1] == "NaN" and [float(x) for x in br if x != "NaN"] == wb[~np.isnan(wb)].tolist()
    # matrix-builder re-exports the option (DistanceMatrixBuilderMain.java: addSubTool) -- same vectors through the whole run
    wd2 = tmp_path / "wd2"
    r = subprocess.run([os.path.join(ROOT, "metafast.sh"), "-k", str(k), "-b", "1", "-l", "60", "-b1", "40", "-b2", "2000", "-i", *reads_files, "--selected", *sel_paths,
                        "-w", str(wd2)], capture_output=True, text=True, timeout=600, cwd=tmp_path)
    assert r.returncode == 0, r.stderr[-3000:]
    assert (wd2 / "component-cutter" / "components.bin").read_bytes() == (tmp_path / "components.bin").read_bytes()
    for s in range(3):
        assert (wd2 / "features-calculator" / "vectors" / ("s%d.vec" % s)).read_bytes() == (wd / "vectors" / ("s%d.vec" % s)).read_bytes()


def test_cami_example_parameters_against_the_oracle(gpu_ctx, oracle, tmp_path):
    """the reference's worked example runs `-k 23 -b 5 -l 1200` (Example.md:18-21, example_scripts/run_metafast.sh:2): the whole path at
    those parameters -- k = 23 takes the k-specialised neighbour kernels -- against the oracle, components and matrix identical"""
    from test_shapes_gpu import _samples_against_the_oracle
    _samples_against_the_oracle(gpu_ctx, oracle, tmp_path, S=3, k=23, b=5, l=1200, b1=1000, b2=10000, n=1_000_000, min_thr=1)


@pytest.mark.parametrize("k", [25, 27, 29, 22])
def test_pipeline_against_the_oracle_other_k(gpu_ctx, oracle, tmp_path, k):
    """k = 25, 27, 29: compile-time-k neighbour kernels (25 and 29 at five waves per SIMD, the others at six); 22: the generic build"""
    from test_shapes_gpu import _samples_against_the_oracle
    _samples_against_the_oracle(gpu_ctx, oracle, tmp_path, S=2, k=k, b=1, l=100, b1=500, b2=5000, n=600_000, min_thr=2)


# ---- the device parser (mf_dparse.hip) against the oracle's readers ----
def _fa_text(rng, n_rec, wrap, crlf=False, n_rate=0.0, lower=False, iupac=False, comments=False, lead_seq=False, empty_lines=False, no_final_nl=False):
    nl = b"\r\n" if crlf else b"\n"
    al = b"ACGT" + (b"acgt" if lower else b"") + (b"RYMKSWHBVDrymkswhbvd" if iupac else b"")
    out = []
    if lead_seq:
        out.append(bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=37)]) + nl)
    for i in range(n_rec):
        if comments and i % 7 == 3:
            out.append(b";a comment > with ; marks" + nl)
        out.append(b">read_%d some description N n . @ +" % i + nl)
        ln = int(rng.integers(0, 260))
        s = np.frombuffer(al, dtype=np.uint8)[rng.integers(0, len(al), size=ln)].copy()
        if n_rate and rng.random() < n_rate and ln:
            s[int(rng.integers(0, ln))] = ord("N") if rng.random() < 0.5 else ord("n")
        s = s.tobytes()
        w = wrap if wrap else max(ln, 1)
        for j in range(0, max(ln, 1), w):
            out.append(s[j:j + w] + nl)
        if empty_lines and i % 11 == 5:
            out.append(nl)
    t = b"".join(out)
    if no_final_nl and t.endswith(nl):
        t = t[:-len(nl)]
    return t


def _fq_text(rng, n_rec, crlf=False, n_rate=0.0, phred0_rate=0.0, qoff=33, odd_headers=False, no_final_nl=False, lower=False):
    nl = b"\r\n" if crlf else b"\n"
    out = []
    for i in range(n_rec):
        ln = int(rng.integers(1, 260))
        al = b"ACGTacgt" if lower else b"ACGT"
        s = np.frombuffer(al, dtype=np.uint8)[rng.integers(0, len(al), size=ln)].copy()
        q = rng.integers(qoff + 1, min(qoff + 42, 127), size=ln).astype(np.uint8)
        if n_rate and rng.random() < n_rate:
            s[int(rng.integers(0, ln))] = int(rng.choice([ord("N"), ord("n"), ord(".")]))
        if phred0_rate and rng.random() < phred0_rate:
            q[int(rng.integers(0, ln))] = qoff
        if odd_headers and i % 5 == 2:
            q[0] = ord("@") if qoff < 64 else q[0]                      # a quality line that starts with '@'
        out.append((b"+odd%d" if odd_headers and i % 9 == 4 else b"@r%d") % i + nl + s.tobytes() + nl + (b"@again" if odd_headers and i % 4 == 1 else b"+") + nl + q.tobytes() + nl)
    t = b"".join(out)
    if no_final_nl:
        t = t[:-len(nl)]
    return t


def _same_reads(ctx, oracle, path):
    """device parser == host parser == the oracle's reader, byte for byte; -> whether the device parser took the file"""
    ob, oo = oracle.read_file(str(path))
    ctx.set_option("device_parse", 0)
    hb, ho = ctx.load_reads([str(path)])
    ctx.set_option("device_parse", 1)
    before = ctx.stat("device_parsed_files"), ctx.stat("device_parser_stepped_back")
    db, do = ctx.load_reads([str(path)])
    took = ctx.stat("device_parsed_files") == before[0] + 1
    assert took or ctx.stat("device_parser_stepped_back") == before[1] + 1
    assert np.array_equal(ho, oo) and np.array_equal(hb, ob), "host parser != oracle"
    assert np.array_equal(do, oo), (path, len(do), len(oo), np.flatnonzero(do[:min(len(do), len(oo))] != oo[:min(len(do), len(oo))])[:5])
    assert np.array_equal(db, ob), (path, np.flatnonzero(db != ob)[:5])
    return took


def test_gz_files_inflate_into_hbm(gpu_ctx, oracle, tmp_path):
    """Round 5 (mf_inflate.h, mf_dparse_gz): a .fa.gz / .fq.gz file is inflated by the host on many threads -- one gzip member cut into pieces at
    block starts found by trying bit positions, pieces decoded without their 32 KB of history -- straight into the pinned staging chunks, and the
    text is parsed in HBM.  With the limits lowered so that files of a few MB go this way (pieces of 64 KB): the oracle's reads, byte for byte;
    FASTQ with both quality offsets; a file of two members and a BGZF file step back to the host's inflater, a damaged file fails as before."""
    import gzip, zlib
    rng = np.random.default_rng(31)
    al = np.frombuffer(b"ACGT", dtype=np.uint8)
    n = 40000
    reads = al[rng.integers(0, 4, (n, 150))]
    reads[rng.random((n, 150)) < 0.002] = ord("N")
    fa = b"".join(b">read_%d some text\n" % i + reads[i].tobytes() + b"\n" for i in range(n))
    files = {"a.fa.gz": gzip.compress(fa, 6), "b.fasta.gz": gzip.compress(fa, 1)}
    plain = {"a.fa.gz": fa, "b.fasta.gz": fa, "two_members.fa.gz": fa}
    for name, qlo in (("c.fq.gz", 33), ("d.fastq.gz", 64)):
        q = rng.integers(qlo + 2, qlo + 41, (n, 150), dtype=np.uint8)
        fq = b"".join(b"@r%d\n" % i + reads[i].tobytes() + b"\n+\n" + q[i].tobytes() + b"\n" for i in range(n))
        files[name] = gzip.compress(fq, 4); plain[name] = fq
    files["two_members.fa.gz"] = gzip.compress(fa[: len(fa) // 2], 6) + gzip.compress(fa[len(fa) // 2:], 6)
    want = {}
    try:
        gpu_ctx.set_option("gz_device_min_bytes", 0); gpu_ctx.set_option("gz_piece_bytes", 65536); gpu_ctx.set_option("device_parse_min_bytes", 1)
        for name, blob in files.items():
            p = tmp_path / name
            p.write_bytes(blob)
            (tmp_path / name[:-3]).write_bytes(plain[name])                        # (the oracle's reader takes plain files)
            ob, oo = want[name] = oracle.read_file(str(tmp_path / name[:-3]))
            before = gpu_ctx.stat("gz_files_inflated_into_hbm")
            db, do = gpu_ctx.load_reads([str(p)])
            assert np.array_equal(do, oo) and np.array_equal(db, ob), name
            assert gpu_ctx.stat("gz_files_inflated_into_hbm") == before + (0 if name.startswith("two") else 1), name
        # two files of one library, one of them that way
        (ob1, oo1), (ob2, oo2) = want["a.fa.gz"], want["c.fq.gz"]
        db, do = gpu_ctx.load_reads([str(tmp_path / "a.fa.gz"), str(tmp_path / "c.fq.gz")])
        assert np.array_equal(db, np.concatenate([ob1, ob2])) and np.array_equal(do, np.concatenate([oo1, oo2[1:] + oo1[-1]]))
        bad = bytearray(files["a.fa.gz"]); bad[len(bad) // 2] ^= 0x10
        (tmp_path / "bad.fa.gz").write_bytes(bytes(bad))
        from metafast_amd.lib import MetafastError
        with pytest.raises(MetafastError, match="GZIP|gzip|corrupt"):
            gpu_ctx.load_reads([str(tmp_path / "bad.fa.gz")])
    finally:
        gpu_ctx.set_option("gz_device_min_bytes", 32 << 20); gpu_ctx.set_option("gz_piece_bytes", 2 << 20); gpu_ctx.set_option("device_parse_min_bytes", 1 << 20)


def test_device_parser_fasta_and_fastq(gpu_ctx, oracle, tmp_path):
    rng = np.random.default_rng(21)
    gpu_ctx.set_option("profile", 1)
    gpu_ctx.set_option("device_parse_min_bytes", 1)
    try:
        cases = []
        for i, kw in enumerate([dict(wrap=0), dict(wrap=70), dict(wrap=60, crlf=True), dict(wrap=0, n_rate=0.05), dict(wrap=80, n_rate=0.2, lower=True, iupac=True),
                                dict(wrap=0, comments=True, lead_seq=True), dict(wrap=70, empty_lines=True, comments=True), dict(wrap=0, no_final_nl=True),
                                dict(wrap=0, crlf=True, no_final_nl=True, n_rate=0.1), dict(wrap=13, lead_seq=True, n_rate=0.3)]):
            for n_rec in (1, 3, 40, 5000):
                p = tmp_path / ("a%d_%d.fa" % (i, n_rec))
                p.write_bytes(_fa_text(rng, n_rec, **kw))
                cases.append(p)
        for i, kw in enumerate([dict(), dict(crlf=True), dict(n_rate=0.1), dict(phred0_rate=0.1), dict(qoff=64, phred0_rate=0.05, n_rate=0.05), dict(odd_headers=True),
                                dict(no_final_nl=True), dict(crlf=True, no_final_nl=True, n_rate=0.2), dict(lower=True, n_rate=0.02)]):
            for n_rec in (1, 2, 50, 5000):
                p = tmp_path / ("q%d_%d.fq" % (i, n_rec))
                p.write_bytes(_fq_text(rng, n_rec, **kw))
                cases.append(p)
        took = sum(_same_reads(gpu_ctx, oracle, p) for p in cases)
        assert took == len(cases), (took, len(cases))                        # (every one of these is a file the device parser is sure about)
        # chunk borders: records, header lines and CRLF pairs that straddle the 4 KB tiles and the 256 KB chunks
        big = tmp_path / "big.fa"
        big.write_bytes(_fa_text(rng, 40000, 0, n_rate=0.01) + _fa_text(rng, 20000, 61, crlf=True, n_rate=0.02, comments=True))
        assert _same_reads(gpu_ctx, oracle, big)
        bigq = tmp_path / "big.fastq"
        bigq.write_bytes(_fq_text(rng, 30000, n_rate=0.02, phred0_rate=0.02) )
        assert _same_reads(gpu_ctx, oracle, bigq)
        # two files in one call (a paired library): the device pieces are joined in HBM
        ob1, oo1 = oracle.read_file(str(big)); ob2, oo2 = oracle.read_file(str(bigq))
        b, o = gpu_ctx.load_reads([str(big), str(bigq)])
        assert np.array_equal(b, np.concatenate([ob1, ob2])) and np.array_equal(o, np.concatenate([oo1, oo2[1:] + oo1[-1]]))
    finally:
        gpu_ctx.set_option("profile", 0)
        gpu_ctx.set_option("device_parse_min_bytes", 1 << 20)


def test_device_parser_steps_back(gpu_ctx, oracle, tmp_path):
    """files the device parser is not sure about go to the host readers: same reads as the oracle, or the reference's error"""
    from metafast_amd import lib as L
    rng = np.random.default_rng(22)
    gpu_ctx.set_option("profile", 1)
    gpu_ctx.set_option("device_parse_min_bytes", 1)
    try:
        good_fa = _fa_text(rng, 300, 0)
        good_fq = _fq_text(rng, 300)
        odd = {"lone_cr.fa": good_fa.replace(b"\n>read_7 ", b"\r>read_7 ", 1), "lone_cr_in_header.fa": good_fa.replace(b"read_9 some", b"read_9\rACGT", 1),
               "empty_lines.fq": good_fq.replace(b"\n@r20\n", b"\n\n@r20\n", 1), "leading_empty.fq": b"\n" + good_fq, "crlf_empty.fq": good_fq.replace(b"\n@r30\n", b"\n\r\n@r30\n", 1)}
        for name, text in odd.items():
            p = tmp_path / name
            p.write_bytes(text)
            assert not _same_reads(gpu_ctx, oracle, p), name                  # (same reads, made by the host parser)
        bad = {"bad_char.fa": good_fa.replace(b"\nA", b"\nJ", 1), "bad_char.fq": good_fq.replace(b"\nA", b"\n#", 1), "truncated.fq": good_fq[: len(good_fq) // 2],
               "lengths.fq": good_fq.replace(b"\n+\n", b"\n+\nI", 1), "structure.fq": good_fq.replace(b"\n@r5\n", b"\nr5\n", 1), "bad_qual.fq": good_fq[:-2] + b"\x1f\n"}
        for name, text in bad.items():
            p = tmp_path / name
            p.write_bytes(text)
            msgs = []
            for dp in (0, 1):
                gpu_ctx.set_option("device_parse", dp)
                with pytest.raises(L.MetafastError) as e:
                    gpu_ctx.load_reads([str(p)])
                msgs.append(str(e.value))
            assert msgs[0] == msgs[1], (name, msgs)                            # (the host reader's message either way)
            with pytest.raises(Exception):
                oracle.read_file(str(p))
    finally:
        gpu_ctx.set_option("device_parse", 1)
        gpu_ctx.set_option("profile", 0)
        gpu_ctx.set_option("device_parse_min_bytes", 1 << 20)


@pytest.mark.parametrize("k", [21, 23, 25])
def test_low_complexity_reads_with_the_short_minimizer(gpu_ctx, oracle, k):
    """k <= 25 takes 13-mers as minimizers (mf_skm_m), with their own hash seed: homopolymer and microsatellite stretches between random flanks --
    thousands of distinct k-mers around a few low-complexity M-mers -- are counted exactly (heavy partitions in several passes or through the
    k-mer path), and lookups find them"""
    rng = np.random.default_rng(100 + k)
    n_reads = 30000
    lut = np.frombuffer(b"AGCT", dtype=np.uint8)
    fl = lut[rng.integers(0, 4, size=(n_reads, 90))]
    units = [b"A", b"T", b"AC", b"AG", b"ACG", b"AAT", b"C", b"GT"]
    mids = np.stack([np.frombuffer((units[i % len(units)] * 30)[:24], dtype=np.uint8) for i in range(n_reads)])
    arr = np.concatenate([fl[:, :45], mids, fl[:, 45:]], axis=1)
    b = np.concatenate([arr.reshape(-1), np.zeros(64, dtype=np.uint8)])
    o = np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(arr.shape[1])
    from util import to_device
    tb, to = to_device(b[: n_reads * arr.shape[1]], o)
    t = gpu_ctx.count_device(tb.data_ptr(), to.data_ptr(), n_reads, int(o[-1]), k, 0)
    gk, gc = t.export()
    ok, ov = oracle.Table().count_buffer(b[: n_reads * arr.shape[1]], o, k).export()
    assert np.array_equal(gk, ok) and np.array_equal(gc.astype(np.int32), ov)
    pick = rng.choice(len(ok), size=3000, replace=False)
    got = t.lookup(np.concatenate([ok[pick], rng.integers(0, 1 << (2 * k), size=300, dtype=np.uint64) | np.uint64(1 << 62)]))
    assert np.array_equal(got[:3000], ov[pick]) and np.all(got[3000:] == -1)


@pytest.mark.parametrize("k,read_len", [(31, 250), (25, 250), (21, 400)])
def test_long_reads_are_planned_by_the_pilot(gpu_ctx, oracle, k, read_len):
    """Round 5: reads of 8 k bases or more were taken for assembled sequences by their length alone (no pilot, units planned for all-distinct
    k-mers: 250-base reads at k = 25 ran 2.6 x slower than 150-base reads of the same volume, profiles/r05az_probe_shapes.txt).  Now the pilot
    looks at them like at any reads (stat pilot_runs); sequences that come with a length filter (the cutter's input) are still planned as
    assembled.  Counts equal the oracle's either way."""
    from util import genome_reads, to_device
    rng = np.random.default_rng(40 + k)
    b, o = genome_reads(rng, 2_000_000, 30_000_000 // read_len, read_len, err=0.005)        # (3e7 bases: enough for a plan of two radix levels)
    tb, to = to_device(b, o)
    ok, ov = oracle.Table().count_buffer(b, o, k).export()
    before = gpu_ctx.stat("pilot_runs")
    t = gpu_ctx.count_device(tb.data_ptr(), to.data_ptr(), len(o) - 1, int(o[-1]), k, 0)
    gk, gc = t.export()
    assert np.array_equal(gk, ok) and np.array_equal(gc.astype(np.int32), ov)
    assert gpu_ctx.stat("pilot_runs") == before + 1
    t2 = gpu_ctx.count_device(tb.data_ptr(), to.data_ptr(), len(o) - 1, int(o[-1]), k, read_len)        # (a length filter: the cutter's call)
    gk, gc = t2.export()
    assert np.array_equal(gk, ok) and np.array_equal(gc.astype(np.int32), ov)
    assert gpu_ctx.stat("pilot_runs") == before + 1


@pytest.mark.parametrize("kind,kt,vt", [(0, np.uint64, np.uint16), (1, np.uint32, np.uint32), (2, np.uint32, np.uint64), (3, np.uint64, np.uint32), (4, np.uint64, np.uint64)])
def test_radix_sort_kernels(gpu_ctx, kind, kt, vt):
    """mf_sort.hip (hand-written since round 5): stable, ascending by the low `bits` key bits and by nothing above them, for every (key, value)
    width the file seams use, at sizes around the kernels' borders (a wave's 4096 elements, a workgroup's 16384, one and several passes)"""
    import ctypes as C
    from metafast_amd import lib as L
    fn = L.lib().mf_debug_sort
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(40 + kind)
    kbits = 8 * np.dtype(kt).itemsize
    for n in (1, 63, 64, 65, 4095, 4096, 4097, 16384, 16385, 100_003, 3_000_017):
        for bits in sorted({1, 5, 8, 9, 21, kbits - 2, kbits}):
            # few distinct keys (long runs of equal ones: stability), or all bits random; bits above `bits` set on purpose
            keys = rng.integers(0, 1 << min(bits, 12), size=n, dtype=np.uint64) if rng.random() < 0.4 else rng.integers(0, 1 << 63, size=n, dtype=np.uint64) >> np.uint64(63 - min(bits, 62))
            junk = rng.integers(0, 1 << 63, size=n, dtype=np.uint64) << np.uint64(bits) if bits < kbits else np.zeros(n, dtype=np.uint64)
            keys = ((keys & np.uint64((1 << bits) - 1)) | junk).astype(kt)
            vals = np.arange(n, dtype=np.uint64).astype(vt) if np.dtype(vt).itemsize >= 4 else (np.arange(n) % 65521).astype(vt)
            ko, vo = np.empty_like(keys), np.empty_like(vals)
            assert fn(gpu_ctx.h, kind, keys.ctypes.data, vals.ctypes.data, n, bits, ko.ctypes.data, vo.ctypes.data) == 0
            order = np.argsort(keys.astype(np.uint64) & np.uint64((1 << bits) - 1), kind="stable")
            assert np.array_equal(ko, keys[order]) and np.array_equal(vo, vals[order]), (kind, n, bits)

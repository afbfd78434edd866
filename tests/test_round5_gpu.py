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
    mats = sorted((wd / "matrices").glob("dist_matrix_*_original_order.txt"))
    assert len(mats) == 1
    out["matrix"] = mats[0].read_bytes()
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
    r = _run_cli(tmp_path, many, files, ["--devices", devices, "-v"])
    assert "5 libraries on 2 device contexts" in r.stderr
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


def test_cli_two_contexts_on_one_gpu(oracle, tmp_path):
    """metafast.sh --devices 0,0: two contexts (two host threads, two streams) on the one GPU of this pool"""
    _devices_case(oracle, tmp_path, "0,0")


def test_cli_two_gpus(oracle, tmp_path):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU on this box")
    _devices_case(oracle, tmp_path, "0,1")


def test_cli_default_is_every_visible_device(tmp_path):
    """without --device(s) the driver takes every device the process sees (mf_device_count)"""
    import torch
    from metafast_amd import lib as L
    assert L.lib().mf_device_count() == torch.cuda.device_count()
    files = []
    for s in range(2):
        p = tmp_path / ("d%d.fa" % s)
        _write_fasta(p, s, 5000)
        files.append(str(p))
    r = _run_cli(tmp_path, tmp_path / "wd", files, ["-v"])
    if torch.cuda.device_count() >= 2:
        assert "2 libraries on 2 device contexts" in r.stderr
    else:
        assert "device contexts" not in r.stderr
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

"""Pins the CPU oracle against the reference's only golden vector for this path
(test_data/meta_test_matrix.txt) and the restatement-derived known answers of BASELINE.md."""
import os

import numpy as np

from conftest import REF_DATA


def _golden_matrix():
    rows = {}
    with open(os.path.join(REF_DATA, "meta_test_matrix.txt")) as f:
        names = f.readline().split()[1:]
        for line in f:
            p = line.split()
            rows[p[0]] = dict(zip(names, map(float, p[1:])))
    return rows


def test_golden_matrix_exact(oracle, ref_files):
    r = oracle.run_pipeline(ref_files)         # defaults k=31 b=1 l=100 b1=1000 b2=10000
    g = _golden_matrix()
    names = ["meta_test_1", "meta_test_2", "meta_test_3"]
    for i, a in enumerate(names):
        for j, b in enumerate(names):
            assert r["matrix"][i, j] == g[a][b], (a, b)     # all 16 printed digits
    assert r["matrix"][0, 1] == 0.5691162409506898
    assert r["matrix"][0, 2] == 0.2981399448537721
    assert r["matrix"][1, 2] == 0.8448331091037222


def test_known_answers_defaults(oracle, ref_files):
    r = oracle.run_pipeline(ref_files)
    got = [(s["n_distinct"], s["n_good"], len(s["seqs"]), s["seqs"].total_len()) for s in r["samples"]]
    assert got == [(17063, 16918, 15, 17322), (8176, 7321, 29, 7910), (14042, 11351, 25, 11123)]
    assert len(r["cutter"]) == 17061
    assert [(a, b, c) for a, b, c, _ in r["comps"].all()] == [
        (6240, 12783, 1), (5713, 11265, 1), (3020, 5977, 1), (2088, 4260, 1)]
    assert r["vecs"].tolist() == [[41935, 38354, 20375, 14211], [20208, 0, 0, 11337], [6517, 34484, 20359, 749]]


def test_known_answers_split(oracle, ref_files):
    """b1=50, b2=500 forces threshold-2 splits (ComponentsBuilder.java:157-180)."""
    r = oracle.run_pipeline(ref_files, b1=50, b2=500)
    cs = [(a, b, c) for a, b, c, _ in r["comps"].all()]
    assert len(cs) == 37
    assert cs[:4] == [(448, 1076, 2), (456, 1066, 2), (426, 939, 2), (467, 934, 2)]
    assert r["matrix"][0, 1] == 0.40246783273019704


def test_occurrence_counts(oracle, ref_files):
    occ = []
    for f in ref_files:
        b, o = oracle.read_file(f)
        lens = np.diff(o.astype(np.int64))
        occ.append(int(np.maximum(lens - 31 + 1, 0).sum()))
    assert occ == [115020, 32400, 64800]


def test_fastq_reader(oracle):
    # quality '#' (35) < 64 -> Illumina parse fails -> Sanger, phred 2 -> reads kept (ReadersUtils.java:63-77)
    b, o = oracle.read_file(os.path.join(REF_DATA, "tinytest_A.fastq"))
    assert bytes(b) == b"AACATAAGCGAAGCCAAC" and o.tolist() == [0, 9, 18]
    t = oracle.Table().count_files([os.path.join(REF_DATA, "tinytest_A.fastq")], 5)
    keys, vals = t.export()
    assert len(keys) == 10 and vals.tolist() == [1] * 10


def test_readers_edge_cases(oracle, tmp_path):
    fa = tmp_path / "x.fa"
    fa.write_text(">r1\nACGT\nACGT\n;comment\n>r2\nACNGT\n>r3\nacgtac\n\n>r4\n")
    b, o = oracle.read_file(str(fa))
    assert bytes(b) == b"ACGTACGTACGTAC" and o.tolist() == [0, 8, 14]       # multi-line joined, N read dropped
    fq = tmp_path / "y.fq"
    fq.write_text("@a\nACGT\n+\nIIII\n@b\nACNT\n+\nIIII\n@c\nGGGG\n+\nI!II\n@d\nTTTT\n+\nIIII\n")
    b, o = oracle.read_file(str(fq))
    assert bytes(b) == b"ACGTTTTT"      # N read and phred-0 ('!' in Sanger) read dropped


def test_saturation_and_polya(oracle):
    from util import pack_reads
    bases, off = pack_reads(["A" * 40000])
    t = oracle.Table().count_buffer(bases, off, 31)
    keys, vals = t.export()
    assert keys.tolist() == [0] and vals.tolist() == [32767]        # key 0 (poly-A == poly-T) saturates


def test_kmers_bin_roundtrip(oracle, ref_files, tmp_path):
    t = oracle.Table().count_files([ref_files[1]], 31)
    kb, st = tmp_path / "s.kmers.bin", tmp_path / "s.stat.txt"
    good = t.write_kmers(1, str(kb), str(st))
    assert good == 7321 and os.path.getsize(kb) == 73210
    t2 = oracle.Table().load_kmers([str(kb)], 0)
    k1, v1 = t.export(1)
    k2, v2 = t2.export()
    assert np.array_equal(k1, k2) and np.array_equal(v1, v2)
    lines = st.read_text().split("\n")
    assert lines[0] == "# k-mer frequency\tnumber of such k-mers" and lines[-2:] == ["", ""]
    assert sum(int(x.split("\t")[1]) for x in lines[1:-2]) == 8176


def test_cpu_baseline_matches(oracle, ref_files):
    b, o = oracle.read_file(ref_files[0])
    d, occ = oracle.cpu_baseline_count(b, o, 31, 4)
    assert (d, occ) == (17063, 115020)

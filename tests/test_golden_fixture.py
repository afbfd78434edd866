"""Committed known answers (tests/golden/known_answers.json, made by tests/golden/make_golden.py from the oracle that the
reference's golden matrix pins).  CPU: the oracle still reproduces them.  GPU: the HIP path reproduces them WITHOUT the
oracle in the loop."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from util import canon_seq

KA = json.load(open(os.path.join(GOLDEN, "known_answers.json")))


def _digest_table(keys, vals):
    return hashlib.sha256(np.asarray(keys, dtype="<u8").tobytes() + np.asarray(vals, dtype="<u2").tobytes()).hexdigest()


def _digest_seqs(seqs):
    return hashlib.sha256("\n".join(sorted(f"{canon_seq(s)} {a} {mn} {mx}" for s, a, mn, mx in seqs)).encode()).hexdigest()


def _members_digest(comps):
    return hashlib.sha256(b"".join(np.asarray(km, dtype="<u8").tobytes() for _, _, _, km in comps)).hexdigest()


def test_oracle_matches_fixture(oracle, ref_files):
    for f, s in zip(ref_files, KA["samples"]):
        keys, vals = oracle.Table().count_files([f], 31).export()
        assert (len(keys), _digest_table(keys, vals)) == (s["n_distinct"], s["counts_sha256"])
    r = oracle.run_pipeline(ref_files)
    p = KA["pipelines"]["default"]
    assert [[a, w, t] for a, w, t, _ in r["comps"].all()] == p["components"]
    assert r["vecs"].tolist() == p["vectors"] and r["matrix"].tolist() == p["matrix"]
    assert p["matrix"][0][1] == 0.5691162409506898         # the reference's own golden value


def _gpu_pipeline_matches(gpu_ctx, p, tables):
    """cutter -> components -> features -> matrix on the GPU against the fixture entry p (no oracle in the loop)"""
    import torch
    from metafast_amd import lib as L
    from metafast_amd import pipeline as P
    goods = [t.filter(KA["b"]) for t in tables]
    seqs = [gpu_ctx.build_unitigs(t, KA["b"], KA["l"]) for t in tables]
    parts_b, parts_o, nb, ns = [], [], 0, 0
    for sq in seqs:
        v = sq.device_view()
        parts_b.append(P.device_tensor(v["bases"], v["n_bases"], "cuda"))
        parts_o.append(P.device_tensor(v["offsets"], (v["n"] + 1) * 8, "cuda").view(torch.int64)[:-1] + nb)
        nb += v["n_bases"]; ns += v["n"]
    gpu_ctx.synchronize()
    allb = torch.cat(parts_b + [torch.zeros(64, dtype=torch.uint8, device="cuda")])
    allo = torch.cat(parts_o + [torch.tensor([nb], dtype=torch.int64, device="cuda")])
    torch.cuda.synchronize()
    cutter = gpu_ctx.count_device(allb.data_ptr(), allo.data_ptr(), ns, nb, KA["k"], KA["l"])
    ck, cc = cutter.export()
    assert (len(ck), _digest_table(ck, cc)) == (p["cutter_size"], p["cutter_sha256"])
    comps = gpu_ctx.cut_components(cutter, p["b1"], p["b2"])
    got = comps.export()
    assert [[a, w, t] for a, w, t, _ in got] == p["components"]
    assert _members_digest(got) == p["members_sha256"]
    vecs = np.array([gpu_ctx.features(comps, g, 0)[0] for g in goods])
    assert vecs.tolist() == p["vectors"]
    m = L.bray_curtis(vecs)
    assert np.abs(m - np.array(p["matrix"])).max() <= 1e-6 and m.tolist() == p["matrix"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["default", "split"])
def test_gpu_matches_fixture(gpu_ctx, ref_files, name):
    tables = []
    for f, s in zip(ref_files, KA["samples"]):
        t = gpu_ctx.count_reads([f], KA["k"])
        keys, cnts = t.export()
        assert (len(keys), _digest_table(keys, cnts)) == (s["n_distinct"], s["counts_sha256"])
        sq = gpu_ctx.build_unitigs(t, KA["b"], KA["l"])
        got = sq.export()
        assert (len(got), sum(len(x[0]) for x in got), _digest_seqs(got)) == (s["n_unitigs"], s["unitig_nt"], s["unitigs_sha256"])
        tables.append(t)
    _gpu_pipeline_matches(gpu_ctx, KA["pipelines"][name], tables)


def test_oracle_matches_branchy_fixture(oracle):
    from util import branchy_reads, emission_census
    for seed, e in KA["branchy"].items():
        b, o = branchy_reads(int(seed))
        keys, vals = oracle.Table().count_buffer(b, o, 31).export()
        assert (len(keys), _digest_table(keys, vals)) == (e["n_distinct"], e["counts_sha256"])
        g = oracle.Table()
        for kk, vv in zip(keys[vals > 1].tolist(), vals[vals > 1].tolist()):
            g.add(kk, vv)
        seqs = oracle.build_unitigs(g, 31, 1, 100).all()
        long_enough = oracle.unitig_census()[1]
        once, twice = emission_census(seqs)
        assert [once, twice, long_enough // 2 - once - twice] == e["census"] and _digest_seqs(seqs) == e["unitigs_sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", ["7", "8", "9"])
def test_gpu_matches_branchy_fixture(gpu_ctx, seed):
    """counts and unitigs (with the 0 / 1 / 2-emission rule at work) against committed digests: no oracle in the loop"""
    from util import branchy_reads, emission_census, gpu_count
    e = KA["branchy"][seed]
    b, o = branchy_reads(int(seed))
    t = gpu_count(gpu_ctx, b, o, 31)
    keys, cnts = t.export()
    assert (len(keys), _digest_table(keys, cnts)) == (e["n_distinct"], e["counts_sha256"])
    got = gpu_ctx.build_unitigs(t, 1, 100).export()
    assert (len(got), _digest_seqs(got)) == (e["n_unitigs"], e["unitigs_sha256"])
    assert list(emission_census(got)) == e["census"][:2]


@pytest.mark.gpu
def test_gpu_matches_levels_fixture(gpu_ctx):
    """threshold levels up to 6 (components larger than b2 split again and again) against the committed answers"""
    from util import branchy_reads, gpu_count
    p = KA["pipelines"]["levels"]
    tables = [gpu_count(gpu_ctx, *branchy_reads(rs, genome_seed=p["genome_seed"], n=p["n_reads"]), 31) for rs in p["read_seeds"]]
    assert max(t for _, _, t in p["components"]) >= 3
    _gpu_pipeline_matches(gpu_ctx, p, tables)

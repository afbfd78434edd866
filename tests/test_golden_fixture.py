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


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["default", "split"])
def test_gpu_matches_fixture(gpu_ctx, ref_files, name):
    import torch
    from metafast_amd import lib as L
    from metafast_amd import pipeline as P
    p = KA["pipelines"][name]
    tables, goods, seqs = [], [], []
    for f, s in zip(ref_files, KA["samples"]):
        t = gpu_ctx.count_reads([f], KA["k"])
        keys, cnts = t.export()
        assert (len(keys), _digest_table(keys, cnts)) == (s["n_distinct"], s["counts_sha256"])
        sq = gpu_ctx.build_unitigs(t, KA["b"], KA["l"])
        got = sq.export()
        assert (len(got), sum(len(x[0]) for x in got), _digest_seqs(got)) == (s["n_unitigs"], s["unitig_nt"], s["unitigs_sha256"])
        tables.append(t); goods.append(t.filter(KA["b"])); seqs.append(sq)
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

"""Regenerates tests/golden/known_answers.json from the CPU oracle (which is itself pinned by the reference's golden
matrix, tests/test_oracle_golden.py).  Inputs: the reference's own test_data files copied under ref_test_data/.
    python tests/golden/make_golden.py
The fixture holds digests and small vectors only (sorted (k-mer,count) lists -> SHA-256; unitig multisets -> SHA-256;
component (size, weight, thr) lists; feature vectors; matrices)."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402


def canon_seq(s):
    rc = s[::-1].translate(str.maketrans("ACGT", "TGCA"))
    return min(s, rc)


def digest_table(keys, vals):
    return hashlib.sha256(np.asarray(keys, dtype="<u8").tobytes() + np.asarray(vals, dtype="<u2").tobytes()).hexdigest()


def digest_seqs(seqs):
    items = sorted(f"{canon_seq(s)} {a} {mn} {mx}" for s, a, mn, mx in seqs)
    return hashlib.sha256("\n".join(items).encode()).hexdigest()


def main():
    files = [os.path.join(HERE, "ref_test_data", f"meta_test_{i}.fa") for i in (1, 2, 3)]
    out = {"k": 31, "b": 1, "l": 100, "samples": [], "pipelines": {}}
    for f in files:
        t = O.Table().count_files([f], 31)
        keys, vals = t.export()
        g = O.Table()
        for kk, vv in zip(keys[vals > 1].tolist(), vals[vals > 1].tolist()):
            g.add(kk, vv)
        seqs = O.build_unitigs(g, 31, 1, 100).all()
        out["samples"].append(dict(file=os.path.basename(f), n_distinct=int(len(keys)), n_good=int((vals > 1).sum()),
                                   counts_sha256=digest_table(keys, vals), n_unitigs=len(seqs),
                                   unitig_nt=sum(len(s[0]) for s in seqs), unitigs_sha256=digest_seqs(seqs)))
    for name, (b1, b2) in {"default": (1000, 10000), "split": (50, 500)}.items():
        r = O.run_pipeline(files, b1=b1, b2=b2)
        comps = r["comps"].all()
        ck, cv = r["cutter"].export()
        out["pipelines"][name] = dict(
            b1=b1, b2=b2, cutter_size=len(r["cutter"]), cutter_sha256=digest_table(ck, cv),
            components=[[int(a), int(w), int(t)] for a, w, t, _ in comps],
            members_sha256=hashlib.sha256(b"".join(np.asarray(km, dtype="<u8").tobytes() for _, _, _, km in comps)).hexdigest(),
            vectors=r["vecs"].tolist(), matrix=[[float(x) for x in row] for row in r["matrix"]])
    # ---- branchy synthetic samples (tests/util.py: branchy_reads): the 0 / 1 / 2-emission rule and threshold levels >= 3
    sys.path.insert(0, os.path.dirname(HERE))
    from util import branchy_reads, emission_census
    import tempfile
    out["branchy"] = {}
    for seed in (7, 8, 9):
        b, o = branchy_reads(seed)
        keys, vals = O.Table().count_buffer(b, o, 31).export()
        g = O.Table()
        for kk, vv in zip(keys[vals > 1].tolist(), vals[vals > 1].tolist()):
            g.add(kk, vv)
        seqs = O.build_unitigs(g, 31, 1, 100).all()
        started, long_enough, emitted = O.unitig_census()
        once, twice = emission_census(seqs)
        out["branchy"][str(seed)] = dict(n_distinct=int(len(keys)), counts_sha256=digest_table(keys, vals), n_unitigs=len(seqs),
                                         unitigs_sha256=digest_seqs(seqs), census=[once, twice, long_enough // 2 - once - twice])
    td = tempfile.mkdtemp()
    lf = []
    for i, rs in enumerate((107, 117, 127, 137)):
        b, o = branchy_reads(rs, genome_seed=7, n=6000)
        f = os.path.join(td, f"s{i}.fa")
        with open(f, "wb") as fh:
            for j in range(len(o) - 1):
                fh.write(b">r\n" + b[int(o[j]):int(o[j + 1])].tobytes() + b"\n")
        lf.append(f)
    r = O.run_pipeline(lf, b1=100, b2=1000)
    comps = r["comps"].all()
    ck, cv = r["cutter"].export()
    out["pipelines"]["levels"] = dict(
        b1=100, b2=1000, read_seeds=[107, 117, 127, 137], genome_seed=7, n_reads=6000, cutter_size=len(r["cutter"]), cutter_sha256=digest_table(ck, cv),
        components=[[int(a), int(w), int(t)] for a, w, t, _ in comps],
        members_sha256=hashlib.sha256(b"".join(np.asarray(km, dtype="<u8").tobytes() for _, _, _, km in comps)).hexdigest(),
        vectors=r["vecs"].tolist(), matrix=[[float(x) for x in row] for row in r["matrix"]])
    with open(os.path.join(HERE, "known_answers.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote known_answers.json")


if __name__ == "__main__":
    main()

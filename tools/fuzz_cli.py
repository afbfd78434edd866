"""Differential fuzzing of the metafast.sh-compatible driver (every step through the reference's files) against the oracle's
whole pipeline: random communities of 2-4 samples that share genomes; the distance matrix must be identical.
python3 tools/fuzz_cli.py [seconds] [seed]"""
import os, subprocess, sys, tempfile, time, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
O.build()
AL = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGT", b"TGCA"):
    COMP[a] = b
td = tempfile.mkdtemp(prefix="mf_fuzz_cli_")
t_end = time.time() + budget
it = 0
while time.time() < t_end:
    it += 1
    pool = [AL[rng.integers(0, 4, size=int(rng.integers(2000, 20000)))] for _ in range(int(rng.integers(2, 6)))]
    ns = int(rng.integers(2, 5))
    files = []
    for s in range(ns):
        path = os.path.join(td, "s%d_%d.fa" % (it, s))
        with open(path, "wb") as f:
            for g in pool:
                cov = float(rng.choice([0, 3, 10, 30]))
                Lr = 100
                for i in range(int(cov * len(g) / Lr)):
                    st = int(rng.integers(0, len(g) - Lr + 1))
                    r = g[st:st + Lr].copy()
                    if rng.integers(0, 2):
                        r = COMP[r[::-1]]
                    m = rng.random(Lr) < 0.005
                    r[m] = AL[rng.integers(0, 4, size=int(m.sum()))]
                    f.write(b">r\n" + r.tobytes() + b"\n")
        if os.path.getsize(path) == 0:
            open(path, "wb").write(b">r\n" + pool[0][:150].tobytes() + b"\n")
        files.append(path)
    k = int(rng.choice([21, 25, 31])); b = int(rng.choice([0, 1, 2])); l = int(rng.choice([40, 100])); b1 = int(rng.choice([10, 200])); b2 = int(rng.choice([2000, 10000]))
    wd = os.path.join(td, "w%d" % it)
    r = subprocess.run([os.path.join(ROOT, "metafast.sh"), "-k", str(k), "-b", str(b), "-l", str(l), "-b1", str(b1), "-b2", str(b2), "-i", *files, "-w", wd,
                        "--output-format", "%s"], capture_output=True, text=True, cwd=td)
    ref = O.run_pipeline(files, k=k, b=b, l=l, b1=b1, b2=b2)
    tag = f"it={it} k={k} b={b} l={l} b1={b1} b2={b2} samples={ns}"
    if ref["matrix"] is None or len(ref["comps"]) == 0:
        assert r.returncode == 1, "expected failure (no components / sequences) " + tag + r.stderr[-300:]
    else:
        if r.returncode != 0:
            # two samples without any feature give 0/0 = NaN; the reference's clustering then throws this very error
            # (src/algo/FullHeatMap.java:264-266) after the matrix in the original order has been written
            assert np.isnan(ref["matrix"]).any() and "Wrong minDist index" in r.stderr, tag + r.stderr[-500:]
        mats = [p for p in os.listdir(os.path.join(wd, "matrices")) if p.endswith("_original_order.txt")]
        rows = open(os.path.join(wd, "matrices", mats[0])).read().splitlines()[1:]
        got = np.array([[float(x) for x in row.split("\t")[1:]] for row in rows])
        assert np.array_equal(got, ref["matrix"], equal_nan=True), tag + f"\n{got}\n{ref['matrix']}"
    shutil.rmtree(wd, ignore_errors=True)
    for p in files:
        os.remove(p)
    if it % 5 == 0:
        print("ok", tag, "components", len(ref["comps"]), flush=True)
print("fuzz_cli done:", it, "communities, no mismatch")

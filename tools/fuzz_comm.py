"""Differential fuzzing of the sharded component cutter behind the C-ABI (mf_cut_components_sharded over the local communicator: W ranks as
threads of this process on one GPU) against the CPU oracle: random communities (several samples of one genome with repeats and errors), k, cut,
minimal length, component window, number of ranks; every rank must end with the oracle's components (ComponentsBuilder.java:58-270 over the
cutter table of ALL samples' unitigs, ComponentCutterMain.java:78-114).  python3 tools/fuzz_comm.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle as O
import test_distributed_gpu as T

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
O.build()
AL = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGT", b"TGCA"):
    COMP[a] = b


def community():
    glen = int(rng.integers(2000, 50000))
    g = AL[rng.integers(0, 4, size=glen)]
    for _ in range(int(rng.integers(0, 4))):
        L0 = int(rng.integers(40, min(500, glen // 2)))
        s, d = int(rng.integers(0, glen - L0)), int(rng.integers(0, glen - L0))
        g[d:d + L0] = g[s:s + L0]
    out = []
    for _ in range(int(rng.integers(1, 6))):
        n = int(rng.integers(200, 5000)); rl = int(rng.integers(60, 160)); err = float(rng.choice([0.0, 0.003, 0.01]))
        starts = rng.integers(0, glen - rl + 1, size=n)
        reads = np.stack([g[s:s + rl] for s in starts])
        flip = rng.integers(0, 2, size=n).astype(bool)
        reads[flip] = COMP[reads[flip][:, ::-1]]
        if err:
            m = rng.random(reads.shape) < err
            reads[m] = AL[rng.integers(0, 4, size=int(m.sum()))]
        out.append((reads.reshape(-1).copy(), np.arange(n + 1, dtype=np.uint64) * np.uint64(rl)))
    return out


t_end = time.time() + budget
it = 0
while time.time() < t_end:
    it += 1
    inputs = community()
    k = int(rng.choice([20, 21, 23, 25, 27, 31]))
    b = int(rng.choice([0, 1, 2])); l = int(rng.choice([k, 40, 60, 100]))
    b1 = int(rng.choice([1, 10, 50])); b2 = b1 + int(rng.choice([20, 300, 5000]))
    W = int(rng.choice([2, 4, 8]))
    tag = f"it={it} seed={seed} samples={len(inputs)} k={k} b={b} l={l} b1={b1} b2={b2} ranks={W}"
    want = T._oracle_components(O, inputs, b1, b2, k=k, b=b, l=l)
    res = T._virtual_ranks(W, inputs, b1, b2, k=k, b=b, l=l, one_call=True)
    for r, (comps, info) in enumerate(res):
        assert not isinstance(comps, str), f"rank {r} gave up: {info} " + tag
        assert [(a, w, t) for a, w, t, _ in comps] == [(a, w, t) for a, w, t, _ in want], f"components (rank {r}) " + tag
        assert all(np.array_equal(np.sort(g[3]), np.sort(x[3])) for g, x in zip(comps, want)), f"members (rank {r}) " + tag
    if it % 10 == 0:
        print("ok", tag, "components", len(want), flush=True)
print("fuzz_comm done:", it, "communities, no mismatch")

"""Differential fuzzing of the whole path against the CPU oracle (test infrastructure): random genomes with repeats,
coverage, error rates, read lengths, k, thresholds and component windows; counts, unitigs (strand-normalised multisets
with weights), components (size, weight, thr, member sets) and feature vectors must be identical.
python3 tools/fuzz.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from metafast_amd import lib as L
from oracle import oracle as O
from util import canon_seq, to_device

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
O.build()
ctx = L.Context(0, stream=torch.cuda.current_stream())
for kv in filter(None, os.environ.get("MF_OPTIONS", "").split(",")):      # e.g. MF_OPTIONS=ut_double_after=1: the long-path route of the unitigs on every case
    ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
AL = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGT", b"TGCA"):
    COMP[a] = b


def make_reads():
    glen = int(rng.integers(200, 60000 * int(os.environ.get("MF_FUZZ_SCALE", "1"))))
    g = AL[rng.integers(0, 4, size=glen)]
    for _ in range(int(rng.integers(0, 4))):                      # repeats: copies of a stretch elsewhere (branches)
        L0 = int(rng.integers(20, min(400, glen // 2)))
        s, d = int(rng.integers(0, glen - L0)), int(rng.integers(0, glen - L0))
        g[d:d + L0] = g[s:s + L0]
    if rng.random() < 0.3:                                        # low-complexity stretch
        L0 = int(rng.integers(10, 80)); d = int(rng.integers(0, glen - L0)); g[d:d + L0] = AL[rng.integers(0, 4)]
    n = int(rng.integers(1, 4000 * int(os.environ.get("MF_FUZZ_SCALE", "1"))))
    lo = int(rng.integers(1, 120)); hi = lo + int(rng.integers(0, 200))
    lens = np.minimum(rng.integers(lo, hi + 1, size=n), glen)
    err = float(rng.choice([0.0, 0.002, 0.01, 0.05]))
    reads = []
    for Lr in lens:
        s = int(rng.integers(0, glen - Lr + 1))
        r = g[s:s + Lr].copy()
        if rng.integers(0, 2):
            r = COMP[r[::-1]]
        if err:
            m = rng.random(Lr) < err
            r[m] = AL[rng.integers(0, 4, size=int(m.sum()))]
        reads.append(r)
    off = np.zeros(n + 1, dtype=np.uint64); off[1:] = np.cumsum([len(r) for r in reads])
    return (np.concatenate(reads) if reads else np.zeros(0, np.uint8)).astype(np.uint8), off


t_end = time.time() + budget
it = 0
while time.time() < t_end:
    it += 1
    b, o = make_reads()
    k = int(rng.choice([1, 3, 8, 15, 16, 19, 20, 21, 24, 27, 30, 31]))
    min_len = int(rng.choice([0, 0, 30, 100]))
    thr = int(rng.choice([0, 1, 2]))
    l = int(rng.choice([k, k + 5, 60, 100]))
    b1 = int(rng.choice([1, 5, 50])); b2 = b1 + int(rng.choice([10, 200, 5000]))
    ctx.set_option("skm", int(rng.choice([1, 1, 0]))); ctx.set_option("skm_dyn", int(rng.choice([0, 1, 2])))
    ctx.set_option("part_target", int(rng.choice([1, 16, 128, 3072])))
    ctx.set_option("skm_slices", int(rng.choice([0, 0, 2, 4]))); ctx.set_option("skm_shared", int(rng.choice([0, 1, 2])))
    ctx.set_option("skm_dedupe", int(rng.choice([0, 1, 5, 5])))
    tag = f"it={it} k={k} reads={len(o)-1} bases={len(b)} min_len={min_len} thr={thr} l={l} b1={b1} b2={b2}"
    tb, to = to_device(b, o)
    gt = ctx.count_device(tb.data_ptr(), to.data_ptr(), len(o) - 1, int(o[-1]), k, min_len)
    ot = O.Table().count_buffer(b, o, k, min_len)
    gk, gc = gt.export(); ok, ov = ot.export()
    assert np.array_equal(gk, ok) and np.array_equal(gc.astype(np.int32), ov), "counts " + tag
    ga, n_all = ctx.count_device_above(tb.data_ptr(), to.data_ptr(), len(o) - 1, int(o[-1]), k, thr, min_len)
    ak, ac = ga.export(); okt, ovt = ot.export(thr)
    assert n_all == len(ok) and np.array_equal(ak, okt) and np.array_equal(ac.astype(np.int32), ovt), "cut " + tag
    gs = ctx.build_unitigs(ga, thr, l)
    os_ = O.build_unitigs(ot, k, thr, l)
    norm = lambda seqs: sorted((canon_seq(s), a, mn, mx) for s, a, mn, mx in seqs)
    gsl, osl = gs.export(), os_.all()
    assert norm(gsl) == norm(osl), "unitigs " + tag
    if gsl:
        sv = gs.device_view()
        cut_g = ctx.count_device(sv["bases"], sv["offsets"], sv["n"], sv["n_bases"], k, l)
        sb, so = np.frombuffer("".join(s[0] for s in osl).encode(), dtype=np.uint8), np.zeros(len(osl) + 1, dtype=np.uint64)
        so[1:] = np.cumsum([len(s[0]) for s in osl])
        cut_o = O.Table().count_buffer(sb.copy(), so, k, l)
        ck, cc = cut_g.export(); cok, cov = cut_o.export()
        assert np.array_equal(ck, cok) and np.array_equal(cc.astype(np.int32), cov), "cutter " + tag
        gcomp = ctx.cut_components(cut_g, b1, b2); ocomp = O.cut_components(cut_o, k, b1, b2)
        key = lambda c: (c[2], -c[1], -c[0], int(c[3][0]) if len(c[3]) else 0)
        G = sorted(((s_, w, t_, tuple(km.tolist())) for s_, w, t_, km in gcomp.export()), key=key)
        R = sorted(((s_, w, t_, tuple(km.tolist())) for s_, w, t_, km in ocomp.all()), key=key)
        assert G == R, "components " + tag
        if len(ocomp) and len(okt) <= 60000:
            # features see the sample's counts > thr only (what the .kmers.bin holds)
            mt = O.Table()
            for kk, vv in zip(okt.tolist(), ovt.tolist()):
                mt.add(int(kk), int(vv))
            ovv, ob = ocomp.features(mt, 0)
            gv, gb = ctx.features(gcomp, ga, 0)
            gmap = {tuple(km.tolist()): (int(v), float(x)) for (s_, w, t_, km), v, x in zip(gcomp.export(), gv, gb)}
            omap = {tuple(km.tolist()): (int(v), float(x)) for (s_, w, t_, km), v, x in zip(ocomp.all(), ovv, ob)}
            assert gmap == omap, "features " + tag
        gcomp.close(); cut_g.close()
    gs.close(); ga.close(); gt.close()
    if it % 20 == 0:
        print("ok", tag, flush=True)
print("fuzz done:", it, "cases, no mismatch")

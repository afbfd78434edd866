"""Differential fuzzing of the k = 32 .. 63 extension (mf_wide.hip + mf_wgraph.hip; NO-REFERENCE EXTENSION: the reference rejects k > 31) against
the 128-bit build of the CPU oracle's own text (oracle/mf_oracle_wide.c): random genomes with repeats and low-complexity stretches, coverage,
error rates, ragged read lengths, k, thresholds, component windows, forced numbers of passes and bucket routes of the wide count; cut tables,
unitigs (strand-normalised multisets with weights), cutter tables, components and feature vectors must be identical.
python3 tools/fuzz_wide.py [seconds] [seed]"""
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
AL = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGT", b"TGCA"):
    COMP[a] = b


def make_reads():
    glen = int(rng.integers(300, 40000))
    g = AL[rng.integers(0, 4, size=glen)]
    for _ in range(int(rng.integers(0, 4))):                      # repeats: copies of a stretch elsewhere (branches)
        L0 = int(rng.integers(40, min(600, glen // 2)))
        s, d = int(rng.integers(0, glen - L0)), int(rng.integers(0, glen - L0))
        g[d:d + L0] = g[s:s + L0]
    if rng.random() < 0.3:                                        # low-complexity stretches: homopolymer / (AT)n -- palindromes at even k
        L0 = int(rng.integers(40, 160)); d = int(rng.integers(0, glen - L0))
        g[d:d + L0] = AL[rng.integers(0, 4)] if rng.random() < 0.5 else np.tile(np.frombuffer(b"AT", dtype=np.uint8), L0)[:L0]
    n = int(rng.integers(1, 3000))
    lo = int(rng.integers(20, 160)); hi = lo + int(rng.integers(0, 200))
    lens = np.minimum(rng.integers(lo, hi + 1, size=n), glen)
    err = float(rng.choice([0.0, 0.002, 0.01]))
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
    return np.concatenate(reads).astype(np.uint8), off


def same_table(wt, keys, vals):
    hi, lo, cnt = wt.export()
    return np.array_equal(hi, keys["hi"]) and np.array_equal(lo, keys["lo"]) and np.array_equal(cnt.astype(np.int32), vals)


norm = lambda seqs: sorted((canon_seq(s), a, mn, mx) for s, a, mn, mx in seqs)
t_end = time.time() + budget
it = 0
while time.time() < t_end:
    it += 1
    b, o = make_reads()
    k = int(rng.integers(32, 64))
    thr = int(rng.choice([0, 1, 2]))
    l = int(rng.choice([k, k + 7, 100, 150]))
    b1 = int(rng.choice([1, 5, 50])); b2 = b1 + int(rng.choice([10, 200, 5000]))
    ctx.set_option("wide_passes", int(rng.choice([0, 0, 3, 16])))
    ctx.set_option("wide_big_bucket", int(rng.choice([256, 256, 4, 32]))); ctx.set_option("wide_distinct", int(rng.choice([1280, 1280, 8])))
    ctx.set_option("ut_double_after", int(rng.choice([4, 4, 1])))
    ctx.set_option("wide_skm", int(rng.choice([1, 2, 2, 0]))); ctx.set_option("wide_skm_min", 1); ctx.set_option("wide_skm_lead", int(rng.choice([1, 1, 2, 0]))); ctx.set_option("wide_skm_merge", int(rng.choice([1, 1, 0]))); ctx.set_option("wide_skm_pack", int(rng.choice([1, 1, 2, 0]))); ctx.set_option("wide_skm_fine", int(rng.choice([1, 8, 32]))); ctx.set_option("wide_skm_unit", int(rng.choice([4000, 4000, 300, 64])))
    tag = f"it={it} seed={seed} k={k} reads={len(o)-1} bases={len(b)} thr={thr} l={l} b1={b1} b2={b2}"
    tb, to = to_device(b, o)
    ga, n_all = ctx.count_wide_above(tb.data_ptr(), to.data_ptr(), len(o) - 1, int(o[-1]), k, thr)
    ot = O.WTable().count_buffer(b, o, k)
    og = ot.good(thr)
    assert n_all == len(ot) and same_table(ga, *og.export()), "cut table " + tag
    gs = ctx.build_unitigs_wide(ga, thr, l)
    os_ = O.wide_build_unitigs(og, k, thr, l)
    gsl, osl = gs.export(), os_.all()
    assert norm(gsl) == norm(osl), "unitigs " + tag
    if gsl:
        sv = gs.device_view()
        cut_g = ctx.count_wide_table(sv["bases"], sv["offsets"], sv["n"], sv["n_bases"], k, l)
        cut_o = O.WTable().count_seqs(os_, k, l)
        assert same_table(cut_g, *cut_o.export()), "cutter " + tag
        gcomp = ctx.cut_components_wide(cut_g, b1, b2); ocomp = O.wide_cut_components(cut_o, k, b1, b2)
        ge, oe = gcomp.export(), ocomp.all()
        assert [(int(a), int(w), int(t)) for a, w, t in zip(ge["sizes"], ge["weights"], ge["thr"])] == [(a, w, t) for a, w, t, _ in oe], "components " + tag
        at = 0
        for n_, _, _, km in oe:
            assert np.array_equal(ge["hi"][at:at + n_], km["hi"]) and np.array_equal(ge["lo"][at:at + n_], km["lo"]), "members " + tag
            at += n_
        if len(oe):
            gv, gb = ctx.features_wide(gcomp, ga, 0)
            ov, ob = ocomp.features(og, 0)
            assert np.array_equal(gv, ov) and np.array_equal(gb, ob), "features " + tag
        gcomp.close(); cut_g.close()
    gs.close(); ga.close()
    if it % 20 == 0:
        print("ok", tag, flush=True)
print("fuzz_wide done:", it, "cases, no mismatch")

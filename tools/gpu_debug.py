"""Staged GPU bring-up: each stage runs in its own process under a short timeout (see tools/gpu_debug.sh)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

stage = sys.argv[1]
t0 = time.time()
def log(*a):
    print(f"[{stage} +{time.time()-t0:6.2f}s]", *a, flush=True)

import torch
log("torch imported, cuda:", torch.cuda.is_available(), torch.cuda.get_device_name(0) if torch.cuda.is_available() else None)
from metafast_amd import lib as L
from oracle import oracle as O
from util import gpu_count, pack_reads, genome_reads, random_reads
ctx = L.Context(0, stream=torch.cuda.current_stream())
ctx.set_option("verbose", 2)
log("ctx created")

def check(bases, off, k, min_len=0):
    t = gpu_count(ctx, bases, off, k, min_len)
    log("count_device returned; distinct", len(t), "occ", t.occurrences())
    gk, gc = t.export()
    ok, ov = O.Table().count_buffer(bases, off, k, min_len).export()
    same = len(gk) == len(ok) and np.array_equal(gk, ok) and np.array_equal(gc.astype(np.int32), ov)
    log("PARITY", "OK" if same else f"MISMATCH gpu={len(gk)} oracle={len(ok)}")
    if not same and len(gk) and len(ok):
        sg, so = set(gk.tolist()), set(ok.tolist())
        log("  only gpu:", len(sg - so), "only oracle:", len(so - sg), "sum gpu", int(gc.sum()), "sum oracle", int(ov.sum()))
    return same

REF = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_test_data")
if stage == "synth":
    n, rl = 2000, 150
    tb = torch.zeros(n * rl + 64, dtype=torch.uint8, device="cuda")
    to = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    ctx.synth_reads_device(1234, 1, 10, n, rl, 20000, tb.data_ptr(), to.data_ptr())
    hb, ho = L.synth_reads_host(1234, 1, 10, n, rl, 20000)
    log("synth match:", np.array_equal(tb[: n * rl].cpu().numpy(), hb), np.array_equal(to.cpu().numpy().astype(np.uint64), ho))
elif stage == "tiny":
    b, o = pack_reads(["ACGTACGTACGTACGTACGTACGTACGTACGTACGTA", "TTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTT"])
    check(b, o, 31); check(b, o, 5)
elif stage in ("ref_direct", "ref_staged", "ref_l1_3", "ref_2lvl", "ref_2lvl_direct"):
    b, o = O.read_file(os.path.join(REF, "meta_test_1.fa"))
    if stage == "ref_direct": ctx.set_option("scatter_staged", 0)
    if stage == "ref_l1_3": ctx.set_option("l1_bits", 3); ctx.set_option("l2_bits", 0)
    if stage == "ref_2lvl": ctx.set_option("l1_bits", 3); ctx.set_option("l2_bits", 4); ctx.set_option("l1_blocks", 2)
    if stage == "ref_2lvl_direct": ctx.set_option("l1_bits", 3); ctx.set_option("l2_bits", 4); ctx.set_option("scatter_staged", 0)
    check(b, o, 31)
elif stage == "medium":
    rng = np.random.default_rng(11)
    b, o = genome_reads(rng, 500_000, 50_000, 150, err=0.005)
    check(b, o, 31)
    ctx.set_option("part_target", 256)
    check(b, o, 31)
elif stage == "ragged":
    rng = np.random.default_rng(3)
    b, o = random_reads(rng, 3000, 0, 90)
    for k in (1, 5, 16, 31):
        check(b, o, k); check(b, o, k, 40)
elif stage == "table_ops":
    b, o = O.read_file(os.path.join(REF, "meta_test_3.fa"))
    t = gpu_count(ctx, b, o, 31)
    keys, cnts = t.export()
    log("lookup", t.lookup(keys[:5]).tolist(), cnts[:5].tolist(), t.lookup(np.array([1, 2], dtype=np.uint64)).tolist())
    f = t.filter(1); log("filter", len(f), int((cnts > 1).sum()))
    log("stats", t.stats())
log("stage done")

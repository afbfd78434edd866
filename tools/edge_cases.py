import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from metafast_amd import lib as L
from oracle import oracle as O
from util import gpu_count, pack_reads
ctx = L.Context(0, stream=torch.cuda.current_stream())
rng = np.random.default_rng(1)
def check(b, o, k, min_len=0, tag=""):
    t = gpu_count(ctx, b, o, k, min_len)
    gk, gc = t.export()
    ok, ov = O.Table().count_buffer(b, o, k, min_len).export()
    same = len(gk) == len(ok) and np.array_equal(gk, ok) and np.array_equal(gc.astype(np.int32), ov)
    print(tag, k, "distinct", len(gk), "records", t.records(), "OK" if same else "MISMATCH", flush=True)
    assert same
al = np.frombuffer(b"ACGT", dtype=np.uint8)
# one long read (a chromosome-like sequence), plus short ones around it
long_read = al[rng.integers(0, 4, size=20_000_000)]
b = np.concatenate([al[rng.integers(0, 4, size=70)], long_read, al[rng.integers(0, 4, size=31)], al[rng.integers(0, 4, size=30)]]).astype(np.uint8)
o = np.array([0, 70, 70 + len(long_read), 70 + len(long_read) + 31, len(b)], dtype=np.uint64)
for k in (31, 21, 15):
    check(b, o, k, tag="long read")
check(b, o, 31, min_len=100, tag="long read min_len")
# reads of exactly k, k-1, k+1 bases; a single read; tandem repeats across a word boundary
for k in (20, 31):
    reads = ["".join("ACGT"[c] for c in rng.integers(0, 4, size=n)) for n in [k, k - 1, k + 1] * 50]
    bb, oo = pack_reads(reads); check(bb, oo, k, tag="len k")
    bb, oo = pack_reads(["ACGTTGCATG" * 30]); check(bb, oo, k, tag="single tandem")
    bb, oo = pack_reads(["A" * 200 + "C" * 200 + "AC" * 150]); check(bb, oo, k, tag="low complexity")
print("all ok")

import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from metafast_amd import lib as L
from oracle import oracle as O
from util import random_reads
ctx = L.Context(0, stream=torch.cuda.current_stream())
ctx.set_option("verbose", 1)
rng = np.random.default_rng(5)
# a genome with coverage so that partitions hold many records
g = rng.integers(0, 4, 200000)
reads = []
for i in range(60000):
    s = rng.integers(0, len(g) - 150); reads.append(g[s:s + 150])
b = np.frombuffer(b"ACGT", dtype=np.uint8)[np.concatenate(reads)]
off = np.arange(0, len(reads) * 150 + 1, 150).astype(np.uint64)
for opts in ({}, {"part_target": 1024}, {"part_target": 256}, {"skm_batches": 1}, {"part_target": 16384}):
    for k_, v in {"part_target": 6144, "skm_batches": 0}.items(): ctx.set_option(k_, v)
    for k_, v in opts.items(): ctx.set_option(k_, v)
    db = torch.from_numpy(np.concatenate([b, np.zeros(64, np.uint8)])).cuda(); do = torch.from_numpy(off.astype(np.int64)).cuda()
    t = ctx.count_device(db.data_ptr(), do.data_ptr(), len(reads), len(b), 31, 0)
    gk, gc = t.export(-1)
    ok, ov = O.Table().count_buffer(b, off, 31, 0).export(-1)
    same_keys = len(gk) == len(ok) and np.array_equal(gk, ok)
    d = gc.astype(np.int64) - ov.astype(np.int64) if same_keys else None
    print(opts, "keys", same_keys, len(gk), len(ok), "count diffs", None if d is None else (int((d != 0).sum()), int(d.sum()), int(d.min()), int(d.max())), flush=True)

import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from metafast_amd import lib as L, pipeline as P
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
ctx = L.Context(0, stream=torch.cuda.current_stream())
for kv in filter(None, os.environ.get("MF_OPTIONS", "").split(",")):
    name, val = kv.split("="); ctx.set_option(name, int(val))
bases = torch.zeros(n * 150 + 64, dtype=torch.uint8, device="cuda"); offs = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
ctx.synth_reads_device(bench.SEED, 0, 0, n, 150, 1_000_000, bases.data_ptr(), offs.data_ptr(), 82)
torch.cuda.synchronize()
fa = "/tmp/e2e_probe.fa"
size = bench._write_fasta(bases, n, 150, fa)
del bases, offs
for rep in range(3):
    tm = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = P.run_samples(ctx, [(fa,)], k=31, timings=tm)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("rep", rep, round(dt, 4), {k: round(v * 1e3, 1) for k, v in tm.items()}, flush=True)
    for x in r["goods"] + r["seqss"] + [r["cutter"], r["comps"]]: x.close()
os.remove(fa)

"""Counting stage on LOW-coverage data (random reads: nearly every k-mer distinct): python3 tools/prof_lowcov.py [reads]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metafast_amd import lib as L
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
rl = 150
ctx = L.Context(0, stream=torch.cuda.current_stream())
ctx.set_option("verbose", 1); ctx.set_option("profile", 1)
g = torch.Generator(device="cuda"); g.manual_seed(1)
codes = torch.randint(0, 4, (n_reads * rl + 64,), device="cuda", generator=g, dtype=torch.uint8)
lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device="cuda")
bases = lut[codes.long()] if n_reads <= 2_000_000 else torch.empty_like(codes)
if n_reads > 2_000_000:
    for i in range(0, codes.numel(), 1 << 28):
        bases[i:i + (1 << 28)] = lut[codes[i:i + (1 << 28)].long()]
del codes
offsets = (torch.arange(n_reads + 1, device="cuda", dtype=torch.int64) * rl)
for it in range(2):
    ctx.reset_timers()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    t = ctx.count_device(bases.data_ptr(), offsets.data_ptr(), n_reads, n_reads * rl, 31, 0)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("distinct", len(t), "occ", t.occurrences(), "records", t.records(), f"{dt*1e3:.1f} ms", flush=True)
    print({k: round(v[1], 1) for k, v in sorted(ctx.kernel_report().items(), key=lambda kv: -kv[1][1])[:8]})
    t.close()

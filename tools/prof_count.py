"""Counting stage only on a synthetic sample (for rocprofv3 --pmc passes): python3 tools/prof_count.py [reads] [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metafast_amd import lib as L
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rl = 150
ctx = L.Context(0, stream=torch.cuda.current_stream())
if os.environ.get("MF_VERBOSE"): ctx.set_option("verbose", int(os.environ["MF_VERBOSE"]))
bases = torch.zeros(n_reads * rl + 64, dtype=torch.uint8, device="cuda")
offsets = torch.zeros(n_reads + 1, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
ctx.synth_reads_device(0x4D45544146415354, 0, 0, n_reads, rl, 1_000_000, bases.data_ptr(), offsets.data_ptr())
for _ in range(iters):
    t = ctx.count_device(bases.data_ptr(), offsets.data_ptr(), n_reads, n_reads * rl, int(os.environ.get("MF_K", "31")), 0)
    torch.cuda.synchronize()
    print("distinct", len(t), "occ", t.occurrences(), flush=True)
    t.close()
if os.environ.get("MF_VERBOSE"):
    ctx.set_option("profile", 1)
    t = ctx.count_device(bases.data_ptr(), offsets.data_ptr(), n_reads, n_reads * rl, int(os.environ.get("MF_K", "31")), 0)
    torch.cuda.synchronize()
    print({k: (v[0], round(v[1], 1)) for k, v in sorted(ctx.kernel_report().items(), key=lambda kv: -kv[1][1])[:10]})

"""Unitig stage only on the bench sample, kernel timers: python3 tools/prof_unitigs.py [reads]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metafast_amd import lib as L
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
rl = 150
ctx = L.Context(0, stream=torch.cuda.current_stream())
bases = torch.zeros(n_reads * rl + 64, dtype=torch.uint8, device="cuda")
offsets = torch.zeros(n_reads + 1, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
ctx.synth_reads_device(0x4D45544146415354, 0, 0, n_reads, rl, 1_000_000, bases.data_ptr(), offsets.data_ptr())
good, nd = ctx.count_device_above(bases.data_ptr(), offsets.data_ptr(), n_reads, n_reads * rl, 31, 1)
ctx.set_option("profile", 1)
for rep_no in range(2):
    ctx.reset_timers()
    s = ctx.build_unitigs(good, 1, 100)
    n = len(s); s.close()
    rep = ctx.kernel_report()
    print("unitigs", n, {k: round(v[1], 2) for k, v in rep.items() if k.startswith("k_ut")}, flush=True)

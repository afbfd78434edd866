"""Ablation of k_l1_scatter (diagnostic): python3 tools/ablate.py [reads]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metafast_amd import lib as L
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
rl = 150
ctx = L.Context(0, stream=torch.cuda.current_stream())
ctx.set_option("profile", 1)
bases = torch.zeros(n_reads * rl + 64, dtype=torch.uint8, device="cuda")
offsets = torch.zeros(n_reads + 1, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
ctx.synth_reads_device(0x4D45544146415354, 0, 0, n_reads, rl, 1_000_000, bases.data_ptr(), offsets.data_ptr())
for ab in (0, 3, 4, 5, 0):
    ctx.set_option("ablate", ab)
    ctx.reset_timers()
    try:
        t = ctx.count_device(bases.data_ptr(), offsets.data_ptr(), n_reads, n_reads * rl, 31, 0)
        t.close()
    except Exception as e:
        print("ablate", ab, "error (expected for 1/2):", str(e)[:80])
    rep = ctx.kernel_report()
    print("ablate", ab, {k: round(v[1], 2) for k, v in rep.items() if k.startswith("k_skm") or k in ("k_l1_hist", "k_l1_scatter", "k_split", "k_count")}, flush=True)

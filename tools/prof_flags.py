"""k_ut_flags ablations on one synthetic sample: python3 tools/prof_flags.py [reads]"""
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
good, n_all = ctx.count_device_above(bases.data_ptr(), offsets.data_ptr(), n_reads, n_reads * rl, 31, 1)
ctx.set_option("profile", 1)
for name, opts in (("global index only", {"nbr_global": 1}), ("partition-local", {}), ("no remote probes", {"ablate": 1}), ("no local probes", {"ablate": 2}),
                   ("no LDS build, no local probes", {"ablate": 6}), ("no probes at all", {"ablate": 7})):
    ctx.set_option("nbr_global", 0); ctx.set_option("ablate", 0)
    for k_, v in opts.items(): ctx.set_option(k_, v)
    for rep in range(2):
        ctx.reset_timers()
        try:
            s = ctx.build_unitigs(good, 1, 100)
            s.close()
        except Exception as e:
            print("   (", str(e)[:60], ")")
    print(f"{name:32s} k_ut_flags {ctx.kernel_report().get('k_ut_flags', (0, 0, 0))[1]:.2f} ms", flush=True)

"""Counting stage under forced partition plans, on reads of a chosen sequencing depth:
    python3 tools/plan_sweep.py <reads> <genome_scale_bp | random> [auto | l1,l2 ...]
(genome_scale 1000000 = the benchmark's 83-fold depth at 100 M reads, 16000000 = 5-fold, `random` = every k-mer distinct).
One line per plan: time of the call and of the counting kernels (HIP events)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metafast_amd import lib as L

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
scale = sys.argv[2] if len(sys.argv) > 2 else "1000000"
plans = sys.argv[3:] or ["auto"]
rl, k = 150, int(os.environ.get("K", "31"))
ctx = L.Context(0, stream=torch.cuda.current_stream())
ctx.set_option("profile", 1)
if os.environ.get("VERBOSE"):
    ctx.set_option("verbose", 1)
for kv in filter(None, os.environ.get("MF_OPTIONS", "").split(",")):
    name, val = kv.split("=")
    ctx.set_option(name, int(val))
bases = torch.zeros(n_reads * rl + 64, dtype=torch.uint8, device="cuda")
offsets = torch.zeros(n_reads + 1, dtype=torch.int64, device="cuda")
if scale == "random":
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device="cuda")
    for i in range(0, n_reads * rl, 1 << 28):
        j = min(n_reads * rl, i + (1 << 28))
        bases[i:j] = lut[torch.randint(0, 4, (j - i,), device="cuda", generator=g, dtype=torch.uint8).long()]
    offsets.copy_(torch.arange(n_reads + 1, device="cuda", dtype=torch.int64) * rl)
else:
    ctx.synth_reads_device(0x4D45544146415354, 0, 0, n_reads, rl, int(scale), bases.data_ptr(), offsets.data_ptr())
torch.cuda.synchronize()
for plan in plans:
    if plan == "auto":
        ctx.set_option("l1_bits", -1); ctx.set_option("l2_bits", -1)
    else:
        a, b = plan.split(",")
        ctx.set_option("l1_bits", int(a)); ctx.set_option("l2_bits", int(b))
    for it in range(2):
        ctx.reset_timers()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        try:
            t, n_all = ctx.count_device_above(bases.data_ptr(), offsets.data_ptr(), n_reads, n_reads * rl, k, 1)
        except L.MetafastError as e:
            print(plan, "failed:", e, flush=True)
            break
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if it == 1:
            rep = {kk: round(v[1], 1) for kk, v in sorted(ctx.kernel_report().items(), key=lambda kv: -kv[1][1])[:7]}
            print(f"plan {plan:>6}: {dt*1e3:8.1f} ms  kept {len(t)} of {n_all} distinct, occ {t.occurrences()}, records {t.records()[0]}  {rep}", flush=True)
        t.close()

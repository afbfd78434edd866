"""What the cutter + components cost on the union of N samples' unitigs (what EVERY rank repeats at N GPUs), measured
on one GPU: samples are processed one after the other, only their unitigs are kept.  python3 tools/sim_union.py [N] [reads]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metafast_amd import lib as L
from metafast_amd import pipeline as P
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
rl, k = 150, 31
ctx = L.Context(0, stream=torch.cuda.current_stream())
ctx.set_option("profile", 1)
if os.environ.get("MF_VERBOSE"): ctx.set_option("verbose", int(os.environ["MF_VERBOSE"]))
if os.environ.get("MF_PTL"): ctx.set_option("part_target_long", int(os.environ["MF_PTL"]))
bases = torch.zeros(n_reads * rl + 64, dtype=torch.uint8, device="cuda")
offsets = torch.zeros(n_reads + 1, dtype=torch.int64, device="cuda")
parts_b, parts_o = [], []
for s in range(N):
    torch.cuda.synchronize()
    ctx.synth_reads_device(0x4D45544146415354, s, 0, n_reads, rl, 1_000_000, bases.data_ptr(), offsets.data_ptr())
    t = ctx.count_device(bases.data_ptr(), offsets.data_ptr(), n_reads, n_reads * rl, k, 0)
    g = t.filter(1)
    seqs = ctx.build_unitigs(g, 1, 100)
    v = seqs.device_view()
    sb = P.device_tensor(v["bases"], v["n_bases"], "cuda").clone()
    so = P.device_tensor(v["offsets"], (v["n"] + 1) * 8, "cuda").view(torch.int64).clone()
    parts_b.append(sb); parts_o.append(so)
    print("sample", s, "unitigs", v["n"], "bases", v["n_bases"], flush=True)
    seqs.close(); g.close(); t.close()
del bases, offsets
for n in sorted({1, 2, 4, N}):
    nb = sum(int(p.numel()) for p in parts_b[:n]); ns = sum(int(p.numel()) - 1 for p in parts_o[:n])
    allb = torch.zeros(nb + 64, dtype=torch.uint8, device="cuda"); allo = torch.zeros(ns + 1, dtype=torch.int64, device="cuda")
    pb = po = 0
    for b, o in zip(parts_b[:n], parts_o[:n]):
        m = int(o.numel()) - 1
        allb[pb:pb + b.numel()] = b; allo[po:po + m] = o[:-1] + pb; pb += int(b.numel()); po += m
    allo[ns] = nb
    for rep in range(2):
        ctx.reset_timers()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        cutter = ctx.count_device(allb.data_ptr(), allo.data_ptr(), ns, nb, k, 100)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        comps = ctx.cut_components(cutter, 1000, 10000)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        nk = len(cutter); nc = len(comps)
        comps.close(); cutter.close()
    rep_k = ctx.kernel_report()
    print("   ", {kk: (v[0], round(v[1], 1)) for kk, v in sorted(rep_k.items(), key=lambda kv: -kv[1][1])[:12]})
    print(f"union of {n}: {ns} unitigs, {nb} bases, cutter k-mers {nk}, components {nc}: cutter count {1e3*(t1-t0):.1f} ms, components {1e3*(t2-t1):.1f} ms", flush=True)

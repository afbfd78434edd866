"""What the cutter + components cost at N GPUs, measured on one: the N samples are processed one after the other, only
their unitigs are kept; then (a) the replicated variant -- every rank builds the cutter table of the union and all
components -- and (b) the sharded one (pipeline.distributed_components) with N virtual ranks on this GPU, one computing at
a time (per-rank wall time between the exchanges; the exchanges themselves are memory copies here).
python3 tools/sim_union.py [N] [reads]"""
import os, sys, time, threading
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
for n in ([N] if os.environ.get('MF_SIM_SHARDED_ONLY') else sorted({1, 2, 4, N})):
    nb = sum(int(p.numel()) for p in parts_b[:n]); ns = sum(int(p.numel()) - 1 for p in parts_o[:n])
    allb = torch.zeros(nb + 64, dtype=torch.uint8, device="cuda"); allo = torch.zeros(ns + 1, dtype=torch.int64, device="cuda")
    pb = po = 0
    for b, o in zip(parts_b[:n], parts_o[:n]):
        m = int(o.numel()) - 1
        allb[pb:pb + b.numel()] = b; allo[po:po + m] = o[:-1] + pb; pb += int(b.numel()); po += m
    allo[ns] = nb
    for rep in range(0 if os.environ.get('MF_SIM_SHARDED_ONLY') else 2):
        ctx.reset_timers()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.set_option("union_samples", n)
        cutter = ctx.count_device(allb.data_ptr(), allo.data_ptr(), ns, nb, k, 100)
        ctx.set_option("union_samples", 0)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        comps = ctx.cut_components(cutter, 1000, 10000)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        nk = len(cutter); nc = len(comps)
        comps.close(); cutter.close()
    if os.environ.get('MF_SIM_SHARDED_ONLY'): continue
    rep_k = ctx.kernel_report()
    print("   ", {kk: (v[0], round(v[1], 1)) for kk, v in sorted(rep_k.items(), key=lambda kv: -kv[1][1])[:12]})
    print(f"union of {n}: {ns} unitigs, {nb} bases, cutter k-mers {nk}, components {nc}: cutter count {1e3*(t1-t0):.1f} ms, components {1e3*(t2-t1):.1f} ms", flush=True)

# ---- (b) sharded: N virtual ranks, every rank has all unitigs (the all-gather) and counts the k-mers it owns
ctx.trim(); torch.cuda.empty_cache()      # (the first context's arena still holds what the samples needed)
ctxs = [L.Context(0) for _ in range(N)]
comms = L.Comm.local(ctxs)                 # (round 6: the library's own communicator of threads; the ranks run side by side on the one GPU)
res = [None] * N
def work(rank):
    try:
        torch.cuda.set_device(0)
        c2, comm = ctxs[rank], comms[rank]
        c2.bind_thread()
        c2.set_option("profile", 1)
        for rep in range(2):
            c2.reset_timers()
            tm = {}
            torch.cuda.synchronize(); t0 = time.perf_counter()
            shard = c2.count_device_shard(allb.data_ptr(), allo.data_ptr(), ns, nb, k, 100, rank, N)
            c2.synchronize(); tm["cutter_count"] = time.perf_counter() - t0
            info = {}
            comm.reset_stats()
            t0 = time.perf_counter()
            comps = P.distributed_components(c2, comm, shard, k, 1000, 10000, info=info)
            c2.synchronize(); tm["components"] = time.perf_counter() - t0
            info["components"] = len(comps); info["exchange_ms"] = round(1e3 * comm.stats()["seconds"], 1)
            comps.close(); shard.close()
        res[rank] = (tm, info, c2.kernel_report())
    except BaseException as e:
        res[rank] = e
        raise
th = [threading.Thread(target=work, args=(r,)) for r in range(N)]
[t.start() for t in th]; [t.join() for t in th]
for r, x in enumerate(res):
    if isinstance(x, BaseException): raise x
    tm, info, rep_k = x
    print(f"rank {r}:", {kk: round(1e3 * v, 1) for kk, v in tm.items()}, "total", round(1e3 * sum(tm.values()), 1), "ms", info, flush=True)
tm, info, rep_k = res[0]
print("    rank 0 kernels:", {kk: (v[0], round(v[1], 1)) for kk, v in sorted(rep_k.items(), key=lambda kv: -kv[1][1])[:30]})
stages = sorted({kk for x in res for kk in x[0]})
print(f"    exchanges of rank 0: {res[0][1]['collectives']} collectives over {res[0][1]['levels']} levels, {res[0][1]['MB_received']} MB received, "
      f"{res[0][1]['exchange_ms']} ms inside them (barrier waits for the slowest rank included)")
print("sharded, max over ranks (the ranks run SIDE BY SIDE on one GPU here: not a per-rank cost):", {kk: round(1e3 * max(x[0][kk] for x in res), 1) for kk in stages})

"""The multi-GPU path end to end on ONE GPU: two ranks (fresh child processes) share cuda:0 and talk through gloo; every rank
must end with the same components and the full matrix, equal to the oracle's pipeline on both samples.  And the RCCL code
path itself at world size 1 (MF_FORCE_DIST=1: the collectives run although nobody else is there)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = r'''
import json, os, sys
sys.path.insert(0, os.environ["MF_ROOT"]); sys.path.insert(0, os.path.join(os.environ["MF_ROOT"], "tests"))
import numpy as np, torch, torch.distributed as dist
from util import branchy_reads, to_device
from metafast_amd import lib as L, pipeline as P
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = int(os.environ.get("MF_DEVICE", "0"))                      # (one GPU box: every rank on cuda:0; a multi-GPU box: LOCAL_RANK)
torch.cuda.set_device(dev)
dist.init_process_group(os.environ["MF_BACKEND"], rank=rank, world_size=world, **({"device_id": torch.device("cuda", dev)} if os.environ["MF_BACKEND"] == "nccl" else {}))
ctx = L.Context(dev, stream=torch.cuda.current_stream())
if os.environ.get("MF_TEST_FAIL_SHARD_RANK") == str(rank):        # (this rank's shard count fails: the ranks must fall back TOGETHER)
    def no_shard(*a, **k):
        raise L.MetafastError("injected: the shard count failed on this rank")
    ctx.count_device_shard = no_shard
spg = int(os.environ.get("MF_SPG", "1"))
def samples():
    for j in range(spg):
        b, o = branchy_reads(107 + 10 * (rank * spg + j), genome_seed=7, n=6000)
        db, do = to_device(b, o)
        yield db, do, len(o) - 1, len(b)
kk = int(os.environ.get("MF_K", "31"))
if kk >= 32:                                                      # NO-REFERENCE EXTENSION: a sample per rank, the replicated wide cutter
    r = P.run_samples_wide(ctx, samples(), k=kk, b=1, l=100, b1=100, b2=1000, device=torch.device("cuda", dev))
    e = r["comps"].export()
    off = e["offsets"].astype(int)
    out = dict(components=[[int(a), int(w), int(t)] for a, w, t in zip(e["sizes"], e["weights"], e["thr"])],
               members=[[(int(h) << 64) | int(l_) for h, l_ in zip(e["hi"][off[i]:off[i + 1]], e["lo"][off[i]:off[i + 1]])] for i in range(len(e["sizes"]))],
               vecs=r["vecs"].tolist(), matrix=r["matrix"].tolist(), comm={k: (v if isinstance(v, str) else float(v)) for k, v in r["comm"].items()})
    json.dump(out, open(os.path.join(os.environ["MF_OUT"], f"rank{rank}.json"), "w"))
    dist.destroy_process_group()
    sys.exit(0)
r = P.run_samples(ctx, samples(), k=31, b=1, l=100, b1=100, b2=1000, device=torch.device("cuda", dev))
comps = r["comps"].export()
out = dict(components=[[int(a), int(w), int(t)] for a, w, t, _ in comps], members=[[int(x) for x in km] for _, _, _, km in comps],
           vecs=r["vecs"].tolist(), matrix=r["matrix"].tolist(), comm={k: (v if isinstance(v, str) else float(v)) for k, v in r["comm"].items()})
json.dump(out, open(os.path.join(os.environ["MF_OUT"], f"rank{rank}.json"), "w"))
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _n_gpus():
    """GPUs of this box, without initialising one in the pytest process (device_count() does not, on this image)"""
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


def _run(world, backend, tmp_path, spg=1, extra_env=None, one_gpu_per_rank=False):
    port = _free_port()
    procs = []
    for rank in range(world):
        dev = str(rank) if one_gpu_per_rank else "0"
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=dev, MF_DEVICE=dev, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   MF_ROOT=ROOT, MF_OUT=str(tmp_path), MF_BACKEND=backend, MF_SPG=str(spg), HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-2000:]
    return [json.load(open(tmp_path / f"rank{r}.json")) for r in range(world)]


def _oracle_pipeline(oracle, tmp_path, seeds):
    from util import branchy_reads
    files = []
    for i, rs in enumerate(seeds):
        b, o = branchy_reads(rs, genome_seed=7, n=6000)
        f = tmp_path / f"s{i:02d}.fa"
        with open(f, "wb") as fh:
            for j in range(len(o) - 1):
                fh.write(b">r\n" + b[int(o[j]):int(o[j + 1])].tobytes() + b"\n")
        files.append(str(f))
    r = oracle.run_pipeline(files, b1=100, b2=1000)
    comps = r["comps"].all()
    return dict(components=[[int(a), int(w), int(t)] for a, w, t, _ in comps], members=[[int(x) for x in km] for _, _, _, km in comps],
                vecs=r["vecs"].tolist(), matrix=r["matrix"].tolist())


def _same(got, want):
    assert got["components"] == want["components"]
    assert [sorted(m) for m in got["members"]] == [sorted(m) for m in want["members"]]
    assert got["vecs"] == want["vecs"]
    assert np.abs(np.array(got["matrix"]) - np.array(want["matrix"])).max() <= 1e-6


def test_two_ranks_one_gpu_gloo(oracle, tmp_path):
    """pipeline.run_samples across two ranks (one sample each): both ranks get the oracle's components and 2 x 2 matrix"""
    res = _run(2, "gloo", tmp_path)
    want = _oracle_pipeline(oracle, tmp_path, [107, 117])
    for r in res:
        _same(r, want)


def test_two_ranks_fall_back_together_when_one_shard_count_fails(oracle, tmp_path):
    """rank 1 cannot count its shard: both ranks leave the sharded cutter at the same gather (DistAbort), build the whole
    cutter table each (the replicated path) and still end with the oracle's components and matrix"""
    res = _run(2, "gloo", tmp_path, extra_env={"MF_TEST_FAIL_SHARD_RANK": "1"})
    want = _oracle_pipeline(oracle, tmp_path, [107, 117])
    for r in res:
        _same(r, want)


def test_two_ranks_two_samples_each(oracle, tmp_path):
    """more samples than ranks (BASELINE config 5's shape): every rank takes two samples; 4 x 4 matrix in rank-major order"""
    res = _run(2, "gloo", tmp_path, spg=2)
    want = _oracle_pipeline(oracle, tmp_path, [107, 117, 127, 137])
    for r in res:
        _same(r, want)


@pytest.mark.parametrize("k,spg", [(47, 1), (63, 2)])
def test_two_ranks_wide_kmers_gloo(oracle, tmp_path, k, spg):
    """NO-REFERENCE EXTENSION over two ranks (BASELINE config 4: a sample per GPU at k = 63): unitigs gathered through the library's
    communicator, the wide cutter replicated on both ranks, the rows all-gathered -- both ranks end with the 128-bit oracle's components and matrix"""
    from util import branchy_reads
    res = _run(2, "gloo", tmp_path, spg=spg, extra_env={"MF_K": str(k)})
    samples = [branchy_reads(107 + 10 * i, genome_seed=7, n=6000) for i in range(2 * spg)]
    w = oracle.run_pipeline_wide(samples, k, b=1, l=100, b1=100, b2=1000)
    wc = w["comps"].all()
    assert len(wc) >= 3
    want = dict(components=[[a, ww, t] for a, ww, t, _ in wc], members=[oracle.w128_to_ints(km) for _, _, _, km in wc], vecs=w["vecs"].tolist(), matrix=w["matrix"].tolist())
    for r in res:
        _same(r, want)
        assert r["comm"]["kind"] == "external" and r["comm"]["collectives"] >= 5


def test_rccl_path_world1(oracle, tmp_path):
    """MF_FORCE_DIST=1: the exchanges go through RCCL ("nccl" backend) although the world has one rank"""
    res = _run(1, "nccl", tmp_path, extra_env={"MF_FORCE_DIST": "1"})
    want = _oracle_pipeline(oracle, tmp_path, [107])
    _same(res[0], want)


# ---- the sharded cutter with W ranks as THREADS of one process, a context each on the one GPU: the library's local communicator
# (mf_comm_create_local) -- what metafast.sh --devices a,b,... runs; on a multi-GPU box the same copies cross xGMI ----
def _virtual_ranks(world, inputs, b1, b2, k=31, b=1, l=100, fail_rank=None, fail_at=None, options=None, one_call=False):
    """inputs: (bases, offsets) host arrays of the samples.  one_call=False: every rank counts its shard of ALL samples' unitigs itself and
    runs mf_cut_components_of_shard (failures can be injected: fail_at = "shard" / "merge" / "merge:3" / "level_local:3"); one_call=True:
    rank r holds the unitigs of samples r, r + W, ... and calls mf_cut_components_sharded (gather + shard count + protocol in one call).
    -> per rank (components export, info) or ("abort", message)"""
    import threading
    import torch
    from util import to_device
    from metafast_amd import lib as L, pipeline as P
    ctx0 = L.Context(0)
    per_sample = []
    for bases, offsets in inputs:
        db, do = to_device(bases, offsets)
        t = ctx0.count_device(db.data_ptr(), do.data_ptr(), len(offsets) - 1, len(bases), k, 0)
        g = t.filter(b)
        sq = ctx0.build_unitigs(g, b, l)
        v = sq.device_view()
        per_sample.append((P.device_tensor(v["bases"], v["n_bases"], "cuda").clone(), P.device_tensor(v["offsets"], (v["n"] + 1) * 8, "cuda").view(torch.int64).clone(), v["n_bases"]))
        sq.close(); g.close(); t.close()

    def cat(samples):
        bs, os_, nb = [], [], 0
        for sb, so, n in samples:
            bs.append(sb); os_.append(so[:-1] + nb); nb += n
        allb = torch.zeros(nb + 64, dtype=torch.uint8, device="cuda")
        if bs:
            allb[:nb] = torch.cat(bs)
        allo = torch.cat(os_ + [torch.tensor([nb], dtype=torch.int64, device="cuda")])
        return allb, allo, nb
    allb, allo, nb = cat(per_sample)
    mine = [cat(per_sample[r::world]) for r in range(world)]
    torch.cuda.synchronize()
    ctxs = [L.Context(0) for _ in range(world)]
    comms = L.Comm.local(ctxs)
    out, errs = [None] * world, []

    def work(rank):
        try:
            torch.cuda.set_device(0)
            ctx, comm = ctxs[rank], comms[rank]
            ctx.bind_thread()
            for name, val in (options or {}).items():
                ctx.set_option(name, val)
            if one_call:
                mb, mo, mnb = mine[rank]
                try:
                    comps = comm.cut_components_sharded(mb.data_ptr(), mo.data_ptr(), int(mo.numel()) - 1, mnb, k, l, b1, b2)
                except L.DistAbort as e:
                    out[rank] = ("abort", str(e))
                    return
                out[rank] = (comps.export(), dict(comm.stats(), kind=comm.kind))
                return
            shard = ctx.count_device_shard(allb.data_ptr(), allo.data_ptr(), int(allo.numel()) - 1, nb, k, l, rank, world)
            info = {}
            if rank == fail_rank and fail_at == "shard":
                shard = None                                     # (the count failed on this rank)
            elif rank == fail_rank and fail_at:                  # (the n-th call of a library function inside the protocol fails on this rank only)
                which, _, nth = fail_at.partition(":")
                ctx.set_option("dcc_test_fail", {"merge": 1000, "level_local": 2000}[which] + int(nth or 1))
            try:
                comps = P.distributed_components(ctx, comm, shard, k, b1, b2, info=info)
            except L.DistAbort as e:
                out[rank] = ("abort", str(e))
                return
            info["shard_len"] = len(shard)
            out[rank] = (comps.export(), info)
        except BaseException as e:          # (a rank that dies must not leave the others waiting: the barrier gives up after a while)
            errs.append(e)

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for c in comms:
        c.close()
    if errs:
        raise errs[0]
    return out


def _oracle_components(oracle, inputs, b1, b2, k=31, b=1, l=100):
    o_cutter = oracle.Table()
    for bases, offsets in inputs:
        t = oracle.Table().count_buffer(bases, offsets, k)
        keys, vals = t.export(b)
        g = oracle.Table()
        for kk, vv in zip(keys.tolist(), vals.tolist()):
            g.add(kk, vv)
        o_cutter.count_seqs(oracle.build_unitigs(g, k, b, l), k, l)
    return oracle.cut_components(o_cutter, k, b1, b2).all()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_cutter_virtual_ranks(oracle, world):
    """W ranks each own a shard of the cutter table (threshold levels 1..6 on the branchy genome): every rank ends with the
    oracle's components, bit for bit"""
    from util import branchy_reads
    seeds = [107, 117, 127, 137]
    inputs = [branchy_reads(rs, genome_seed=7, n=6000) for rs in seeds]
    want = _oracle_components(oracle, inputs, 100, 1000)
    res = _virtual_ranks(world, inputs, 100, 1000)
    assert len(want) == 13
    for comps, info in res:
        assert [(a, w, t) for a, w, t, _ in comps] == [(a, w, t) for a, w, t, _ in want]
        assert all(np.array_equal(np.sort(g[3]), np.sort(x[3])) for g, x in zip(comps, want))
        assert info["levels"] == 6
    assert sum(i["shard_len"] for _, i in res) == res[0][1]["vertices"]
    assert all(i["shard_len"] > 0 for _, i in res)


@pytest.mark.parametrize("fail_at", ["shard", "merge", "merge:3", "level_local:3"])
def test_sharded_cutter_ranks_abort_together(fail_at):
    """one rank cannot do its part (its shard count failed / a call in the middle of a level fails -- the first level's or a later one's):
    EVERY rank leaves mf_cut_components_of_shard with MF_ERR_TOGETHER at the same gather -- nobody is left waiting inside an exchange
    (pipeline.run_samples then takes the replicated cutter on all ranks)"""
    from util import branchy_reads
    inputs = [branchy_reads(rs, genome_seed=7, n=6000) for rs in (107, 117, 127, 137)]
    res = _virtual_ranks(4, inputs, 100, 1000, fail_rank=2, fail_at=fail_at)
    assert all(r[0] == "abort" for r in res), res
    assert all("rank(s) 2 failed" in r[1] for r in res), res
    assert "injected failure" in res[2][1] or fail_at == "shard"          # (the failing rank also says why)


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_cutter_in_one_call(oracle, world):
    """mf_cut_components_sharded: rank r brings the unitigs of ITS samples, the call gathers them, counts the rank's shard and runs the
    protocol -- what metafast.sh's component-cutter does per device entry: every rank ends with the oracle's components"""
    from util import branchy_reads
    inputs = [branchy_reads(rs, genome_seed=7, n=6000) for rs in [107, 117, 127, 137, 147]]
    want = _oracle_components(oracle, inputs, 100, 1000)
    res = _virtual_ranks(world, inputs, 100, 1000, one_call=True)
    for comps, info in res:
        assert [(a, w, t) for a, w, t, _ in comps] == [(a, w, t) for a, w, t, _ in want]
        assert all(np.array_equal(np.sort(g[3]), np.sort(x[3])) for g, x in zip(comps, want))
        assert info["kind"] == "local" and info["collectives"] > 10 and info["bytes_in"] > 0


def test_comm_primitives_local():
    """the three primitives of a local communicator (4 threads, one GPU): ragged all-gather with an empty contribution, non-uniform
    all-to-all with a rank that neither sends nor receives, integer gather; mismatched sizes are an error on every rank, not a hang"""
    import threading
    import torch
    from metafast_amd import lib as L
    W = 4
    ctxs = [L.Context(0) for _ in range(W)]
    comms = L.Comm.local(ctxs)
    m = np.array([[0, 5, 0, 2], [1, 0, 0, 7], [0, 0, 0, 0], [4, 3, 0, 0]])          # m[src][dst] elements of 8 bytes
    sizes = [3, 0, 4, 1]
    out, errs = [None] * W, []

    def work(r):
        try:
            torch.cuda.set_device(0)
            ctxs[r].bind_thread()
            c = comms[r]
            ints = c.gather_ints([r, 10 * r])
            send = (torch.arange(int(m[r].sum()), dtype=torch.int64, device="cuda") + 1000 * r)
            recv = torch.full((int(m[:, r].sum()) + 1,), -7, dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            c.all_to_all(send.data_ptr(), 8 * m[r], recv.data_ptr(), 8 * m[:, r])
            ag_in = torch.full((sizes[r],), r, dtype=torch.int32, device="cuda")
            ag_out = torch.full((sum(sizes) + 1,), -1, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            c.all_gather(ag_in.data_ptr(), ag_out.data_ptr(), [4 * x for x in sizes])
            ctxs[r].synchronize()
            out[r] = (ints.tolist(), recv.cpu().tolist(), ag_out.cpu().tolist(), c.stats())
        except BaseException as e:
            errs.append(e)

    th = [threading.Thread(target=work, args=(r,)) for r in range(W)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for r in range(W):
        ints, recv, ag, st = out[r]
        assert ints == [[0, 0], [1, 10], [2, 20], [3, 30]]
        want = []
        for src in range(W):
            o = int(m[src][:r].sum())
            want += [1000 * src + o + i for i in range(int(m[src][r]))]
        assert recv == want + [-7]
        assert ag == [0, 0, 0, 2, 2, 2, 2, 3, -1]
        assert st["collectives"] == 3 and st["bytes_in"] == 8 * 8 + 8 * int(m[:, r].sum()) + 4 * sum(sizes)
    # sizes that do not match: an error on the ranks that see it, the others are woken up -- nobody hangs
    res = [None] * W

    def bad(r):
        torch.cuda.set_device(0)
        ctxs[r].bind_thread()
        buf = torch.zeros(64, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        try:
            comms[r].all_to_all(buf.data_ptr(), [8, 8, 8, 8], buf.data_ptr() + 256, [8, 8, 8, 16 if r == 0 else 8])
            res[r] = "ok"
        except L.MetafastError as e:
            res[r] = str(e)
    th = [threading.Thread(target=bad, args=(r,)) for r in range(W)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert all(x != "ok" for x in res), res
    for c in comms:
        c.close()


RCCL_WORKER = r'''
import json, os, sys
sys.path.insert(0, os.environ["MF_ROOT"])
import numpy as np, torch
from metafast_amd import lib as L
torch.cuda.set_device(0)
ctx = L.Context(0)
c = L.Comm.rccl(ctx, L.Comm.rccl_id(), 0, 1)
a = torch.arange(1000, dtype=torch.int64, device="cuda")
b = torch.zeros(1000, dtype=torch.int64, device="cuda"); d = torch.zeros(1000, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
ints = c.gather_ints([5, 6, 7]).tolist()
c.all_gather(a.data_ptr(), b.data_ptr(), [8000])
c.all_to_all(a.data_ptr(), [8000], d.data_ptr(), [8000])
ctx.synchronize()
rows = c.features_allgather(np.arange(12, dtype=np.int64).reshape(3, 4)).tolist()
print(json.dumps(dict(kind=c.kind, ints=ints, ag=bool((a == b).all()), aa=bool((a == d).all()), rows=rows, stats=c.stats())))
'''


def test_comm_rccl_world1():
    """the library's own RCCL communicator (librccl through dlopen, ncclCommInitRank, grouped send / recv) at world size 1 with MF_FORCE_DIST=1:
    the transport really runs (more than one rank needs more than one GPU: the two-GPU tests below)"""
    env = dict(os.environ, MF_ROOT=ROOT, MF_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", RCCL_WORKER], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])      # (RCCL prints a banner of its own on stdout)
    assert d["kind"] == "rccl" and d["ints"] == [[5, 6, 7]] and d["ag"] and d["aa"]
    assert d["rows"] == np.arange(12).reshape(3, 4).tolist()
    assert d["stats"]["collectives"] >= 5 and d["stats"]["bytes_in"] >= 16000


@pytest.mark.skipif(not os.environ.get("MF_TRY_RCCL_2RANKS"), reason="two RCCL ranks on ONE device: RCCL refuses duplicate GPUs on most builds (opt-in: MF_TRY_RCCL_2RANKS=1)")
def test_two_ranks_one_gpu_rccl(oracle, tmp_path):
    res = _run(2, "nccl", tmp_path)
    want = _oracle_pipeline(oracle, tmp_path, [107, 117])
    for r in res:
        _same(r, want)


def test_dist_cutter_rejects_bad_arguments(gpu_ctx):
    """error behaviour of the sharded cutter's entry points: a clean MetafastError, nothing half-built"""
    from util import branchy_reads, to_device
    from metafast_amd import lib as L
    b, o = branchy_reads(107, genome_seed=7, n=6000)
    db, do = to_device(b, o)
    args = (db.data_ptr(), do.data_ptr(), len(o) - 1, len(b), 31, 0)
    with pytest.raises(L.MetafastError):
        gpu_ctx.count_device_shard(*args, 0, 3)                       # world sizes are powers of two
    with pytest.raises(L.MetafastError):
        gpu_ctx.count_device_shard(*args, 2, 2)                       # rank < world
    with pytest.raises(L.MetafastError):
        gpu_ctx.count_device_shard(db.data_ptr(), do.data_ptr(), len(o) - 1, len(b), 15, 0, 0, 2)     # k >= 20: minimizer partitions decide the owner
    shard = gpu_ctx.count_device_shard(*args, 1, 2)
    n = len(shard)
    assert n > 0
    with pytest.raises(L.MetafastError):
        L.DistCutter(gpu_ctx, shard, 1, 2, [0, 5, 5 + n + 1])         # base[] must match the shard
    whole = gpu_ctx.count_device(*args)
    with pytest.raises(L.MetafastError):
        L.DistCutter(gpu_ctx, whole, 0, 2, [0, len(whole), len(whole)])               # a table that holds other ranks' k-mers too
    D = L.DistCutter(gpu_ctx, shard, 1, 2, [0, 5, 5 + n])
    q = D.queries()
    assert q[1] == 0 and q[0] > 0                                     # every foreign neighbour belongs to rank 0
    with pytest.raises(L.MetafastError):
        D.set_answers(0, 0)                                           # as many answers as queries
    D.close()


def test_sharded_cutter_with_nothing_to_cut():
    """samples without a single unitig (random reads, no k-mer seen twice): empty shards on every rank, no components, no hang"""
    rng = np.random.default_rng(5)
    from util import random_reads
    inputs = [random_reads(rng, 300, 120, 160) for _ in range(2)]
    res = _virtual_ranks(2, inputs, 100, 1000)
    for comps, info in res:
        assert comps == [] and info["vertices"] == 0 and info["members"] == 0 and info["levels"] == 1


def test_sharded_cutter_small_graph_many_ranks(oracle):
    """a graph of a few thousand k-mers over 8 ranks (some threshold levels leave ranks with nothing alive): the oracle's components"""
    from util import branchy_reads
    inputs = [branchy_reads(107, genome_seed=7, n=2500)]
    want = _oracle_components(oracle, inputs, 20, 400)
    res = _virtual_ranks(8, inputs, 20, 400)
    assert len(want) > 0
    for comps, info in res:
        assert [(a, w, t) for a, w, t, _ in comps] == [(a, w, t) for a, w, t, _ in want]
        assert all(np.array_equal(np.sort(g[3]), np.sort(x[3])) for g, x in zip(comps, want))


@pytest.mark.parametrize("seed", [107, 211, 307])
def test_sharded_cutter_sparse_setup_every_level(oracle, seed):
    """after the first level the arrays over all vertex ids are reset only where this level's pairs and this rank's own
    fragment roots touch them (option dcc_sparse forces that path at every level).  Small graphs over 8 ranks leave remnants of an
    oversize component that live on ONE rank, some of them under the root that summed up the whole component a level earlier:
    every rank must still see the owner's size for them (no stale sums), i.e. agree on the kept list and leave the level loop
    together -- and end with the oracle's components."""
    from util import branchy_reads
    inputs = [branchy_reads(seed, genome_seed=7, n=2500), branchy_reads(seed + 10, genome_seed=7, n=2500)]
    want = _oracle_components(oracle, inputs, 20, 300)
    res = _virtual_ranks(8, inputs, 20, 300, options={"dcc_sparse": 1})
    assert len(want) > 0
    for comps, info in res:
        assert not isinstance(comps, str), (comps, info)
        assert [(a, w, t) for a, w, t, _ in comps] == [(a, w, t) for a, w, t, _ in want]
        assert all(np.array_equal(np.sort(g[3]), np.sort(x[3])) for g, x in zip(comps, want))
    assert max(i["levels"] for _, i in res) >= 2


# ---- a box with more than one GPU tests itself (this pool's boxes have one: the tests below skip there) ----
@pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs (RCCL over xGMI with one rank per GPU)")
@pytest.mark.parametrize("spg", [1, 2])
def test_two_ranks_two_gpus_rccl(oracle, tmp_path, spg):
    """one rank per GPU over RCCL: the sharded cutter's all-to-all / all-gathers between two devices, against the oracle's pipeline
    (KmersCounterForManyFilesMain.java:80-108 + ComponentCutterMain.java:78-114 on all samples)"""
    res = _run(2, "nccl", tmp_path, spg=spg, one_gpu_per_rank=True)
    want = _oracle_pipeline(oracle, tmp_path, [107 + 10 * i for i in range(2 * spg)])
    for r in res:
        _same(r, want)
        assert r["comm"]["collectives"] > 0 and r["comm"]["bytes_in"] > 0


@pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs")
def test_bench_two_gpus_under_the_launcher():
    """bench.py exactly as the driver launches it for N = 2 (the launcher starts before anything touches a GPU; the ranks are its
    child processes): one JSON line, n_gpus = 2, the sharded cutter's collectives counted, weak scaling (50 M-read config scaled down)"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--reads", "5000000", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.strip()]
    assert len(lines) == 1, lines[:3]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["comm"]["collectives_per_step"] > 0 and d["comm"]["MB_received_per_step"] > 0 and d["comm"]["kind"] == "rccl"
    assert d["stats"]["n_occ"] == 5000000 * 120 and "components" in d["stage_ms_per_step"]

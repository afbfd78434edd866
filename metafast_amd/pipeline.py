"""Device-resident matrix-builder pipeline, one sample per GPU (one process per GPU).

Host-side mirror of the step wiring in DistanceMatrixBuilderMain (src/tools/DistanceMatrixBuilderMain.java:88-175):
kmer-counter -> seq-builder -> component-cutter -> features-calculator -> dist-matrix-calculator, with the
files between the steps replaced by buffers that stay in HBM.  Steps 1, 2 and 4 are independent per sample
(KmersCounterForManyFilesMain.java:80-108, SeqBuilderForManyFilesMain.java:82-94, FeaturesCalculatorMain.java:137-162);
step 3 joins all samples (ComponentCutterMain.java:81), so the ranks exchange their unitigs once (all-gather over
RCCL / xGMI) and every rank builds the same cutter table and components; the per-sample feature vectors are
all-gathered for the Bray-Curtis matrix.  torch is used for device memory and torch.distributed only.
"""
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from . import lib as L


# MF_FORCE_DIST=1: run the collectives even at world size 1 (lets a 1-GPU box exercise the RCCL code path)
def _force():
    """MF_FORCE_DIST=1 and an initialised process group (checked at call time: the variable alone must not send a
    single-process run into collectives without a group)"""
    return bool(os.environ.get("MF_FORCE_DIST")) and dist.is_available() and dist.is_initialized()


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def all_gather_ragged(t):
    """all-gather of 1-D tensors of different lengths -> list of per-rank tensors (same device/dtype as t).
    Sizes are exchanged first; payloads are padded to the largest one (one collective each)."""
    rank, world = _world()
    if world == 1 and not _force():
        return [t]
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    m = max(max(sizes), 1)
    pad = torch.zeros(m, dtype=t.dtype, device=t.device)
    pad[: t.numel()] = t
    outs = [torch.empty(m, dtype=t.dtype, device=t.device) for _ in range(world)]
    dist.all_gather(outs, pad)
    return [o[:s] for o, s in zip(outs, sizes)]


def gather_sequences(bases, offsets):
    """bases: uint8[n_bases], offsets: int64[n+1] of this rank's unitigs -> concatenation over all ranks
    (rank order), offsets rebased; the result has 64 bytes of slack after the last base."""
    parts_b = all_gather_ragged(bases)
    parts_o = all_gather_ragged(offsets)
    nb = sum(int(p.numel()) for p in parts_b)
    ns = sum(int(p.numel()) - 1 for p in parts_o)
    allb = torch.zeros(nb + 64, dtype=torch.uint8, device=bases.device)
    allo = torch.zeros(ns + 1, dtype=torch.int64, device=bases.device)
    pb = po = 0
    for b, o in zip(parts_b, parts_o):
        n = int(o.numel()) - 1
        allb[pb:pb + b.numel()] = b
        allo[po:po + n] = o[:-1] + pb
        pb += int(b.numel())
        po += n
    allo[ns] = nb
    return allb, allo, ns, nb


def gather_vectors(vec):
    """vec: int64[C] -> int64[world, C] (every rank has the same C)"""
    rank, world = _world()
    if world == 1 and not _force():
        return vec.reshape(1, -1)
    outs = [torch.empty_like(vec) for _ in range(world)]
    dist.all_gather(outs, vec)
    return torch.stack(outs)


def device_tensor(ptr, nbytes, device):
    """zero-copy torch uint8 view of a library-owned device buffer"""
    if nbytes == 0:
        return torch.zeros(0, dtype=torch.uint8, device=device)

    class _H:
        pass

    h = _H()
    h.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 3}
    return torch.as_tensor(h, device=device)


def run_sample(ctx, d_bases, d_offsets, n_reads, n_bases, k=31, b=1, l=100, b1=1000, b2=10000, device="cuda",
               timings=None):
    """One sample on this rank's GPU, joined with the other ranks for the cutter and the matrix.
    d_bases / d_offsets: torch tensors in HBM (ASCII bases, int64 offsets).  Returns a dict of results."""
    t0 = time.perf_counter()

    def mark(name):
        nonlocal t0
        if timings is not None:
            torch.cuda.synchronize() if torch.cuda.is_available() else None
            t1 = time.perf_counter()
            timings[name] = timings.get(name, 0.0) + (t1 - t0)
            t0 = t1

    # kmer-counter: k-mers with count > b go on (IOUtils.printKmers); the others are dropped inside the counting kernels
    good, n_distinct = ctx.count_device_above(d_bases.data_ptr(), d_offsets.data_ptr(), n_reads, n_bases, k, b)
    # ... and the histogram of ALL counts, dropped k-mers included (the .stat.txt of IOUtils.printKmers, src/io/IOUtils.java:45-71)
    hist = good.hist()
    mark("count")
    seqs = ctx.build_unitigs(good, b, l)
    mark("unitigs")
    v = seqs.device_view()
    sb = device_tensor(v["bases"], v["n_bases"], device)
    so = device_tensor(v["offsets"], (v["n"] + 1) * 8, device).view(torch.int64)
    ctx.synchronize()
    allb, allo, ns, nb = gather_sequences(sb, so)
    if torch.cuda.is_available():
        torch.cuda.current_stream().synchronize()
    mark("exchange_unitigs")
    cutter = ctx.count_device(allb.data_ptr(), allo.data_ptr(), ns, nb, k, l)
    mark("cutter_count")
    comps = ctx.cut_components(cutter, b1, b2)
    mark("components")
    vec, breadth = ctx.features(comps, good, 0)
    vt = torch.from_numpy(vec).to(device)
    vecs = gather_vectors(vt).cpu().numpy()
    matrix = L.bray_curtis(vecs) if vecs.shape[1] else np.zeros((vecs.shape[0], vecs.shape[0]))
    mark("features_matrix")
    return dict(good=good, seqs=seqs, cutter=cutter, comps=comps, vec=vec, breadth=breadth, vecs=vecs,
                matrix=matrix, n_occ=good.occurrences(), n_distinct=n_distinct, hist=hist)

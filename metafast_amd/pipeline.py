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
    """all-gather of 1-D tensors of different lengths -> list of per-rank tensors (views of ONE buffer of sum-of-sizes
    elements; same device/dtype as t).  Sizes are exchanged first; every rank's payload then lands at its offset of the
    pre-sized buffer: one all_gather_into_tensor when the sizes agree, else one broadcast per rank into its view (the same
    bytes on the wire as an all-gather; nothing is padded to the largest payload)."""
    rank, world = _world()
    if world == 1 and not _force():
        return [t]
    if t.is_cuda and dist.get_backend() == "gloo":          # (tests: two ranks on one GPU talk through gloo, on the host)
        return [p.to(t.device) for p in all_gather_ragged(t.cpu())]
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    buf = torch.empty(max(sum(sizes), 1), dtype=t.dtype, device=t.device)
    outs, at = [], 0
    for s_ in sizes:
        outs.append(buf[at:at + s_]); at += s_
    if len(set(sizes)) == 1:
        if at:
            dist.all_gather_into_tensor(buf[:at], t.contiguous())
    else:
        outs[rank].copy_(t)
        works = [dist.broadcast(outs[r], src=r, async_op=True) for r in range(world) if sizes[r]]
        for w in works:
            w.wait()
    return outs


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
    if vec.is_cuda and dist.get_backend() == "gloo":
        return gather_vectors(vec.cpu()).to(vec.device)
    outs = [torch.empty_like(vec) for _ in range(world)]
    dist.all_gather(outs, vec)
    return torch.stack(outs)


def gather_vector_rows(rows):
    """rows: int64[n_local_samples, C] -> int64[n_all_samples, C], rank-major (ranks may hold different numbers of samples)"""
    rank, world = _world()
    if world == 1 and not _force():
        return rows
    c = rows.shape[1]
    parts = all_gather_ragged(rows.reshape(-1))
    return torch.cat([p.reshape(-1, c) for p in parts]) if c else torch.zeros((sum(int(p.numel()) for p in parts), 0), dtype=rows.dtype, device=rows.device)


def device_tensor(ptr, nbytes, device):
    """zero-copy torch uint8 view of a library-owned device buffer"""
    if nbytes == 0:
        return torch.zeros(0, dtype=torch.uint8, device=device)

    class _H:
        pass

    h = _H()
    h.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 3}
    return torch.as_tensor(h, device=device)


def run_samples(ctx, samples, k=31, b=1, l=100, b1=1000, b2=10000, device="cuda", timings=None):
    """This rank's samples (KmersCounterForManyFilesMain.java:80-108 loops over all libraries: with more samples than GPUs a
    rank takes several, one after the other), joined with the other ranks' for the cutter and the matrix.
    samples: iterable of (d_bases, d_offsets, n_reads, n_bases) -- torch tensors in HBM (ASCII bases, int64 offsets); it may
    be a generator that produces a sample only when it is asked for (the reads of one sample at a time in HBM).
    The global sample order is rank-major: rank 0's samples, then rank 1's, ...  Returns a dict of results."""
    t0 = time.perf_counter()

    def mark(name):
        nonlocal t0
        if timings is not None:
            torch.cuda.synchronize() if torch.cuda.is_available() else None
            t1 = time.perf_counter()
            timings[name] = timings.get(name, 0.0) + (t1 - t0)
            t0 = t1

    goods, seqss, hists, n_occ, n_distinct = [], [], [], 0, 0
    for d_bases, d_offsets, n_reads, n_bases in samples:
        # kmer-counter: k-mers with count > b go on (IOUtils.printKmers); the others are dropped inside the counting kernels
        good, nd = ctx.count_device_above(d_bases.data_ptr(), d_offsets.data_ptr(), n_reads, n_bases, k, b)
        # ... and the histogram of ALL counts, dropped k-mers included (the .stat.txt of IOUtils.printKmers, src/io/IOUtils.java:45-71)
        hists.append(good.hist())
        mark("count")
        seqss.append(ctx.build_unitigs(good, b, l))
        mark("unitigs")
        goods.append(good); n_occ += good.occurrences(); n_distinct += nd
    # this rank's unitigs, all samples one after the other
    views = [sq.device_view() for sq in seqss]
    parts_b = [device_tensor(v["bases"], v["n_bases"], device) for v in views]
    parts_o, nb = [], 0
    for v in views:
        parts_o.append(device_tensor(v["offsets"], (v["n"] + 1) * 8, device).view(torch.int64)[:-1] + nb)
        nb += v["n_bases"]
    ctx.synchronize()
    if len(views) == 1:
        sb, so = parts_b[0], device_tensor(views[0]["offsets"], (views[0]["n"] + 1) * 8, device).view(torch.int64)
    else:
        sb = torch.cat(parts_b) if parts_b else torch.zeros(0, dtype=torch.uint8, device=device)
        so = torch.cat(parts_o + [torch.tensor([nb], dtype=torch.int64, device=device)])
    allb, allo, ns, nbt = gather_sequences(sb, so)
    if torch.cuda.is_available():
        torch.cuda.current_stream().synchronize()
    mark("exchange_unitigs")
    cutter = ctx.count_device(allb.data_ptr(), allo.data_ptr(), ns, nbt, k, l)
    mark("cutter_count")
    comps = ctx.cut_components(cutter, b1, b2)
    mark("components")
    vecs_local, breadths = [], []
    for good in goods:
        vec, breadth = ctx.features(comps, good, 0)
        vecs_local.append(vec); breadths.append(breadth)
    vt = torch.from_numpy(np.stack(vecs_local) if vecs_local else np.zeros((0, len(comps)), dtype=np.int64)).to(device)
    vecs = gather_vector_rows(vt).cpu().numpy()
    matrix = L.bray_curtis(vecs) if vecs.shape[1] else np.zeros((vecs.shape[0], vecs.shape[0]))
    mark("features_matrix")
    return dict(goods=goods, seqss=seqss, cutter=cutter, comps=comps, vecs_local=vecs_local, breadths=breadths, vecs=vecs,
                matrix=matrix, n_occ=n_occ, n_distinct=n_distinct, hists=hists)


def run_sample(ctx, d_bases, d_offsets, n_reads, n_bases, k=31, b=1, l=100, b1=1000, b2=10000, device="cuda",
               timings=None):
    """One sample on this rank's GPU (run_samples with a single sample; the result keys of one sample)."""
    r = run_samples(ctx, [(d_bases, d_offsets, n_reads, n_bases)], k=k, b=b, l=l, b1=b1, b2=b2, device=device, timings=timings)
    return dict(good=r["goods"][0], seqs=r["seqss"][0], cutter=r["cutter"], comps=r["comps"], vec=r["vecs_local"][0],
                breadth=r["breadths"][0], vecs=r["vecs"], matrix=r["matrix"], n_occ=r["n_occ"], n_distinct=r["n_distinct"],
                hist=r["hists"][0])

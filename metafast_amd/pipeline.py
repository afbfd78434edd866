"""Device-resident matrix-builder pipeline, one sample per GPU (one process per GPU).

Host-side mirror of the step wiring in DistanceMatrixBuilderMain (src/tools/DistanceMatrixBuilderMain.java:88-175):
kmer-counter -> seq-builder -> component-cutter -> features-calculator -> dist-matrix-calculator, with the
files between the steps replaced by buffers that stay in HBM.  Steps 1, 2 and 4 are independent per sample
(KmersCounterForManyFilesMain.java:80-108, SeqBuilderForManyFilesMain.java:82-94, FeaturesCalculatorMain.java:137-162);
step 3 joins all samples (ComponentCutterMain.java:81): the ranks exchange their unitigs once (all-gather over RCCL / xGMI), every rank counts
the k-mers it OWNS (a shard of the cutter table, Context.count_device_shard), and the components are found with every
rank working on its shard (distributed_components); the per-sample feature vectors are all-gathered for the
Bray-Curtis matrix.  torch is used for device memory and torch.distributed only.
"""
import os
import sys
import threading
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
    elements; same device/dtype as t).  Sizes are exchanged first; how the payloads travel: _gather_sized."""
    rank, world = _world()
    if world == 1 and not _force():
        return [t]
    if t.is_cuda and dist.get_backend() == "gloo":          # (tests: two ranks on one GPU talk through gloo, on the host)
        return [p.to(t.device) for p in all_gather_ragged(t.cpu())]
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    buf = _gather_sized(t, sizes, rank, world)
    outs, at = [], 0
    for s_ in sizes:
        outs.append(buf[at:at + s_]); at += s_
    return outs


def _gather_sized(t, sizes, rank, world):
    """1-D tensors of known sizes -> their concatenation in rank order.  Equal sizes: one all_gather_into_tensor.  Nearly
    equal sizes (the largest within 12.5 % of the mean: the usual case, samples and shards are balanced): ONE collective on
    payloads padded to the largest, then compacted -- a collective costs tens of microseconds before the first byte, and the
    threshold levels of the cutter issue hundreds.  Otherwise: one broadcast per rank into its slice of the pre-sized buffer
    (exactly the sum of the sizes on the wire, nothing padded)."""
    tot, mx = sum(sizes), max(sizes) if sizes else 0
    buf = _alloc(lambda: torch.empty(max(tot, 1), dtype=t.dtype, device=t.device))
    if not tot:
        return buf[:0]
    if len(set(sizes)) == 1:
        dist.all_gather_into_tensor(buf[:tot], t.contiguous())
    elif mx * world <= tot + tot // 8 + 4096:
        inp = _alloc(lambda: torch.empty(mx, dtype=t.dtype, device=t.device))
        inp[:t.numel()] = t
        pad = _alloc(lambda: torch.empty(mx * world, dtype=t.dtype, device=t.device))
        dist.all_gather_into_tensor(pad, inp)
        at = 0
        for r, n in enumerate(sizes):
            buf[at:at + n] = pad[r * mx:r * mx + n]; at += n
    else:
        at, works = 0, []
        for r, n in enumerate(sizes):
            if n:
                if r == rank:
                    buf[at:at + n].copy_(t)
                works.append(dist.broadcast(buf[at:at + n], src=r, async_op=True))
            at += n
        for w in works:
            w.wait()
    return buf[:tot]


# torch allocations of the exchange helpers go through this hook: run_samples points it at _with_room (the library's arena may hold all
# of the device in idle regions -- 4 x 380 M reads: 42 MB free when the gathered unitigs wanted 264 MB), anybody else gets a plain call
# -- per THREAD (pipeline.ThreadComm runs several ranks as threads, each with a context of its own that only its thread may touch), and only
# for the duration of that thread's run_samples call
_TLS = threading.local()


def _alloc(fn):
    hook = getattr(_TLS, "alloc", None)
    return hook(fn) if hook is not None else fn()


def gather_sequences(bases, offsets):
    """bases: uint8[n_bases], offsets: int64[n+1] of this rank's unitigs -> concatenation over all ranks
    (rank order), offsets rebased; the result has 64 bytes of slack after the last base."""
    parts_b = all_gather_ragged(bases)
    parts_o = all_gather_ragged(offsets)
    nb = sum(int(p.numel()) for p in parts_b)
    ns = sum(int(p.numel()) - 1 for p in parts_o)
    allb = _alloc(lambda: torch.zeros(nb + 64, dtype=torch.uint8, device=bases.device))
    allo = _alloc(lambda: torch.zeros(ns + 1, dtype=torch.int64, device=bases.device))
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


class TorchComm:
    """The exchange steps of the distributed cutter over torch.distributed (nccl = RCCL over xGMI; gloo in the CPU-side
    tests, staged through the host)."""

    def __init__(self):
        self.rank, self.world = _world()
        self.stats = dict(collectives=0, bytes_in=0, seconds=0.0)     # what a step exchanged (bench.py reports it)

    def _account(self, t0, nbytes_in):
        if torch.cuda.is_available():
            torch.cuda.current_stream().synchronize()
        self.stats["collectives"] += 1
        self.stats["bytes_in"] += int(nbytes_in)
        self.stats["seconds"] += time.perf_counter() - t0

    def all_gather_ints(self, vals):
        """small host vectors (same length on every rank) -> int64 ndarray [world, len]"""
        if self.world == 1 and not _force():
            return np.asarray([vals], dtype=np.int64)
        t0 = time.perf_counter()
        dev = "cpu" if dist.get_backend() == "gloo" else "cuda"
        t = torch.tensor(list(vals), dtype=torch.int64, device=dev)
        out = torch.empty(self.world * t.numel(), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(out, t)
        res = out.cpu().numpy().reshape(self.world, -1)
        self._account(t0, out.numel() * 8)
        return res

    def all_gather(self, t, sizes):
        """1-D tensors, sizes[r] elements on rank r (known to all) -> their concatenation in rank order"""
        sizes = [int(x) for x in sizes]
        if self.world == 1 and not _force():
            return t
        if t.is_cuda and dist.get_backend() == "gloo":
            return self.all_gather(t.cpu(), sizes).to(t.device)
        t0 = time.perf_counter()
        out = _gather_sized(t, sizes, self.rank, self.world)
        self._account(t0, out.numel() * out.element_size())
        return out

    def all_reduce_min(self, t):
        if self.world == 1 and not _force():
            return t
        if t.is_cuda and dist.get_backend() == "gloo":
            return self.all_reduce_min(t.cpu()).to(t.device)
        if t.numel():
            t0 = time.perf_counter()
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            self._account(t0, t.numel() * t.element_size())
        return t

    def all_to_all(self, t, matrix):
        """t: this rank's payload grouped by destination, matrix[src][dst] = elements (known to all) -> what the others sent here,
        grouped by source"""
        send = [int(x) for x in matrix[self.rank]]
        recv = [int(matrix[r][self.rank]) for r in range(self.world)]
        if self.world == 1 and not _force():
            return t
        if t.is_cuda and dist.get_backend() == "gloo":         # (tests: staged through the host, the same call below)
            return self.all_to_all(t.cpu(), matrix).to(t.device)
        t0 = time.perf_counter()
        out = _alloc(lambda: torch.empty(sum(recv), dtype=t.dtype, device=t.device))
        dist.all_to_all_single(out, t.contiguous(), recv, send)
        self._account(t0, out.numel() * out.element_size())
        return out


    def all_to_all_v(self, t, send, recv):
        """all_to_all where a rank knows only ITS OWN split sizes: send[d] elements go to rank d, recv[s] come from rank s"""
        send, recv = [int(x) for x in send], [int(x) for x in recv]
        if self.world == 1 and not _force():
            return t
        if t.is_cuda and dist.get_backend() == "gloo":
            return self.all_to_all_v(t.cpu(), send, recv).to(t.device)
        t0 = time.perf_counter()
        out = _alloc(lambda: torch.empty(sum(recv), dtype=t.dtype, device=t.device))
        dist.all_to_all_single(out, t.contiguous(), recv, send)
        self._account(t0, out.numel() * out.element_size())
        return out


DistAbort = L.DistAbort        # (every rank raises it from the same call: see lib.DistAbort)


def make_comm(ctx, device="cuda"):
    """The communicator of this rank (mf_comm, include/metafast_hip.h): the exchange steps of the path run INSIDE the library on it.
    One process per GPU under torch.distributed: the library's own RCCL communicator (mf_comm_create_rccl; rank 0's id travels through
    torch's process group, which is only the rendezvous) -- or, with the gloo backend (tests: several ranks on one GPU) or MF_COMM=torch,
    an external communicator whose three primitives are TorchComm's.  A single process: a local communicator of one rank."""
    rank, world = _world()
    if not (dist.is_available() and dist.is_initialized()):
        return L.Comm.local([ctx])[0]
    backend = dist.get_backend()
    if backend == "nccl" and os.environ.get("MF_COMM", "rccl") != "torch":
        # the library's own RCCL communicator; torch's process group is the rendezvous for the id.  Whatever goes wrong on ANY rank (no librccl
        # to be found, ncclCommInitRank refusing, a first small exchange that fails) sends ALL ranks to torch's collectives instead, together
        ids = [None]
        if rank == 0:
            try:
                ids = [L.Comm.rccl_id()]
            except L.MetafastError as e:
                print("[metafast_amd] no RCCL communicator of the library's own (%s): torch.distributed's collectives instead" % e, file=sys.stderr)
        dist.broadcast_object_list(ids, src=0)
        comm, ok = None, 0
        if ids[0] is not None:
            try:
                comm = L.Comm.rccl(ctx, ids[0], rank, world)
                ok = int(comm.gather_ints([rank + 1])[:, 0].tolist() == list(range(1, world + 1)))
            except L.MetafastError as e:
                print("[metafast_amd] rank %d: the library's RCCL communicator failed (%s)" % (rank, e), file=sys.stderr)
        flag = torch.tensor([ok], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            return comm
        if comm is not None:
            comm.close()
    tc = TorchComm()
    dev = torch.device(device) if not isinstance(device, torch.device) else device

    def gather_ints(vals):
        return tc.all_gather_ints(vals)

    def all_gather(d_send, d_recv, nbytes):
        send = device_tensor(d_send, nbytes[rank], dev)
        out = tc.all_gather(send, nbytes)
        if sum(nbytes):
            device_tensor(d_recv, sum(nbytes), dev).copy_(out)
        torch.cuda.current_stream().synchronize()

    def all_to_all(d_send, sb, d_recv, rb):
        send = device_tensor(d_send, sum(sb), dev)
        out = tc.all_to_all_v(send, sb, rb)
        if sum(rb):
            device_tensor(d_recv, sum(rb), dev).copy_(out)
        torch.cuda.current_stream().synchronize()

    return L.Comm.external(ctx, rank, world, gather_ints, all_gather, all_to_all)


def distributed_components(ctx, comm, shard, k, b1, b2, device="cuda", timings=None, info=None):
    """Component cutter with every rank owning a shard of the cutter table (include/metafast_hip.h, "A9-A11 on several GPUs"): since
    round 6 the exchange protocol runs inside the library (mf_cut_components_of_shard, mf_comm.hip) on `comm` (lib.Comm).  shard: this
    rank's Context.count_device_shard of all samples' unitigs (None: the count failed here -- the ranks then raise DistAbort
    together).  Returns the components: the same object on every rank, identical to cut_components on the whole table
    (ComponentsBuilder.splitStrategy, src/algo/ComponentsBuilder.java:24-32)."""
    comps, inf = comm.cut_components_of_shard(shard, k, b1, b2)
    if info is not None:
        st = comm.stats()
        info.update(inf, shard=len(shard) if shard is not None else 0, collectives=st["collectives"], MB_received=round(st["bytes_in"] / 1e6, 2))
    return comps


def _with_room(ctx, fn):
    """fn() allocates torch tensors beside the library's workspace arena.  When the device is full of idle arena regions
    (samples of hundreds of millions of reads leave 100 GB record buffers cached), hand some back and try again: small
    regions first, because the library pays 35 ms per GiB for a region it has to allocate again."""
    oom = getattr(torch, "OutOfMemoryError", RuntimeError)
    for want in (2 << 30, 16 << 30, None):
        try:
            return fn()
        except oom as e:
            if "out of memory" not in str(e).lower():
                raise
            ctx.synchronize()
            ctx.trim(want)
    return fn()


def run_samples(ctx, samples, k=31, b=1, l=100, b1=1000, b2=10000, device="cuda", timings=None):
    """_run_samples with the exchange helpers' allocations of THIS thread going through _with_room(ctx, .) while it runs"""
    prev = getattr(_TLS, "alloc", None)
    _TLS.alloc = lambda fn: _with_room(ctx, fn)
    try:
        return _run_samples(ctx, samples, k=k, b=b, l=l, b1=b1, b2=b2, device=device, timings=timings)
    finally:
        _TLS.alloc = prev


def _run_samples(ctx, samples, k=31, b=1, l=100, b1=1000, b2=10000, device="cuda", timings=None):
    """This rank's samples (KmersCounterForManyFilesMain.java:80-108 loops over all libraries: with more samples than GPUs a
    rank takes several, one after the other), joined with the other ranks' for the cutter and the matrix.
    samples: iterable of (d_bases, d_offsets, n_reads, n_bases) -- torch tensors in HBM (ASCII bases, int64 offsets) -- or of
    tuples of file names (FASTA / FASTQ / .gz / .bz2 / .binq: the library reads, parses and uploads them); it may
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
    # Several samples on this rank, MF_OVERLAP_SAMPLES=1 (round 6, OFF by default): sample i's unitigs are built while sample i + 1 is counted.
    # The idea: the walk / hook kernels of the graph stages are latency-bound (< 10 % VALU busy), the counting kernels
    # issue-bound (65 - 73 %), two streams should fill the device better than one.  A context is driven by one thread at a time and owns its
    # tables, so the samples ALTERNATE between this context and a peer on the same device (made when the second sample arrives, if the device
    # has room for a second workspace), and a sample's unitigs run in a worker thread on the sample's own context while the main thread counts
    # the next sample on the other one.  MEASURED (profiles/r06z_shape_config5_*): config 5 as specified 1296.2 -> 1292.7 ms per step, 8 x 5 M
    # reads 272 -> 254 (most of it the features of the two contexts side by side): the kernels do run side by side -- k_skm_count's event time
    # goes from 278 to 366 ms -- and slow each other down by what the overlap hides.  Not worth a second workspace and a thread by default.
    ctxs, running = [ctx], {}

    def wait(ci):
        w = running.pop(ci, None)
        if w is not None:
            w[0].join()
            if w[1]:
                raise w[1][0]

    def unitigs_of(c, good, slot, drop):
        def work(err):
            try:
                c.bind_thread()                                # (a thread of its own: HIP's current device belongs to the thread)
                seqss[slot] = c.build_unitigs(good, b, l)
                if drop:
                    # several samples on this rank: the sample's lookup index (3-6 times its table) is not needed again before its feature
                    # vector, where it is rebuilt (mf_table_drop_index) -- 4 samples of > 2^32 distinct k-mers each (BASELINE config 5) fit
                    # one GPU's HBM this way
                    good.drop_index()
                c.synchronize()
            except BaseException as e:
                err.append(e)
        return work

    it = iter(samples)
    sample = next(it, None)
    si = -1
    overlap_on = os.environ.get("MF_OVERLAP_SAMPLES", "0") == "1" and torch.cuda.is_available()
    while sample is not None:
        si += 1
        if si == 1 and len(ctxs) == 1 and overlap_on:
            peer = getattr(ctx, "_mf_peer", None)
            if peer is None or peer.h is None:
                free_b = torch.cuda.mem_get_info(ctx.device)[0]
                if free_b > 1.3 * ctx.stat("arena_bytes") + (8 << 30):          # (room for a second workspace of the size the first sample needed)
                    try:
                        peer = L.Context(ctx.device)
                        for name, val in getattr(ctx, "options", {}).items():
                            peer.set_option(name, val)
                        ctx._mf_peer = peer
                    except L.MetafastError:
                        peer = None
            if peer is not None:
                ctxs.append(peer)
        ci = si % len(ctxs)
        c = ctxs[ci]
        wait(ci)                                              # (the unitigs of the sample before the last: the same context)
        if len(ctxs) == 1 and si:
            goods[-1].drop_index()
        # kmer-counter: k-mers with count > b go on (IOUtils.printKmers); the others are dropped inside the counting kernels
        if isinstance(sample[0], (str, bytes, os.PathLike)):
            # a sample handed over as its read files (IOUtils.loadReads, src/io/IOUtils.java:772-803: all files of a library into one
            # table): read + parse + H2D inside the library (mf_count_reads_above)
            good, nd = c.count_reads_above([os.fspath(f) for f in sample], k, b)
        else:
            d_bases, d_offsets, n_reads, n_bases = sample
            if c is not ctx:
                torch.cuda.current_stream().synchronize()      # (the reads were written on torch's stream, the peer launches on its own)
            good, nd = c.count_device_above(d_bases.data_ptr(), d_offsets.data_ptr(), n_reads, n_bases, k, b)
        # ... and the histogram of ALL counts, dropped k-mers included (the .stat.txt of IOUtils.printKmers, src/io/IOUtils.java:45-71)
        hists.append(good.hist())
        c.synchronize()                                        # (the reads are free for the next sample from here on)
        mark("count")
        seqss.append(None)
        goods.append(good); n_occ += good.occurrences(); n_distinct += nd
        work = unitigs_of(c, good, si, drop=si > 0)
        # the next sample, now that this one's reads are free (a generator may refill the same buffer); none: this sample's unitigs run right
        # here -- a single sample (the benchmark's step) never meets a thread
        sample = next(it, None)
        if sample is not None and overlap_on and (len(ctxs) > 1 or si == 0):
            err = []
            th = threading.Thread(target=work, args=(err,))
            th.start()
            running[ci] = (th, err)
        else:
            err = []
            work(err)
            if err:
                raise err[0]
            mark("unitigs")
    for ci in list(running):
        wait(ci)
    mark("unitigs")
    # this rank's unitigs, all samples one after the other
    views = [sq.device_view() for sq in seqss]
    ctx.synchronize()
    keep = None
    if len(views) == 1:
        v = views[0]
        sb, so, ns_l, nb_l = v["bases"] or 0, v["offsets"] or 0, v["n"], v["n_bases"]
    else:
        parts_b = [device_tensor(v["bases"], v["n_bases"], device) for v in views]
        parts_o, nb_l = [], 0
        for v in views:
            parts_o.append(device_tensor(v["offsets"], (v["n"] + 1) * 8, device).view(torch.int64)[:-1] + nb_l)
            nb_l += v["n_bases"]
        tb = _with_room(ctx, lambda: torch.cat(parts_b + [torch.zeros(64, dtype=torch.uint8, device=device)]))
        to = _with_room(ctx, lambda: torch.cat(parts_o + [torch.tensor([nb_l], dtype=torch.int64, device=device)]))
        torch.cuda.current_stream().synchronize()
        keep = (tb, to)
        sb, so, ns_l = tb.data_ptr(), to.data_ptr(), int(to.numel()) - 1
    rank, world = _world()
    # the exchanges run inside the library on this rank's communicator (mf_comm): the sequences of all ranks, the sharded cutter, the rows
    comm = getattr(ctx, "_mf_comm", None)                      # (made once per context: an RCCL communicator costs a rendezvous)
    if comm is None or comm.h is None:
        comm = ctx._mf_comm = make_comm(ctx, device)
    comm.reset_stats()
    allr = comm.gather_sequences(sb, so, ns_l, nb_l)            # every rank's unitigs on every rank (ComponentCutterMain.java:81 over all libraries)
    av = allr.device_view()
    del keep
    mark("exchange_unitigs")
    n_all = int(comm.gather_ints([len(goods)]).sum())
    sharded = (world > 1 or _force()) and k >= 20 and world & (world - 1) == 0 and world <= 64 and not os.environ.get("MF_REPLICATED_CUTTER")
    cutter = comps = None
    if sharded:
        # every rank owns a shard of the cutter table and of the components step (mf_cut_components_of_shard)
        try:
            cutter = ctx.count_device_shard(av["bases"], av["offsets"], av["n"], av["n_bases"], k, l, rank, world)
        except L.MetafastError:
            cutter = None                    # (e.g. a partition too rich for the shard path on this rank: the ranks agree below)
        mark("cutter_count")
        try:
            comps = distributed_components(ctx, comm, cutter, k, b1, b2, device=device)
        except DistAbort as e:
            # all ranks are here together: the replicated cutter instead (every rank counts all unitigs and cuts all components)
            print("[metafast_amd] %s -- every rank builds the whole cutter table" % e, file=sys.stderr)
            if cutter is not None:
                cutter.close()
            cutter = None
        mark("components")
    if comps is None:
        # (world sizes that are not a power of two, k < 20, a rank that could not do its part: every rank builds the whole cutter table)
        ctx.set_option("union_samples", n_all)          # (planning hint: many samples share most of their unitig k-mers)
        try:
            cutter = ctx.count_device(av["bases"], av["offsets"], av["n"], av["n_bases"], k, l)
        finally:
            ctx.set_option("union_samples", 0)
        mark("cutter_count")
        comps = ctx.cut_components(cutter, b1, b2)
        mark("components")
    allr.close()
    vecs_local, breadths = [None] * len(goods), [None] * len(goods)

    def features_of(c, idx, err):
        # (on the context that owns the samples' tables: this one or its peer -- the peer's samples in a thread of their own beside these)
        try:
            if c is not ctx:
                c.bind_thread()
            for i in idx:
                vecs_local[i], breadths[i] = c.features(comps, goods[i], 0)
                if len(goods) > 1:
                    goods[i].drop_index()
        except BaseException as e:
            err.append(e)
    ferr, fth = [], []
    for c in ctxs[1:]:
        idx = [i for i, g in enumerate(goods) if g.ctx is c]
        if idx:
            fth.append(threading.Thread(target=features_of, args=(c, idx, ferr)))
            fth[-1].start()
    features_of(ctx, [i for i, g in enumerate(goods) if g.ctx is ctx], ferr)
    for t in fth:
        t.join()
    if ferr:
        raise ferr[0]
    rows = np.stack(vecs_local) if vecs_local else np.zeros((0, len(comps)), dtype=np.int64)
    vecs = comm.features_allgather(rows)                        # north_star's all-gather of the per-sample feature vectors
    matrix = L.bray_curtis(vecs) if vecs.shape[1] else np.zeros((vecs.shape[0], vecs.shape[0]))
    mark("features_matrix")
    comm_stats = dict(comm.stats(), kind=comm.kind)
    return dict(goods=goods, seqss=seqss, cutter=cutter, comps=comps, vecs_local=vecs_local, breadths=breadths, vecs=vecs,
                matrix=matrix, n_occ=n_occ, n_distinct=n_distinct, hists=hists, comm=comm_stats)


def run_samples_wide(ctx, samples, k=63, b=1, l=100, b1=1000, b2=10000, device="cuda", timings=None):
    """NO-REFERENCE EXTENSION (the reference rejects k > 31, src/tools/KmersCounterMain.java:66-73; BASELINE config 4 names a k = 63
    leg): the steps of _run_samples for 32 <= k <= 63 on THIS rank's samples -- counts with the cut inside the pass, unitigs, the cutter
    table over all unitigs, components, features, matrix (mf_wide.hip, mf_wgraph.hip).  samples: (d_bases, d_offsets, n_reads, n_bases)
    tuples of torch tensors in HBM.  Several ranks (one process per GPU, BASELINE config 4: a sample per GPU): the per-sample steps need no
    exchange; the unitigs of all ranks are gathered on every rank (mf_comm_gather_sequences: sequences are sequences whatever k is) and every
    rank cuts the components of the whole cutter table itself -- the REPLICATED cutter: a sharded one is not built for wide k-mers, their
    tables have no minimizer partitions to own --; the rows of all ranks' samples are all-gathered for the matrix (mf_features_allgather)."""
    t0 = time.perf_counter()

    def mark(name):
        nonlocal t0
        if timings is not None:
            ctx.synchronize()
            t1 = time.perf_counter()
            timings[name] = timings.get(name, 0.0) + (t1 - t0)
            t0 = t1

    goods, seqss, n_occ, n_distinct = [], [], 0, []
    for si, (d_bases, d_offsets, n_reads, n_bases) in enumerate(samples):
        if si:
            goods[-1].drop_index()
        good, nd = ctx.count_wide_above(d_bases.data_ptr(), d_offsets.data_ptr(), n_reads, n_bases, k, b)
        mark("count")
        seqss.append(ctx.build_unitigs_wide(good, b, l))
        mark("unitigs")
        goods.append(good); n_occ += good.stats()[1]; n_distinct.append(nd)
    if len(goods) > 1:
        goods[-1].drop_index()
    views = [sq.device_view() for sq in seqss]
    if len(views) == 1:
        v = views[0]
        sb, so, ns, nb = v["bases"], v["offsets"], v["n"], v["n_bases"]
        keep = None
    else:
        parts_b = [device_tensor(v["bases"], v["n_bases"], device) for v in views]
        parts_o, nb = [], 0
        for v in views:
            parts_o.append(device_tensor(v["offsets"], (v["n"] + 1) * 8, device).view(torch.int64)[:-1] + nb)
            nb += v["n_bases"]
        ctx.synchronize()
        tb = _with_room(ctx, lambda: torch.cat(parts_b + [torch.zeros(64, dtype=torch.uint8, device=device)]))
        to = _with_room(ctx, lambda: torch.cat(parts_o + [torch.tensor([nb], dtype=torch.int64, device=device)]))
        torch.cuda.current_stream().synchronize()
        keep = (tb, to)
        sb, so, ns = tb.data_ptr(), to.data_ptr(), int(to.numel()) - 1
    rank, world = _world()
    comm, allr = None, None
    if world > 1 or _force():
        comm = getattr(ctx, "_mf_comm", None)
        if comm is None or comm.h is None:
            comm = ctx._mf_comm = make_comm(ctx, device)
        comm.reset_stats()
        ctx.synchronize()
        allr = comm.gather_sequences(sb or 0, so or 0, ns, nb)
        av = allr.device_view()
        sb, so, ns, nb = av["bases"], av["offsets"], av["n"], av["n_bases"]
        mark("exchange_unitigs")
    cutter = ctx.count_wide_table(sb, so, ns, nb, k, l)
    if allr is not None:
        allr.close()
    mark("cutter_count")
    comps = ctx.cut_components_wide(cutter, b1, b2)
    cutter.drop_index()
    mark("components")
    vecs_local, breadths = [], []
    for good in goods:
        vec, br = ctx.features_wide(comps, good, 0)
        if len(goods) > 1:
            good.drop_index()
        vecs_local.append(vec); breadths.append(br)
    vecs_local = np.stack(vecs_local) if vecs_local else np.zeros((0, len(comps)), dtype=np.int64)
    vecs = comm.features_allgather(vecs_local) if comm is not None else vecs_local
    matrix = L.bray_curtis(vecs) if vecs.shape[1] else np.zeros((vecs.shape[0], vecs.shape[0]))
    mark("features_matrix")
    del keep
    return dict(goods=goods, seqss=seqss, cutter=cutter, comps=comps, vecs=vecs, vecs_local=vecs_local, breadths=breadths, matrix=matrix, n_occ=n_occ,
                n_distinct=n_distinct, comm=dict(comm.stats(), kind=comm.kind) if comm is not None else None)


def run_sample(ctx, d_bases, d_offsets, n_reads, n_bases, k=31, b=1, l=100, b1=1000, b2=10000, device="cuda",
               timings=None):
    """One sample on this rank's GPU (run_samples with a single sample; the result keys of one sample)."""
    r = run_samples(ctx, [(d_bases, d_offsets, n_reads, n_bases)], k=k, b=b, l=l, b1=b1, b2=b2, device=device, timings=timings)
    return dict(good=r["goods"][0], seqs=r["seqss"][0], cutter=r["cutter"], comps=r["comps"], vec=r["vecs_local"][0],
                breadth=r["breadths"][0], vecs=r["vecs"], matrix=r["matrix"], n_occ=r["n_occ"], n_distinct=r["n_distinct"],
                hist=r["hists"][0])

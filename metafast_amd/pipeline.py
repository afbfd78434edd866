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


class ThreadGroup:
    """W virtual ranks inside ONE process (one thread each, all on the same GPU): the distributed cutter end to end on a
    1-GPU box -- tests and tools/sim_union.py.  serial=True lets only one rank compute at a time (clean per-rank timings)."""

    def __init__(self, world, serial=True):
        import threading
        self.world, self.slots = world, [None] * world
        self.barrier = threading.Barrier(world)
        self.turn = threading.Lock() if serial else None


class ThreadComm:
    def __init__(self, group, rank):
        self.g, self.rank, self.world = group, rank, group.world
        self.waited = 0.0            # seconds spent waiting for the other ranks (not this rank's work)
        self.stats = dict(collectives=0, bytes_in=0)          # (as TorchComm.stats: what the protocol exchanged)
        if group.turn:
            group.turn.acquire()

    def done(self):
        if self.g.turn:
            self.g.turn.release()

    def _exchange(self, x):
        g = self.g
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        g.slots[self.rank] = x
        tw = time.perf_counter()
        if g.turn:
            g.turn.release()
        try:
            g.barrier.wait()
            out = list(g.slots)
            g.barrier.wait()
        finally:
            if g.turn:
                g.turn.acquire()
        self.waited += time.perf_counter() - tw
        return out

    def _timed(self, fn):
        """self.exchange: seconds this rank spent inside the exchanges (their copies; waiting for the others excluded)"""
        t0, w0 = time.perf_counter(), self.waited
        out = fn()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        self.exchange = getattr(self, "exchange", 0.0) + (time.perf_counter() - t0) - (self.waited - w0)
        self.stats["collectives"] += 1
        self.stats["bytes_in"] += int(out.numel() * out.element_size()) if torch.is_tensor(out) else int(out.nbytes)
        return out

    def all_gather_ints(self, vals):
        return self._timed(lambda: np.asarray(self._exchange([int(v) for v in vals]), dtype=np.int64).reshape(self.world, -1))

    def all_gather(self, t, sizes):
        return self._timed(lambda: torch.cat(self._exchange(t)))

    def all_reduce_min(self, t):
        return self._timed(lambda: torch.stack(self._exchange(t)).min(dim=0).values)

    def all_to_all(self, t, matrix):
        def f():
            parts = self._exchange(t)
            out = []
            for r in range(self.world):
                o = int(sum(matrix[r][:self.rank]))
                out.append(parts[r][o:o + int(matrix[r][self.rank])])
            return torch.cat(out)
        return self._timed(f)


    def all_to_all_v(self, t, send, recv):
        def f():
            parts = self._exchange((t, [int(x) for x in send]))
            out = []
            for r in range(self.world):
                pt, ps = parts[r]
                o = int(sum(ps[:self.rank]))
                assert ps[self.rank] == int(recv[r]), "all_to_all_v: rank %d sends %d, rank %d expects %d" % (r, ps[self.rank], self.rank, int(recv[r]))
                out.append(pt[o:o + ps[self.rank]])
            return torch.cat(out)
        return self._timed(f)


def _i64(n, device):
    return _alloc(lambda: torch.empty(max(int(n), 1), dtype=torch.int64, device=device))


class DistAbort(L.MetafastError):
    """a rank could not do its part of the sharded cutter; EVERY rank raises this at the same point of the protocol (the status
    rides on the integer gathers), so that all of them can take the replicated path together instead of one rank raising
    while its peers wait inside a collective"""


def distributed_components(ctx, comm, shard, k, b1, b2, device="cuda", timings=None, info=None):
    """Component cutter with every rank owning a shard of the cutter table (include/metafast_hip.h, "A9-A11 on several
    GPUs").  shard: this rank's Context.count_device_shard of all samples' unitigs (None: the count failed here -- the
    ranks then raise DistAbort together).  Returns the components: the same object on every rank, identical to
    cut_components on the whole table (ComponentsBuilder.splitStrategy, src/algo/ComponentsBuilder.java:24-32).

    Collectives: 1 + 3 once (shard sizes; query counts, queries, answers), then per threshold level 2 integer gathers (half
    pairs per destination + the level before's oversize count; records per rank) + 1 all-to-all (half pairs) + 2 all-gathers (completed pairs; per-component
    records) -- from the gathered records EVERY rank derives all kept components and the number of oversize ones itself
    (round 2: three more collectives per level for lists every rank could compute) --, and 2 + 1 + 1 at the end (members'
    k-mers and roots, their count, the components' smallest k-mers)."""
    W, me = comm.world, comm.rank
    t0, w0 = time.perf_counter(), getattr(comm, "waited", 0.0)
    err = []                                 # the first library error on this rank: announced with the next integer gather

    def buf(n, _device=None):
        """an exchange buffer of n int64; on a rank whose library call has failed it is ZEROED: the rank keeps the collectives going with
        buffers of the agreed sizes until the next status gather, and what its healthy peers read from them meanwhile must be in-range
        indices and ranks, not whatever the allocator left there"""
        return _alloc(lambda: torch.zeros(max(int(n), 1), dtype=torch.int64, device=device)) if err else _i64(n, device)

    def mark(name):
        nonlocal t0, w0
        if timings is not None:
            torch.cuda.synchronize()
            t1, w1 = time.perf_counter(), getattr(comm, "waited", 0.0)
            timings[name] = timings.get(name, 0.0) + (t1 - t0) - (w1 - w0)      # (virtual ranks: the others' turns are not this rank's time)
            t0, w0 = t1, w1

    # torch fills the exchange buffers on ITS current stream, the library reads and writes them on the context's: a stream synchronisation stands
    # between the two -- unless they are the same stream (bench.py, run_samples' callers: the context was made with torch's current stream), where
    # the order is the stream's own (round 5: eight host waits per threshold level less)
    same_stream = getattr(ctx, "stream_handle", None) is not None and torch.cuda.is_available() and ctx.stream_handle == int(torch.cuda.current_stream().cuda_stream)

    def sync():
        if not same_stream:
            torch.cuda.current_stream().synchronize()

    def call(fn, default=None):
        """a library call that may fail (memory on ONE rank, a capacity limit): after the first failure this rank only keeps the
        collectives going (buffers of the agreed sizes, contents irrelevant) until the next integer gather tells everybody"""
        if err:
            return default
        try:
            return fn()
        except L.MetafastError as e:
            err.append(e)
            return default

    def gather_ints(vals, n):
        """all_gather_ints with the status in front; vals: callable -> n integers"""
        v = call(lambda: [int(x) for x in vals()], None)
        m = comm.all_gather_ints([0 if err else 1] + (v if v is not None else [0] * n))
        if not m[:, 0].all():
            bad = [int(r) for r in np.nonzero(m[:, 0] == 0)[0]]
            raise DistAbort("sharded component cutter: rank(s) %s failed%s" % (bad, (": %s" % err[0]) if err else ""))
        return m[:, 1:]

    if shard is None:
        err.append(L.MetafastError("no shard"))
    ns = gather_ints(lambda: [len(shard)], 1)[:, 0]
    base = np.concatenate([[0], np.cumsum(ns)])
    if int(base[-1]) >= 0xFFFFFFFF:
        raise L.MetafastError("components: more than 2^32 vertices over all ranks is not supported")      # (every rank sees the same total)
    D = call(lambda: L.DistCutter(ctx, shard, me, W, base))
    try:
        # ---- neighbours in other shards
        qm = gather_ints(lambda: D.queries(), W)
        nq = int(qm[me].sum())
        q = buf(2 * nq, device); sync()
        call(lambda: D.queries_fill(q.data_ptr()))
        rq = comm.all_to_all(q[:2 * nq], 2 * qm); sync()
        na = int(rq.numel()) // 2
        a = buf(2 * na, device); sync()
        call(lambda: D.answer(rq.data_ptr(), na, a.data_ptr()))
        ra = comm.all_to_all(a[:2 * na], 2 * qm.T); sync()
        call(lambda: D.set_answers(ra.data_ptr(), nq))
        del q, rq, a, ra
        mark("cutter_adjacency")
        # ---- threshold levels (ComponentsBuilder.java:86-150)
        # Round 5: the sizes of a level's exchanges ride IN BAND where the level before bounds them.
        #  * per-component records: every rank's slice has the capacity 2 x its count of the level before + 1024 and says its real
        #    count in the slice's first record (a count beyond the capacity -- components can split -- is answered by ONE more
        #    all-gather with the sizes everybody has just read): the integer gather of round 4 is gone from every level but the first;
        #  * half pairs: a cross edge only ever dies, so a level's counts are bounded by the level before's.  Once a level's pairs are
        #    few (MF_DCC_INBAND_MAX, 2^20 = 8 MB over all ranks: the latency of a collective then costs more than the padding), the
        #    all-to-all and the all-gather of the pairs keep the split sizes of the last level that was counted, unused room is filled
        #    with a pair no kernel takes (0xFFFFFFFF, 0xFFFFFFFF), and the status every rank owes its peers travels in the slices' first
        #    element: no integer gather at all, 3 collectives per level (5 in round 4), and the host reads the result of a collective
        #    twice per level (the status, the records' counts) where it read it five times.
        inband_max = int(os.environ.get("MF_DCC_INBAND_MAX", str(1 << 20)))
        stat_mul, stat_add = (int(x) for x in os.environ.get("MF_DCC_STATS_ROOM", "2,1024").split(","))      # (tests: "0,0" makes every level overflow its room)
        SENT = -1                                                    # int64 view of the pair (0xFFFFFFFF, 0xFFFFFFFF)
        kept, levels, per_level = [], 0, []
        n_big = -1
        cap_send = cap_recv = cap_tot = None                         # split sizes of the last COUNTED level: to each rank / from each rank / every rank's total
        prev_nstat = None                                            # every rank's record count of the level before
        host_reads = 0

        def stats_exchange(n_stats):
            """all ranks' per-component records, rank order, contiguous; -> (tensor, counts per rank)"""
            nonlocal host_reads
            if prev_nstat is None:                                   # (the first level: nothing bounds the counts yet)
                sm = gather_ints(lambda: [n_stats], 1)[:, 0]
                st = buf(2 * n_stats, device); sync()
                call(lambda: D.stats_fill(st.data_ptr()))
                alls = comm.all_gather(st[:2 * n_stats], 2 * sm); sync()
                host_reads += 1
                return alls, sm
            caps = [stat_mul * int(x) + stat_add for x in prev_nstat]
            st = buf(2 * max(n_stats, 1), device); sync()
            call(lambda: D.stats_fill(st.data_ptr()))
            mine = _alloc(lambda: torch.zeros(2 * (caps[me] + 1), dtype=torch.int64, device=device))
            mine[0] = -1 if err else n_stats                         # (a failed rank says so here)
            m = min(n_stats, caps[me]) if not err else 0
            mine[2:2 + 2 * m] = st[:2 * m]
            allg = comm.all_gather(mine, [2 * (c + 1) for c in caps]); sync()
            starts = np.concatenate([[0], np.cumsum([2 * (c + 1) for c in caps])])
            sm = allg[torch.as_tensor(starts[:-1], device=allg.device)].cpu().numpy(); host_reads += 1
            if (sm < 0).any():
                raise DistAbort("sharded component cutter: rank(s) %s failed%s" % ([int(r) for r in np.nonzero(sm < 0)[0]], (": %s" % err[0]) if err else ""))
            if any(int(sm[r]) > caps[r] for r in range(W)):          # (rare: a level that splits one component into thousands) -- the sizes are known now
                alls = comm.all_gather(st[:2 * n_stats], 2 * sm); sync()
                return alls, sm
            alls = torch.cat([allg[int(starts[r]) + 2: int(starts[r]) + 2 + 2 * int(sm[r])] for r in range(W)]) if int(sm.sum()) else allg[:0]
            return alls, sm

        for thr in range(1, 1 << 16):
            inband = cap_tot is not None and int(np.sum(cap_tot)) <= inband_max and n_big > 0
            mode = "in-band" if inband else "counted"
            if not inband:
                # (the gather that opens a counted level also closes the one before: it carries that level's oversize count -- the same on
                # every rank -- and the status of the calls since the last gather, so all ranks leave the loop, or abort, together)
                pm = gather_ints(lambda: [n_big] + ([int(x) for x in D.level_local()] if n_big else [0] * W), W + 1)
                host_reads += 1
                if len(set(int(x) for x in pm[:, 0])) != 1:         # (every rank derives the count from the same gathered records)
                    raise DistAbort("sharded component cutter: the ranks disagree on a level's oversize components: %s" % [int(x) for x in pm[:, 0]])
                if int(pm[me][0]) == 0:
                    break
                pm = pm[:, 1:]
                nsend = int(pm[me].sum())
                hp = buf(nsend, device); sync()
                call(lambda: D.pairs_fill(hp.data_ptr()))
                rp = comm.all_to_all(hp[:nsend], pm); sync()
                nr = int(rp.numel())
                if nr:
                    rp = rp.contiguous()
                    call(lambda: D.pairs_complete(rp.data_ptr(), nr))
                allp = comm.all_gather(rp, pm.sum(axis=0)); sync()
                cap_send, cap_recv, cap_tot = pm[me].copy(), pm[:, me].copy(), pm.sum(axis=0)
            else:
                cnt = call(lambda: [int(x) for x in D.level_local()], [0] * W)
                nsend = int(sum(cnt))
                hp = buf(nsend, device); sync()
                call(lambda: D.pairs_fill(hp.data_ptr()))
                send = [int(c) + 1 for c in cap_send]               # (one status element in front of every slice)
                out = _alloc(lambda: torch.full((sum(send),), SENT, dtype=torch.int64, device=device))
                so, po = 0, 0
                for d in range(W):
                    c = cnt[d] if not err else 0
                    if c > int(cap_send[d]):
                        err.append(L.MetafastError("sharded component cutter: %d half pairs for rank %d, %d at the level before" % (c, d, int(cap_send[d]))))
                        c = 0
                    out[so] = 0 if err else 1
                    if c:
                        out[so + 1: so + 1 + c] = hp[po: po + c]
                    so += send[d]; po += cnt[d]
                if err:                                              # (a failure found while filling: every slice says so)
                    so = 0
                    for d in range(W):
                        out[so] = 0; so += send[d]
                recv = [int(c) + 1 for c in cap_recv]
                rp = comm.all_to_all_v(out, send, recv); sync()
                heads = np.concatenate([[0], np.cumsum(recv)])[:-1]
                ht = torch.as_tensor(heads, device=rp.device)
                status = rp[ht].cpu().numpy(); host_reads += 1
                if not (status == 1).all():
                    raise DistAbort("sharded component cutter: rank(s) %s failed%s" % ([int(r) for r in np.nonzero(status != 1)[0]], (": %s" % err[0]) if err else ""))
                rp = rp.contiguous()
                rp[ht] = SENT                                        # (the status elements become pairs nobody takes)
                nr = int(rp.numel())
                call(lambda: D.pairs_complete(rp.data_ptr(), nr))
                allp = comm.all_gather(rp, [int(c) + W for c in cap_tot]); sync()
            n_stats = call(lambda: D.merge(allp.data_ptr(), int(allp.numel())), 0)
            alls, sm = stats_exchange(n_stats)
            prev_nstat = sm
            seg = np.concatenate([[0], np.cumsum(sm)])
            alls = alls.contiguous()
            n_kept, n_big = call(lambda: D.classify(alls.data_ptr(), int(alls.numel()) // 2, seg, n_stats, b1, b2, thr, me), (0, 0))
            kb = buf(2 * n_kept, device); sync()
            call(lambda: D.kept_fill(kb.data_ptr()))
            allk = kb[:2 * n_kept].cpu().numpy()
            if allk.size:
                r = allk.reshape(-1, 2)
                r = r[np.argsort(r[:, 0] & 0xFFFFFFFF, kind="stable")]        # by root: the order every rank agrees on
                kept.append((r[:, 0] & 0xFFFFFFFF, (r[:, 0] >> 32) & 0xFFFFFFFF, r[:, 1], np.full(len(r), thr, dtype=np.int32)))
            levels = thr
            per_level.append((int(allp.numel()), int(sm.sum()), int(n_kept), int(n_big), mode))
            if inband and n_big == 0:
                break                                                # (every rank has derived the same count from the same records; the members' gather below carries the status)
        mark("cutter_levels")
        # ---- members of the kept components, everywhere
        # ---- members of the kept components, everywhere: 8 bytes per member (the k-mers, sorted by component on the rank) + one
        # (root, count) record per component and rank
        nm, nr = call(lambda: D.members_grouped(), (0, 0))
        mm = gather_ints(lambda: [nm, nr], 2)
        mk = buf(nm, device); mr = buf(nr, device); sync()
        call(lambda: D.members_grouped_fill(mk.data_ptr(), mr.data_ptr()))
        allmk = comm.all_gather(mk[:nm], mm[:, 0]); allmr = comm.all_gather(mr[:nr], mm[:, 1]); sync()
        cat = (lambda i, dt: np.concatenate([x[i] for x in kept]).astype(dt)) if kept else (lambda i, dt: np.zeros(0, dtype=dt))
        roots = cat(0, np.uint32)
        mn = buf(len(roots), device); sync()
        call(lambda: D.minkeys(roots, mn.data_ptr()))
        mn = comm.all_reduce_min(mn[:len(roots)]).cpu().numpy().astype(np.uint64)
        comps = call(lambda: D.finish_grouped(allmk.data_ptr(), int(allmk.numel()), allmr.data_ptr(), int(allmr.numel()), roots, cat(1, np.uint32),
                                              cat(2, np.int64), cat(3, np.int32), mn))
        gather_ints(lambda: [len(comps)], 1)          # (the last status: every rank has its components, or all raise)
        if info is not None:
            info.update(levels=levels, per_level=per_level, shard=int(ns[me]), vertices=int(base[-1]), queries=nq, members=int(allmk.numel()), host_reads_in_levels=host_reads,
                        collectives=int(comm.stats["collectives"]), MB_received=round(comm.stats["bytes_in"] / 1e6, 2))
    finally:
        if D is not None:
            D.close()
    mark("cutter_members")
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
    comm_stats = dict(collectives=0, bytes_in=0, seconds=0.0)
    for si, sample in enumerate(samples):
        if si:
            # several samples on this rank: the previous sample's lookup index (3-6 times its table) is not needed again before
            # its feature vector, where it is rebuilt (mf_table_drop_index) -- 4 samples of > 2^32 distinct k-mers each
            # (BASELINE config 5) fit one GPU's HBM this way
            goods[-1].drop_index()
        # kmer-counter: k-mers with count > b go on (IOUtils.printKmers); the others are dropped inside the counting kernels
        if isinstance(sample[0], (str, bytes, os.PathLike)):
            # a sample handed over as its read files (IOUtils.loadReads, src/io/IOUtils.java:772-803: all files of a library into one
            # table): read + parse + H2D inside the library (mf_count_reads_above)
            good, nd = ctx.count_reads_above([os.fspath(f) for f in sample], k, b)
        else:
            d_bases, d_offsets, n_reads, n_bases = sample
            good, nd = ctx.count_device_above(d_bases.data_ptr(), d_offsets.data_ptr(), n_reads, n_bases, k, b)
        # ... and the histogram of ALL counts, dropped k-mers included (the .stat.txt of IOUtils.printKmers, src/io/IOUtils.java:45-71)
        hists.append(good.hist())
        mark("count")
        seqss.append(ctx.build_unitigs(good, b, l))
        if si:
            good.drop_index()
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
        sb = _with_room(ctx, lambda: torch.cat(parts_b) if parts_b else torch.zeros(0, dtype=torch.uint8, device=device))
        so = _with_room(ctx, lambda: torch.cat(parts_o + [torch.tensor([nb], dtype=torch.int64, device=device)]))
    rank, world = _world()
    sharded = (world > 1 or _force()) and k >= 20 and world & (world - 1) == 0 and world <= 64 and not os.environ.get("MF_REPLICATED_CUTTER")
    if sharded:
        # every rank owns a shard of the cutter table and of the components step (distributed_components)
        allb, allo, ns, nbt = gather_sequences(sb, so)
        torch.cuda.current_stream().synchronize()
        mark("exchange_unitigs")
        comm = TorchComm()
        try:
            cutter = ctx.count_device_shard(allb.data_ptr(), allo.data_ptr(), ns, nbt, k, l, rank, world)
        except L.MetafastError:
            cutter = None                    # (e.g. a partition too rich for the shard path on this rank: the ranks agree below)
        mark("cutter_count")
        try:
            comps = distributed_components(ctx, comm, cutter, k, b1, b2, device=device, timings=timings)
        except DistAbort as e:
            # all ranks are here together: the replicated cutter instead (every rank counts all unitigs and cuts all components)
            print("[metafast_amd] %s -- every rank builds the whole cutter table" % e, file=sys.stderr)
            if cutter is not None:
                cutter.close()
            ctx.set_option("union_samples", int(comm.all_gather_ints([len(goods)]).sum()))
            try:
                cutter = ctx.count_device(allb.data_ptr(), allo.data_ptr(), ns, nbt, k, l)
            finally:
                ctx.set_option("union_samples", 0)
            comps = ctx.cut_components(cutter, b1, b2)
        comm_stats = comm.stats
        t0 = time.perf_counter()
    else:
        # (world sizes that are not a power of two, k < 20: every rank builds the whole cutter table and all components)
        allb, allo, ns, nbt = gather_sequences(sb, so)
        if torch.cuda.is_available():
            torch.cuda.current_stream().synchronize()
        mark("exchange_unitigs")
        n_all = int(TorchComm().all_gather_ints([len(goods)]).sum())
        ctx.set_option("union_samples", n_all)          # (planning hint: many samples share most of their unitig k-mers)
        try:
            cutter = ctx.count_device(allb.data_ptr(), allo.data_ptr(), ns, nbt, k, l)
        finally:
            ctx.set_option("union_samples", 0)
        mark("cutter_count")
        comps = ctx.cut_components(cutter, b1, b2)
        mark("components")
    vecs_local, breadths = [], []
    for good in goods:
        vec, breadth = ctx.features(comps, good, 0)
        if len(goods) > 1:
            good.drop_index()
        vecs_local.append(vec); breadths.append(breadth)
    vt = torch.from_numpy(np.stack(vecs_local) if vecs_local else np.zeros((0, len(comps)), dtype=np.int64)).to(device)
    vecs = gather_vector_rows(vt).cpu().numpy()
    matrix = L.bray_curtis(vecs) if vecs.shape[1] else np.zeros((vecs.shape[0], vecs.shape[0]))
    mark("features_matrix")
    return dict(goods=goods, seqss=seqss, cutter=cutter, comps=comps, vecs_local=vecs_local, breadths=breadths, vecs=vecs,
                matrix=matrix, n_occ=n_occ, n_distinct=n_distinct, hists=hists, comm=comm_stats)


def run_samples_wide(ctx, samples, k=63, b=1, l=100, b1=1000, b2=10000, device="cuda", timings=None):
    """NO-REFERENCE EXTENSION (the reference rejects k > 31, src/tools/KmersCounterMain.java:66-73; BASELINE config 4 names a k = 63
    leg): the steps of _run_samples for 32 <= k <= 63 on THIS rank's samples -- counts with the cut inside the pass, unitigs, the cutter
    table over all unitigs, components, features, matrix (mf_wide.hip, mf_wgraph.hip).  samples: (d_bases, d_offsets, n_reads, n_bases)
    tuples of torch tensors in HBM.  One rank only: the per-sample steps need no exchange, the join of several ranks' unitigs is not
    built for wide k-mers."""
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
    cutter = ctx.count_wide_table(sb, so, ns, nb, k, l)
    mark("cutter_count")
    comps = ctx.cut_components_wide(cutter, b1, b2)
    cutter.drop_index()
    mark("components")
    vecs, breadths = [], []
    for good in goods:
        vec, br = ctx.features_wide(comps, good, 0)
        if len(goods) > 1:
            good.drop_index()
        vecs.append(vec); breadths.append(br)
    vecs = np.stack(vecs) if vecs else np.zeros((0, len(comps)), dtype=np.int64)
    matrix = L.bray_curtis(vecs) if vecs.shape[1] else np.zeros((vecs.shape[0], vecs.shape[0]))
    mark("features_matrix")
    del keep
    return dict(goods=goods, seqss=seqss, cutter=cutter, comps=comps, vecs=vecs, breadths=breadths, matrix=matrix, n_occ=n_occ,
                n_distinct=n_distinct)


def run_sample(ctx, d_bases, d_offsets, n_reads, n_bases, k=31, b=1, l=100, b1=1000, b2=10000, device="cuda",
               timings=None):
    """One sample on this rank's GPU (run_samples with a single sample; the result keys of one sample)."""
    r = run_samples(ctx, [(d_bases, d_offsets, n_reads, n_bases)], k=k, b=b, l=l, b1=b1, b2=b2, device=device, timings=timings)
    return dict(good=r["goods"][0], seqs=r["seqss"][0], cutter=r["cutter"], comps=r["comps"], vec=r["vecs_local"][0],
                breadth=r["breadths"][0], vecs=r["vecs"], matrix=r["matrix"], n_occ=r["n_occ"], n_distinct=r["n_distinct"],
                hist=r["hists"][0])

"""N>1 path on CPU: the two exchanges of the multi-GPU pipeline (ragged all-gather of unitigs, all-gather of feature
vectors) under world_size-2 gloo."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from metafast_amd import pipeline as P
    # rank r owns r+2 "unitigs" of different lengths (rank 1 has more bases than rank 0)
    seqs = [("ACGT" * (3 + rank + i))[: 10 + 7 * rank + i] for i in range(rank + 2)]
    bases = torch.from_numpy(np.frombuffer("".join(seqs).encode(), dtype=np.uint8).copy())
    offs = torch.tensor(np.concatenate([[0], np.cumsum([len(s) for s in seqs])]), dtype=torch.int64)
    allb, allo, ns, nb = P.gather_sequences(bases, offs)
    vec = torch.tensor([10 * rank + 1, 7, 100 * (rank + 1)], dtype=torch.int64)
    vecs = P.gather_vectors(vec)
    rag = P.all_gather_ragged(torch.arange(rank * 3, dtype=torch.int64))     # rank 0 contributes an EMPTY tensor
    rows = P.gather_vector_rows(torch.full((rank + 1, 3), rank + 1, dtype=torch.int64))       # rank r holds r + 1 samples
    np.save(os.path.join(out_dir, f"w{rank}.npy"), rows.numpy())
    same = P.all_gather_ragged(torch.arange(4, dtype=torch.int64) + 10 * rank)                # equal sizes: one collective
    np.save(os.path.join(out_dir, f"s{rank}.npy"), torch.cat(same).numpy())
    # the three ways a ragged gather travels: equal sizes (above), nearly equal (one padded collective), far apart (a broadcast per rank)
    near = P.all_gather_ragged(torch.arange(100000 + 50 * rank, dtype=torch.int64) * (rank + 1))
    far = P.all_gather_ragged(torch.arange(100 + 200000 * rank, dtype=torch.int64) + rank)
    np.save(os.path.join(out_dir, f"n{rank}.npy"), np.array([int(x.sum()) for x in near] + [len(x) for x in near] + [int(x.sum()) for x in far] + [len(x) for x in far]))
    # the exchanges of the sharded cutter (pipeline.TorchComm), on host tensors
    comm = P.TorchComm()
    ints = comm.all_gather_ints([rank + 1, 10 * rank])
    m = np.array([[1, 2], [3, 0]])                                                # matrix[src][dst] elements
    a2a = comm.all_to_all(torch.arange(int(m[rank].sum()), dtype=torch.int64) + 100 * rank, m)
    ag = comm.all_gather(torch.full((3 + 2 * rank,), rank + 7, dtype=torch.int32), [3, 5])
    mn = comm.all_reduce_min(torch.tensor([5 - rank, 3 + rank, 0x7FFFFFFFFFFFFFFF], dtype=torch.int64))
    np.save(os.path.join(out_dir, f"c{rank}.npy"), np.array(ints.reshape(-1).tolist() + a2a.tolist() + ag.tolist() + mn.tolist(), dtype=np.int64))
    np.save(os.path.join(out_dir, f"b{rank}.npy"), allb.numpy())
    np.save(os.path.join(out_dir, f"o{rank}.npy"), allo.numpy())
    np.save(os.path.join(out_dir, f"v{rank}.npy"), vecs.numpy())
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.array([len(x) for x in rag]))
    dist.destroy_process_group()


def test_exchanges_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    exp_seqs = []
    for rank in range(world):
        exp_seqs += [("ACGT" * (3 + rank + i))[: 10 + 7 * rank + i] for i in range(rank + 2)]
    exp_b = "".join(exp_seqs).encode()
    exp_o = np.concatenate([[0], np.cumsum([len(s) for s in exp_seqs])])
    for rank in range(world):
        b = np.load(tmp_path / f"b{rank}.npy")
        o = np.load(tmp_path / f"o{rank}.npy")
        v = np.load(tmp_path / f"v{rank}.npy")
        r = np.load(tmp_path / f"r{rank}.npy")
        assert bytes(b[: len(exp_b)]) == exp_b and len(b) == len(exp_b) + 64     # 64 bytes of slack for the kernels
        assert o.tolist() == exp_o.tolist()                                      # every rank sees the same, rebased offsets
        assert v.tolist() == [[1, 7, 100], [11, 7, 200]]
        assert r.tolist() == [0, 3]
        assert np.load(tmp_path / f"w{rank}.npy").tolist() == [[1, 1, 1], [2, 2, 2], [2, 2, 2]]
        assert np.load(tmp_path / f"s{rank}.npy").tolist() == [0, 1, 2, 3, 10, 11, 12, 13]
        # rank 0 receives [1 from rank 0 | 3 from rank 1], rank 1 receives [2 from rank 0 | nothing]
        want_a2a = [0, 100, 101, 102] if rank == 0 else [1, 2]
        assert np.load(tmp_path / f"c{rank}.npy").tolist() == [1, 0, 2, 10] + want_a2a + [7] * 3 + [8] * 5 + [4, 3, 0x7FFFFFFFFFFFFFFF]
        a0, a1, f0, f1 = np.arange(100000), np.arange(100050) * 2, np.arange(100), np.arange(200100) + 1
        assert np.load(tmp_path / f"n{rank}.npy").tolist() == [a0.sum(), a1.sum(), 100000, 100050, f0.sum(), f1.sum(), 100, 200100]


def _worker4(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from metafast_amd import pipeline as P
    comm = P.TorchComm()
    # matrix[src][dst]: non-uniform, rank 2 neither sends nor receives anything, rank 3 only sends
    m = np.array([[0, 5, 0, 2], [1, 0, 0, 7], [0, 0, 0, 0], [4, 3, 0, 0]])
    send = torch.arange(int(m[rank].sum()), dtype=torch.int64) + 1000 * rank
    got = comm.all_to_all(send, m)
    # the same exchange where a rank knows only its own row and column (round 5: the split sizes of a level ride in band)
    got_v = comm.all_to_all_v(send, m[rank], m[:, rank])
    assert got_v.tolist() == got.tolist()
    sizes = [3, 0, 4, 1]                                                     # an all-gather with an EMPTY contribution (rank 1)
    ag = comm.all_gather(torch.full((sizes[rank],), rank, dtype=torch.int32), sizes)
    ints = comm.all_gather_ints([rank, 1 if rank != 2 else 0])
    mn = comm.all_reduce_min(torch.tensor([10 + rank, 10 - rank], dtype=torch.int64))
    seqs = ["ACGTACGTAC"[: 4 + rank]] * (rank % 2)                           # ranks 0 and 2 have NO unitigs
    bases = torch.from_numpy(np.frombuffer("".join(seqs).encode(), dtype=np.uint8).copy())
    offs = torch.tensor(np.concatenate([[0], np.cumsum([len(x) for x in seqs])]), dtype=torch.int64)
    allb, allo, ns, nb = P.gather_sequences(bases, offs)
    np.save(os.path.join(out_dir, f"q{rank}.npy"), np.array(got.tolist() + [-1] + ag.tolist() + [-1] + ints.reshape(-1).tolist() + [-1] + mn.tolist()
                                                             + [-1, ns, nb] + allo.tolist() + [comm.stats["collectives"]], dtype=np.int64))
    dist.destroy_process_group()


def test_exchanges_world4_nonuniform_and_empty_ranks(tmp_path):
    world = 4
    mp.spawn(_worker4, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    m = np.array([[0, 5, 0, 2], [1, 0, 0, 7], [0, 0, 0, 0], [4, 3, 0, 0]])
    for rank in range(world):
        want = []
        for src in range(world):                                             # what src sent to `rank`: the slice of its payload behind the earlier destinations
            o = int(m[src][:rank].sum())
            want += [1000 * src + o + i for i in range(int(m[src][rank]))]
        want += [-1] + [0, 0, 0, 2, 2, 2, 2, 3] + [-1] + [0, 1, 1, 1, 2, 0, 3, 1] + [-1] + [10, 7] + [-1, 2, 12, 0, 5, 12]
        got = np.load(tmp_path / f"q{rank}.npy").tolist()
        assert got[:-1] == want, (rank, got, want)
        assert got[-1] >= 5                                                  # (the exchanges were counted)


def test_single_process_paths():
    from metafast_amd import pipeline as P
    t = torch.arange(5)
    assert P.all_gather_ragged(t)[0] is t
    assert P.gather_vectors(torch.tensor([1, 2])).tolist() == [[1, 2]]
    b, o, ns, nb = P.gather_sequences(torch.zeros(6, dtype=torch.uint8), torch.tensor([0, 2, 6]))
    assert (ns, nb, o.tolist()) == (2, 6, [0, 2, 6])

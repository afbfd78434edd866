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
torch.cuda.set_device(0)
dist.init_process_group(os.environ["MF_BACKEND"], rank=rank, world_size=world, **({"device_id": torch.device("cuda", 0)} if os.environ["MF_BACKEND"] == "nccl" else {}))
ctx = L.Context(0, stream=torch.cuda.current_stream())
spg = int(os.environ.get("MF_SPG", "1"))
def samples():
    for j in range(spg):
        b, o = branchy_reads(107 + 10 * (rank * spg + j), genome_seed=7, n=6000)
        db, do = to_device(b, o)
        yield db, do, len(o) - 1, len(b)
r = P.run_samples(ctx, samples(), k=31, b=1, l=100, b1=100, b2=1000)
comps = r["comps"].export()
out = dict(components=[[int(a), int(w), int(t)] for a, w, t, _ in comps], members=[[int(x) for x in km] for _, _, _, km in comps],
           vecs=r["vecs"].tolist(), matrix=r["matrix"].tolist())
json.dump(out, open(os.path.join(os.environ["MF_OUT"], f"rank{rank}.json"), "w"))
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, backend, tmp_path, spg=1, extra_env=None):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
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


def test_two_ranks_two_samples_each(oracle, tmp_path):
    """more samples than ranks (BASELINE config 5's shape): every rank takes two samples; 4 x 4 matrix in rank-major order"""
    res = _run(2, "gloo", tmp_path, spg=2)
    want = _oracle_pipeline(oracle, tmp_path, [107, 117, 127, 137])
    for r in res:
        _same(r, want)


def test_rccl_path_world1(oracle, tmp_path):
    """MF_FORCE_DIST=1: the exchanges go through RCCL ("nccl" backend) although the world has one rank"""
    res = _run(1, "nccl", tmp_path, extra_env={"MF_FORCE_DIST": "1"})
    want = _oracle_pipeline(oracle, tmp_path, [107])
    _same(res[0], want)

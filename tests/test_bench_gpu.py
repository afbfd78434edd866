"""bench.py's contract on a small workload: ONE JSON line on stdout with the fields the driver reads, plain and under the
launcher the driver uses for N > 1 (here with one rank and MF_FORCE_DIST=1, so the RCCL path of the pipeline runs)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

FIELDS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
          "data", "config", "roofline", "cpu_baseline")


def _check(out, n_gpus):
    lines = [ln for ln in out.strip().splitlines() if ln.strip()]
    assert len(lines) == 1, lines[:3]
    d = json.loads(lines[0])
    for f in FIELDS:
        assert f in d, f
    assert d["metric"] == "k-mers/s counted+graphed at k=31, 150 bp reads" and d["unit"] == "k-mers/s" and d["n_gpus"] == n_gpus
    assert d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "u64" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "lds+valu") and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert d["value"] > 0 and abs(d["value"] - d["stats"]["n_occ"] / (d["ms_per_step"] / 1e3)) / d["value"] < 1e-3
    return d


def test_bench_line_small():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--reads", "300000", "--steps", "2", "--warmup", "1", "--cpu-sample-reads", "20000",
                        "--cpu-count-only-reads", "20000"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _check(p.stdout, 1)
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["with_reader"]["reads"] == 20000 and c["count_only"]["value"] > 0


def test_bench_under_the_launcher_rccl_world1():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MF_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "1", "--reads", "300000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _check(p.stdout, 1)
    assert "cutter_adjacency" in d["stage_ms_per_step"]           # the sharded cutter ran (through RCCL, one rank)

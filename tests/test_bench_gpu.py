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
    """... and the two lines that have the reader in them: `end_to_end` (FASTA file -> matrix in one process) and
    `cli` (metafast.sh on two samples through the reference's files, seconds per tool from the workDir's log)"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--reads", "300000", "--steps", "2", "--warmup", "1", "--cpu-sample-reads", "20000",
                        "--cpu-count-only-reads", "20000", "--genome-scale", "4000", "--b1", "50", "--b2", "5000", "--e2e-reads", "200000", "--cli-reads", "150000"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _check(p.stdout, 1)
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["with_reader"]["reads"] == 20000 and c["count_only"]["value"] > 0
    e = d["end_to_end"]
    assert e["unit"] == "k-mers/s" and e["reads"] == 200000 and e["value"] > 0 and abs(e["value"] - 200000 * 120 / e["seconds"]) / e["value"] < 2e-2
    c = d["cli"]
    assert "error" not in c, c
    assert c["unit"] == "k-mers/s" and c["samples"] == 2 and c["reads_per_sample"] == 150000 and c["value"] > 0
    assert abs(c["value"] - 2 * 150000 * 120 / c["seconds"]) / c["value"] < 2e-2
    for tool in ("kmer-counter-many", "seq-builder-many", "component-cutter", "features-calculator"):
        assert tool in c["step_seconds"], c["step_seconds"]
    assert e["value"] < d["value"] * 1.5 and c["value"] < d["value"]              # neither is the headline figure


def test_bench_under_the_launcher_rccl_world1():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MF_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "1", "--reads", "300000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-end-to-end"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _check(p.stdout, 1)
    # the sharded cutter ran, its exchanges inside the library on the library's own RCCL communicator (one rank)
    assert d["comm"]["kind"] == "rccl" and d["comm"]["collectives_per_step"] >= 20 and "components" in d["stage_ms_per_step"]

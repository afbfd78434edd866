"""End-to-end time of the metafast.sh-compatible driver on synthetic FASTA files (file-level seams: every step reads and
writes the reference's files): python3 tools/cli_rate.py [samples] [reads per sample]"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metafast_amd import lib as L
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = []
for s in range(ns):
    b, o = L.synth_reads_host(0x4D45544146415354, s, 0, n, 150, 1_000_000)
    path = "/tmp/mf_cli_%d.fa" % s
    rows = b.reshape(n, 150)
    out = np.empty((n, 3 + 150 + 1), dtype=np.uint8)
    out[:, :3] = np.frombuffer(b">r\n", dtype=np.uint8); out[:, 3:153] = rows; out[:, 153] = 10
    out.tofile(path)
    files.append(path)
    print(path, os.path.getsize(path) / 1e9, "GB", flush=True)
wd = "/tmp/mf_cli_wd"
os.environ["MF_IO_TIMING"] = "1"
for variant in os.environ.get("MF_VARIANTS", "").split(";"):        # MF_VARIANTS="a=1,b=2;c=3": one extra run per MF_OPTIONS value (the last run is the default)
    if not variant:
        continue
    subprocess.run(["rm", "-rf", wd])
    t0 = time.perf_counter()
    r = subprocess.run([os.path.join(root, "metafast.sh"), "-k", "31", "-i", *files, "-w", wd], capture_output=True, text=True, cwd="/tmp", env=dict(os.environ, MF_OPTIONS=variant))
    print("MF_OPTIONS=%s: exit %d total %.2f s" % (variant, r.returncode, time.perf_counter() - t0))
    print("\n".join(l for l in r.stderr.splitlines() if "write_" in l or "count_reads (1" in l or "driver" in l))
subprocess.run(["rm", "-rf", wd])
t0 = time.perf_counter()
r = subprocess.run([os.path.join(root, "metafast.sh"), "-k", "31", "-i", *files, "-w", wd, "-v"], capture_output=True, text=True, cwd="/tmp")
t1 = time.perf_counter()
print("exit", r.returncode, "total %.2f s" % (t1 - t0))
log = open(os.path.join(wd, "log")).read().splitlines()
for l in log:
    if any(w in l for w in ("Running tool", "printed to", "Sequences printed", "Components saved", "Features for", "matrix")):
        print(l)
print("\n".join(l for l in r.stderr.splitlines() if l.startswith("[mf]")))

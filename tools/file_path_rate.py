"""End-to-end rate of the file-level seam (host parse + H2D + count) vs the device-resident path: python3 tools/file_path_rate.py [reads]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from metafast_amd import lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
ctx = L.Context(0, stream=torch.cuda.current_stream()); ctx.set_option("verbose", 1)
b, o = L.synth_reads_host(0x4D45544146415354, 0, 0, n, 150, 1_000_000)
path = "/tmp/mf_rate.fa"
with open(path, "wb") as f:
    rows = b.reshape(n, 150)
    hdr = np.frombuffer(b">r\n", dtype=np.uint8)
    out = np.empty((n, 3 + 150 + 1), dtype=np.uint8)
    out[:, :3] = hdr; out[:, 3:153] = rows; out[:, 153] = 10
    f.write(out.tobytes())
sz = os.path.getsize(path)
for rep in range(2):
    t0 = time.perf_counter(); t = ctx.count_reads([path], 31); ctx.synchronize(); t1 = time.perf_counter()
    occ = t.occurrences(); t.close()
    print(f"file path : {sz/1e9:.2f} GB FASTA, {occ} k-mers in {t1-t0:.3f} s = {occ/(t1-t0):.3e} k-mers/s, {sz/(t1-t0)/1e9:.2f} GB/s of FASTA")
tb = torch.zeros(len(b) + 64, dtype=torch.uint8, device="cuda"); to = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
t0 = time.perf_counter(); tb[: len(b)] = torch.from_numpy(b); to.copy_(torch.from_numpy(o.view(np.int64))); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"H2D only  : {len(b)/1e9:.2f} GB in {t1-t0:.3f} s = {len(b)/(t1-t0)/1e9:.1f} GB/s (pageable)")
for rep in range(2):
    t0 = time.perf_counter(); t = ctx.count_device(tb.data_ptr(), to.data_ptr(), n, len(b), 31, 0); ctx.synchronize(); t1 = time.perf_counter()
    print(f"device path: {t.occurrences()/(t1-t0):.3e} k-mers/s"); t.close()

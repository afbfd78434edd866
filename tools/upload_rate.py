"""The device parser's upload (file in the page cache -> HBM) under its options: python3 tools/upload_rate.py [reads]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from metafast_amd import lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
ctx = L.Context(0, stream=torch.cuda.current_stream())
b, o = L.synth_reads_host(0x4D45544146415354, 0, 0, n, 150, 1_000_000)
path = "/tmp/mf_up.fa"
out = np.empty((n, 3 + 150 + 1), dtype=np.uint8)
out[:, :3] = np.frombuffer(b">r\n", dtype=np.uint8); out[:, 3:153] = b.reshape(n, 150); out[:, 153] = 10
out.tofile(path); del out
sz = os.path.getsize(path)
import ctypes as C
ctx.close()
for name, opts in (("16 MB x 16", dict(device_parse_piece_bytes=16 << 20, device_parse_threads=16)), ("8 MB x 16", dict(device_parse_piece_bytes=8 << 20, device_parse_threads=16)),
                   ("8 MB x 12", dict(device_parse_piece_bytes=8 << 20, device_parse_threads=12)), ("8 MB x 8", dict(device_parse_piece_bytes=8 << 20, device_parse_threads=8)),
                   ("4 MB x 16", dict(device_parse_piece_bytes=4 << 20, device_parse_threads=16)), ("4 MB x 8", dict(device_parse_piece_bytes=4 << 20, device_parse_threads=8)),
                   ("16 MB x 8", dict(device_parse_piece_bytes=16 << 20, device_parse_threads=8)),
                   ("pinned 8 MB x 8", dict(host_pinned=1, device_parse_piece_bytes=8 << 20, device_parse_threads=8)),
                   ("host parser", dict(device_parse=0))):
    ctx = L.Context(0, stream=torch.cuda.current_stream())        # (a context of its own: the FIRST load pays for the staging pool's pages, as a short process does)
    for k, v in opts.items():
        ctx.set_option(k, v)
    ts = []
    for rep in range(3):
        h = C.c_void_p()
        t0 = time.perf_counter()
        L._check(L.lib().mf_reads_load(ctx.h, L._cfiles([path]), 1, C.byref(h)))
        ctx.synchronize()
        ts.append(time.perf_counter() - t0)
        L.lib().mf_reads_destroy(h)
    print(f"{name:16s}: {sz/1e9:.2f} GB FASTA -> reads in HBM: first load {ts[0]:.3f} s = {sz/ts[0]/1e9:.1f} GB/s, then {min(ts[1:]):.3f} s = {sz/min(ts[1:])/1e9:.1f} GB/s", flush=True)
    ctx.close()

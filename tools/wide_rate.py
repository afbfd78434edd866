"""k = 63 (no-reference extension) beyond toy size: canonical 63-mer counts of one synthetic sample, seconds and k-mers/s:
python3 tools/wide_rate.py [reads] [k]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json
import torch
from metafast_amd import lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 63
rl = 150
ctx = L.Context(0, stream=torch.cuda.current_stream()); ctx.set_option("profile", 1); ctx.set_option("verbose", int(os.environ.get("MF_VERBOSE", "0"))); ctx.set_option("wide_ablate", int(os.environ.get("MF_WIDE_DEBUG", "0")))
bases = torch.zeros(n * rl + 64, dtype=torch.uint8, device="cuda"); offsets = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
ctx.synth_reads_device(0x4D45544146415354, 0, 0, n, rl, 1_000_000, bases.data_ptr(), offsets.data_ptr())
torch.cuda.synchronize()
import ctypes as C
t = C.c_void_p()
res = []
for it in range(2):
    ctx.reset_timers()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    L._check(L.lib().mf_count_wide_device(ctx.h, C.c_void_p(bases.data_ptr()), C.c_void_p(offsets.data_ptr()), n, n * rl, k, 0, C.byref(t)))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    nd, occ, kk = C.c_uint64(), C.c_uint64(), C.c_int()
    L._check(L.lib().mf_wtable_stats(t, C.byref(nd), C.byref(occ), C.byref(kk)))
    L.lib().mf_wtable_destroy(t)
    res.append(dict(seconds=round(dt, 3), n_occ=occ.value, n_distinct=nd.value, kmers_per_s=round(occ.value / dt, 1),
                    hipmalloc=dict(calls=ctx.stat('hipmalloc_calls'), GB=round(ctx.stat('hipmalloc_bytes') / 1e9, 1), seconds=round(ctx.stat('hipmalloc_us') / 1e6, 3)),
                    kernels={kk_: round(v[1], 1) for kk_, v in ctx.kernel_report().items()}))
print(json.dumps(dict(what="NO-REFERENCE EXTENSION: canonical %d-mer counts of one sample of %d synthetic 150 bp reads on one MI355X (mf_count_wide_device: class-range passes, radix passes over the leading 32 bits, buckets ordered in LDS)" % (k, n),
                      reads=n, k=k, runs=res)))

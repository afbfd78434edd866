"""The hand-written radix sort (mf_sort.hip) against torch.sort on the same device: python3 tools/sort_rate.py [elements]"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from metafast_amd import lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
ctx = L.Context(0, stream=torch.cuda.current_stream()); ctx.set_option("profile", 1)
fn = L.lib().mf_debug_sort
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]
rng = np.random.default_rng(1)
for kind, kt, vt, bits, what in ((0, np.uint64, np.uint16, 62, "(k-mer, count): .kmers.bin order"), (3, np.uint64, np.uint32, 64, "(start k-mer, index): unitig order"),
                                 (1, np.uint32, np.uint32, 20, "(partition, position): a loaded table's partitions"), (4, np.uint64, np.uint64, 64, "(low word, high word): k = 63")):
    keys = (rng.integers(0, 1 << 63, size=n, dtype=np.uint64) >> np.uint64(64 - bits if bits < 64 else 0)).astype(kt)
    vals = np.arange(n).astype(vt)
    ko, vo = np.empty_like(keys), np.empty_like(vals)
    ctx.reset_timers()
    assert fn(ctx.h, kind, keys.ctypes.data, vals.ctypes.data, n, bits, ko.ctypes.data, vo.ctypes.data) == 0
    calls, ms = ctx.kernel_time("k_radix_sort")
    assert np.all(ko[1:] >= ko[:-1])
    tk = torch.from_numpy(keys.astype(np.int64) if kt == np.uint64 else keys.astype(np.int64)).cuda()
    torch.cuda.synchronize(); t0 = time.perf_counter(); sk, si = torch.sort(tk, stable=True); torch.cuda.synchronize(); tt = time.perf_counter() - t0
    print(f"{what}: {n} elements, {bits} bits: mf_sort {ms:.1f} ms = {n / ms / 1e6:.2f} G elements/s; torch.sort of the keys alone (a library sort): {1e3 * tt:.1f} ms", flush=True)
    del tk, sk, si

"""Partition plans / variants of the hash-count kernel on one synthetic sample:
python3 tools/count_ab.py [reads] [k] [pt[:option=value]*,...]      e.g. 6144,6144:count_group=1,12288
Every variant must give the same table (order-independent checksum over (k-mer, count)) and the same count histogram."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metafast_amd import lib as L
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
variants = []
for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["3072", "6144", "12288"]):
    f = v.split(":")
    variants.append((int(f[0]), [(o.split("=")[0], int(o.split("=")[1])) for o in f[1:]]))
defaults = {}
rl = 150
hip = C.CDLL("libamdhip64.so")
ctx = L.Context(0, stream=torch.cuda.current_stream())
bases = torch.zeros(n_reads * rl + 64, dtype=torch.uint8, device="cuda")
offsets = torch.zeros(n_reads + 1, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
ctx.synth_reads_device(0x4D45544146415354, 0, 0, n_reads, rl, 1_000_000, bases.data_ptr(), offsets.data_ptr())
ctx.set_option("profile", 1)


def checksum(t):
    kp, cp, n = t.device_view()
    if not n:
        return 0, 0
    keys = torch.empty(n, dtype=torch.int64, device="cuda")
    cnts = torch.empty(n, dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    assert hip.hipMemcpy(C.c_void_p(keys.data_ptr()), C.c_void_p(kp), C.c_size_t(n * 8), 3) == 0
    assert hip.hipMemcpy(C.c_void_p(cnts.data_ptr()), C.c_void_p(cp), C.c_size_t(n * 2), 3) == 0
    h = (keys * -7046029254386353131) ^ (keys >> 29)
    h = h * (cnts.to(torch.int64) * 2 + 1)
    return int(h.sum().item()), int(cnts.to(torch.int64).sum().item())


ref = None
for pt, opts in variants:
    for name in defaults:
        ctx.set_option(name, -1 if name in ("l1_bits", "l2_bits") else 0)
    for name, val in opts:
        defaults[name] = 0
        ctx.set_option(name, val)
    ab = " ".join(f"{n}={v}" for n, v in opts)
    ctx.set_option("verbose", int(os.environ.get("MF_VERBOSE", "1")))
    ctx.set_option("part_target", pt)
    for rep in range(2):
        ctx.reset_timers()
        t, n_all = ctx.count_device_above(bases.data_ptr(), offsets.data_ptr(), n_reads, n_reads * rl, k, 1)
        torch.cuda.synchronize()
        rep_k = ctx.kernel_report()
        if rep == 0:
            t.close()
    cs = checksum(t)
    h = t.hist()
    sig = (len(t), n_all, cs, int(h[1]), int(h[2]), int(h.sum()), int((h * torch.arange(32768).numpy()).sum() % (1 << 61)))
    t.close()
    tot = sum(v[1] for v in rep_k.values())
    print(f"part_target={pt} {ab}: k_skm_count {rep_k.get('k_skm_count', (0, 0))[1]:.2f} ms, all kernels {tot:.1f} ms  "
          + " ".join(f"{n}={v[1]:.1f}" for n, v in sorted(rep_k.items(), key=lambda kv: -kv[1][1])[:6]), flush=True)
    print("   signature", sig, "OK" if ref is None or sig == ref else "MISMATCH", flush=True)
    if ref is None:
        ref = sig

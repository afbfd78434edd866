"""Which share of a k-mer's graph neighbours lives in ANOTHER minimizer partition -- the lookups of k_ut_flags_part / k_cc_adjacency_part that
leave the wave's LDS table and go to the HBM index -- as a function of k and of the minimizer length M (the build has M = 15 compiled in:
a 21-mer has 7 15-mers where a 31-mer has 17).  Measured on the kept k-mers (count > 1) of a synthetic sample, in numpy, with the library's
own M-mer order (mf_mmer_hash, mf_common.h); M != 15 is a what-if: nothing in the library changes.
python3 tools/nbr_locality.py [reads] [sampled k-mers]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from metafast_amd import lib as L

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
n_samp = int(sys.argv[2]) if len(sys.argv) > 2 else 400_000
ctx = L.Context(0, stream=torch.cuda.current_stream())
rng = np.random.default_rng(3)
U = np.uint64


def revcomp(x, k):
    x = x.copy()
    r = np.zeros_like(x)
    for _ in range(k):
        r = (r << U(2)) | (U(3) - (x & U(3)))
        x >>= U(2)
    return r


def min_hash(x, k, M):
    """smallest mf_mmer_hash over the canonical M-mers of the k-mers x"""
    mm = U((1 << (2 * M)) - 1)
    best = np.full(len(x), 0xFFFFFFFF, dtype=np.uint64)
    for j in range(k - M, -1, -1):
        f = (x >> U(2 * j)) & mm
        r = revcomp(f, M)
        c = np.minimum(f, r)
        h = (((c ^ U(0x051E6720)) & U(0xFFFFFFFF)) * U(0x9E3779B1)) & U(0xFFFFFFFF)
        best = np.minimum(best, h)
    return best


for k in (21, 23, 25, 31):
    bases = torch.zeros(n_reads * 150 + 64, dtype=torch.uint8, device="cuda"); offs = torch.zeros(n_reads + 1, dtype=torch.int64, device="cuda")
    ctx.synth_reads_device(0x4D45544146415354, 0, 0, n_reads, 150, max(n_reads // 100, 1000), bases.data_ptr(), offs.data_ptr())
    ctx.synchronize()
    t, _ = ctx.count_device_above(bases.data_ptr(), offs.data_ptr(), n_reads, n_reads * 150, k, 1)
    keys, _ = t.export()
    t.close(); del bases, offs
    keys = np.sort(keys.astype(np.uint64))
    x = keys[rng.integers(0, len(keys), size=min(n_samp, len(keys)))]
    mask = U((1 << (2 * k)) - 1)
    nb = []
    for c in range(4):
        nb.append(((x << U(2)) | U(c)) & mask)                      # right neighbours
        nb.append((x >> U(2)) | (U(c) << U(2 * k - 2)))             # left neighbours
    nb = np.stack(nb, axis=1).reshape(-1)
    nbc = np.minimum(nb, revcomp(nb, k))
    pos = np.searchsorted(keys, nbc).clip(max=len(keys) - 1)
    present = keys[pos] == nbc
    line = "k = %d: %d kept k-mers, %.2f present neighbours per k-mer;" % (k, len(keys), present.sum() / len(x))
    for M in (11, 13, 15, 17):
        if M > k - 4:
            continue
        hx = np.repeat(min_hash(x, k, M), 8)[present]
        hn = min_hash(nbc[present], k, M)
        line += "  M = %d: %.1f %% of them in another partition (%d M-mers per k-mer)" % (M, 100.0 * float((hx != hn).mean()), k - M + 1)
    print(line, flush=True)

"""compressed input: metafast.sh kmer-counter on a .fa.gz / .fq.gz / .fa.bz2 file against the decompressor alone (gzip -dc | wc -c): how much of the step is
the one-stream inflate the format imposes (the reference reads through GZIPInputStream, single-threaded as well).  python3 tools/gz_rate.py [reads]"""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metafast_amd import lib as L
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
rl = 150
ctx = L.Context(0, stream=torch.cuda.current_stream())
bases = torch.zeros(n * rl + 64, dtype=torch.uint8, device="cuda"); offs = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
ctx.synth_reads_device(0x4D45544146415354, 0, 0, n, rl, 1_000_000, bases.data_ptr(), offs.data_ptr())
torch.cuda.synchronize()
td = tempfile.mkdtemp(prefix="mf_gz_")
FQ = os.environ.get("MF_GZ_FASTQ") == "1"                    # FASTQ with random qualities (what sequencers write): twice the text, literals all over
fa = os.path.join(td, "s.fq" if FQ else "s.fa")
b = bases[: n * rl].view(n, rl).cpu().numpy()
import numpy as _np
_rng = _np.random.default_rng(5)
with open(fa, "wb") as f:
    for i in range(0, n, 1_000_000):
        blk = b[i:i + 1_000_000]
        if FQ:
            q = _rng.integers(35, 75, blk.shape, dtype=_np.uint8)
            f.write(b"".join(b"@r\n" + r.tobytes() + b"\n+\n" + qq.tobytes() + b"\n" for r, qq in zip(blk, q)))
        else:
            f.write(b"".join(b">r\n" + r.tobytes() + b"\n" for r in blk))
del bases, offs
size = os.path.getsize(fa)
subprocess.run(["gzip", "-1", "-k", fa], check=True)
subprocess.run(["bzip2", "-1", "-k", fa], check=True) if n <= 5_000_000 else None
for ext, dec in ((".gz", ["gzip", "-dc"]),) + (((".bz2", ["bzip2", "-dc"]),) if n <= 5_000_000 else ()) + (("", ["cat"]),):
    f = fa + ext
    t0 = time.perf_counter(); subprocess.run(dec + [f], stdout=subprocess.DEVNULL, check=True); td_ = time.perf_counter() - t0
    wd = os.path.join(td, "wd" + ext.replace(".", "_"))
    t0 = time.perf_counter()
    p = subprocess.run([os.path.join(ROOT, "metafast.sh"), "-t", "kmer-counter", "-k", "31", "-i", f, "-w", wd], capture_output=True, text=True, env=dict(os.environ, MF_IO_TIMING="1"))
    dt = time.perf_counter() - t0
    line = [ln for ln in p.stderr.splitlines() if "count_reads" in ln][-1:] or [p.stderr[-300:]]
    for ln in p.stderr.splitlines():
        if "inflate" in ln or ": read " in ln or (os.environ.get("MF_GZ_ALL") and ln.startswith("[mf]")): print("      " + ln[:200])
    print("%-5s %.2f GB on disk, %.2f GB of FASTA: %s alone %.2f s = %.2f GB/s; kmer-counter %.2f s (exit %d)  %s" % (ext or "plain", os.path.getsize(f) / 1e9, size / 1e9, dec[0], td_, size / 1e9 / td_, dt, p.returncode, line[0][:200]))
# a library of two compressed files (a paired-end sample): the two streams inflate side by side
half = n // 2
fa2 = [os.path.join(td, "s_%d.%s" % (h, "fq" if FQ else "fa")) for h in (1, 2)]
for h, f2 in enumerate(fa2):
    with open(f2, "wb") as f:
        blk = b[h * half:(h + 1) * half]
        for i in range(0, len(blk), 1_000_000):
            if FQ:
                qq = _rng.integers(35, 75, blk[i:i + 1_000_000].shape, dtype=_np.uint8)
                f.write(b"".join(b"@r\n" + r.tobytes() + b"\n+\n" + x.tobytes() + b"\n" for r, x in zip(blk[i:i + 1_000_000], qq)))
            else:
                f.write(b"".join(b">r\n" + r.tobytes() + b"\n" for r in blk[i:i + 1_000_000]))
    subprocess.run(["gzip", "-1", f2], check=True)
wd = os.path.join(td, "wd_pair")
t0 = time.perf_counter()
p = subprocess.run([os.path.join(ROOT, "metafast.sh"), "-t", "kmer-counter", "-k", "31", "-i", fa2[0] + ".gz", fa2[1] + ".gz", "-w", wd], capture_output=True, text=True, env=dict(os.environ, MF_IO_TIMING="1"))
dt = time.perf_counter() - t0
line = [ln for ln in p.stderr.splitlines() if "count_reads" in ln][-1:] or [p.stderr[-300:]]
print("two .gz files of one library (the same reads): kmer-counter %.2f s (exit %d)  %s" % (dt, p.returncode, line[0][:200]))
# the drop-in on two compressed libraries (the two halves as two samples): matrix-builder, every step
wd = os.path.join(td, "wd_two_libs")
t0 = time.perf_counter()
p = subprocess.run([os.path.join(ROOT, "metafast.sh"), "-k", "31", "-i", fa2[0] + ".gz", fa2[1] + ".gz", "-w", wd, "--separate-libs"] if False else
                   [os.path.join(ROOT, "metafast.sh"), "-k", "31", "-i", fa2[0] + ".gz", fa2[1] + ".gz", "-w", wd], capture_output=True, text=True, env=dict(os.environ, MF_IO_TIMING="1"))
dt = time.perf_counter() - t0
print("metafast.sh -k 31 -i s_1.fa.gz s_2.fa.gz (two libraries of %d reads, matrix-builder): %.2f s (exit %d)" % (half, dt, p.returncode))
for ln in p.stderr.splitlines():
    if "inflated into HBM" in ln or "driver:" in ln: print("      " + ln[:200])
import shutil; shutil.rmtree(td, ignore_errors=True)

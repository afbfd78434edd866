"""Differential fuzzing of the file readers (streaming reader with tiny pieces, whole-file reader) against the oracle's
serial readers: random FASTA / FASTQ files with multi-line records, comments, CRLF, empty lines, N / lower-case / IUPAC
characters, phred-0 qualities, quality lines starting with '@' or '+'.  python3 tools/fuzz_files.py [seconds] [seed]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from metafast_amd import lib as L
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
O.build()
ctx = L.Context(0, stream=torch.cuda.current_stream())
td = tempfile.mkdtemp(prefix="mf_fuzz_")


def seq(n):
    al = b"ACGT" if rng.random() < 0.7 else b"ACGTacgtN"
    return bytes(np.frombuffer(al, dtype=np.uint8)[rng.integers(0, len(al), size=n)])


def make_fasta(path):
    nl = b"\r\n" if rng.random() < 0.2 else b"\n"
    with open(path, "wb") as f:
        if rng.random() < 0.1:
            f.write(seq(int(rng.integers(1, 80))) + nl)               # sequence before any header
        for i in range(int(rng.integers(1, 3000))):
            if rng.random() < 0.05:
                f.write(b";comment" + nl)
            f.write(b">r%d desc" % i + nl)
            Lr = int(rng.integers(0, 400)) if rng.random() < 0.95 else int(rng.integers(2000, 30000))
            s = seq(Lr); w = int(rng.choice([60, 70, 80, 100000]))
            for j in range(0, max(Lr, 1), w):
                f.write(s[j:j + w] + nl)
            if rng.random() < 0.02:
                f.write(nl)


def make_fastq(path):
    quals = np.frombuffer(b"@+IIIIFFFF5555#!", dtype=np.uint8)
    with open(path, "wb") as f:
        for i in range(int(rng.integers(1, 3000))):
            Lr = int(rng.integers(1, 300))
            s = seq(Lr).replace(b"a", b"A").replace(b"c", b"C").replace(b"g", b"G").replace(b"t", b"T")
            q = bytes(quals[rng.integers(0, len(quals) - (0 if rng.random() < 0.05 else 1), size=Lr)])
            f.write(b"@read%d\n" % i + s + b"\n+\n" + q + b"\n")
            if rng.random() < 0.002:
                f.write(b"\n")


t_end = time.time() + budget
it = 0
while time.time() < t_end:
    it += 1
    fq = rng.random() < 0.4
    path = os.path.join(td, "f%d.%s" % (it % 4, "fq" if fq else "fa"))
    (make_fastq if fq else make_fasta)(path)
    k = int(rng.choice([5, 15, 21, 31])); ml = int(rng.choice([0, 0, 50]))
    ok, ov = O.Table().count_files([path], k, ml).export()
    # the device parser (mf_dparse.hip) against the oracle's reader: the reads themselves, byte for byte (a file it is not sure about goes
    # through the host readers inside the same call)
    ctx.set_option("device_parse", 1); ctx.set_option("device_parse_min_bytes", 1)
    ob, oo = O.read_file(path)
    gb, go = ctx.load_reads([path])
    assert np.array_equal(go, oo) and np.array_equal(gb, ob), f"it={it} {path}: device parser"
    gk, gc = ctx.count_reads([path], k, ml).export()
    assert np.array_equal(gk, ok) and np.array_equal(gc.astype(np.int32), ov), f"it={it} {path} k={k} min_len={ml} device parser"
    # the same file gzip-compressed (round 5): inflated by mf_inflate.h -- on several threads in pieces of 16 .. 64 KB where the file gives enough
    # of them -- into the staging chunks and parsed in HBM (mf_dparse_gz), or from the host buffer (mf_dparse_mem)
    import gzip as _gzip
    gzp = path + ".gz"
    with open(path, "rb") as f, open(gzp, "wb") as g:
        g.write(_gzip.compress(f.read(), int(rng.choice([1, 6, 9]))))
    ctx.set_option("gz_device_min_bytes", int(rng.choice([0, 1 << 40]))); ctx.set_option("gz_piece_bytes", int(rng.choice([16384, 65536])))
    gb, go = ctx.load_reads([gzp])
    assert np.array_equal(go, oo) and np.array_equal(gb, ob), f"it={it} {gzp}: gz"
    ctx.set_option("gz_device_min_bytes", 32 << 20); ctx.set_option("gz_piece_bytes", 2 << 20)
    os.remove(gzp)
    ctx.set_option("device_parse", 0)
    for sr, piece, slack in ((1, int(rng.choice([4096, 8192, 65536])), int(rng.choice([1024, 4096, 32768]))), (0, 8 << 20, 1 << 20)):
        ctx.set_option("stream_reader", sr); ctx.set_option("stream_piece_bytes", piece); ctx.set_option("stream_slack_bytes", slack)
        gk, gc = ctx.count_reads([path], k, ml).export()
        assert np.array_equal(gk, ok) and np.array_equal(gc.astype(np.int32), ov), f"it={it} {path} k={k} min_len={ml} stream={sr} piece={piece} slack={slack}"
    if it % 25 == 0:
        print("ok", it, path, os.path.getsize(path), flush=True)
print("fuzz_files done:", it, "files, no mismatch")

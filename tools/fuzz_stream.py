"""Differential fuzzing of the STREAMED count (mf_stream.hip: pieces cut at record borders, a sample of chunks, one level-1 scatter per piece)
against the oracle's count of the same files: FASTA / FASTQ files of a few MB in pieces of 1 - 3 MB, one or two files, wrapped lines, CRLF,
comments, N / lower case, quality lines that start with '@' or '+', both quality offsets, records near the piece borders of every length;
files the device parser is not sure about (empty lines, lone CR) step back inside the same call.  python3 tools/fuzz_stream.py [seconds] [seed]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from metafast_amd import lib as L
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
O.build()
ctx = L.Context(0, stream=torch.cuda.current_stream())
td = tempfile.mkdtemp(prefix="mf_fuzz_stream_")
AL = np.frombuffer(b"ACGT", dtype=np.uint8)


def reads(n, genome):
    g = rng.integers(0, 4, genome, dtype=np.uint8)
    out = []
    for _ in range(n):
        ln = int(rng.integers(1, 400)) if rng.random() < 0.97 else int(rng.integers(400, 5000))
        p = int(rng.integers(0, genome - ln))
        r = g[p:p + ln].copy()
        if rng.random() < 0.5:
            r = (3 - r)[::-1]
        a = AL[r].copy()
        m = rng.random(ln) < 0.004
        a[m] = AL[rng.integers(0, 4, int(m.sum()))]
        out.append(a)
    return out


def make_fasta(path, rs, odd):
    nl = b"\r\n" if rng.random() < 0.2 else b"\n"
    w = int(rng.choice([60, 70, 1 << 30]))
    lower = rng.random() < 0.2
    parts = []
    for i, a in enumerate(rs):
        s = a.tobytes()
        if rng.random() < 0.01: s = s[:len(s) // 2] + (b"N" if rng.random() < 0.5 else b"n") + s[len(s) // 2 + 1:]
        if lower and i % 3 == 0: s = s.lower()
        if rng.random() < 0.02: parts.append(b";a comment > with marks" + nl)
        parts.append(b">r%d some text @ + >" % i + nl)
        for j in range(0, max(len(s), 1), w):
            parts.append(s[j:j + w] + nl)
        if odd and i == len(rs) * 2 // 3: parts.append(nl)                 # an empty line: the host reader's business
    open(path, "wb").write(b"".join(parts))


def make_fastq(path, rs, odd):
    qoff = 33 if rng.random() < 0.7 else 64
    nl = b"\r\n" if rng.random() < 0.15 else b"\n"
    parts = []
    for i, a in enumerate(rs):
        s = a.tobytes()
        if rng.random() < 0.01: s = s[:len(s) // 2] + b"N" + s[len(s) // 2 + 1:]
        q = rng.integers(qoff + 1, qoff + 41, len(s)).astype(np.uint8)
        if rng.random() < 0.01: q[int(rng.integers(0, len(s)))] = qoff
        if qoff == 33 and rng.random() < 0.3: q[0] = ord("@") if rng.random() < 0.6 else ord("+")
        parts.append(b"@r%d\n".replace(b"\n", nl) % i + s + nl + (b"+" if rng.random() < 0.9 else b"+r%d" % i) + nl + q.tobytes() + nl)
        if odd and i == len(rs) * 2 // 3: parts.append(nl)
    open(path, "wb").write(b"".join(parts))


t_end = time.time() + budget
it = streamed = back = 0
while time.time() < t_end:
    it += 1
    fq = rng.random() < 0.5
    nf = 1 if rng.random() < 0.7 else 2
    odd = rng.random() < 0.1
    files = []
    for f in range(nf):
        path = os.path.join(td, "f%d_%d.%s" % (it % 3, f, "fq" if fq else "fa"))
        rs = reads(int(rng.integers(6000, 30000)), int(rng.choice([20000, 200000, 2000000])))
        (make_fastq if fq else make_fasta)(path, rs, odd and f == nf - 1)
        files.append(path)
    k = int(rng.choice([21, 23, 25, 27, 31])); thr = int(rng.choice([-1, 0, 1, 3, 40]))
    ok, ov = O.Table().count_files(files, k, 0).export(thr if thr >= 0 else -(2 ** 31))
    ctx.set_option("device_parse_min_bytes", 1)
    ctx.set_option("stream_count_min_bytes", 1); ctx.set_option("stream_count_piece_bytes", int(rng.integers(1 << 20, 3 << 20)))
    ctx.set_option("part_target", int(rng.choice([32, 64, 256])))               # (small counting units: a plan of two levels for a few million k-mers)
    ctx.set_option("device_parse_piece_bytes", int(rng.choice([1 << 16, 1 << 18, 8 << 20])))
    ctx.set_option("stream_count_test_pct", int(rng.choice([100, 100, 100, 60])))
    b0, b1 = ctx.stat("streamed_counts"), ctx.stat("streamed_counts_stepped_back")
    if thr >= 0:
        t, n_all = ctx.count_reads_above(files, k, thr)
    else:
        t = ctx.count_reads(files, k)
    gk, gc = t.export()
    o = np.argsort(gk, kind="stable")
    assert np.array_equal(gk[o], ok) and np.array_equal(gc[o].astype(np.int64), ov.astype(np.int64)), f"it={it} {files} k={k} thr={thr}"
    t.close()
    streamed += ctx.stat("streamed_counts") - b0; back += ctx.stat("streamed_counts_stepped_back") - b1
    if it % 10 == 0:
        print("ok", it, files, [os.path.getsize(f) for f in files], "k", k, "thr", thr, "streamed", streamed, "stepped back", back, flush=True)
for name, v in (("part_target", 0), ("device_parse_piece_bytes", 8 << 20), ("stream_count_test_pct", 100)):
    pass
print("fuzz_stream done:", it, "cases,", streamed, "streamed,", back, "stepped back, no mismatch")

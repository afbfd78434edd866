"""The host-side code that reads files it has never seen, under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only;
SURVEY.md section 5 names this as the build's equivalent of the reference's JVM bounds checks):

  * metafast_amd/csrc/mf_parse.h -- the FASTA / FASTQ / .gz / .bz2 / .binq readers, the record cutters of the parallel and the
    streaming reader, the header walk of components.bin -- through tests/host/parse_harness.cpp (the same header mf_io.hip compiles);
  * metafast_amd/cli/metafast_main.cpp -- option parsing, in.properties, the view / bin2fasta readers of .kmers.bin and
    components.bin, the matrix parser of heatmap-maker -- linked against tests/host/mf_stub.cpp (no GPU entry points).

Valid files must parse to what the oracle's reader gives (FastaReader.java:53-104, FastqReader.java:53-82); truncated, bit-flipped
and spliced files must end in exit code 0 (parsed) or 1 (rejected with a message) -- a sanitizer finding exits with 99, a crash
with a signal."""
import bz2
import gzip
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT

HOST = os.path.join(ROOT, "tests", "host")
BUILD = os.path.join(HOST, "build")
SAN = ["-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all"]
ENV = dict(os.environ, ASAN_OPTIONS="exitcode=99:detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="exitcode=99:halt_on_error=1:print_stacktrace=1")


_CWD = [None]


@pytest.fixture(autouse=True)
def _scratch_cwd(tmp_path):
    _CWD[0] = str(tmp_path)
    yield
    _CWD[0] = None


@pytest.fixture(scope="module")
def binaries():
    if not shutil.which("g++"):
        pytest.skip("no g++")
    os.makedirs(BUILD, exist_ok=True)
    ph, cli = os.path.join(BUILD, "parse_harness"), os.path.join(BUILD, "metafast_san")
    jobs = [(ph, [os.path.join(HOST, "parse_harness.cpp")], ["-lz", "-ldl", "-lpthread"], [os.path.join(ROOT, "metafast_amd", "csrc", "mf_parse.h")]),
            (cli, [os.path.join(ROOT, "metafast_amd", "cli", "metafast_main.cpp"), os.path.join(HOST, "mf_stub.cpp")], ["-lpthread"], [])]
    for out, srcs, libs, deps in jobs:
        newest = max(os.path.getmtime(f) for f in srcs + deps)
        if os.path.exists(out) and os.path.getmtime(out) >= newest:
            continue
        r = subprocess.run(["g++", *SAN, *srcs, "-o", out, *libs], capture_output=True, text=True)
        if r.returncode != 0 and "sanitize" in r.stderr and "cannot find" in r.stderr:
            pytest.skip("g++ has no sanitizer runtime here")
        assert r.returncode == 0, r.stderr[-3000:]
    return ph, cli


def _run(cmd, **kw):
    kw.setdefault("input", "y\n")             # (a used workDir asks "rewrite them?")
    kw.setdefault("cwd", _CWD[0])              # (a tool run without -w makes ./workDir: not in the repository)
    r = subprocess.run(cmd, capture_output=True, text=True, errors="replace", env=ENV, timeout=120, **kw)
    assert r.returncode in (0, 1), "exit %s\n%s\n%s" % (r.returncode, " ".join(map(str, cmd)), (r.stdout + r.stderr)[-3000:])
    return r


# ---- generators (the shapes tools/fuzz_files.py throws at the GPU readers) ----
def _seq(rng, n, weird=True):
    al = np.frombuffer(b"ACGTacgt", dtype=np.uint8)
    s = bytearray(al[rng.integers(0, 8 if weird else 4, size=n)].tobytes())
    if weird and n and rng.random() < 0.15:
        s[int(rng.integers(0, n))] = ord(rng.choice(list("NnRYKMSWBDHV")))
    return bytes(s)


def _fasta(rng, n_reads):
    nl = b"\r\n" if rng.random() < 0.2 else b"\n"
    out = bytearray()
    for i in range(n_reads):
        out += b">r%d some description" % i + nl
        L = int(rng.integers(0, 400)) if rng.random() < 0.95 else int(rng.integers(2000, 9000))
        s, w = _seq(rng, L), int(rng.choice([60, 70, 80, 100000]))
        for j in range(0, max(L, 1), w):
            out += s[j:j + w] + nl
        if rng.random() < 0.02:
            out += nl
    return bytes(out)


def _fastq(rng, n_reads):
    quals = np.frombuffer(b"@+IIIIFFFF5555#!", dtype=np.uint8)
    out = bytearray()
    for i in range(n_reads):
        L = int(rng.integers(1, 300))
        s = _seq(rng, L, weird=False).upper()
        q = bytes(quals[rng.integers(0, len(quals) - (0 if rng.random() < 0.05 else 1), size=L)])
        out += b"@read%d\n" % i + s + b"\n+\n" + q + b"\n"
    return bytes(out)


def _binq(rng, n_reads):
    out = bytearray()
    for i in range(n_reads):
        L = int(rng.integers(0, 120))
        if i % 9 == 0:
            out += b"\xff" * int(rng.integers(1, 4))
        out += L.to_bytes(4, "big") + bytes(rng.integers(0, 256, size=L, dtype=np.uint8).tobytes())
    return bytes(out) + b"\xff"


def _mutations(rng, blob, n):
    """truncations, bit flips, a spliced-in run of random bytes, a doubled tail"""
    for _ in range(n):
        kind = int(rng.integers(0, 4))
        b = bytearray(blob)
        if kind == 0 or len(b) < 8:
            yield bytes(b[: int(rng.integers(0, len(b) + 1))])
        elif kind == 1:
            for _ in range(int(rng.integers(1, 6))):
                b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
            yield bytes(b)
        elif kind == 2:
            at = int(rng.integers(0, len(b)))
            yield bytes(b[:at]) + rng.integers(0, 256, size=int(rng.integers(1, 64)), dtype=np.uint8).tobytes() + bytes(b[at:])
        else:
            at = int(rng.integers(0, len(b)))
            yield bytes(b) + bytes(b[at:])


def _dump(path):
    raw = open(path, "rb").read()
    nr, nb = np.frombuffer(raw[:16], dtype=np.uint64)
    offs = np.frombuffer(raw[16:16 + 8 * (int(nr) + 1)], dtype=np.uint64)
    bases = np.frombuffer(raw[16 + 8 * (int(nr) + 1):], dtype=np.uint8)
    assert len(bases) == int(nb)
    return bases, offs


def test_valid_files_parse_to_what_the_oracle_reads(binaries, oracle, tmp_path):
    ph, _ = binaries
    rng = np.random.default_rng(11)
    for it in range(6):
        fq = it % 3 == 2
        blob = _fastq(rng, 400) if fq else _fasta(rng, 300)
        plain = tmp_path / ("v%d.%s" % (it, "fq" if fq else "fa"))
        plain.write_bytes(blob)
        try:
            wb, wo = oracle.read_file(str(plain))
        except Exception:
            wb = None                       # (an IUPAC code the generator placed where the reference rejects the file: both must reject)
        variants = [plain]
        gz = tmp_path / (plain.name + ".gz"); gz.write_bytes(gzip.compress(blob[: len(blob) // 2], 1) + gzip.compress(blob[len(blob) // 2:], 9)); variants.append(gz)
        bz = tmp_path / (plain.name + ".bz2"); bz.write_bytes(bz2.compress(blob, 5)); variants.append(bz)
        for f in variants:
            for threads in (1, 7):
                d = tmp_path / "dump.bin"
                r = _run([ph, "reads", str(f), str(threads), str(d)])
                if wb is None:
                    assert r.returncode == 1, r.stdout
                    continue
                if r.returncode == 1 and f.suffix == ".bz2" and "libbz2" in r.stdout:
                    continue                # (no libbz2 on this box: rejected with a message, fine)
                assert r.returncode == 0, r.stdout
                gb, go = _dump(d)
                assert np.array_equal(go, wo) and np.array_equal(gb, wb), (f, threads)
    # the reference's own data
    for name in ("meta_test_1.fa", "tinytest_A.fastq"):
        f = os.path.join(ROOT, "tests", "golden", "ref_test_data", name)
        d = tmp_path / "dump.bin"
        assert _run([ph, "reads", f, "3", str(d)]).returncode == 0
        gb, go = _dump(d)
        wb, wo = oracle.read_file(f)
        assert np.array_equal(go, wo) and np.array_equal(gb, wb)


def test_damaged_read_files_are_parsed_or_rejected_cleanly(binaries, tmp_path):
    ph, _ = binaries
    rng = np.random.default_rng(12)
    seeds = [("fa", _fasta(rng, 120)), ("fq", _fastq(rng, 150)), ("binq", _binq(rng, 200))]
    seeds.append(("fa.gz", gzip.compress(seeds[0][1], 6)))
    seeds.append(("fq.bz2", bz2.compress(seeds[1][1], 9)))
    n = 0
    for ext, blob in seeds:
        for m in _mutations(rng, blob, 14):
            f = tmp_path / ("m%d.%s" % (n, ext))
            f.write_bytes(m)
            _run([ph, "reads", str(f), str(1 + n % 5)])
            if ext in ("fa", "fq"):
                _run([ph, "cuts", str(f), "1" if ext == "fa" else "2", str(int(rng.choice([64, 257, 4096]))), str(int(rng.choice([16, 100, 1024])))])
            n += 1
    assert n == 70
    for blob in (b"", b">", b">\n", b"@\n", b"@r\nAC\n+\n", b"\n\n\n", b"\r", b">a\r>b\r\r"):       # the shortest files there are
        for ext in ("fa", "fq"):
            f = tmp_path / ("tiny." + ext)
            f.write_bytes(blob)
            _run([ph, "reads", str(f), "2"])
            _run([ph, "cuts", str(f), "1" if ext == "fa" else "2", "1", "1"])


def _components_bin(rng, n):
    out = bytearray(n.to_bytes(4, "big"))
    for _ in range(n):
        sz = int(rng.integers(0, 50))
        out += sz.to_bytes(4, "big") + int(rng.integers(0, 1 << 40)).to_bytes(8, "big")
        out += rng.integers(0, 1 << 62, size=sz, dtype=np.uint64).astype(">u8").tobytes()
    return bytes(out)


def test_components_and_kmers_files(binaries, tmp_path):
    ph, cli = binaries
    rng = np.random.default_rng(13)
    good = _components_bin(rng, 40)
    f = tmp_path / "components.bin"
    f.write_bytes(good)
    r = _run([ph, "comps", str(f)])
    assert r.returncode == 0 and r.stdout.startswith("components 40 ")
    assert _run([cli, "-t", "view", "-k", "31", "--components-file", str(f), "-w", str(tmp_path / "w0")]).returncode == 0
    assert _run([cli, "-t", "bin2fasta", "-k", "31", "--components-file", str(f), "-w", str(tmp_path / "w0"), "--output-file", str(tmp_path / "c.fa")]).returncode == 0
    huge = (0xFFFFFFFF).to_bytes(4, "big") + good[4:]                     # a count the file cannot hold: rejected, nothing allocated for it
    for i, m in enumerate([huge, good[:3], good[:4], good[:17]] + list(_mutations(rng, good, 16))):
        g = tmp_path / ("c%d.bin" % i)
        g.write_bytes(m)
        r = _run([ph, "comps", str(g)])
        if i < 4:
            assert r.returncode == 1 and "corrupted" in r.stdout, r.stdout
        _run([cli, "-t", "view", "-k", "31", "--components-file", str(g), "-w", str(tmp_path / "w1")])
        _run([cli, "-t", "bin2fasta", "-k", "31", "--components-file", str(g), "-w", str(tmp_path / "w1"), "--output-file", str(tmp_path / "c.fa")])
    keys = np.sort(rng.integers(0, 1 << 62, size=300, dtype=np.uint64))
    rec = b"".join(int(k).to_bytes(8, "big") + int(c).to_bytes(2, "big") for k, c in zip(keys, rng.integers(1, 500, size=300)))
    kf = tmp_path / "s.kmers.bin"
    kf.write_bytes(rec)
    r = _run([cli, "-t", "view", "-k", "31", "--kmers-file", str(kf), "-w", str(tmp_path / "w2")])
    assert r.returncode == 0 and len(r.stdout.splitlines()) >= 300
    for i, m in enumerate(_mutations(rng, rec, 10)):
        g = tmp_path / ("k%d.kmers.bin" % i)
        g.write_bytes(m)
        _run([cli, "-t", "view", "-k", str(int(rng.choice([1, 15, 31]))), "--kmers-file", str(g), "-w", str(tmp_path / "w2")])
        _run([cli, "-t", "bin2fasta", "-k", "31", "--kmers-file", str(g), "-w", str(tmp_path / "w2"), "--output-file", str(tmp_path / "k.fa")])


def test_driver_options_properties_and_matrices(binaries, tmp_path):
    _, cli = binaries
    rng = np.random.default_rng(14)
    # (a copy: heatmap-maker writes <input>_renumbered.txt NEXT TO its input by default, and the golden directory is tracked)
    golden = str(tmp_path / "meta_test_matrix.txt")
    shutil.copyfile(os.path.join(ROOT, "tests", "golden", "ref_test_data", "meta_test_matrix.txt"), golden)
    wd = tmp_path / "hm"
    assert _run([cli, "-t", "heatmap-maker", "-i", golden, "-w", str(wd)]).returncode == 0
    text = open(golden, "rb").read()
    for i, m in enumerate(list(_mutations(rng, text, 12)) + [b"", b"\n", b"a\tb\n", b"\t1\n1\tx\n", b"s1\t0.0\tnan\ns2\tinf\t0.0\n"]):
        f = tmp_path / ("m%d.txt" % i)
        f.write_bytes(m)
        _run([cli, "-t", "heatmap-maker", "-i", str(f), "-w", str(tmp_path / ("hm%d" % i))])
    # in.properties of an earlier run, read back under -c: separators, escapes, comments, a line without a value, no final newline, binary junk
    props = [b"k = 31\nreads = /a/b.fa\nreads = /c d.fa\n", b"k:31\r\nreads=\\\\x\\=y\r\n# comment\n! another\n\n", b"k", b"=\n:\n = \n", b"k = " + b"9" * 400 + b"\n",
             bytes(rng.integers(0, 256, size=300, dtype=np.uint8).tobytes()), b"matrix-file = " + golden.encode() + b"\noutput-format = %.4f"]
    for i, p in enumerate(props):
        w = tmp_path / ("p%d" % i)
        os.makedirs(w, exist_ok=True)
        (w / "in.properties").write_bytes(p)
        (w / "SUCCESS").write_bytes(b"")
        _run([cli, "-t", "heatmap-maker", "-i", golden, "-w", str(w), "-c"], input="y\n")
        _run([cli, "-t", "heatmap-maker", "-w", str(w), "-c"], input="n\n")
    # command lines: unknown tools and options, options without their value, repeated and empty ones; GPU steps stop at the stub with exit 1
    for args in (["-t"], ["-t", "nonsense"], ["-k"], ["-k", "x", "-i"], ["-i", "a.fa", "-i", "b.fa", "-k", "31", "-w", str(tmp_path / "g")], ["--work-dir"], ["-t", "view"],
                 ["-t", "view", "-k", "99", "--kmers-file", "/nonexistent"], ["-h"], ["--help-all"], ["-t", "kmer-counter", "-k", "0", "-i", golden, "-w", str(tmp_path / "g2")],
                 ["-m", "4G", "-ea", "-Xmx1g", "--tools"], ["", "", ""], ["-w", str(tmp_path / "g3"), "-s", "nonsense", "-i", golden], ["-t", "dist-matrix-calculator", "--features", golden, "-w", str(tmp_path / "g4")],
                 ["-i", "a.fa", "b.fa", "c.fa", "--devices", "0,1,0", "-w", str(tmp_path / "g5")], ["-i", "a.fa", "--devices", "0,,1", "-w", str(tmp_path / "g6")],
                 ["-i", "a.fa", "--devices", "x", "-w", str(tmp_path / "g7")], ["-i", "a.fa", "--devices", "-w", str(tmp_path / "g8")],
                 ["-t", "features-calculator", "-k", "31", "-cm", golden, "-ka", golden, "--selected", golden, "--devices", "0,1", "-w", str(tmp_path / "g9")]):
        r = subprocess.run([cli, *args], capture_output=True, text=True, errors="replace", env=ENV, timeout=60, input="", cwd=str(tmp_path))
        assert r.returncode in (0, 1), (args, r.returncode, (r.stdout + r.stderr)[-2000:])


def _record_starts(blob, fq):
    """byte offsets where a record begins: FASTA -- header / comment lines; FASTQ -- every fourth line"""
    starts, pos, line = [], 0, 0
    while pos < len(blob):
        if fq:
            if line % 4 == 0:
                starts.append(pos)
        elif blob[pos:pos + 1] in (b">", b";"):
            starts.append(pos)
        nl = blob.find(b"\n", pos)
        if nl < 0:
            break
        pos = nl + 1; line += 1
    return starts


def test_streamed_count_piece_plan_and_sample(binaries, tmp_path):
    """Round 6 (mf_stream.hip; the host half is st_plan_pieces / st_sample in mf_parse.h): the pieces tile every file, each one starts where a record
    starts -- also when quality lines start with '@' or '+' --, the sample is whole records of the files; damaged files are planned or rejected cleanly"""
    ph, _ = binaries
    rng = np.random.default_rng(17)
    planned = 0
    for it in range(6):
        fq = it % 2 == 1
        blobs = [(_fastq(rng, 9000) if fq else _fasta(rng, 7000)) for _ in range(1 + it % 2 * (it // 2 % 2))]
        files = []
        for j, b in enumerate(blobs):
            f = tmp_path / ("s%d_%d.%s" % (it, j, "fq" if fq else "fa")); f.write_bytes(b); files.append(f)
        piece = int(rng.choice([1 << 20, (1 << 20) + 12345, 3 << 19]))
        out = tmp_path / "sample.bin"
        r = _run([ph, "stream", "2" if fq else "1", str(piece), "8", str(out), *map(str, files)])
        if r.returncode == 1:
            # (a FASTA file with a 9000-base record wrapped at 60 has no shorter lines than the window... not with these sizes: a plan is expected)
            assert "rejected" in r.stdout
            continue
        planned += 1
        pieces = [tuple(map(int, ln.split()[1:])) for ln in r.stdout.splitlines() if ln.startswith("piece ")]
        for j, b in enumerate(blobs):
            mine = [(o, n) for f, o, n in pieces if f == j]
            assert mine[0][0] == 0 and sum(n for _, n in mine) == len(b) and all(mine[i][0] + mine[i][1] == mine[i + 1][0] for i in range(len(mine) - 1))
            starts = set(_record_starts(b, fq))
            assert all(o in starts for o, _ in mine), (it, j)
            assert all(n >= piece for _, n in mine[:-1])
        sample = out.read_bytes()
        assert len(sample) > 0 and sample.endswith(b"\n")
        # the sample: whole records of the files (every record of it is a record of some file)
        recs = set()
        for b in blobs:
            st = _record_starts(b, fq) + [len(b)]
            recs.update(b[st[i]:st[i + 1]] for i in range(len(st) - 1))
        sst = _record_starts(sample, fq) + [len(sample)]
        assert sst[0] == 0
        assert all(sample[sst[i]:sst[i + 1]] in recs for i in range(len(sst) - 1)), it
    assert planned >= 5, planned
    # damaged files: any outcome but a sanitizer report
    blob = _fastq(rng, 9000)
    for j, b in enumerate(_mutations(rng, blob, 12)):
        f = tmp_path / ("d%d.fq" % j); f.write_bytes(b)
        _run([ph, "stream", "2", str(1 << 20), "8", str(tmp_path / "sample.bin"), str(f)])
    for j, b in enumerate(_mutations(rng, _fasta(rng, 7000), 12)):
        f = tmp_path / ("d%d.fa" % j); f.write_bytes(b)
        _run([ph, "stream", "1", str(1 << 20), "8", str(tmp_path / "sample.bin"), str(f)])

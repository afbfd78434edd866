"""mf_inflate.h (round 5): the whole-buffer DEFLATE / gzip decoder the compressed-input readers try first (FastaGZReader.java / FastqGZReader.java read
through java.util.zip.GZIPInputStream).  Host code: runs without a GPU.  Against Python's zlib on every kind of block (stored, fixed, dynamic), every
level, long distances, sub-table codes (skewed alphabets), concatenated members, header fields, empty members -- and on damaged files, where it must
refuse (the product then lets zlib word the error)."""
import ctypes as C
import gzip
import io
import os
import zlib

import numpy as np
import pytest


@pytest.fixture(scope="module")
def gunzip():
    from metafast_amd import lib as L
    lib = L.lib()
    fn = lib.mf_debug_gunzip
    fn.restype = C.c_int
    fn.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.mf_debug_free.restype = None
    lib.mf_debug_free.argtypes = [C.c_void_p]

    def run(data, mode):
        out, n = C.c_void_p(), C.c_uint64()
        rc = fn(data, len(data), mode, C.byref(out), C.byref(n))
        if rc != 0:
            return None
        try:
            return C.string_at(out, n.value)
        finally:
            lib.mf_debug_free(out)
    return run


def _gz(data, level=6, **kw):
    b = io.BytesIO()
    with gzip.GzipFile(fileobj=b, mode="wb", compresslevel=level, **kw) as f:
        f.write(data)
    return b.getvalue()


def _corpus():
    rng = np.random.default_rng(7)
    dna = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 3_000_000)].tobytes()
    fq = b"".join(b"@read_%d\n" % i + dna[i * 150:(i + 1) * 150] + b"\n+\n" + bytes(rng.integers(33, 74, 150, dtype=np.uint8)) + b"\n" for i in range(12000))
    skew = bytes(np.minimum(rng.geometric(0.02, 400_000), 255).astype(np.uint8))           # long codes: sub-tables behind the 11-bit table
    rep = (b"abcdefghij" * 7 + b"Z") * 30000                                                # matches at many distances, lengths up to 258
    return {"empty": b"", "one": b"A", "dna": dna, "fastq": fq, "random": bytes(rng.integers(0, 256, 500_000, dtype=np.uint8)), "skewed": skew, "repeats": rep,
            "zeros": bytes(2_000_000), "far": dna[:40000] + bytes(rng.integers(0, 256, 31000, dtype=np.uint8)) + dna[:40000]}


def test_every_block_kind_and_level(gunzip):
    for name, data in _corpus().items():
        for level in (0, 1, 2, 4, 6, 9):
            z = _gz(data, level)
            assert gunzip(z, 1) == data, (name, level)
            assert gunzip(z, 2) == data and gunzip(z, 0) == data
        # fixed-Huffman blocks and raw strategies
        for strategy in (zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
            co = zlib.compressobj(6, zlib.DEFLATED, 31, 9, strategy)
            z = co.compress(data) + co.flush()
            assert gunzip(z, 1) == data, (name, strategy)
        # many small blocks (a flush after every piece)
        co = zlib.compressobj(5, zlib.DEFLATED, 31)
        z = b"".join(co.compress(data[i:i + 7001]) + co.flush(zlib.Z_FULL_FLUSH if (i // 7001) % 2 else zlib.Z_SYNC_FLUSH) for i in range(0, len(data), 7001)) + co.flush()
        assert gunzip(z, 1) == data, name


def test_members_and_header_fields(gunzip):
    c = _corpus()
    parts = [c["fastq"][:100000], b"", c["dna"][:50000], c["one"], c["repeats"][:300000]]
    z = b"".join(_gz(p, lv) for p, lv in zip(parts, (1, 6, 9, 0, 3)))
    assert gunzip(z, 1) == b"".join(parts)
    # FNAME, mtime; FEXTRA + FCOMMENT + FHCRC by hand
    z = _gz(parts[0], 6, filename="reads_1.fastq", mtime=12345)
    assert gunzip(z, 1) == parts[0]
    raw = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = raw.compress(parts[2]) + raw.flush()
    hdr = bytes([0x1F, 0x8B, 8, 4 | 8 | 16 | 2, 0, 0, 0, 0, 0, 255]) + (5).to_bytes(2, "little") + b"EXTRA" + b"name\0" + b"a comment\0" + b"\x12\x34"
    z = hdr + body + zlib.crc32(parts[2]).to_bytes(4, "little") + (len(parts[2]) & 0xFFFFFFFF).to_bytes(4, "little")
    assert gunzip(z, 1) == parts[2]


def test_damaged_files_are_refused(gunzip):
    c = _corpus()
    data = c["fastq"][:400000]
    z = bytearray(_gz(data, 6))
    assert gunzip(bytes(z), 1) == data
    rng = np.random.default_rng(3)
    refused = 0
    for _ in range(300):
        y = bytearray(z)
        kind = int(rng.integers(0, 4))
        if kind == 0:
            y[int(rng.integers(0, len(y)))] ^= 1 << int(rng.integers(0, 8))                # one flipped bit anywhere
        elif kind == 1:
            del y[int(rng.integers(1, len(y))):]                                           # truncated
        elif kind == 2:
            at = int(rng.integers(10, len(y) - 8)); y[at:at + 4] = bytes(rng.integers(0, 256, 4, dtype=np.uint8))
        else:
            y += bytes(rng.integers(1, 256, int(rng.integers(1, 40)), dtype=np.uint8))      # garbage behind the member
        got = gunzip(bytes(y), 1)
        if got is None:
            refused += 1
        else:
            assert got == data, kind                                                       # (a flipped header byte -- mtime, OS -- changes nothing)
        # the product's order never returns anything but what zlib returns
        try:
            want = gzip.decompress(bytes(y))
        except Exception:
            want = None
        both = gunzip(bytes(y), 2)
        assert both == want or (want is None and both is None) or (kind == 3 and both is None), kind
    assert refused > 150
    for bad in (b"", b"\x1f", b"\x1f\x8b\x08", b"not a gzip file at all, just text\n" * 10):
        assert gunzip(bad, 1) is None


def test_random_streams(gunzip):
    """many small streams of random shape: alphabets of 1 .. 256 symbols with flat, geometric and two-level distributions (code lengths up to 15:
    sub-tables; pairs of short codes), runs, copies at random distances, every level and strategy, stored / fixed / dynamic blocks mixed by flushes"""
    rng = np.random.default_rng(2024)
    for it in range(1500):
        n = int(rng.integers(0, 30000))
        k = int(rng.integers(1, 257))
        kind = int(rng.integers(0, 4))
        if kind == 0:
            a = rng.integers(0, k, n)
        elif kind == 1:
            a = np.minimum(rng.geometric(float(rng.uniform(0.01, 0.6)), n) - 1, k - 1)
        elif kind == 2:
            a = np.where(rng.random(n) < 0.97, rng.integers(0, min(k, 4), n), rng.integers(0, k, n))
        else:
            a = np.repeat(rng.integers(0, k, n // 50 + 1), rng.integers(1, 120, n // 50 + 1))[:n]
        data = bytearray(a.astype(np.uint8).tobytes())
        for _ in range(int(rng.integers(0, 6))):                                           # copies: matches at chosen distances
            if len(data) > 600:
                L0 = int(rng.integers(3, 300)); src = int(rng.integers(0, len(data) - L0)); dst = int(rng.integers(0, len(data) - L0))
                data[dst:dst + L0] = data[src:src + L0]
        data = bytes(data)
        level = int(rng.integers(0, 10))
        strategy = int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED]))
        co = zlib.compressobj(level, zlib.DEFLATED, 31, int(rng.integers(1, 10)), strategy)
        cut = int(rng.integers(0, len(data) + 1))
        z = co.compress(data[:cut]) + (co.flush(zlib.Z_SYNC_FLUSH) if rng.random() < 0.5 else b"") + co.compress(data[cut:]) + co.flush()
        assert gunzip(z, 1) == data, (it, n, k, kind, level, strategy)


def test_one_member_on_several_threads(gunzip):
    """the speculative form (mf_inflate.h: inflate_parallel), forced onto small inputs with pieces of 48 KB of compressed data: block starts found
    by trying bit positions, pieces decoded without their 32 KB of history (markers), windows handed on, markers resolved.  Pieces shorter than
    a window, stored and fixed blocks among the dynamic ones, matches that copy markers; several members: the form steps back (mode 3 says so)
    and one thread does it."""
    c = _corpus()
    rng = np.random.default_rng(5)
    ran = 0
    for name in ("dna", "fastq", "random", "skewed", "repeats", "far", "zeros"):
        data = c[name]
        for level in (1, 4, 6, 9):
            z = _gz(data, level)
            got = gunzip(z, 3)
            if got is not None:
                ran += 1
                assert got == data, (name, level)
            assert gunzip(z, 1) == data
        # blocks of every kind in one stream: pieces of it compressed at different levels and strategies, joined by full flushes
        co = [zlib.compressobj(lv, zlib.DEFLATED, -15, 8, st) for lv, st in ((6, zlib.Z_DEFAULT_STRATEGY), (0, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (9, zlib.Z_FILTERED))]
        co = zlib.compressobj(6, zlib.DEFLATED, 31, 8)
        z = b"".join(co.compress(data[i:i + 90001]) + co.flush(zlib.Z_FULL_FLUSH if (i // 90001) % 3 == 0 else zlib.Z_SYNC_FLUSH) for i in range(0, len(data), 90001)) + co.flush()
        got = gunzip(z, 3)
        assert got is None or got == data, name
        assert gunzip(z, 1) == data
    assert ran >= 12                                                                       # (compressible data gives too few pieces to cut: those ran on one thread)
    # text with long-range repeats at random distances <= 32 KB: matches reach into the unknown window all the time
    words = [bytes(rng.integers(97, 123, int(rng.integers(3, 40)), dtype=np.uint8)) for _ in range(3000)]
    text = b" ".join(words[int(i)] for i in rng.integers(0, len(words), 700_000))
    for level in (1, 6, 9):
        z = _gz(text, level)
        assert gunzip(z, 3) == text, level
    # ... and into a sink (mode 4: what mf_dparse_gz does with the pinned staging chunks): slots of 64 KB, three workers
    for level in (1, 6):
        assert gunzip(_gz(text, level), 4) == text, level
    assert gunzip(_gz(c["fastq"], 6), 4) == c["fastq"]
    # several members: not for this form
    z = _gz(text[:2_000_000], 6) + _gz(text[2_000_000:], 6)
    assert gunzip(z, 3) is None and gunzip(z, 4) is None and gunzip(z, 1) == text and gunzip(z, 2) == text
    # damaged: never anything but the content or a refusal
    z = bytearray(_gz(text, 6))
    for _ in range(40):
        y = bytearray(z)
        y[int(rng.integers(20, len(y) - 8))] ^= 1 << int(rng.integers(0, 8))
        got = gunzip(bytes(y), 3)
        assert got is None or got == text


def _bgzf(data, level=6, block=0xFF00):
    out = []
    for i in list(range(0, len(data), block)) + [None]:                                    # (+ the empty end-of-file block bgzip appends)
        chunk = b"" if i is None else data[i:i + block]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = co.compress(chunk) + co.flush()
        bsize = 12 + 6 + len(body) + 8
        out.append(bytes([0x1F, 0x8B, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0]) + b"BC" + (2).to_bytes(2, "little") + (bsize - 1).to_bytes(2, "little") + body +
                   zlib.crc32(chunk).to_bytes(4, "little") + len(chunk).to_bytes(4, "little"))
    return b"".join(out)


def test_bgzf_members_in_parallel(gunzip):
    """bgzip's blocked gzip: members of <= 64 KB with their length in the FEXTRA field -- counted off by the headers, inflated side by side"""
    c = _corpus()
    for name in ("fastq", "dna", "random", "zeros", "one", "empty"):
        for level in (1, 6):
            z = _bgzf(c[name], level)
            assert gzip.decompress(z) == c[name]
            assert gunzip(z, 1) == c[name] and gunzip(z, 2) == c[name], (name, level)
    z = bytearray(_bgzf(c["fastq"], 6))
    rng = np.random.default_rng(9)
    for _ in range(60):
        y = bytearray(z)
        y[int(rng.integers(0, len(y)))] ^= 1 << int(rng.integers(0, 8))
        got = gunzip(bytes(y), 1)
        assert got is None or got == c["fastq"]
        try:
            want = gzip.decompress(bytes(y))
        except Exception:
            want = None
        assert gunzip(bytes(y), 2) == want


def test_large_member_in_parallel_crc(gunzip):
    rng = np.random.default_rng(11)
    dna = np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.integers(0, 5, 60_000_000)].tobytes()
    z = _gz(dna, 1)
    assert gunzip(z, 1) == dna


def test_decoder_under_asan_ubsan(tmp_path):
    """tools/inflate_sanitized.cpp: mf_inflate.h alone under AddressSanitizer + UBSan -- one thread and the several-thread form forced with 16 KB
    pieces -- on valid, bit-flipped, truncated and overwritten gzip / BGZF files: no report, nothing but content or a refusal"""
    import shutil, subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "inflate_sanitized"
    r = subprocess.run(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-o", str(exe), os.path.join(root, "tools", "inflate_sanitized.cpp"), "-lz", "-lpthread"],
                       capture_output=True, text=True, cwd=os.path.join(root, "tools"))
    if r.returncode != 0:
        pytest.skip("no sanitizer runtime here: " + r.stderr[-200:])
    rng = np.random.default_rng(17)
    words = [bytes(rng.integers(97, 123, int(rng.integers(3, 40)), dtype=np.uint8)) for _ in range(800)]
    text = b" ".join(words[int(i)] for i in rng.integers(0, len(words), 90000))
    dna = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 600_000)].tobytes()
    files = []
    for name, d in (("text", text), ("dna", dna), ("zero", bytes(300000))):
        for lvl in (0, 1, 6):
            for kind, z in (("gz", _gz(d, lvl)), ("bgzf", _bgzf(d, lvl))):
                for v in range(7):
                    y = bytearray(z)
                    if v == 1:
                        y[int(rng.integers(0, len(y)))] ^= 1 << int(rng.integers(0, 8))
                    elif v == 2:
                        del y[int(rng.integers(1, len(y))):]
                    elif v == 3:
                        at = int(rng.integers(0, max(1, len(y) - 8))); y[at:at + 6] = bytes(rng.integers(0, 256, 6, dtype=np.uint8))
                    elif v >= 4:
                        y[int(rng.integers(10, len(y)))] ^= 1 << int(rng.integers(0, 8))
                    f = tmp_path / f"{name}_{lvl}_{kind}_{v}.gz"
                    f.write_bytes(bytes(y)); files.append(str(f))
    r = subprocess.run([str(exe)] + files, capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0 and "ERROR" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-2000:]
    assert "accepted" in r.stdout

"""ctypes wrapper around the CPU oracle (oracle/mf_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (metafast_amd/) must never
import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "build", "libmf_oracle.so")
_lib = None


def build(force=False):
    """Compile the oracle with gcc (a few seconds)."""
    srcs = [os.path.join(_HERE, f) for f in ("mf_oracle.c", "mf_oracle_wide.c", "mf_oracle_core.inc", "mf_oracle.h")]
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(f) for f in srcs)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    u64, i64, i32, vp, cp = C.c_uint64, C.c_int64, C.c_int, C.c_void_p, C.c_char_p
    pu64 = C.POINTER(C.c_uint64)

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    sig("or_last_error", cp)
    sig("or_read_file", i32, cp, C.POINTER(vp), C.POINTER(vp), pu64, pu64)
    sig("or_free", None, vp)
    sig("or_table_new", vp)
    sig("or_table_free", None, vp)
    sig("or_count_buffer", i32, vp, vp, vp, u64, i32, i32)
    sig("or_count_wide", i32, vp, vp, u64, i32, i32, vp, vp, vp, u64, vp, vp)
    sig("or_count_files", i32, vp, C.POINTER(cp), i32, i32, i32)
    sig("or_table_size", u64, vp)
    sig("or_table_export", u64, vp, i32, vp, vp, u64)
    sig("or_table_get", i64, vp, u64)
    sig("or_table_add", i32, vp, u64, i32)
    sig("or_write_kmers", i32, vp, i32, cp, cp, pu64)
    sig("or_load_kmers", i32, vp, C.POINTER(cp), i32, i32)
    sig("or_build_unitigs", vp, vp, i32, i32, i32)
    sig("or_seqs_free", None, vp)
    sig("or_unitig_census", None, pu64)
    sig("or_seqs_count", u64, vp)
    sig("or_seqs_total_len", u64, vp)
    sig("or_seqs_get", i32, vp, u64, C.POINTER(vp), pu64, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32))
    sig("or_seqs_write_fasta", i32, vp, cp)
    sig("or_write_distribution", i32, vp, cp)
    sig("or_count_seqs", i32, vp, vp, i32, i32)
    sig("or_cut_components", vp, vp, i32, i32, i32)
    sig("or_comps_free", None, vp)
    sig("or_comps_count", u64, vp)
    sig("or_comps_get", i32, vp, u64, pu64, C.POINTER(i64), C.POINTER(i32), C.POINTER(vp))
    sig("or_comps_write", i32, vp, cp, cp)
    sig("or_comps_load", vp, cp)
    sig("or_features", i32, vp, vp, i32, vp, vp)
    sig("or_features_reads", i32, vp, vp, vp, u64, i32, i32, vp, vp)
    sig("or_features_selected", i32, vp, vp, i32, vp, vp, vp)
    sig("or_features_reads_selected", i32, vp, vp, vp, u64, i32, i32, vp, vp, vp)
    sig("or_bray_curtis", i32, vp, i32, i32, vp)
    sig("or_revcomp", u64, u64, i32)
    sig("or_canonical", u64, u64, i32)
    # NO-REFERENCE EXTENSION, k <= 63 (mf_oracle_wide.c: the core compiled for 128-bit keys)
    sig("orw_last_error", cp)
    sig("orw_table_new", vp)
    sig("orw_table_free", None, vp)
    sig("orw_table_size", u64, vp)
    sig("orw_table_export", u64, vp, i32, vp, vp, u64)
    sig("orw_table_get2", i64, vp, u64, u64)
    sig("orw_table_add2", i32, vp, u64, u64, i32)
    sig("orw_count_buffer", i32, vp, vp, vp, u64, i32, i32)
    sig("orw_build_unitigs", vp, vp, i32, i32, i32)
    sig("orw_unitig_census", None, pu64)
    sig("orw_seqs_free", None, vp)
    sig("orw_seqs_count", u64, vp)
    sig("orw_seqs_total_len", u64, vp)
    sig("orw_seqs_get", i32, vp, u64, C.POINTER(vp), pu64, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32))
    sig("orw_count_seqs", i32, vp, vp, i32, i32)
    sig("orw_cut_components", vp, vp, i32, i32, i32)
    sig("orw_comps_free", None, vp)
    sig("orw_comps_count", u64, vp)
    sig("orw_comps_get", i32, vp, u64, pu64, C.POINTER(i64), C.POINTER(i32), C.POINTER(vp))
    sig("orw_features_selected", i32, vp, vp, i32, vp, vp, vp)
    sig("orw_features_reads_selected", i32, vp, vp, vp, u64, i32, i32, vp, vp, vp)
    sig("orw_revcomp2", None, u64, u64, i32, pu64)
    sig("or_cpu_baseline_count", u64, vp, vp, u64, i32, i32, pu64)
    sig("or_cpu_baseline_file", i32, cp, i32, i32, i32, cp, pu64, C.POINTER(C.c_double))
    _lib = L
    return L


class OracleError(RuntimeError):
    pass


def _check(rc):
    if rc < 0:
        raise OracleError(lib().or_last_error().decode())


def _cfiles(files):
    arr = (C.c_char_p * len(files))(*[os.fsencode(f) for f in files])
    return arr


def read_file(path):
    """-> (bases uint8[n_bases] ASCII, offsets uint64[n_reads+1])"""
    L = lib()
    b, o = C.c_void_p(), C.c_void_p()
    nr, nb = C.c_uint64(), C.c_uint64()
    _check(L.or_read_file(os.fsencode(path), C.byref(b), C.byref(o), C.byref(nr), C.byref(nb)))
    bases = np.ctypeslib.as_array(C.cast(b, C.POINTER(C.c_uint8)), shape=(max(nb.value, 1),))[: nb.value].copy()
    offs = np.ctypeslib.as_array(C.cast(o, C.POINTER(C.c_uint64)), shape=(nr.value + 1,)).copy()
    L.or_free(b)
    L.or_free(o)
    return bases, offs


class Table:
    def __init__(self):
        self.h = lib().or_table_new()

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().or_table_free(self.h)
                self.h = None
        except Exception:            # (interpreter shutdown: the module is already torn down)
            pass

    def count_buffer(self, bases, offsets, k, min_len=0):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        _check(lib().or_count_buffer(self.h, bases.ctypes.data, offsets.ctypes.data,
                                     len(offsets) - 1, k, min_len))
        return self

    def count_files(self, files, k, min_len=0):
        _check(lib().or_count_files(self.h, _cfiles(files), len(files), k, min_len))
        return self

    def load_kmers(self, files, freq_threshold=0):
        _check(lib().or_load_kmers(self.h, _cfiles(files), len(files), freq_threshold))
        return self

    def add(self, key, inc=1):
        lib().or_table_add(self.h, int(key), int(inc))

    def get(self, key):
        return lib().or_table_get(self.h, int(key))

    def __len__(self):
        return lib().or_table_size(self.h)

    def export(self, threshold=-(2 ** 31)):
        """-> (keys uint64[n] ascending, vals int32[n]) of entries with value > threshold"""
        L = lib()
        n = L.or_table_export(self.h, threshold, None, None, 0)
        keys = np.empty(n, dtype=np.uint64)
        vals = np.empty(n, dtype=np.int32)
        if n:
            L.or_table_export(self.h, threshold, keys.ctypes.data, vals.ctypes.data, n)
        return keys, vals

    def write_kmers(self, threshold, kmers_bin, stat_txt=None):
        g = C.c_uint64()
        _check(lib().or_write_kmers(self.h, threshold, os.fsencode(kmers_bin),
                                    os.fsencode(stat_txt) if stat_txt else None, C.byref(g)))
        return g.value

    def write_distribution(self, path):
        _check(lib().or_write_distribution(self.h, os.fsencode(path)))

    def count_seqs(self, seqs, k, min_len):
        _check(lib().or_count_seqs(self.h, seqs.h, k, min_len))
        return self


class Seqs:
    def __init__(self, h):
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().or_seqs_free(self.h)
                self.h = None
        except Exception:            # (interpreter shutdown: the module is already torn down)
            pass

    def __len__(self):
        return lib().or_seqs_count(self.h)

    def total_len(self):
        return lib().or_seqs_total_len(self.h)

    def get(self, i):
        p = C.c_void_p()
        n = C.c_uint64()
        a, mn, mx = C.c_int(), C.c_int(), C.c_int()
        _check(lib().or_seqs_get(self.h, i, C.byref(p), C.byref(n), C.byref(a), C.byref(mn), C.byref(mx)))
        return C.string_at(p, n.value).decode(), a.value, mn.value, mx.value

    def all(self):
        return [self.get(i) for i in range(len(self))]

    def write_fasta(self, path):
        _check(lib().or_seqs_write_fasta(self.h, os.fsencode(path)))


def unitig_census():
    """of the last build_unitigs call: (walks started, walks of at least min_len nucleotides, walks emitted)"""
    c = (C.c_uint64 * 3)()
    lib().or_unitig_census(c)
    return int(c[0]), int(c[1]), int(c[2])


def build_unitigs(table, k, freq_threshold, min_len):
    return Seqs(lib().or_build_unitigs(table.h, k, freq_threshold, min_len))


class Comps:
    def __init__(self, h):
        if not h:
            raise OracleError(lib().or_last_error().decode())
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().or_comps_free(self.h)
                self.h = None
        except Exception:            # (interpreter shutdown: the module is already torn down)
            pass

    def __len__(self):
        return lib().or_comps_count(self.h)

    def get(self, i):
        """-> (size, weight, thr, kmers uint64[] ascending)"""
        sz, w, t, p = C.c_uint64(), C.c_int64(), C.c_int(), C.c_void_p()
        _check(lib().or_comps_get(self.h, i, C.byref(sz), C.byref(w), C.byref(t), C.byref(p)))
        n = sz.value
        km = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(n,)).copy() if n else np.empty(0, np.uint64)
        return n, w.value, t.value, km

    def all(self):
        return [self.get(i) for i in range(len(self))]

    def write(self, components_bin, stat_txt=None):
        _check(lib().or_comps_write(self.h, os.fsencode(components_bin),
                                    os.fsencode(stat_txt) if stat_txt else None))

    def features(self, sample_table, threshold=0, selected=None):
        """selected: Table of the --selected k-mers (FeaturesCalculatorMain.java:113-116) or None"""
        n = len(self)
        vec = np.zeros(n, dtype=np.int64)
        br = np.zeros(n, dtype=np.float64)
        _check(lib().or_features_selected(self.h, sample_table.h, threshold, selected.h if selected is not None else None,
                                          vec.ctypes.data, br.ctypes.data))
        return vec, br


def features_from_reads(comps, bases, offsets, k, threshold=0, selected=None):
    """FeaturesCalculatorMain --reads branch: occurrences of the component k-mers in the reads (long counts)"""
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = len(comps)
    vec = np.zeros(n, dtype=np.int64)
    br = np.zeros(n, dtype=np.float64)
    _check(lib().or_features_reads_selected(comps.h, bases.ctypes.data, offsets.ctypes.data, len(offsets) - 1, k, threshold,
                                            selected.h if selected is not None else None, vec.ctypes.data, br.ctypes.data))
    return vec, br


def cut_components(table, k, b1, b2):
    return Comps(lib().or_cut_components(table.h, k, b1, b2))


def load_components(path):
    return Comps(lib().or_comps_load(os.fsencode(path)))


def bray_curtis(vecs):
    vecs = np.ascontiguousarray(vecs, dtype=np.int64)
    s, c = vecs.shape
    out = np.zeros((s, s), dtype=np.float64)
    _check(lib().or_bray_curtis(vecs.ctypes.data, s, c, out.ctypes.data))
    return out


def heatmap_order(matrix):
    """Leaf order of the heat map's dendrogram = the permutation the reference renumbers the matrix with.
    Restates FullHeatMap.clusterObjects (src/algo/FullHeatMap.java:221-296: average linkage over the ORIGINAL distances,
    O(n^3); the first closest pair in row-major order is merged with a strict '<'; the new node takes the smaller index,
    old node on the left) and renumber (:327-337: left subtree first).  Pure Python: n is the number of samples."""
    m = [[float(x) for x in row] for row in matrix]
    n = len(m)
    nodes = [("leaf", i) for i in range(n)]

    def group(node):
        if node is None:
            return []
        if node[0] == "leaf":
            return [node[1]]
        return group(node[1]) + group(node[2])

    def between(g1, g2):
        if not g1 or not g2:
            return -1.0
        s = 0.0
        for a in g1:
            for b in g2:
                s += m[a][b]
        return s / len(g1) / len(g2)

    dist = [[0.0] * n for _ in range(n)]
    for i in range(n):
        for j in range(i + 1, n):
            dist[i][j] = dist[j][i] = between([i], [j])
    count, root = n, (nodes[0] if n else None)
    while count > 1:
        best, bi, bj = float("inf"), -1, -1          # (Double.MAX_VALUE in the reference)
        for i in range(n):
            for j in range(i + 1, n):
                if nodes[i] is not None and nodes[j] is not None and dist[i][j] < best:
                    best, bi, bj = dist[i][j], i, j
        assert bi >= 0 and best >= 0
        root = ("node", nodes[bi], nodes[bj])
        nodes[bi], nodes[bj] = root, None
        g1 = group(root)
        for i in range(n):
            dist[i][bj] = dist[bj][i] = -1.0
            if i != bi:
                dist[i][bi] = dist[bi][i] = between(g1, group(nodes[i]))
        count -= 1
    return group(root)


def count_wide(bases, offsets, k, min_len=0):
    """NO-REFERENCE EXTENSION (32 <= k <= 63; the reference rejects k > 31): -> (hi, lo, counts, n_occ), ascending 2k-bit k-mers"""
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    cap = max(int(len(bases)), 1)
    hi = np.empty(cap, dtype=np.uint64); lo = np.empty(cap, dtype=np.uint64); cnt = np.empty(cap, dtype=np.int32)
    n = C.c_uint64(); n_occ = C.c_uint64()
    _check(lib().or_count_wide(bases.ctypes.data, offsets.ctypes.data, len(offsets) - 1, k, min_len, hi.ctypes.data, lo.ctypes.data,
                               cnt.ctypes.data, cap, C.byref(n), C.byref(n_occ)))
    return hi[:n.value].copy(), lo[:n.value].copy(), cnt[:n.value].copy(), n_occ.value


# ---------------------------------------------------------------------------------------------------------------------
# NO-REFERENCE EXTENSION: k <= 63 (oracle/mf_oracle_wide.c = mf_oracle_core.inc compiled for unsigned __int128 keys).
# A k-mer is a Python int on this side; arrays of k-mers are structured numpy arrays W128 (low word first).
# ---------------------------------------------------------------------------------------------------------------------
W128 = np.dtype([("lo", "<u8"), ("hi", "<u8")])


def w128_to_ints(a):
    return [(int(h) << 64) | int(l) for l, h in zip(a["lo"].tolist(), a["hi"].tolist())]


def _wcheck(rc):
    if rc < 0:
        raise OracleError(lib().orw_last_error().decode())


class WTable:
    """Table with 128-bit keys (any k <= 63)"""

    def __init__(self):
        self.h = lib().orw_table_new()

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().orw_table_free(self.h)
                self.h = None
        except Exception:
            pass

    def count_buffer(self, bases, offsets, k, min_len=0):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        _wcheck(lib().orw_count_buffer(self.h, bases.ctypes.data, offsets.ctypes.data, len(offsets) - 1, k, min_len))
        return self

    def add(self, key, inc=1):
        key = int(key)
        lib().orw_table_add2(self.h, key >> 64, key & 0xFFFFFFFFFFFFFFFF, int(inc))

    def get(self, key):
        key = int(key)
        return lib().orw_table_get2(self.h, key >> 64, key & 0xFFFFFFFFFFFFFFFF)

    def __len__(self):
        return lib().orw_table_size(self.h)

    def export(self, threshold=-(2 ** 31)):
        """-> (keys W128[n] ascending, vals int32[n]) of entries with value > threshold"""
        L = lib()
        n = L.orw_table_export(self.h, threshold, None, None, 0)
        keys = np.empty(n, dtype=W128)
        vals = np.empty(n, dtype=np.int32)
        if n:
            L.orw_table_export(self.h, threshold, keys.ctypes.data, vals.ctypes.data, n)
        return keys, vals

    def good(self, b):
        """the table KmersCounterMain hands on: entries with value > b (IOUtils.printKmers), as a new table"""
        keys, vals = self.export(b)
        g = WTable()
        for (lo, hi), v in zip(keys.tolist(), vals.tolist()):
            lib().orw_table_add2(g.h, hi, lo, v)
        return g

    def count_seqs(self, seqs, k, min_len):
        _wcheck(lib().orw_count_seqs(self.h, seqs.h, k, min_len))
        return self


class WSeqs:
    def __init__(self, h):
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().orw_seqs_free(self.h)
                self.h = None
        except Exception:
            pass

    def __len__(self):
        return lib().orw_seqs_count(self.h)

    def get(self, i):
        p = C.c_void_p()
        n = C.c_uint64()
        a, mn, mx = C.c_int(), C.c_int(), C.c_int()
        _wcheck(lib().orw_seqs_get(self.h, i, C.byref(p), C.byref(n), C.byref(a), C.byref(mn), C.byref(mx)))
        return C.string_at(p, n.value).decode(), a.value, mn.value, mx.value

    def all(self):
        return [self.get(i) for i in range(len(self))]


def wide_build_unitigs(table, k, freq_threshold, min_len):
    return WSeqs(lib().orw_build_unitigs(table.h, k, freq_threshold, min_len))


def wide_unitig_census():
    c = (C.c_uint64 * 3)()
    lib().orw_unitig_census(c)
    return int(c[0]), int(c[1]), int(c[2])


class WComps:
    def __init__(self, h):
        if not h:
            raise OracleError(lib().orw_last_error().decode())
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().orw_comps_free(self.h)
                self.h = None
        except Exception:
            pass

    def __len__(self):
        return lib().orw_comps_count(self.h)

    def get(self, i):
        """-> (size, weight, thr, kmers W128[] ascending)"""
        sz, w, t, p = C.c_uint64(), C.c_int64(), C.c_int(), C.c_void_p()
        _wcheck(lib().orw_comps_get(self.h, i, C.byref(sz), C.byref(w), C.byref(t), C.byref(p)))
        n = sz.value
        km = np.frombuffer(C.string_at(p, 16 * n), dtype=W128).copy() if n else np.empty(0, W128)
        return n, w.value, t.value, km

    def all(self):
        return [self.get(i) for i in range(len(self))]

    def features(self, sample_table, threshold=0, selected=None):
        n = len(self)
        vec = np.zeros(n, dtype=np.int64)
        br = np.zeros(n, dtype=np.float64)
        _wcheck(lib().orw_features_selected(self.h, sample_table.h, threshold, selected.h if selected is not None else None,
                                            vec.ctypes.data, br.ctypes.data))
        return vec, br


def wide_cut_components(table, k, b1, b2):
    return WComps(lib().orw_cut_components(table.h, k, b1, b2))


def wide_revcomp(kmer, k):
    kmer = int(kmer)
    out = (C.c_uint64 * 2)()
    lib().orw_revcomp2(kmer >> 64, kmer & 0xFFFFFFFFFFFFFFFF, k, out)
    return (int(out[1]) << 64) | int(out[0])


def run_pipeline_wide(samples, k, b=1, l=100, b1=1000, b2=10000):
    """run_pipeline on reads in memory with 128-bit keys (any k <= 63): samples = list of (bases, offsets).
    The steps of DistanceMatrixBuilderMain.java:88-175 exactly as run_pipeline wires them."""
    out = []
    for bases, offsets in samples:
        t = WTable().count_buffer(bases, offsets, k, 0)
        good = t.good(b)
        seqs = wide_build_unitigs(good, k, b, l)
        out.append(dict(table=t, good=good, seqs=seqs, n_distinct=len(t), n_good=len(good)))
    cutter = WTable()
    for s in out:
        cutter.count_seqs(s["seqs"], k, l)
    comps = wide_cut_components(cutter, k, b1, b2)
    vecs = np.array([comps.features(s["good"], 0)[0] for s in out], dtype=np.int64).reshape(len(out), len(comps))
    mat = bray_curtis(vecs) if len(comps) else None
    return dict(samples=out, cutter=cutter, comps=comps, vecs=vecs, matrix=mat)


def revcomp(kmer, k):
    return lib().or_revcomp(int(kmer), k)


def canonical(kmer, k):
    return lib().or_canonical(int(kmer), k)


def cpu_baseline_count(bases, offsets, k, threads):
    """Multi-threaded restatement of the reference counting loop; -> (n_distinct, n_occ)"""
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    occ = C.c_uint64()
    d = lib().or_cpu_baseline_count(bases.ctypes.data, offsets.ctypes.data, len(offsets) - 1, k, threads, C.byref(occ))
    return d, occ.value


def cpu_baseline_file(fasta, k, threads, bcut=1, kmers_bin=None):
    """kmer-counter on a FASTA file the way the reference runs it: serial reader inside the dispatcher's monitor, P counting
    workers, single-threaded dump.  -> dict(n_occ, distinct, written, reads, load_s, dump_s)"""
    res = (C.c_uint64 * 4)()
    sec = (C.c_double * 2)()
    rc = lib().or_cpu_baseline_file(fasta.encode(), k, threads, bcut, kmers_bin.encode() if kmers_bin else None, res, sec)
    if rc != 0:
        raise RuntimeError(f"cannot read {fasta}")
    return dict(n_occ=res[0], distinct=res[1], written=res[2], reads=res[3], load_s=sec[0], dump_s=sec[1])


def run_pipeline(files, k=31, b=1, l=100, b1=1000, b2=10000):
    """Whole default matrix-builder path on the oracle (DistanceMatrixBuilderMain.java:88-175).
    -> dict with per-sample stats, components, vectors, matrix."""
    files = sorted(files)                      # KmersCounterForManyFilesMain.java:73-74
    samples = []
    for f in files:
        t = Table().count_files([f], k, 0)
        keys, vals = t.export()
        good = Table()
        gk, gv = keys[vals > b], vals[vals > b]
        for kk, vv in zip(gk.tolist(), gv.tolist()):
            good.add(kk, vv)
        seqs = build_unitigs(good, k, b, l)
        samples.append(dict(file=f, table=t, good=good, seqs=seqs,
                            n_distinct=len(keys), n_good=len(gk)))
    cutter = Table()
    for s in samples:
        cutter.count_seqs(s["seqs"], k, l)
    comps = cut_components(cutter, k, b1, b2)
    vecs = []
    for s in samples:
        v, br = comps.features(s["good"], 0)
        vecs.append(v)
    vecs = np.array(vecs, dtype=np.int64).reshape(len(samples), len(comps))
    mat = bray_curtis(vecs) if len(comps) else None
    return dict(samples=samples, cutter=cutter, comps=comps, vecs=vecs, matrix=mat)

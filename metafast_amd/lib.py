"""ctypes binding of libmetafast_hip.so (the C-ABI in include/metafast_hip.h).

This is the host-side mirror of the reference's seams for the hot path
(IOUtils.loadReads / printKmers / loadKmers, SequencesFinders.thresholdStrategy,
ComponentsBuilder.splitStrategy, FeaturesCalculatorMain.buildAndPrintVector,
DistanceMatrixCalculatorMain.brayCurtisDistance).  There is no CPU fallback: if the
HIP library is missing or no GPU is present, calls fail loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("METAFAST_HIP_LIB") or os.path.join(_HERE, "lib", "libmetafast_hip.so")     # (override: A/B runs of two builds)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "metafast_hip.h")
_lib = None

u64, i64, i32, vp, cp = C.c_uint64, C.c_int64, C.c_int, C.c_void_p, C.c_char_p
pu64, pvp = C.POINTER(C.c_uint64), C.POINTER(C.c_void_p)

_SIGS = {
    "mf_last_error": (cp, []),
    "mf_version": (cp, []),
    "mf_ctx_create": (i32, [i32, i32, pvp]),
    "mf_ctx_destroy": (None, [vp]),
    "mf_ctx_set_stream": (i32, [vp, vp]),
    "mf_ctx_set_option": (i32, [vp, cp, i64]),
    "mf_ctx_synchronize": (i32, [vp]),
    "mf_ctx_trim": (i32, [vp]),
    "mf_ctx_trim_bytes": (i32, [vp, u64, C.POINTER(C.c_uint64)]),
    "mf_ctx_kernel_time": (i64, [vp, cp, C.POINTER(C.c_double)]),
    "mf_ctx_kernel_report": (i32, [vp, cp, u64]),
    "mf_ctx_reset_timers": (i32, [vp]),
    "mf_count_reads": (i32, [vp, C.POINTER(cp), i32, i32, i32, pvp]),
    "mf_table_drop_index": (i32, [vp]),
    "mf_count_wide_device": (i32, [vp, vp, vp, u64, u64, i32, i32, pvp]),
    "mf_wtable_destroy": (None, [vp]),
    "mf_wtable_stats": (i32, [vp, pu64, pu64, C.POINTER(C.c_int)]),
    "mf_wtable_export": (i32, [vp, vp, vp, vp, u64, pu64]),
    "mf_wtable_pieces": (i32, [vp, C.POINTER(C.c_uint32)]),
    "mf_wtable_piece_view": (i32, [vp, C.c_uint32, pvp, pvp, pvp, pu64]),
    "mf_count_wide_device_above": (i32, [vp, vp, vp, u64, u64, i32, i32, i32, pvp, pu64]),
    "mf_wtable_filter": (i32, [vp, i32, pvp]),
    "mf_wtable_drop_index": (i32, [vp]),
    "mf_wtable_lookup": (i32, [vp, vp, vp, u64, vp]),
    "mf_build_unitigs_wide_device": (i32, [vp, vp, i32, i32, pvp]),
    "mf_cut_components_wide_device": (i32, [vp, vp, i32, i32, pvp]),
    "mf_wcomps_destroy": (None, [vp]),
    "mf_wcomps_stats": (i32, [vp, pu64, pu64]),
    "mf_wcomps_export": (i32, [vp, vp, vp, vp, vp, vp, vp]),
    "mf_features_wide_device": (i32, [vp, vp, vp, i32, vp, vp]),
    "mf_comm_create_local": (i32, [vp, i32, vp]),
    "mf_comm_rccl_id": (i32, [vp]),
    "mf_comm_create_rccl": (i32, [vp, vp, i32, i32, pvp]),
    "mf_comm_create_external": (i32, [vp, i32, i32, vp, vp, pvp]),
    "mf_comm_destroy": (None, [vp]),
    "mf_comm_rank": (i32, [vp]),
    "mf_comm_world": (i32, [vp]),
    "mf_comm_kind": (cp, [vp]),
    "mf_comm_stat": (i64, [vp, cp]),
    "mf_comm_reset_stats": (i32, [vp]),
    "mf_comm_gather_ints": (i32, [vp, vp, i32, vp]),
    "mf_comm_all_gather": (i32, [vp, vp, vp, vp]),
    "mf_comm_all_to_all": (i32, [vp, vp, vp, vp, vp]),
    "mf_comm_gather_sequences": (i32, [vp, vp, vp, u64, u64, pvp]),
    "mf_cut_components_sharded": (i32, [vp, vp, vp, u64, u64, i32, i32, i32, i32, pvp]),
    "mf_cut_components_sharded_files": (i32, [vp, C.POINTER(cp), i32, i32, i32, i32, i32, cp, cp, pu64]),
    "mf_cut_components_of_shard": (i32, [vp, vp, i32, i32, i32, pvp, vp]),
    "mf_features_allgather": (i32, [vp, vp, u64, u64, vp, u64, pu64]),
    "mf_count_reads_above": (i32, [vp, C.POINTER(cp), i32, i32, i32, i32, pvp, pu64]),
    "mf_count_device": (i32, [vp, vp, vp, u64, u64, i32, i32, pvp]),
    "mf_count_device_above": (i32, [vp, vp, vp, u64, u64, i32, i32, i32, pvp, pu64]),
    "mf_table_destroy": (None, [vp]),
    "mf_table_stats": (i32, [vp, pu64, pu64]),
    "mf_table_occurrences": (i32, [vp, pu64]),
    "mf_table_records": (i32, [vp, pu64, C.POINTER(i32)]),
    "mf_table_hist": (i32, [vp, vp]),
    "mf_table_export": (i32, [vp, i32, vp, vp, u64, pu64]),
    "mf_table_device_view": (i32, [vp, pvp, pvp, pu64]),
    "mf_table_lookup": (i32, [vp, vp, u64, vp]),
    "mf_table_write_kmers": (i32, [vp, i32, cp, cp, pu64]),
    "mf_table_write_kmers_filtered": (i32, [vp, i32, vp, i32, cp, pu64]),
    "mf_table_load_kmers": (i32, [vp, C.POINTER(cp), i32, i32, i32, pvp]),
    "mf_table_filter": (i32, [vp, i32, pvp]),
    "mf_table_from_host": (i32, [vp, vp, vp, u64, i32, pvp]),
    "mf_count_device_shard": (i32, [vp, vp, vp, u64, u64, i32, i32, i32, i32, pvp]),
    "mf_dcc_create": (i32, [vp, vp, i32, i32, vp, pvp]),
    "mf_dcc_destroy": (None, [vp]),
    "mf_dcc_queries": (i32, [vp, vp]),
    "mf_dcc_queries_fill": (i32, [vp, vp]),
    "mf_dcc_answer": (i32, [vp, vp, u64, vp]),
    "mf_dcc_set_answers": (i32, [vp, vp, u64]),
    "mf_dcc_level_local": (i32, [vp, vp]),
    "mf_dcc_pairs_fill": (i32, [vp, vp]),
    "mf_dcc_pairs_complete": (i32, [vp, vp, u64]),
    "mf_dcc_merge": (i32, [vp, vp, u64, pu64]),
    "mf_dcc_stats_fill": (i32, [vp, vp]),
    "mf_dcc_classify": (i32, [vp, vp, u64, vp, u64, u64, i32, i32, i32, pu64, pu64]),
    "mf_dcc_world": (i32, [vp]),
    "mf_dcc_members_grouped": (i32, [vp, pu64, pu64]),
    "mf_dcc_members_grouped_fill": (i32, [vp, vp, vp]),
    "mf_dcc_finish_grouped": (i32, [vp, vp, u64, vp, u64, vp, vp, vp, vp, vp, u64, pvp]),
    "mf_dcc_kept_fill": (i32, [vp, vp]),
    "mf_dcc_members": (i32, [vp, pu64]),
    "mf_dcc_members_fill": (i32, [vp, vp, vp]),
    "mf_dcc_minkeys": (i32, [vp, vp, u64, vp]),
    "mf_dcc_finish": (i32, [vp, vp, vp, u64, vp, vp, vp, vp, vp, u64, pvp]),
    "mf_build_unitigs_device": (i32, [vp, vp, i32, i32, pvp]),
    "mf_seqs_destroy": (None, [vp]),
    "mf_seqs_stats": (i32, [vp, pu64, pu64]),
    "mf_seqs_device_view": (i32, [vp, pvp, pvp, pvp, pvp, pvp, pu64, pu64]),
    "mf_seqs_export": (i32, [vp, vp, vp, vp, vp, vp]),
    "mf_seqs_write_fasta": (i32, [vp, cp]),
    "mf_build_unitigs": (i32, [vp, vp, i32, i32, i32, cp, cp, pu64]),
    "mf_cut_components_device": (i32, [vp, vp, i32, i32, pvp]),
    "mf_comps_destroy": (None, [vp]),
    "mf_comps_stats": (i32, [vp, pu64, pu64]),
    "mf_comps_export": (i32, [vp, vp, vp, vp, vp, vp]),
    "mf_comps_write": (i32, [vp, cp, cp]),
    "mf_comps_load": (i32, [vp, cp, pvp]),
    "mf_cut_components": (i32, [vp, vp, i32, i32, i32, cp, cp, pu64]),
    "mf_features_device": (i32, [vp, vp, vp, i32, vp, vp]),
    "mf_features_reads_device": (i32, [vp, vp, vp, vp, u64, u64, i32, i32, vp, vp]),
    "mf_features_reads": (i32, [vp, cp, C.POINTER(cp), i32, i32, i32, cp, cp]),
    "mf_features": (i32, [vp, cp, cp, i32, i32, cp, cp]),
    "mf_features_device_selected": (i32, [vp, vp, vp, vp, i32, vp, vp]),
    "mf_features_selected": (i32, [vp, cp, cp, i32, i32, vp, cp, cp]),
    "mf_features_reads_device_selected": (i32, [vp, vp, vp, vp, u64, u64, i32, vp, i32, vp, vp]),
    "mf_features_reads_selected": (i32, [vp, cp, C.POINTER(cp), i32, i32, i32, vp, cp, cp]),
    "mf_bray_curtis": (i32, [vp, i32, i32, vp]),
    "mf_reads_load": (i32, [vp, C.POINTER(cp), i32, pvp]),
    "mf_reads_destroy": (None, [vp]),
    "mf_reads_stats": (i32, [vp, pu64, pu64]),
    "mf_reads_device_view": (i32, [vp, pvp, pvp]),
    "mf_reads_export": (i32, [vp, vp, vp]),
    "mf_ctx_stat": (i64, [vp, cp]),
    "mf_device_count": (i32, []),
    "mf_device_memory": (i32, [i32, pu64]),
    "mf_ctx_device": (i32, [vp]),
    "mf_ctx_bind_thread": (i32, [vp]),
    "mf_synth_reads_device": (i32, [vp, u64, i32, u64, u64, i32, u64, vp, vp]),
    "mf_synth_reads_host": (i32, [u64, i32, u64, u64, i32, u64, vp, vp]),
    "mf_synth_reads_device_ex": (i32, [vp, u64, i32, u64, u64, i32, u64, i32, vp, vp]),
    "mf_synth_reads_host_ex": (i32, [u64, i32, u64, u64, i32, u64, i32, vp, vp]),
}


class MetafastError(RuntimeError):
    """Mirrors ExecutionFailedException (itmo!/utils/tool/Tool.java:450-463)."""


class DistAbort(MetafastError):
    """MF_ERR_TOGETHER: a rank could not do its part of an exchange step; EVERY rank raises this from the same call (the status rides on
    the integer gathers), so that all of them can take another route together instead of one rank raising while its peers wait inside
    a collective"""


def exported_symbols():
    """Every entry point declared in include/metafast_hip.h."""
    return sorted(_SIGS)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MetafastError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            f = getattr(L, name)          # AttributeError if the .so does not export a declared symbol
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def _check(rc):
    if rc < 0:
        raise (DistAbort if rc == -2 else MetafastError)(lib().mf_last_error().decode(errors="replace"))
    return rc


def _cfiles(files):
    return (C.c_char_p * len(files))(*[os.fsencode(f) for f in files])


def _opt(path):
    return os.fsencode(path) if path else None


class Context:
    def __init__(self, device=0, host_threads=0, stream=None):
        h = C.c_void_p()
        _check(lib().mf_ctx_create(device, host_threads or (os.cpu_count() or 1), C.byref(h)))
        self.h = h
        self.device = device
        self.stream_handle = None                     # the raw stream the library launches on when the caller gave one (else: its own)
        if stream is not None:
            self.set_stream(stream)

    def close(self):
        c = getattr(self, "_mf_comm", None)             # (the communicator pipeline.py made for this context)
        if c is not None:
            self._mf_comm = None
            c.close()
        peer = getattr(self, "_mf_peer", None)          # (... and the peer context for the samples it overlaps)
        if peer is not None:
            self._mf_peer = None
            peer.close()
        if getattr(self, "h", None) and _lib is not None:
            _lib.mf_ctx_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream):
        """stream: a raw hipStream_t (int), or a torch.cuda.Stream"""
        raw = getattr(stream, "cuda_stream", stream)
        _check(lib().mf_ctx_set_stream(self.h, C.c_void_p(int(raw) if raw else None)))
        self.stream_handle = int(raw) if raw else 0

    def bind_thread(self):
        """a thread other than the one that made the context calls this before its first call on it (HIP's current device is per thread)"""
        _check(lib().mf_ctx_bind_thread(self.h))

    def set_option(self, name, value):
        _check(lib().mf_ctx_set_option(self.h, name.encode(), int(value)))
        if not hasattr(self, "options"):
            self.options = {}
        self.options[name] = int(value)          # (what a peer context of this one is given too: pipeline.py)

    def synchronize(self):
        _check(lib().mf_ctx_synchronize(self.h))

    def trim(self, want_bytes=None):
        """give idle workspace back to the driver: all of it, or (want_bytes) the smallest idle regions until that much is free
        again -- returns the bytes given back in that case"""
        if want_bytes is None:
            _check(lib().mf_ctx_trim(self.h))
            return None
        got = C.c_uint64(0)
        _check(lib().mf_ctx_trim_bytes(self.h, int(want_bytes), C.byref(got)))
        return int(got.value)

    def reset_timers(self):
        peer = getattr(self, "_mf_peer", None)
        if peer is not None and peer.h is not None:
            peer.reset_timers()
        _check(lib().mf_ctx_reset_timers(self.h))

    def stat(self, name):
        """counters and gauges of the context: slice_restarts, device_parsed_files, device_parser_stepped_back, hipmalloc_calls, ..."""
        v = lib().mf_ctx_stat(self.h, name.encode())
        if v < 0:
            raise MetafastError(lib().mf_last_error().decode(errors="replace"))
        return int(v)

    def kernel_time(self, name):
        ms = C.c_double()
        n = lib().mf_ctx_kernel_time(self.h, name.encode(), C.byref(ms))
        return int(n), float(ms.value)

    def kernel_report(self):
        """name -> (launches, total ms, longest launch ms); the peer context's launches (pipeline.py: overlapped samples) are added in"""
        buf = C.create_string_buffer(1 << 16)
        _check(lib().mf_ctx_kernel_report(self.h, buf, len(buf)))
        out = {}
        for line in buf.value.decode().splitlines():
            name, n, ms, mx = line.split("\t")
            out[name] = (int(n), float(ms), float(mx))
        peer = getattr(self, "_mf_peer", None)
        if peer is not None and peer.h is not None:
            for name, (n, ms, mx) in peer.kernel_report().items():
                a = out.get(name, (0, 0.0, 0.0))
                out[name] = (a[0] + n, a[1] + ms, max(a[2], mx))
        return out

    # ---- A1-A4 ----
    def load_reads(self, files):
        """the readers alone (ReadersUtils.readDnaLazy): -> (bases uint8[], offsets uint64[n + 1]) on the host"""
        h = C.c_void_p()
        _check(lib().mf_reads_load(self.h, _cfiles(files), len(files), C.byref(h)))
        try:
            n, nb = C.c_uint64(), C.c_uint64()
            _check(lib().mf_reads_stats(h, C.byref(n), C.byref(nb)))
            bases = np.empty(nb.value, dtype=np.uint8)
            off = np.empty(n.value + 1, dtype=np.uint64)
            _check(lib().mf_reads_export(h, bases.ctypes.data, off.ctypes.data))
            return bases, off
        finally:
            lib().mf_reads_destroy(h)

    def count_reads(self, files, k, min_read_len=0):
        """IOUtils.loadReads (src/io/IOUtils.java:772-803)"""
        t = C.c_void_p()
        _check(lib().mf_count_reads(self.h, _cfiles(files), len(files), k, min_read_len, C.byref(t)))
        return Table(self, t)

    def count_reads_above(self, files, k, threshold, min_read_len=0):
        """KmersCounterMain.runImpl (src/tools/KmersCounterMain.java:77-99): load + the cut value > threshold of printKmers,
        made inside the counting kernels; -> (Table of the kept k-mers, distinct k-mers before the cut)"""
        t = C.c_void_p()
        n_all = C.c_uint64()
        _check(lib().mf_count_reads_above(self.h, _cfiles(files), len(files), k, min_read_len, threshold, C.byref(t), C.byref(n_all)))
        return Table(self, t), n_all.value

    def count_device(self, d_bases, d_offsets, n_reads, n_bases, k, min_read_len=0):
        """d_bases / d_offsets: raw device pointers (int) -- e.g. torch tensor.data_ptr()"""
        t = C.c_void_p()
        _check(lib().mf_count_device(self.h, C.c_void_p(d_bases), C.c_void_p(d_offsets), n_reads, n_bases, k,
                                     min_read_len, C.byref(t)))
        return Table(self, t)

    def count_wide_device(self, d_bases, d_offsets, n_reads, n_bases, k, min_read_len=0):
        """NO-REFERENCE EXTENSION, 32 <= k <= 63 (mf_wide.hip) -> dict(hi, lo, counts: ascending 2k-bit k-mers; n_occ)"""
        t = C.c_void_p()
        _check(lib().mf_count_wide_device(self.h, C.c_void_p(d_bases), C.c_void_p(d_offsets), n_reads, n_bases, k, min_read_len, C.byref(t)))
        try:
            n, occ, kk = C.c_uint64(), C.c_uint64(), C.c_int()
            _check(lib().mf_wtable_stats(t, C.byref(n), C.byref(occ), C.byref(kk)))
            hi = np.empty(n.value, dtype=np.uint64); lo = np.empty(n.value, dtype=np.uint64); cnt = np.empty(n.value, dtype=np.uint16)
            m = C.c_uint64()
            _check(lib().mf_wtable_export(t, hi.ctypes.data, lo.ctypes.data, cnt.ctypes.data, n.value, C.byref(m)))
            return dict(hi=hi, lo=lo, counts=cnt, n_occ=occ.value, k=kk.value)
        finally:
            lib().mf_wtable_destroy(t)

    def count_wide_table(self, d_bases, d_offsets, n_reads, n_bases, k, min_read_len=0):
        """NO-REFERENCE EXTENSION, 32 <= k <= 63: the table left in HBM -> WideTable (pieces(): device pointers)"""
        t = C.c_void_p()
        _check(lib().mf_count_wide_device(self.h, C.c_void_p(d_bases), C.c_void_p(d_offsets), n_reads, n_bases, k, min_read_len, C.byref(t)))
        return WideTable(self, t)

    def count_wide_above(self, d_bases, d_offsets, n_reads, n_bases, k, threshold, min_read_len=0):
        """NO-REFERENCE EXTENSION, 32 <= k <= 63: count_wide_table with the cut count > threshold inside the pass
        -> (WideTable, distinct k-mers before the cut)"""
        t = C.c_void_p()
        n_all = C.c_uint64()
        _check(lib().mf_count_wide_device_above(self.h, C.c_void_p(d_bases), C.c_void_p(d_offsets), n_reads, n_bases, k, min_read_len, threshold,
                                                C.byref(t), C.byref(n_all)))
        return WideTable(self, t), n_all.value

    def build_unitigs_wide(self, table, freq_threshold, min_len):
        """NO-REFERENCE EXTENSION: build_unitigs on a WideTable -> Seqs"""
        s = C.c_void_p()
        _check(lib().mf_build_unitigs_wide_device(self.h, table.h, freq_threshold, min_len, C.byref(s)))
        return Seqs(self, s)

    def cut_components_wide(self, cutter, b1, b2):
        """NO-REFERENCE EXTENSION: cut_components on a WideTable -> WideComps"""
        c = C.c_void_p()
        _check(lib().mf_cut_components_wide_device(self.h, cutter.h, b1, b2, C.byref(c)))
        return WideComps(self, c)

    def features_wide(self, comps, sample, threshold=0):
        n = len(comps)
        vec = np.zeros(n, dtype=np.int64)
        br = np.zeros(n, dtype=np.float64)
        _check(lib().mf_features_wide_device(self.h, comps.h, sample.h, threshold, vec.ctypes.data, br.ctypes.data))
        return vec, br

    def count_device_above(self, d_bases, d_offsets, n_reads, n_bases, k, threshold, min_read_len=0):
        """count, keeping only the k-mers with count > threshold (what the k-mer counter hands on, IOUtils.printKmers);
        -> (Table of the kept k-mers, number of distinct k-mers before the cut)"""
        t = C.c_void_p()
        n_all = C.c_uint64()
        _check(lib().mf_count_device_above(self.h, C.c_void_p(d_bases), C.c_void_p(d_offsets), n_reads, n_bases, k,
                                           min_read_len, threshold, C.byref(t), C.byref(n_all)))
        return Table(self, t), n_all.value

    def load_kmers(self, files, freq_threshold, k):
        """IOUtils.loadKmers (src/io/IOUtils.java:369-401)"""
        t = C.c_void_p()
        _check(lib().mf_table_load_kmers(self.h, _cfiles(files), len(files), freq_threshold, k, C.byref(t)))
        return Table(self, t)

    def table_from_host(self, keys, counts, k):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        counts = np.ascontiguousarray(counts, dtype=np.uint16)
        t = C.c_void_p()
        _check(lib().mf_table_from_host(self.h, keys.ctypes.data, counts.ctypes.data, len(keys), k, C.byref(t)))
        return Table(self, t)

    def count_device_shard(self, d_bases, d_offsets, n_seqs, n_bases, k, min_len, rank, world):
        """this rank's shard (k-mers whose minimizer-partition hash starts with `rank`) of the table of all the sequences"""
        t = C.c_void_p()
        _check(lib().mf_count_device_shard(self.h, d_bases, d_offsets, n_seqs, n_bases, k, min_len, rank, world, C.byref(t)))
        return Table(self, t)

    # ---- A7 ----
    def build_unitigs(self, table, freq_threshold, min_len):
        """SequencesFinders.thresholdStrategy (src/algo/SequencesFinders.java:13-31)"""
        s = C.c_void_p()
        _check(lib().mf_build_unitigs_device(self.h, table.h, freq_threshold, min_len, C.byref(s)))
        return Seqs(self, s)

    # ---- A10 ----
    def cut_components(self, cutter_table, b1, b2):
        """ComponentsBuilder.splitStrategy (src/algo/ComponentsBuilder.java:24-32)"""
        c = C.c_void_p()
        _check(lib().mf_cut_components_device(self.h, cutter_table.h, b1, b2, C.byref(c)))
        return Comps(self, c)

    def load_components(self, path):
        c = C.c_void_p()
        _check(lib().mf_comps_load(self.h, os.fsencode(path), C.byref(c)))
        return Comps(self, c)

    # ---- A12 ----
    def features(self, comps, sample_table, threshold=0, selected=None):
        """selected: Table of the --selected k-mers (FeaturesCalculatorMain.java:113-116, 193) or None"""
        n = comps.stats()[0]
        vec = np.zeros(n, dtype=np.int64)
        br = np.zeros(n, dtype=np.float64)
        _check(lib().mf_features_device_selected(self.h, comps.h, sample_table.h, selected.h if selected is not None else None,
                                                 threshold, vec.ctypes.data, br.ctypes.data))
        return vec, br

    def features_reads(self, comps, d_bases, d_offsets, n_reads, n_bases, k, threshold=0, selected=None):
        """features of a sample straight from its reads in HBM (--use-reads-for-calculating-features): 64-bit counts"""
        n = comps.stats()[0]
        vec = np.zeros(n, dtype=np.int64)
        br = np.zeros(n, dtype=np.float64)
        _check(lib().mf_features_reads_device_selected(self.h, comps.h, C.c_void_p(d_bases), C.c_void_p(d_offsets), n_reads, n_bases, k,
                                                       selected.h if selected is not None else None, threshold,
                                                       vec.ctypes.data, br.ctypes.data))
        return vec, br

    def features_reads_files(self, components_bin, files, k, threshold, vec_path, breadth_path, selected=None):
        _check(lib().mf_features_reads_selected(self.h, os.fsencode(components_bin), _cfiles(files), len(files), k, threshold,
                                                selected.h if selected is not None else None, _opt(vec_path), _opt(breadth_path)))

    def features_files(self, components_bin, kmers_bin, k, threshold, vec_path, breadth_path, selected=None):
        _check(lib().mf_features_selected(self.h, os.fsencode(components_bin), os.fsencode(kmers_bin), k, threshold,
                                          selected.h if selected is not None else None, _opt(vec_path), _opt(breadth_path)))

    # ---- synthetic reads ----
    def synth_reads_device(self, seed, sample, first_read, n_reads, read_len, genome_scale_bp, d_bases, d_offsets, sub_per_16384=82):
        """sub_per_16384: substitutions per 16384 bases (82 = 0.5 %, the benchmark's; 164 = 1 %, BASELINE config 5)"""
        _check(lib().mf_synth_reads_device_ex(self.h, seed, sample, first_read, n_reads, read_len, genome_scale_bp, sub_per_16384,
                                              C.c_void_p(d_bases), C.c_void_p(d_offsets)))


class WideTable:
    """NO-REFERENCE EXTENSION: canonical counts of 2k-bit k-mers (32 <= k <= 63) in HBM, in ascending pieces"""

    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h

    def close(self):
        if self.h:
            lib().mf_wtable_destroy(self.h)
            self.h = None

    __del__ = close

    def stats(self):
        n, occ, kk = C.c_uint64(), C.c_uint64(), C.c_int()
        _check(lib().mf_wtable_stats(self.h, C.byref(n), C.byref(occ), C.byref(kk)))
        return n.value, occ.value, kk.value

    def export(self):
        """-> (hi, lo, counts) ascending"""
        n = self.stats()[0]
        hi = np.empty(n, dtype=np.uint64); lo = np.empty(n, dtype=np.uint64); cnt = np.empty(n, dtype=np.uint16)
        m = C.c_uint64()
        _check(lib().mf_wtable_export(self.h, hi.ctypes.data, lo.ctypes.data, cnt.ctypes.data, n, C.byref(m)))
        return hi, lo, cnt

    def filter(self, threshold):
        t = C.c_void_p()
        _check(lib().mf_wtable_filter(self.h, threshold, C.byref(t)))
        return WideTable(self.ctx, t)

    def drop_index(self):
        _check(lib().mf_wtable_drop_index(self.h))

    def lookup(self, kmers):
        """kmers: iterable of Python ints (2k-bit k-mers) -> int32 counts, -1 = absent"""
        ks = [int(x) for x in kmers]
        hi = np.array([x >> 64 for x in ks], dtype=np.uint64); lo = np.array([x & 0xFFFFFFFFFFFFFFFF for x in ks], dtype=np.uint64)
        out = np.empty(len(ks), dtype=np.int32)
        _check(lib().mf_wtable_lookup(self.h, hi.ctypes.data, lo.ctypes.data, len(ks), out.ctypes.data))
        return out

    def pieces(self):
        """[(d_hi, d_lo, d_counts, n)]: raw device pointers of every piece (uint64, uint64, uint16)"""
        m = C.c_uint32()
        _check(lib().mf_wtable_pieces(self.h, C.byref(m)))
        out = []
        for i in range(m.value):
            a, b, c, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64()
            _check(lib().mf_wtable_piece_view(self.h, i, C.byref(a), C.byref(b), C.byref(c), C.byref(n)))
            out.append((a.value or 0, b.value or 0, c.value or 0, n.value))
        return out


class Reads:
    """sequences in HBM (bases + offsets) that the library owns: mf_reads (here: the gathered unitigs of all ranks)"""

    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h

    def close(self):
        if getattr(self, "h", None) and _lib is not None and getattr(self.ctx, "h", None):
            _lib.mf_reads_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stats(self):
        n, nb = C.c_uint64(), C.c_uint64()
        _check(lib().mf_reads_stats(self.h, C.byref(n), C.byref(nb)))
        return n.value, nb.value

    def device_view(self):
        b, o = C.c_void_p(), C.c_void_p()
        _check(lib().mf_reads_device_view(self.h, C.byref(b), C.byref(o)))
        n, nb = self.stats()
        return dict(bases=b.value or 0, offsets=o.value or 0, n=n, n_bases=nb)

    def export(self):
        n, nb = self.stats()
        bases = np.empty(nb, dtype=np.uint8)
        off = np.empty(n + 1, dtype=np.uint64)
        _check(lib().mf_reads_export(self.h, bases.ctypes.data, off.ctypes.data))
        return bases, off


_OPS_GATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_int64))
_OPS_ALLGATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64))
_OPS_ALLTOALL = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint64))


class _CommOps(C.Structure):
    _fields_ = [("gather_ints", _OPS_GATHER), ("all_gather", _OPS_ALLGATHER), ("all_to_all", _OPS_ALLTOALL)]


class Comm:
    """mf_comm (include/metafast_hip.h): the exchanges of the multi-GPU path behind the C-ABI.  local(): one communicator per context for
    threads of this process; rccl(): one process per GPU; external(): the caller's own primitives (torch.distributed in pipeline.py)."""

    def __init__(self, ctx, h, keep=None):
        self.ctx, self.h, self._keep = ctx, h, keep
        self.rank, self.world = lib().mf_comm_rank(h), lib().mf_comm_world(h)
        self.kind = lib().mf_comm_kind(h).decode()

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.mf_comm_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def local(ctxs):
        n = len(ctxs)
        arr = (C.c_void_p * n)(*[c.h for c in ctxs])
        out = (C.c_void_p * n)()
        _check(lib().mf_comm_create_local(arr, n, out))
        return [Comm(ctxs[r], C.c_void_p(out[r])) for r in range(n)]

    @staticmethod
    def rccl_id():
        buf = C.create_string_buffer(128)
        _check(lib().mf_comm_rccl_id(buf))
        return buf.raw

    @staticmethod
    def rccl(ctx, id128, rank, world):
        h = C.c_void_p()
        _check(lib().mf_comm_create_rccl(ctx.h, C.c_char_p(bytes(id128)), rank, world, C.byref(h)))
        return Comm(ctx, h)

    @staticmethod
    def external(ctx, rank, world, gather_ints, all_gather, all_to_all):
        """gather_ints(list of int) -> int64 array [world, n]; all_gather(d_send, d_recv, bytes[world]); all_to_all(d_send, send_bytes[world],
        d_recv, recv_bytes[world]) -- device pointers as ints, the data has arrived when the call returns"""
        def _g(user, vals, n, out):
            try:
                m = np.asarray(gather_ints([int(vals[i]) for i in range(n)]), dtype=np.int64).reshape(-1)
                for i in range(world * n):
                    out[i] = int(m[i])
                return 0
            except Exception as e:              # (a C caller sees a status, not a Python exception)
                print("[metafast_amd] external gather_ints failed: %r" % (e,), file=__import__("sys").stderr)
                return -1

        def _ag(user, d_send, d_recv, nb):
            try:
                all_gather(d_send or 0, d_recv or 0, [int(nb[r]) for r in range(world)])
                return 0
            except Exception as e:
                print("[metafast_amd] external all_gather failed: %r" % (e,), file=__import__("sys").stderr)
                return -1

        def _aa(user, d_send, sb, d_recv, rb):
            try:
                all_to_all(d_send or 0, [int(sb[r]) for r in range(world)], d_recv or 0, [int(rb[r]) for r in range(world)])
                return 0
            except Exception as e:
                print("[metafast_amd] external all_to_all failed: %r" % (e,), file=__import__("sys").stderr)
                return -1
        ops = _CommOps(_OPS_GATHER(_g), _OPS_ALLGATHER(_ag), _OPS_ALLTOALL(_aa))
        h = C.c_void_p()
        _check(lib().mf_comm_create_external(ctx.h, rank, world, C.byref(ops), None, C.byref(h)))
        return Comm(ctx, h, keep=ops)

    def stats(self):
        return dict(collectives=int(lib().mf_comm_stat(self.h, b"collectives")), bytes_in=int(lib().mf_comm_stat(self.h, b"bytes_in")),
                    seconds=lib().mf_comm_stat(self.h, b"us") / 1e6)

    def reset_stats(self):
        _check(lib().mf_comm_reset_stats(self.h))

    def gather_ints(self, vals):
        v = np.ascontiguousarray(vals, dtype=np.int64)
        out = np.zeros((self.world, len(v)), dtype=np.int64)
        _check(lib().mf_comm_gather_ints(self.h, v.ctypes.data, len(v), out.ctypes.data))
        return out

    def all_gather(self, d_send, d_recv, bytes_per_rank):
        b = np.ascontiguousarray(bytes_per_rank, dtype=np.uint64)
        _check(lib().mf_comm_all_gather(self.h, C.c_void_p(d_send), C.c_void_p(d_recv), b.ctypes.data))

    def all_to_all(self, d_send, send_bytes, d_recv, recv_bytes):
        sb = np.ascontiguousarray(send_bytes, dtype=np.uint64); rb = np.ascontiguousarray(recv_bytes, dtype=np.uint64)
        _check(lib().mf_comm_all_to_all(self.h, C.c_void_p(d_send), sb.ctypes.data, C.c_void_p(d_recv), rb.ctypes.data))

    def gather_sequences(self, d_bases, d_offsets, n_seqs, n_bases):
        """every rank's sequences on every rank, rank after rank -> Reads"""
        h = C.c_void_p()
        _check(lib().mf_comm_gather_sequences(self.h, C.c_void_p(d_bases), C.c_void_p(d_offsets), n_seqs, n_bases, C.byref(h)))
        return Reads(self.ctx, h)

    def cut_components_sharded(self, d_bases, d_offsets, n_seqs, n_bases, k, min_len, b1, b2):
        """ComponentCutterMain.runImpl over all ranks in one call (this rank's sequences in, the same Comps on every rank out)"""
        c = C.c_void_p()
        _check(lib().mf_cut_components_sharded(self.h, C.c_void_p(d_bases), C.c_void_p(d_offsets), n_seqs, n_bases, k, min_len, b1, b2, C.byref(c)))
        return Comps(self.ctx, c)

    def cut_components_of_shard(self, shard, k, b1, b2):
        """... on a shard the caller has counted (None: this rank has none; all ranks raise DistAbort) -> (Comps, info)"""
        c = C.c_void_p()
        info = np.zeros(4, dtype=np.uint64)
        _check(lib().mf_cut_components_of_shard(self.h, shard.h if shard is not None else None, k, b1, b2, C.byref(c), info.ctypes.data))
        return Comps(self.ctx, c), dict(levels=int(info[0]), queries=int(info[1]), members=int(info[2]), vertices=int(info[3]))

    def features_allgather(self, rows):
        """rows int64[n_local_samples, C] -> int64[n_all_samples, C], rank-major"""
        rows = np.ascontiguousarray(rows, dtype=np.int64).reshape(len(rows), -1)
        n = C.c_uint64()
        _check(lib().mf_features_allgather(self.h, rows.ctypes.data, rows.shape[0], rows.shape[1], None, 0, C.byref(n)))
        out = np.zeros((n.value, rows.shape[1]), dtype=np.int64)
        _check(lib().mf_features_allgather(self.h, rows.ctypes.data, rows.shape[0], rows.shape[1], out.ctypes.data, n.value, C.byref(n)))
        return out


class WideComps:
    """NO-REFERENCE EXTENSION: connected components of 2k-bit k-mers (32 <= k <= 63)"""

    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h

    def close(self):
        if self.h:
            lib().mf_wcomps_destroy(self.h)
            self.h = None

    __del__ = close

    def stats(self):
        n, nk = C.c_uint64(), C.c_uint64()
        _check(lib().mf_wcomps_stats(self.h, C.byref(n), C.byref(nk)))
        return n.value, nk.value

    def __len__(self):
        return self.stats()[0]

    def export(self):
        """-> dict(sizes, weights, thr, offsets, hi, lo): k-mers grouped by component, ascending inside"""
        n, nk = self.stats()
        sizes = np.zeros(n, dtype=np.uint64); weights = np.zeros(n, dtype=np.int64); thr = np.zeros(n, dtype=np.int32)
        off = np.zeros(n + 1, dtype=np.uint64); hi = np.zeros(nk, dtype=np.uint64); lo = np.zeros(nk, dtype=np.uint64)
        _check(lib().mf_wcomps_export(self.h, sizes.ctypes.data, weights.ctypes.data, thr.ctypes.data, off.ctypes.data, hi.ctypes.data, lo.ctypes.data))
        return dict(sizes=sizes, weights=weights, thr=thr, offsets=off, hi=hi, lo=lo)


class Table:
    """BigLong2ShortHashMap stand-in: canonical k-mer -> saturating count, resident in HBM."""

    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h

    def close(self):
        if getattr(self, "h", None) and _lib is not None and getattr(self.ctx, "h", None):
            _lib.mf_table_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stats(self):
        a, b = C.c_uint64(), C.c_uint64()
        _check(lib().mf_table_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def __len__(self):
        a = C.c_uint64()
        _check(lib().mf_table_stats(self.h, C.byref(a), None))
        return a.value

    def occurrences(self):
        a = C.c_uint64()
        _check(lib().mf_table_occurrences(self.h, C.byref(a)))
        return a.value

    def records(self):
        """-> (records the counting pass partitioned, bytes per record); measurement only"""
        a, b = C.c_uint64(), C.c_int32()
        _check(lib().mf_table_records(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def hist(self):
        """number of distinct k-mers per count over ALL counted k-mers (those a cut dropped included): the .stat.txt rows"""
        h = np.zeros(32768, dtype=np.uint64)
        _check(lib().mf_table_hist(self.h, h.ctypes.data))
        return h

    def export(self, threshold=-1):
        """-> (keys uint64[n] ascending, counts uint16[n]) with count > threshold"""
        n = C.c_uint64()
        _check(lib().mf_table_export(self.h, threshold, None, None, 0, C.byref(n)))
        keys = np.empty(n.value, dtype=np.uint64)
        cnts = np.empty(n.value, dtype=np.uint16)
        if n.value:
            _check(lib().mf_table_export(self.h, threshold, keys.ctypes.data, cnts.ctypes.data, n.value, C.byref(n)))
        return keys, cnts

    def device_view(self):
        k, c, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
        _check(lib().mf_table_device_view(self.h, C.byref(k), C.byref(c), C.byref(n)))
        return k.value, c.value, n.value

    def lookup(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        out = np.empty(len(keys), dtype=np.int32)
        _check(lib().mf_table_lookup(self.h, keys.ctypes.data, len(keys), out.ctypes.data))
        return out

    def drop_index(self):
        """give the lookup index back (rebuilt on the next lookup): several samples per GPU"""
        _check(lib().mf_table_drop_index(self.h))

    def write_kmers(self, threshold, kmers_bin, stat_txt=None):
        """IOUtils.printKmers (src/io/IOUtils.java:45-71)"""
        g = C.c_uint64()
        _check(lib().mf_table_write_kmers(self.h, threshold, os.fsencode(kmers_bin), _opt(stat_txt), C.byref(g)))
        return g.value

    def write_kmers_filtered(self, threshold, filter_table, filter_threshold, kmers_bin):
        """IOUtils.filterAndPrintKmers (src/io/IOUtils.java:101-123)"""
        g = C.c_uint64()
        _check(lib().mf_table_write_kmers_filtered(self.h, threshold, filter_table.h, filter_threshold, os.fsencode(kmers_bin), C.byref(g)))
        return g.value

    def filter(self, threshold):
        t = C.c_void_p()
        _check(lib().mf_table_filter(self.h, threshold, C.byref(t)))
        return Table(self.ctx, t)


class Seqs:
    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h

    def close(self):
        if getattr(self, "h", None) and _lib is not None and getattr(self.ctx, "h", None):
            _lib.mf_seqs_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stats(self):
        a, b = C.c_uint64(), C.c_uint64()
        _check(lib().mf_seqs_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def __len__(self):
        return self.stats()[0]

    def device_view(self):
        p = [C.c_void_p() for _ in range(5)]
        n, nb = C.c_uint64(), C.c_uint64()
        _check(lib().mf_seqs_device_view(self.h, *[C.byref(x) for x in p], C.byref(n), C.byref(nb)))
        return dict(bases=p[0].value, offsets=p[1].value, avg=p[2].value, min=p[3].value, max=p[4].value,
                    n=n.value, n_bases=nb.value)

    def export(self):
        """-> list of (sequence str, avg, min, max)"""
        n, nb = self.stats()
        bases = np.empty(max(nb, 1), dtype=np.uint8)
        off = np.empty(n + 1, dtype=np.uint64)
        a = np.empty(max(n, 1), dtype=np.int32)
        mn = np.empty(max(n, 1), dtype=np.int32)
        mx = np.empty(max(n, 1), dtype=np.int32)
        _check(lib().mf_seqs_export(self.h, bases.ctypes.data, off.ctypes.data, a.ctypes.data, mn.ctypes.data,
                                    mx.ctypes.data))
        raw = bases.tobytes()
        return [(raw[int(off[i]):int(off[i + 1])].decode(), int(a[i]), int(mn[i]), int(mx[i])) for i in range(n)]

    def write_fasta(self, path):
        _check(lib().mf_seqs_write_fasta(self.h, os.fsencode(path)))


class Comps:
    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h

    def close(self):
        if getattr(self, "h", None) and _lib is not None and getattr(self.ctx, "h", None):
            _lib.mf_comps_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stats(self):
        a, b = C.c_uint64(), C.c_uint64()
        _check(lib().mf_comps_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def __len__(self):
        return self.stats()[0]

    def export(self):
        """-> list of (size, weight, thr, kmers uint64[] ascending)"""
        n, nk = self.stats()
        sizes = np.empty(max(n, 1), dtype=np.uint64)
        w = np.empty(max(n, 1), dtype=np.int64)
        thr = np.empty(max(n, 1), dtype=np.int32)
        off = np.empty(n + 1, dtype=np.uint64)
        km = np.empty(max(nk, 1), dtype=np.uint64)
        _check(lib().mf_comps_export(self.h, sizes.ctypes.data, w.ctypes.data, thr.ctypes.data, off.ctypes.data,
                                     km.ctypes.data))
        return [(int(sizes[i]), int(w[i]), int(thr[i]), km[int(off[i]):int(off[i + 1])].copy()) for i in range(n)]

    def write(self, components_bin, stat_txt=None):
        _check(lib().mf_comps_write(self.h, os.fsencode(components_bin), _opt(stat_txt)))


class DistCutter:
    """One rank's part of the distributed component cutter (mf_dcc_* in include/metafast_hip.h): it owns a shard of the
    cutter table; the exchange steps in between are the caller's (metafast_amd/pipeline.py: distributed_components)."""

    def __init__(self, ctx, shard, rank, world, base):
        self.ctx, self.shard = ctx, shard
        base = np.ascontiguousarray(base, dtype=np.uint32)
        h = C.c_void_p()
        _check(lib().mf_dcc_create(ctx.h, shard.h, rank, world, base.ctypes.data, C.byref(h)))
        self.h, self.world, self.rank = h, world, rank

    def close(self):
        if getattr(self, "h", None) and _lib is not None and getattr(self.ctx, "h", None):
            _lib.mf_dcc_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _counts(self, fn):
        c = np.zeros(self.world, dtype=np.uint64)
        _check(fn(self.h, c.ctypes.data))
        return c

    def queries(self):
        return self._counts(lib().mf_dcc_queries)

    def queries_fill(self, d_q):
        _check(lib().mf_dcc_queries_fill(self.h, d_q))

    def answer(self, d_q, n, d_a):
        _check(lib().mf_dcc_answer(self.h, d_q, n, d_a))

    def set_answers(self, d_a, n):
        _check(lib().mf_dcc_set_answers(self.h, d_a, n))

    def level_local(self):
        return self._counts(lib().mf_dcc_level_local)

    def pairs_fill(self, d_out):
        _check(lib().mf_dcc_pairs_fill(self.h, d_out))

    def pairs_complete(self, d_pairs, n):
        _check(lib().mf_dcc_pairs_complete(self.h, d_pairs, n))

    def merge(self, d_pairs, n):
        m = C.c_uint64()
        _check(lib().mf_dcc_merge(self.h, d_pairs, n, C.byref(m)))
        return m.value

    def stats_fill(self, d_out):
        _check(lib().mf_dcc_stats_fill(self.h, d_out))

    def classify(self, d_stats, n, seg_first, own_n, b1, b2, thr, rank):
        """seg_first: first record of every rank (world + 1 entries) -> (kept, oversize) components of the level over ALL ranks"""
        seg = np.ascontiguousarray(seg_first, dtype=np.uint64)
        a, b = C.c_uint64(), C.c_uint64()
        _check(lib().mf_dcc_classify(self.h, d_stats, n, seg.ctypes.data, int(seg[rank]), own_n, b1, b2, thr, C.byref(a), C.byref(b)))
        return a.value, b.value

    def kept_fill(self, d_out):
        _check(lib().mf_dcc_kept_fill(self.h, d_out))

    def members(self):
        n = C.c_uint64()
        _check(lib().mf_dcc_members(self.h, C.byref(n)))
        return n.value

    def members_fill(self, d_keys, d_roots):
        _check(lib().mf_dcc_members_fill(self.h, d_keys, d_roots))

    def members_grouped(self):
        """-> (members, runs): this rank's members sorted by component, 8 B each + one (root, count) record per component"""
        n, r = C.c_uint64(), C.c_uint64()
        _check(lib().mf_dcc_members_grouped(self.h, C.byref(n), C.byref(r)))
        return n.value, r.value

    def members_grouped_fill(self, d_keys, d_runs):
        _check(lib().mf_dcc_members_grouped_fill(self.h, d_keys, d_runs))

    def finish_grouped(self, d_keys, nm, d_runs, n_runs, roots, sizes, weights, thrs, minkeys):
        roots = np.ascontiguousarray(roots, dtype=np.uint32); sizes = np.ascontiguousarray(sizes, dtype=np.uint32)
        weights = np.ascontiguousarray(weights, dtype=np.int64); thrs = np.ascontiguousarray(thrs, dtype=np.int32)
        minkeys = np.ascontiguousarray(minkeys, dtype=np.uint64)
        c = C.c_void_p()
        _check(lib().mf_dcc_finish_grouped(self.h, d_keys, nm, d_runs, n_runs, roots.ctypes.data, sizes.ctypes.data, weights.ctypes.data,
                                           thrs.ctypes.data, minkeys.ctypes.data, len(roots), C.byref(c)))
        return Comps(self.ctx, c)

    def minkeys(self, roots, d_out):
        roots = np.ascontiguousarray(roots, dtype=np.uint32)
        _check(lib().mf_dcc_minkeys(self.h, roots.ctypes.data, len(roots), d_out))

    def finish(self, d_keys, d_roots, nm, roots, sizes, weights, thrs, minkeys):
        roots = np.ascontiguousarray(roots, dtype=np.uint32); sizes = np.ascontiguousarray(sizes, dtype=np.uint32)
        weights = np.ascontiguousarray(weights, dtype=np.int64); thrs = np.ascontiguousarray(thrs, dtype=np.int32)
        minkeys = np.ascontiguousarray(minkeys, dtype=np.uint64)
        c = C.c_void_p()
        _check(lib().mf_dcc_finish(self.h, d_keys, d_roots, nm, roots.ctypes.data, sizes.ctypes.data, weights.ctypes.data,
                                   thrs.ctypes.data, minkeys.ctypes.data, len(roots), C.byref(c)))
        return Comps(self.ctx, c)


def bray_curtis(vecs):
    """DistanceMatrixCalculatorMain.brayCurtisDistance (:140-152) for all sample pairs."""
    vecs = np.ascontiguousarray(vecs, dtype=np.int64)
    s, c = vecs.shape
    out = np.zeros((s, s), dtype=np.float64)
    _check(lib().mf_bray_curtis(vecs.ctypes.data, s, c, out.ctypes.data))
    return out


def synth_reads_host(seed, sample, first_read, n_reads, read_len, genome_scale_bp, sub_per_16384=82):
    bases = np.empty(n_reads * read_len, dtype=np.uint8)
    off = np.empty(n_reads + 1, dtype=np.uint64)
    _check(lib().mf_synth_reads_host_ex(seed, sample, first_read, n_reads, read_len, genome_scale_bp, sub_per_16384,
                                        bases.ctypes.data, off.ctypes.data))
    return bases, off

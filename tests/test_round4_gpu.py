"""Round 4's machinery under forced settings: whatever the plan -- bits after level 1 from the pilot, table partitions per counting
unit, units dealt or claimed, threshold levels dense or on a list, files loaded from disk or handed over in HBM -- the results are
the ones the oracle gives (IOUtils.loadReads src/io/IOUtils.java:772-803, ComponentsBuilder.java:58-270, IOUtils.loadKmers :369-401)."""
import os

import numpy as np
import pytest

from util import to_device

pytestmark = pytest.mark.gpu

SEED = 0x4D45544146415354
DEFAULTS = dict(skm_pilot=1, skm_unit_distinct=2200, skm_unit_records=0, part_good=220, unit_parts_long=3, skm_dynq=1, cc_sparse=1,
                skm_batches=0, l1_bits=-1, l2_bits=-1, file_cache=0)


@pytest.fixture()
def ctx(gpu_ctx):
    for k, v in DEFAULTS.items():
        gpu_ctx.set_option(k, v)
    yield gpu_ctx
    for k, v in DEFAULTS.items():
        gpu_ctx.set_option(k, v)


def _sample(n, scale, sample=5):
    from metafast_amd import lib as L
    return L.synth_reads_host(SEED, sample, 0, n, 150, scale)


@pytest.mark.parametrize("opts", [dict(skm_unit_distinct=150), dict(skm_unit_distinct=64, skm_unit_records=64), dict(part_good=16), dict(part_good=4096),
                                  dict(skm_dynq=0), dict(skm_unit_distinct=3400, skm_unit_records=1 << 20), dict(skm_batches=7, part_good=40)])
def test_count_plans_from_the_pilot(ctx, oracle, opts):
    """the pilot's decisions pushed to their ends -- units of 64 distinct k-mers (three radix levels on 300 000 reads), units as large
    as the table takes (many of them counted in several passes), 1 to 16 table partitions per unit (k_gather_split_n with every s),
    units dealt by a fixed stride -- on 300 000 reads at 6-fold depth: the oracle's table, the cut inside the kernels = filtering"""
    b, o = _sample(300_000, 60_000)
    tb, to = to_device(b, o)
    want_k, want_c = oracle.Table().count_buffer(b, o, 31).export()
    for name, v in opts.items():
        ctx.set_option(name, v)
    t = ctx.count_device(tb.data_ptr(), to.data_ptr(), len(o) - 1, len(b), 31, 0)
    gk, gc = t.export()
    assert np.array_equal(gk, want_k) and np.array_equal(gc.astype(np.int32), want_c)
    t2, n_all = ctx.count_device_above(tb.data_ptr(), to.data_ptr(), len(o) - 1, len(b), 31, 1)
    k2, c2 = t2.export()
    keep = want_c > 1
    assert n_all == len(want_k) and np.array_equal(k2, want_k[keep]) and np.array_equal(c2.astype(np.int32), want_c[keep])
    # the partition structure the graph kernels rely on: every k-mer is found through the partitioned index
    probe = np.concatenate([want_k[keep][::97], want_k[~keep][::211]])
    got = t2.lookup(probe)
    exp = np.concatenate([want_c[keep][::97], np.full(len(want_k[~keep][::211]), -1)])
    assert np.array_equal(got.astype(np.int64), exp.astype(np.int64))


@pytest.mark.parametrize("parts", [0, 1, 2, 4])
def test_assembled_input_units_of_several_table_partitions(ctx, oracle, parts):
    """the cutter's input (sequences counted with a length filter, ComponentCutterMain.java:81) is counted 2^unit_parts_long table
    partitions to a unit: every setting gives the oracle's table and the same components"""
    from util import branchy_reads
    ctx.set_option("unit_parts_long", parts)
    inputs = [branchy_reads(rs, genome_seed=7, n=6000) for rs in (107, 117, 127)]
    o_cutter = oracle.Table()
    seqs_b, seqs_o, nb = [], [np.zeros(1, dtype=np.uint64)], 0
    for bases, offsets in inputs:
        t = oracle.Table().count_buffer(bases, offsets, 31)
        keys, vals = t.export(1)
        g = oracle.Table()
        for kk, vv in zip(keys.tolist(), vals.tolist()):
            g.add(kk, vv)
        sq = oracle.build_unitigs(g, 31, 1, 100)
        o_cutter.count_seqs(sq, 31, 100)
        for text, _a, _mn, _mx in sq.all():
            q = np.frombuffer(text.encode(), dtype=np.uint8)
            seqs_b.append(q); nb += len(q); seqs_o.append(np.array([nb], dtype=np.uint64))
    b = np.concatenate(seqs_b); o = np.concatenate(seqs_o)
    tb, to = to_device(b, o)
    cut = ctx.count_device(tb.data_ptr(), to.data_ptr(), len(o) - 1, len(b), 31, 100)
    gk, gc = cut.export()
    ok, ov = o_cutter.export()
    assert np.array_equal(gk, ok) and np.array_equal(gc.astype(np.int32), ov)
    want = oracle.cut_components(o_cutter, 31, 100, 1000).all()
    comps = ctx.cut_components(cut, 100, 1000).export()
    assert [(a, w, t) for a, w, t, _ in comps] == [(a, w, t) for a, w, t, _ in want]
    assert all(np.array_equal(np.sort(g[3]), np.sort(x[3])) for g, x in zip(comps, want))


@pytest.mark.parametrize("sparse", [0, 1])
def test_threshold_levels_dense_and_on_a_list(ctx, oracle, sparse):
    """ComponentsBuilder.java:86-150 through six threshold levels: with every level visiting all vertices (cc_sparse = 0) and with the
    later levels running on the list of the survivors -- the oracle's components, bit for bit"""
    from util import branchy_reads
    ctx.set_option("cc_sparse", sparse)
    o_cutter = oracle.Table()
    for rs in (107, 117, 127, 137):
        bases, offsets = branchy_reads(rs, genome_seed=7, n=6000)
        t = oracle.Table().count_buffer(bases, offsets, 31)
        keys, vals = t.export(1)
        g = oracle.Table()
        for kk, vv in zip(keys.tolist(), vals.tolist()):
            g.add(kk, vv)
        o_cutter.count_seqs(oracle.build_unitigs(g, 31, 1, 100), 31, 100)
    want = oracle.cut_components(o_cutter, 31, 100, 1000).all()
    assert max(t for _, _, t, _ in want) >= 5
    ck, cv = o_cutter.export()
    cut = ctx.table_from_host(ck, cv.astype(np.uint16), 31)
    comps = ctx.cut_components(cut, 100, 1000).export()
    assert [(a, w, t) for a, w, t, _ in comps] == [(a, w, t) for a, w, t, _ in want]
    assert all(np.array_equal(np.sort(g[3]), np.sort(x[3])) for g, x in zip(comps, want))


def test_file_cache_hands_out_what_was_written_and_notices_a_rewrite(ctx, oracle, tmp_path):
    """option file_cache: mf_table_load_kmers of a file this context has just written returns the table it was written from (another
    handle on the same object, or a filtered copy for a higher threshold) -- the same entries as reading the file; a file that has been
    rewritten since is read; with the cache off the file is read.  IOUtils.loadKmers semantics (src/io/IOUtils.java:369-401) either way."""
    b, o = _sample(200_000, 40_000)
    tb, to = to_device(b, o)
    ot = oracle.Table().count_buffer(b, o, 31)
    path = str(tmp_path / "s.kmers.bin")
    results = {}
    for cache in (0, 4):
        ctx.set_option("file_cache", cache)
        t, _ = ctx.count_device_above(tb.data_ptr(), to.data_ptr(), len(o) - 1, len(b), 31, 1)
        n_good = t.write_kmers(1, path)
        t.close()                                            # (the cache keeps its own handle)
        for thr in (-1, 1, 3):
            l = ctx.load_kmers([path], thr, 31)
            k, c = l.export()
            wk, wc = ot.export(max(thr, 1))
            assert np.array_equal(k, wk) and np.array_equal(c.astype(np.int32), wc), (cache, thr)
            # usable like any table: the unitigs of the loaded table = the oracle's
            if thr == 1:
                sq = ctx.build_unitigs(l, 1, 100)
                results[cache] = sorted(x[0] for x in sq.export())
                sq.close()
            l.close()
        assert n_good == len(ot.export(1)[0])
    assert results[0] == results[4] and len(results[0]) > 0
    # somebody else rewrites the file (the oracle's writer, fewer records): the cached table no longer stands for it
    ctx.set_option("file_cache", 4)
    t, _ = ctx.count_device_above(tb.data_ptr(), to.data_ptr(), len(o) - 1, len(b), 31, 1)
    t.write_kmers(1, path)
    t.close()
    ot.write_kmers(4, path, None)
    l = ctx.load_kmers([path], -1, 31)
    k, c = l.export()
    wk, wc = ot.export(4)
    assert np.array_equal(k, wk) and np.array_equal(c.astype(np.int32), wc)
    l.close()
    ctx.set_option("file_cache", 0)

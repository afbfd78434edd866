"""BASELINE.json config 5's defining shape on one MI355X: ONE sample with more than 2^32 distinct k-mers (380 M synthetic
150 bp reads, k = 31: 4.56e10 occurrences, ~4.3e9 distinct).  The reference grows its map by doubling
(itmo!/structures/map/Long2ShortHashMap.java:191-214) and hands on the entries above the cut (src/io/IOUtils.java:52-60);
here the run is cut into digit-range slices (mf_skm.hip) and the cut is made inside the counting kernels
(mf_count_device_above -- what mf_count_reads_above / `metafast.sh -t kmer-counter` use), so the uncut table never exists.

Checked without a CPU pass: n_distinct > 2^32; N_occ; the all-counts histogram sums to n_distinct and (no saturation) to
N_occ; the kept table = histogram rows above the cut; an INDEPENDENT exact recount in plain torch ops of sampled keys over
ALL reads against lookups in the cut table (kept iff count > threshold).  MF_CONFIG5_READS overrides the size."""
import os

import numpy as np
import pytest

from test_fullsize_gpu import K, RL, SEED, _canon_chunk, _codes_lut, _recount

pytestmark = pytest.mark.gpu

N_READS = int(os.environ.get("MF_CONFIG5_READS", "380000000"))
THR = 1


def test_one_sample_with_more_than_2p32_distinct_kmers(gpu_ctx):
    import torch
    free, total = torch.cuda.mem_get_info()
    need = N_READS * 560                        # reads + sliced record buffers + kept table + the recount's temporaries
    if free < need:
        pytest.skip("needs %.0f GB of free HBM" % (need / 1e9))
    dev = "cuda"
    bases = torch.zeros(N_READS * RL + 64, dtype=torch.uint8, device=dev)
    offsets = torch.zeros(N_READS + 1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    gpu_ctx.synth_reads_device(SEED, 0, 0, N_READS, RL, 1_000_000, bases.data_ptr(), offsets.data_ptr())
    gpu_ctx.synchronize()
    t, n_all = gpu_ctx.count_device_above(bases.data_ptr(), offsets.data_ptr(), N_READS, N_READS * RL, K, THR)
    try:
        n_occ = N_READS * (RL - K + 1)
        if N_READS >= 380_000_000:
            assert n_all > 2 ** 32, n_all
        assert t.occurrences() == n_occ
        h = t.hist().astype(np.int64)
        assert h[0] == 0 and int(h.sum()) == n_all
        weighted = int((h * np.arange(len(h), dtype=np.int64)).sum())
        assert weighted <= n_occ and (h[-1] > 0 or weighted == n_occ)
        assert len(t) == int(h[THR + 1:].sum()) < 2 ** 32
        # exact recount of sampled keys over the whole input
        lut = _codes_lut(torch, dev)
        b2d = bases[: N_READS * RL].view(N_READS, RL)
        g = torch.Generator(device="cpu").manual_seed(11)
        mid = N_READS // 2
        pres = torch.cat([_canon_chunk(torch, lut, b2d[lo:lo + 2048]).reshape(-1)[torch.randperm(2048 * 120, generator=g)[:3000].to(dev)]
                          for lo in (0, mid, N_READS - 2048)])
        absent = torch.randint(0, 1 << 62, (1000,), generator=g, dtype=torch.int64).to(dev)
        sample = torch.unique(torch.cat([pres, absent]))
        exact = _recount(torch, lut, b2d, sample).cpu().numpy()
        assert (exact > THR).sum() > 3000 and (exact == 1).sum() > 100 and (exact == 0).sum() >= 990
        want = np.where(exact > THR, np.minimum(exact, 32767), -1)
        got = t.lookup(sample.cpu().numpy().astype(np.uint64)).astype(np.int64)
        assert np.array_equal(got, want)
    finally:
        t.close()
        del bases, offsets
        gpu_ctx.trim()
        torch.cuda.empty_cache()

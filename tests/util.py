"""Helpers shared by the tests (inputs in the (bases, offsets) layout the C-ABI takes)."""
import numpy as np


def pack_reads(reads):
    """list of str/bytes -> (bases uint8[], offsets uint64[n+1])"""
    bs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
    off = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        off[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    bases = np.frombuffer(b"".join(bs), dtype=np.uint8).copy()
    return bases, off


def random_reads(rng, n, min_len, max_len, alphabet=b"ACGT"):
    al = np.frombuffer(alphabet, dtype=np.uint8)
    lens = rng.integers(min_len, max_len + 1, size=n)
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens, dtype=np.uint64)
    bases = al[rng.integers(0, len(al), size=int(off[-1]))]
    return bases.astype(np.uint8), off


def genome_reads(rng, genome_len, n_reads, read_len, err=0.0, flip=True):
    """reads sampled from one random genome (gives repeated k-mers, i.e. counts > 1)"""
    al = np.frombuffer(b"ACGT", dtype=np.uint8)
    comp = np.zeros(256, dtype=np.uint8)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    g = al[rng.integers(0, 4, size=genome_len)]
    starts = rng.integers(0, genome_len - read_len + 1, size=n_reads)
    out = np.empty((n_reads, read_len), dtype=np.uint8)
    for i, s in enumerate(starts):
        r = g[s:s + read_len]
        if flip and rng.integers(0, 2):
            r = comp[r[::-1]]
        out[i] = r
    if err > 0:
        m = rng.random(out.shape) < err
        out[m] = al[rng.integers(0, 4, size=int(m.sum()))]
    off = (np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len))
    return out.reshape(-1).copy(), off


def to_device(bases, offsets):
    """-> (torch uint8 tensor, torch int64 tensor) on cuda:0; keeps 16-byte slack for the kernels' 16-byte loads"""
    import torch
    tb = torch.zeros(len(bases) + 64, dtype=torch.uint8, device="cuda")
    if len(bases):
        tb[: len(bases)] = torch.from_numpy(np.ascontiguousarray(bases))
    to = torch.from_numpy(np.ascontiguousarray(offsets).view(np.int64)).to("cuda")
    return tb, to


def gpu_count(ctx, bases, offsets, k, min_len=0):
    tb, to = to_device(bases, offsets)
    t = ctx.count_device(tb.data_ptr(), to.data_ptr(), len(offsets) - 1, int(offsets[-1]), k, min_len)
    return t


def canon_seq(s):
    """strand-normalised form of a sequence (min of itself and its reverse complement)"""
    rc = s[::-1].translate(str.maketrans("ACGT", "TGCA"))
    return min(s, rc)


def branchy_reads(seed, genome_seed=None, n=12000):
    """genome with two extra copies of a 400-bp repeat + substitution errors: gives branches, tips and bubbles,
    i.e. paths that the reference's emission rule prints 0, 1 or 2 times.  genome_seed: several
    samples (different `seed`s) of ONE genome, for cutter tables with values > 2."""
    grng = np.random.default_rng(seed if genome_seed is None else genome_seed)
    al = np.frombuffer(b"ACGT", dtype=np.uint8)
    g = al[grng.integers(0, 4, size=60000)]
    rep = g[1000:1400].copy()
    g = np.concatenate([g[:30000], rep, g[30000:45000], rep, g[45000:]])
    rng = grng if genome_seed is None else np.random.default_rng(seed)
    comp = np.zeros(256, dtype=np.uint8)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    rl = 150
    starts = rng.integers(0, len(g) - rl + 1, size=n)
    out = np.empty((n, rl), dtype=np.uint8)
    for i, s in enumerate(starts):
        r = g[s:s + rl]
        out[i] = comp[r[::-1]] if rng.integers(0, 2) else r
    m = rng.random(out.shape) < 0.004
    out[m] = al[rng.integers(0, 4, size=int(m.sum()))]
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(rl)
    return out.reshape(-1).copy(), off


def emission_census(seqs):
    """(printed once, printed twice) over the strand-normalised unitigs"""
    from collections import Counter
    c = Counter(canon_seq(s[0]) for s in seqs)
    assert max(c.values(), default=1) <= 2
    return sum(1 for v in c.values() if v == 1), sum(1 for v in c.values() if v == 2)

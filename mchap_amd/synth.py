"""Synthetic loci for benchmarks and parity tests.

Restates what the reference's test-only simulator does (mchap/testing.py:9-73: sample read
haplotypes, assign qualities, encode with as_probabilistic, resample errors) with a per-unit
deterministic generator so that every rank / shard can build its own units.
"""
import numpy as np

PFEIFFER_ERROR = 0.0024  # reference mchap/constant.py:3


def as_probabilistic(calls, n_alleles, p, error_factor=3.0):
    """int8 calls [R, M] (+ per-call probability p) -> float64 tensor [R, M, A]
    (reference encoding/integer/transcode.py:16-77): called allele p, others (1-p)/3, alleles >= n_alleles[j]
    zero, gaps (call < 0) NaN *before* the zero mask."""
    calls = np.asarray(calls)
    n_alleles = np.asarray(n_alleles)
    p = np.broadcast_to(np.asarray(p, dtype=np.float64), calls.shape)
    A = int(np.max(n_alleles))
    alleles = np.arange(A)
    onehot = calls[..., None] == alleles
    new = ((1.0 - p) / error_factor)[..., None] * ~onehot
    new = np.where(onehot, p[..., None], new)
    new[calls < 0] = np.nan
    new[..., n_alleles[..., None] <= alleles] = 0.0
    return new


def synth_units(n_units, ploidy=4, n_pos=8, n_reads=200, n_alleles=2, first_unit=0, seed=20260101,
                window=(4, 8), qual=(20, 40), dedup=False):
    """Config #2 shape by default: tetraploid, 8 biallelic SNVs, 200 distinct read rows (phred path).

    Returns (reads [U, R, M, A] float64, calls [U, R, M] int8, truth [U, K, M] int8).  With dedup=True the
    phred-ignored encoding (p = 1 - error) is used instead and rows are NOT merged here (see dedup_unit)."""
    A = n_alleles
    reads = np.empty((n_units, n_reads, n_pos, A), dtype=np.float64)
    calls_all = np.empty((n_units, n_reads, n_pos), dtype=np.int8)
    truth = np.empty((n_units, ploidy, n_pos), dtype=np.int8)
    na = np.full(n_pos, A)
    for u in range(n_units):
        rng = np.random.default_rng([seed, first_unit + u])
        while True:
            haps = rng.integers(0, A, size=(ploidy, n_pos)).astype(np.int8)
            if len(np.unique(haps, axis=0)) >= min(3, ploidy, A ** n_pos):
                break
        src = haps[rng.integers(0, ploidy, size=n_reads)]
        if dedup:
            p = np.full(src.shape, 1.0 - PFEIFFER_ERROR)
        else:
            q = rng.integers(qual[0], qual[1] + 1, size=src.shape)
            p = (1.0 - PFEIFFER_ERROR) * (1.0 - 10.0 ** (-q / 10.0))
        flip = rng.random(src.shape) >= p
        calls = np.where(flip, (src + rng.integers(1, A, size=src.shape)) % A if A > 1 else src, src).astype(np.int8)
        lo, hi = window
        wlen = rng.integers(min(lo, n_pos), min(hi, n_pos) + 1, size=n_reads)
        start = (rng.random(n_reads) * (n_pos - wlen + 1)).astype(int)
        pos = np.arange(n_pos)[None, :]
        gap = (pos < start[:, None]) | (pos >= (start + wlen)[:, None])
        calls[gap] = -1
        reads[u] = as_probabilistic(calls, na, p)
        calls_all[u] = calls
        truth[u] = haps
    return reads, calls_all, truth


def dedup_unit(reads):
    """De-duplicate identical read rows (reference application/baseclass.py:207): rows in order of first
    appearance plus their counts."""
    reads = np.asarray(reads)
    if len(reads) == 0 or reads[0].size == 0:
        # no reads, or a locus without positions: every row is the same (empty) row
        n = len(reads)
        return reads[: min(n, 1)], np.full(min(n, 1), n, dtype=np.int64)
    flat = np.ascontiguousarray(reads).reshape(len(reads), -1)
    keys = flat.view("V%d" % (flat.shape[1] * 8)).reshape(-1)
    _, first, inv = np.unique(keys, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")
    rank = np.empty(len(order), dtype=np.int64)
    rank[order] = np.arange(len(order))
    counts = np.bincount(rank[inv.reshape(-1)], minlength=len(order)).astype(np.int64)
    return reads[first[order]], counts

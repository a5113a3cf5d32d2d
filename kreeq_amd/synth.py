"""Deterministic synthetic inputs of the shapes BASELINE.md §3 names (numpy PCG64, fixed seeds).

genome : iid uniform ACGT
reads  : uniform start, strand 50/50, substitution errors at `err` per base
batch  : reads joined by '\n' (any non-ACGT byte separates reads)
"""
import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def genome_codes(length, seed=1):
    return np.random.default_rng(seed).integers(0, 4, length, dtype=np.uint8)


def codes_to_ascii(codes):
    return _ACGT[codes]


def reads_batch(genome, n_reads, read_len, seed=2, err=0.005, chunk=100_000):
    """-> uint8 array of n_reads*(read_len+1)-1 bytes"""
    rng = np.random.default_rng(seed)
    out = np.empty(n_reads * (read_len + 1), dtype=np.uint8)
    view = out.reshape(n_reads, read_len + 1)
    ar = np.arange(read_len, dtype=np.int64)
    for lo in range(0, n_reads, chunk):
        n = min(chunk, n_reads - lo)
        starts = rng.integers(0, len(genome) - read_len + 1, n)
        codes = genome[starts[:, None] + ar[None, :]]
        rev = rng.integers(0, 2, n).astype(bool)
        codes[rev] = 3 - codes[rev][:, ::-1]
        if err > 0:
            e = rng.random(codes.shape) < err
            codes = np.where(e, (codes + rng.integers(1, 4, codes.shape, dtype=np.uint8)) & 3, codes).astype(np.uint8)
        view[lo:lo + n, :read_len] = _ACGT[codes]
        view[lo:lo + n, read_len] = ord("\n")
    return out[:-1]


def mutate(codes, rate, seed=3):
    rng = np.random.default_rng(seed)
    e = rng.random(len(codes)) < rate
    return np.where(e, (codes + rng.integers(1, 4, len(codes), dtype=np.uint8)) & 3, codes).astype(np.uint8)

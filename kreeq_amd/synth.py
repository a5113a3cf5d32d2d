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


# ---- the same shapes generated ON THE GPU (torch RNG, fixed seeds): BASELINE.md configs[2]-[4] never leave HBM ----
def genome_dev(length, dev, seed=1):
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    return torch.randint(0, 4, (length,), dtype=torch.uint8, device=dev, generator=g)


def mutate_dev(codes, rate, seed=3):
    """-> (mutated codes, number of substitutions)"""
    import torch
    g = torch.Generator(device=codes.device)
    g.manual_seed(seed)
    n = codes.numel()
    out = codes.clone()
    n_sub = 0
    for lo in range(0, n, 1 << 28):                     # bounded temporaries
        hi = min(n, lo + (1 << 28))
        mut = torch.rand(hi - lo, device=codes.device, generator=g) < rate
        add = torch.randint(1, 4, (hi - lo,), dtype=torch.uint8, device=codes.device, generator=g)
        out[lo:hi] = torch.where(mut, (codes[lo:hi] + add) & 3, codes[lo:hi])
        n_sub += int(mut.sum())
    return out, n_sub


def ascii_dev(codes):
    import torch
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=codes.device)
    out = torch.empty_like(codes)
    for lo in range(0, codes.numel(), 1 << 28):
        hi = min(codes.numel(), lo + (1 << 28))
        out[lo:hi] = acgt[codes[lo:hi].long()]
    return out


def reads_dev(genome, n_reads, read_len, gen, err=0.005, chunk=1_000_000):
    """one read batch on the device: uint8 tensor of n_reads*(read_len+1)-1 bytes, reads separated by '\\n';
    uniform start, strand 50/50, substitutions at `err` per base.  `gen` = torch.Generator on the genome's device
    (its state advances, so consecutive calls give different reads)."""
    import torch
    dev = genome.device
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    ar = torch.arange(read_len, device=dev)
    out = torch.full((n_reads, read_len + 1), 10, dtype=torch.uint8, device=dev)
    G = genome.numel()
    for lo in range(0, n_reads, chunk):
        n = min(chunk, n_reads - lo)
        starts = torch.randint(0, G - read_len + 1, (n,), device=dev, generator=gen)
        codes = genome[starts[:, None] + ar[None, :]]
        rev = torch.rand(n, device=dev, generator=gen) < 0.5
        codes = torch.where(rev[:, None], 3 - codes.flip(1), codes)
        e = torch.rand((n, read_len), device=dev, generator=gen) < err
        codes = torch.where(e, (codes + torch.randint(1, 4, (n, read_len), dtype=torch.uint8, device=dev, generator=gen)) & 3, codes)
        out[lo:lo + n, :read_len] = acgt[codes.long()]
    return out.reshape(-1)[:-1]


def pack_dev(ascii_t, chunk_units=1 << 24):
    """2-bit packed form of a read batch, made on the device: (codes int32[ceil(n/16)], inv int16[ceil(n/16)]) in the layout of
    kq_pack_bases (include/kreeq_amd.h): base i of a unit at bits 2i (A C G T = 0 1 2 3, case-blind), invalid-base bit i for
    anything else and for the positions behind the end.  3 bits per base instead of 8: what lets a whole 30x read set
    of BASELINE configs[2] stay resident next to its table."""
    import torch
    dev = ascii_t.device
    n = ascii_t.numel()
    units = (n + 15) // 16
    codes = torch.empty(units, dtype=torch.int32, device=dev)
    inv = torch.empty(units, dtype=torch.int16, device=dev)
    sh2 = (2 * torch.arange(16, device=dev, dtype=torch.int32))[None, :]
    sh1 = torch.arange(16, device=dev, dtype=torch.int32)[None, :]
    for lo in range(0, units, chunk_units):
        hi = min(units, lo + chunk_units)
        x = ascii_t[lo * 16:min(n, hi * 16)]
        if x.numel() < (hi - lo) * 16:
            x = torch.cat([x, torch.full(((hi - lo) * 16 - x.numel(),), 10, dtype=torch.uint8, device=dev)])
        x = x.view(-1, 16).to(torch.int32)
        u = x & 0xDF
        ok = (u == 65) | (u == 67) | (u == 71) | (u == 84)
        c = torch.where(ok, ((x >> 1) ^ (x >> 2)) & 3, torch.zeros_like(x))
        codes[lo:hi] = (c << sh2).sum(1, dtype=torch.int32)          # disjoint bit fields: the sum is their OR (wraps into the sign bit)
        inv[lo:hi] = ((~ok).to(torch.int32) << sh1).sum(1, dtype=torch.int32).to(torch.int16)
    return codes, inv

"""Randomised differential test: many small, nasty inputs (ragged reads, every byte value, k from 2
to 32, both count paths, tiny capacity hints, several batches, map-range lookups) against the oracle."""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu

ALPHABETS = [b"ACGT", b"ACGTacgt", b"ACGTN", b"ACGTacgtNn-*", bytes(range(256)), b"AAAAAAAC", b"ACACACAG", b"A"]


def random_batch(rng, n_reads, max_len, alphabet, sep):
    alpha = np.frombuffer(alphabet, dtype=np.uint8)
    reads = []
    for _ in range(n_reads):
        ln = int(rng.integers(0, max_len + 1))
        r = alpha[rng.integers(0, len(alpha), ln)]
        if rng.random() < 0.3 and ln > 40:                     # plant a repeat so that edges and duplicates exist
            a = int(rng.integers(0, ln - 40))
            r[a + 20:a + 40] = r[a:a + 20]
        reads.append(r.tobytes())
    return sep.join(reads)


def _copy_oracle(O, db, k):
    c = O.OracleDB(k, 128)
    c.import_entries(db.export())
    return c


@pytest.mark.parametrize("seed", range(150))
def test_random_differential(seed):
    import kreeq_amd as kq
    from oracle import oracle as O

    rng = np.random.default_rng(1000 + seed)
    k = int(rng.choice([2, 3, 5, 11, 16, 21, 27, 28, 29, 31, 32]))
    path = ["auto", "direct", "partitioned"][seed % 3]
    alphabet = ALPHABETS[int(rng.integers(0, len(ALPHABETS)))]
    sep = [b"\n", b"N", b"\x00", b"\n\n"][int(rng.integers(0, 4))]
    hint = int(rng.choice([0, 1, 5000, 3_000_000]))
    gpu, cpu = kq.KreeqDB(k, 128, capacity_hint=hint), O.OracleDB(k, 128)
    gpu.set_option("count_path", path)
    if rng.random() < 0.3:
        gpu.set_option("slice_kmers", int(rng.integers(1000, 50000)))
    batches = [random_batch(rng, int(rng.integers(1, 400)), int(rng.choice([10, 60, 300])), alphabet, sep) for _ in range(int(rng.integers(1, 4)))]
    for b in batches:
        gpu.count_batch(b)
        cpu.count_batch(b)
        keys, edges = gpu.emit_records(b)
        ok, oe = O.emit_records(k, b)
        assert np.array_equal(keys, ok) and np.array_equal(edges, oe)
    assert gpu.summary(with_hist=True) == cpu.summary(with_hist=True)
    assert H.entries_equal(gpu.export(), cpu.export())
    asm = random_batch(rng, 5, 500, alphabet, b"N")
    cutoff = int(rng.choice([0, 0, 2, 5]))
    lo = int(rng.integers(0, 100))
    hi = int(rng.integers(lo + 1, 129))
    for a, b_ in ((0, 128), (lo, hi)):
        cg, pg = gpu.lookup_sequence(asm, cov_cutoff=cutoff, map_lo=a, map_hi=b_, per_base=True)
        cc, pc = cpu.validate_sequence(asm, cov_cutoff=cutoff, map_lo=a, map_hi=b_, per_base=True)
        assert np.array_equal(cg, cc)
        for f in ("fw", "bw", "cov", "isFw"):
            assert np.array_equal(pg[f], pc[f]), f
        # the same counters through the region-wise lookup (P1 -> level -> k_lookup_regions)
        gpu.set_option("lookup_path", "partitioned")
        cr, _ = gpu.lookup_sequence(asm, cov_cutoff=cutoff, map_lo=a, map_hi=b_)
        gpu.set_option("lookup_path", "auto")
        assert np.array_equal(cr, cc)
    # union by regions into a handle of another geometry == the table itself
    other = kq.KreeqDB(k, 128, capacity_hint=int(rng.choice([0, 2_900_000])))
    other.set_option("merge_path", "partitioned")
    other.merge(gpu)
    assert H.entries_equal(other.export(), cpu.export())
    other.merge(gpu)                                                # every key present: counters double (saturating)
    cpu.merge(cpu2 := _copy_oracle(O, cpu, k))
    assert other.summary(with_hist=True) == cpu.summary(with_hist=True)
    assert H.entries_equal(other.export(), cpu.export())

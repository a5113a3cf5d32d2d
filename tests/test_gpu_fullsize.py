"""BASELINE configs[1] at FULL size (1 M x 150 bp, k=21) through size-independent properties, plus an
oracle comparison on a 10 % slice.  GPU only; ~20 s."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K, N_READS, READ_LEN, GENOME = 21, 1_000_000, 150, 5_000_000


@pytest.fixture(scope="module")
def workload():
    import torch

    import kreeq_amd
    from kreeq_amd import synth

    if not kreeq_amd.device_available():
        pytest.fail("no gfx950 device")
    genome = synth.genome_codes(GENOME, seed=1)
    reads = synth.reads_batch(genome, N_READS, READ_LEN, seed=2, err=0.005)
    return genome, reads, torch.from_numpy(reads).cuda()


@pytest.mark.parametrize("path", ["partitioned", "direct"])
def test_count_properties_full_size(workload, path):
    import torch

    import kreeq_amd
    from kreeq_amd import synth

    genome, reads, d_reads = workload
    n_kmers = N_READS * (READ_LEN - K + 1)
    db = kreeq_amd.KreeqDB(K, 128, capacity_hint=24_000_000)
    db.set_option("trust_capacity", 1)
    db.set_option("count_path", path)
    db.count_batch_dev(d_reads.data_ptr(), d_reads.numel())
    db.sync()
    s1 = db.summary(with_hist=True)
    assert s1["total"] == n_kmers == db.info()["kmers_counted"]          # every window of every read is one instance
    assert s1["distinct"] == db.info()["slots_used"] == sum(s1["hist"].values())
    assert sum(c * n for c, n in s1["hist"].items()) == s1["total"]       # checksum of the histogram
    assert s1["unique"] == s1["hist"][1]
    assert s1["missing"] == 4 ** K - s1["distinct"]
    # linearity: counting the same batch again doubles every coverage and creates no k-mer
    db.count_batch_dev(d_reads.data_ptr(), d_reads.numel())
    db.sync()
    s2 = db.summary(with_hist=True)
    assert s2["total"] == 2 * n_kmers and s2["distinct"] == s1["distinct"] and s2["unique"] == 0
    assert s2["hist"] == {2 * c: n for c, n in s1["hist"].items()}
    assert s2["edges"] == s1["edges"]
    # every read k-mer is in the table, with at least the edges its own read gave it
    ctr = torch.zeros(3, dtype=torch.int64, device="cuda")
    db.lookup_sequence_dev(d_reads.data_ptr(), d_reads.numel(), ctr.data_ptr())
    db.sync()
    assert ctr.cpu().tolist() == [0, n_kmers, 0]
    # the error-free genome: its k-mers are all covered at 30x minus edge effects -> almost nothing missing
    g = torch.from_numpy(synth.codes_to_ascii(genome)).cuda()
    ctr.zero_()
    db.lookup_sequence_dev(g.data_ptr(), g.numel(), ctr.data_ptr())
    db.sync()
    missing, total, edge_missing = ctr.cpu().tolist()
    assert total == GENOME - K + 1 and missing < 2000 and edge_missing == 0
    # map-range passes partition the lookups (src/kreeq.cpp:59-76)
    acc = torch.zeros(3, dtype=torch.int64, device="cuda")
    for lo, hi in ((0, 16), (16, 100), (100, 128)):
        db.lookup_sequence_dev(g.data_ptr(), g.numel(), acc.data_ptr(), map_lo=lo, map_hi=hi)
    db.sync()
    assert acc.cpu().tolist() == [missing, total, edge_missing]


def test_oracle_on_a_tenth(workload):
    import kreeq_amd
    from oracle import oracle as O
    from tests import helpers as H

    _, reads, _ = workload
    n = 100_000
    part = reads[:n * (READ_LEN + 1) - 1].tobytes()
    gpu, cpu = kreeq_amd.KreeqDB(K, 128, capacity_hint=6_000_000), O.OracleDB(K, 128)
    gpu.count_batch(part)
    cpu.count_batch(part, threads=16)
    assert gpu.summary(with_hist=True) == cpu.summary(with_hist=True)
    assert H.entries_equal(gpu.export(), cpu.export())


def test_union_is_linear_full_size(workload):
    """count(A) U count(B) == count(A + B) at full size (two halves of the batch)"""
    import kreeq_amd

    _, reads, d_reads = workload
    half = (N_READS // 2) * (READ_LEN + 1)
    a = kreeq_amd.KreeqDB(K, 128, capacity_hint=16_000_000)
    b = kreeq_amd.KreeqDB(K, 128, capacity_hint=16_000_000)
    whole = kreeq_amd.KreeqDB(K, 128, capacity_hint=24_000_000)
    a.count_batch_dev(d_reads.data_ptr(), half - 1)
    b.count_batch_dev(d_reads.data_ptr() + half, d_reads.numel() - half)
    whole.count_batch_dev(d_reads.data_ptr(), d_reads.numel())
    a.merge(b)
    assert a.summary(with_hist=True) == whole.summary(with_hist=True)


def test_hifi_length_reads_k31():
    """BASELINE configs[4] shape at small scale: 15 kbp reads, k=31, 0.1 % substitutions (WIDE records)"""
    import kreeq_amd
    from kreeq_amd import synth
    from oracle import oracle as O
    from tests import helpers as H

    genome = synth.genome_codes(3_000_000, seed=4)
    reads = synth.reads_batch(genome, 2000, 15000, seed=5, err=0.001, chunk=500).tobytes()
    gpu, cpu = kreeq_amd.KreeqDB(31, 128), O.OracleDB(31, 128)
    gpu.count_batch(reads)
    cpu.count_batch(reads, threads=16)
    assert gpu.summary(with_hist=True) == cpu.summary(with_hist=True)
    assert H.entries_equal(gpu.export(), cpu.export())
    asm = synth.codes_to_ascii(synth.mutate(genome, 1e-4, seed=6)).tobytes()
    cg, _ = gpu.lookup_sequence(asm)
    cc, _ = cpu.validate_sequence(asm, threads=16)
    assert np.array_equal(cg, cc)

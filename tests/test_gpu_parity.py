"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle and against the
reference's committed goldens.  Bit-exact (integer counts, printed QV text)."""
import os

import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kq():
    import kreeq_amd

    if not kreeq_amd.device_available():
        pytest.fail("no gfx950 device: the product path has no CPU fallback")
    return kreeq_amd


@pytest.fixture(scope="module")
def O():
    from oracle import oracle

    return oracle


VALIDATE_TESTS = list(range(0, 35))
DB_READS = {"test1": "random1.fastq", "test2": "random2.fastq", "random5": "random5.fastq", "random6": "random6.fastq",
            "random7": "random7.fastq", "random8": "random8.fastq", "random9": "random9.fastq", "random10": "random10.fastq",
            "random11": "random11.fastq", "random12": "random12.fastq"}


# ---------------------------------------------------------------------------------- K1 emit
@pytest.mark.parametrize("name", ["random1.fastq", "random2.fastq", "random3.N.fastq", "repeat1.fastq", "to_correct.fastq"])
def test_emit_records_fixtures(kq, O, name):
    batch = H.reads_batch([H.golden_input(name)])
    db = kq.KreeqDB(21, 128)
    keys, edges = db.emit_records(batch)
    ok, oe = O.emit_records(21, batch)
    assert np.array_equal(keys, ok) and np.array_equal(edges, oe)


@pytest.mark.parametrize("k", [2, 5, 16, 21, 31, 32])
def test_emit_records_random(kq, O, k):
    batch, _ = H.synth_reads(700, 97, 5000, seed=100 + k, err=0.02, n_rate=0.01)
    db = kq.KreeqDB(k, 128)
    for cut in (len(batch), 4032, 4031, 4033, 8064 + 5, 17, k, k - 1, 0):
        b = batch[:cut]
        keys, edges = db.emit_records(b)
        ok, oe = O.emit_records(k, b)
        assert np.array_equal(keys, ok), (k, cut)
        assert np.array_equal(edges, oe), (k, cut)


def test_emit_odd_bytes(kq, O):
    """every byte value: only ACGTacgt are bases"""
    rng = np.random.default_rng(5)
    raw = rng.integers(0, 256, 20000, dtype=np.uint8)
    acgt = np.frombuffer(b"ACGTacgt", dtype=np.uint8)
    mask = rng.random(20000) < 0.93
    raw = np.where(mask, acgt[rng.integers(0, 8, 20000)], raw).astype(np.uint8)
    b = raw.tobytes()
    db = kq.KreeqDB(11, 128)
    keys, edges = db.emit_records(b)
    ok, oe = O.emit_records(11, b)
    assert len(ok) > 1000
    assert np.array_equal(keys, ok) and np.array_equal(edges, oe)


# ---------------------------------------------------------------------------------- count
@pytest.mark.parametrize("name", sorted(DB_READS))
def test_count_reproduces_fixture_db(kq, name):
    db = kq.KreeqDB(21, 128)
    db.count_batch(H.reads_batch([H.golden_input(DB_READS[name])]))
    assert H.entries_equal(db.export(), H.load_db_table(name))


@pytest.mark.parametrize("idx", VALIDATE_TESTS)
def test_validate_stdout_goldens(kq, O, idx):
    argv, expected = H.parse_tst(os.path.join(H.GOLDEN, "validateFiles", f"test.{idx}.tst"))
    asm, reads = H.parse_validate_cmd(argv)
    db = kq.KreeqDB(21, 128)
    for r in reads:
        db.count_batch(H.reads_batch([H.golden_input(r)]))
    out = H.stats_block(db.summary())
    ctr = np.zeros(3, dtype=np.uint64)
    for _, seq in H.read_fastx(H.golden_input(asm)):
        c, _ = db.lookup_sequence(seq)
        ctr += c
    # the two doubles are computed on the host from the three integers (errorRate, src/kreeq.cpp:36-40)
    out += H.qv_block(int(ctr[0]), int(ctr[1]), int(ctr[2]), 21, O.error_rate, O.qv)
    assert out == expected


def test_union_golden(kq):
    argv, expected = H.parse_tst(os.path.join(H.GOLDEN, "validateFiles", "test.35.tst"))
    a, b = kq.KreeqDB(21, 128), kq.KreeqDB(21, 128)
    a.import_entries(H.load_db_table("test1"))
    b.import_entries(H.load_db_table("test2"))
    assert H.entries_equal(a.export(), H.load_db_table("test1"))
    a.merge(b)
    assert H.stats_block(a.summary()) == expected
    c = kq.KreeqDB(21, 128)
    c.count_batch(H.reads_batch([H.golden_input("random1.fastq")]))
    c.count_batch(H.reads_batch([H.golden_input("random2.fastq")]))
    assert H.entries_equal(a.export(), c.export())
    # importing two databases into one handle is the same union
    d = kq.KreeqDB(21, 128)
    d.import_entries(H.load_db_table("test1"))
    d.import_entries(H.load_db_table("test2"))
    assert H.entries_equal(a.export(), d.export())


@pytest.mark.parametrize("asm,reads,bkwig", [("repeat1.fasta", "repeat1.fastq", "decompressor2.bkwig"),
                                            ("decompressor1.fasta", "random1.fastq", "decompressor1.bkwig")])
def test_bkwig_per_base(kq, asm, reads, bkwig):
    from tests.test_oracle_golden import per_base_triplets, read_bkwig

    k, vals = read_bkwig(H.golden_input(bkwig))
    db = kq.KreeqDB(k, 128)
    db.count_batch(H.reads_batch([H.golden_input(reads)]))
    seqs = dict(H.read_fastx(H.golden_input(asm)))
    for hdr, pos, a in vals:
        _, pb = db.lookup_sequence(seqs[hdr], per_base=True)
        assert np.array_equal(per_base_triplets(pb[pos:pos + len(a)]), a), (hdr, pos)


@pytest.mark.parametrize("k,seed", [(21, 1), (31, 2), (32, 3), (9, 4)])
def test_count_random_vs_oracle(kq, O, k, seed):
    batch, genome = H.synth_reads(20000, 150, 60000, seed=seed, err=0.01, n_rate=0.003)
    gpu, cpu = kq.KreeqDB(k, 128), O.OracleDB(k, 128)
    gpu.count_batch(batch)
    cpu.count_batch(batch, threads=8)
    assert gpu.summary(with_hist=True) == cpu.summary(with_hist=True)
    assert H.entries_equal(gpu.export(), cpu.export())
    # per-map export agrees and is a partition of the whole
    n = 0
    for lo, hi in ((0, 1), (1, 64), (64, 128)):
        e = gpu.export(lo, hi)
        assert np.all((e["key"] % 128 >= lo) & (e["key"] % 128 < hi))
        n += len(e)
    assert n == gpu.summary()["distinct"]
    # lookup: whole genome, with per-base output, cutoff and map ranges
    for cutoff in (0, 3):
        cg, pg = gpu.lookup_sequence(genome, cov_cutoff=cutoff, per_base=True)
        cc, pc = cpu.validate_sequence(genome, cov_cutoff=cutoff, per_base=True, threads=8)
        assert np.array_equal(cg, cc)
        for f in ("fw", "bw", "cov", "isFw"):
            assert np.array_equal(pg[f], pc[f]), f
    ctr = np.zeros(3, dtype=np.uint64)
    pb = np.zeros(len(genome), dtype=kq.DBGBASE_DTYPE)
    for lo, hi in ((0, 40), (40, 41), (41, 128)):          # map-range passes accumulate (src/kreeq.cpp:59-76)
        c, pb = gpu.lookup_sequence(genome, map_lo=lo, map_hi=hi, per_base_buf=pb)
        ctr += c
    cc, pc = cpu.validate_sequence(genome, per_base=True)
    assert np.array_equal(ctr, cc)
    for f in ("fw", "bw", "cov", "isFw"):
        assert np.array_equal(pb[f], pc[f]), f


def test_overflow_to_high_copy(kq, O):
    """u8 -> u32 tier (src/graph-builder.cpp:166-205): order-independent exact sums, hc iff cov >= 255"""
    rng = np.random.default_rng(7)
    keys = np.concatenate([np.full(3000, 5 * 128 + 3, dtype=np.uint64), np.full(254, 9 * 128 + 3, dtype=np.uint64),
                           np.full(255, 11 * 128 + 3, dtype=np.uint64), np.full(70000, 12345678901, dtype=np.uint64),
                           rng.integers(0, 50, 20000).astype(np.uint64)])
    edges = rng.integers(0, 256, len(keys)).astype(np.uint8)
    p = rng.permutation(len(keys))
    gpu, cpu = kq.KreeqDB(21, 128), O.OracleDB(21, 128)
    gpu.insert_records(keys[p], edges[p])
    cpu.insert_records(keys, edges)
    assert H.entries_equal(gpu.export(), cpu.export())
    assert gpu.summary(with_hist=True) == cpu.summary(with_hist=True)
    # homopolymer reads: one k-mer hammered from every lane
    batch = b"\n".join([b"A" * 500, b"T" * 777, b"ACGT" * 100, b"a" * 30])
    g2, c2 = kq.KreeqDB(21, 128), O.OracleDB(21, 128)
    for _ in range(3):
        g2.count_batch(batch)
        c2.count_batch(batch)
    assert H.entries_equal(g2.export(), c2.export())
    cg, pg = g2.lookup_sequence(b"A" * 50 + b"N" + b"T" * 40, per_base=True)
    cc, pc = c2.validate_sequence(b"A" * 50 + b"N" + b"T" * 40, per_base=True)
    assert np.array_equal(cg, cc)
    for f in ("fw", "bw", "cov", "isFw"):
        assert np.array_equal(pg[f], pc[f])


def test_saturation_at_largest(kq, O):
    """counters saturate at 2^32-1 (include/kreeq.h:68; src/graph-builder.cpp:198-204, :316-329)"""
    big = np.zeros(3, dtype=kq.ENTRY_DTYPE)
    big["key"] = [7, 8, 9]
    big["cov"] = [4294967295, 4294967290, 3000000000]
    big["fw"][:, 1] = [4294967295, 4294967290, 2999999999]
    big["bw"][:, 2] = [17, 4294967289, 5]
    big["hc"] = 1
    gpu, cpu = kq.KreeqDB(21, 128), O.OracleDB(21, 128)
    gpu.import_entries(big)
    cpu.import_entries(big)
    assert H.entries_equal(gpu.export(), cpu.export())
    keys = np.repeat(np.array([7, 8, 9], dtype=np.uint64), 20)
    edges = np.full(60, 0b01000010, dtype=np.uint8)      # fw[1] and bw[2]
    gpu.insert_records(keys, edges)
    cpu.insert_records(keys, edges)
    assert H.entries_equal(gpu.export(), cpu.export())
    g2, c2 = kq.KreeqDB(21, 128), O.OracleDB(21, 128)
    g2.import_entries(big)
    c2.import_entries(big)
    gpu.merge(g2)
    cpu.merge(c2)
    out = gpu.export()
    assert H.entries_equal(out, cpu.export())
    assert out["cov"].max() == 4294967295
    assert gpu.summary() == cpu.summary()


def test_table_growth(kq, O):
    """tiny capacity hint: the table must grow by rehashing without losing a count"""
    batch, _ = H.synth_reads(30000, 120, 3_000_000, seed=21, err=0.0)
    gpu, cpu = kq.KreeqDB(25, 128, capacity_hint=1000), O.OracleDB(25, 128)
    third = len(batch) // 3
    cuts = [batch.rfind(b"\n", 0, third), batch.rfind(b"\n", 0, 2 * third)]
    parts = [batch[:cuts[0]], batch[cuts[0] + 1:cuts[1]], batch[cuts[1] + 1:]]
    before = gpu.info()["slots_total"]
    for p in parts:
        gpu.count_batch(p)
        cpu.count_batch(p, threads=8)
    assert gpu.info()["slots_total"] > before
    assert gpu.info()["slots_used"] == cpu.summary()["distinct"]
    assert H.entries_equal(gpu.export(), cpu.export())


def test_device_pointer_entry_points(kq, O):
    """*_dev variants on torch-owned HBM, including a misaligned base pointer"""
    import torch

    batch, genome = H.synth_reads(5000, 150, 40000, seed=33, err=0.01, n_rate=0.002)
    cpu = O.OracleDB(21, 128)
    cpu.count_batch(batch)
    for shift in (0, 1, 7, 15):
        t = torch.zeros(len(batch) + 64, dtype=torch.uint8, device="cuda")
        t[shift:shift + len(batch)] = torch.frombuffer(bytearray(batch), dtype=torch.uint8).cuda()
        gpu = kq.KreeqDB(21, 128)
        gpu.count_batch_dev(t.data_ptr() + shift, len(batch))
        gpu.sync()
        assert H.entries_equal(gpu.export(), cpu.export()), shift
        g = torch.frombuffer(bytearray(genome), dtype=torch.uint8).cuda()
        ctr = torch.zeros(3, dtype=torch.int64, device="cuda")
        gpu.lookup_sequence_dev(g.data_ptr(), len(genome), ctr.data_ptr())
        gpu.sync()
        cc, _ = cpu.validate_sequence(genome)
        assert ctr.cpu().numpy().astype(np.uint64).tolist() == cc.tolist()


def test_emit_partitioned_then_insert(kq, O):
    """the multi-GPU staging path on one GPU: partition by owner, insert part by part"""
    import torch

    batch, _ = H.synth_reads(8000, 150, 50000, seed=44, err=0.01, n_rate=0.002)
    cpu = O.OracleDB(21, 128)
    cpu.count_batch(batch)
    t = torch.frombuffer(bytearray(batch), dtype=torch.uint8).cuda()
    for n_parts in (1, 2, 8, 128):
        src = kq.KreeqDB(21, 128)
        keys = torch.empty(len(batch), dtype=torch.int64, device="cuda")
        edges = torch.empty(len(batch), dtype=torch.uint8, device="cuda")
        counts = src.emit_partitioned_dev(t.data_ptr(), len(batch), n_parts, keys.data_ptr(), edges.data_ptr(), len(batch))
        assert counts.sum() == len(O.emit_records(21, batch)[0])
        hk = keys.cpu().numpy().astype(np.uint64)
        off = 0
        dst = kq.KreeqDB(21, 128)
        for p in range(n_parts):
            n = int(counts[p])
            part = hk[off:off + n]
            owner = (part % 128) * n_parts // 128
            assert np.all(owner == p)
            dst.insert_records_dev(keys.data_ptr() + 8 * off, edges.data_ptr() + off, n)
            off += n
        dst.sync()
        assert H.entries_equal(dst.export(), cpu.export()), n_parts


def test_errors(kq):
    db = kq.KreeqDB(21, 128)
    with pytest.raises(kq.KqError):
        db.lookup_sequence(b"ACGT" * 10, map_lo=5, map_hi=4)
    with pytest.raises(kq.KqError):
        db.export(0, 129)
    other = kq.KreeqDB(31, 128)
    with pytest.raises(kq.KqError) as e:
        db.merge(other)
    assert e.value.code == -7
    db.count_batch(b"")                     # empty and shorter-than-k batches are no-ops (src/graph-builder.cpp:60)
    db.count_batch(b"ACGTACGT")
    assert db.summary()["distinct"] == 0
    c, _ = db.lookup_sequence(b"ACGT")
    assert c.tolist() == [0, 0, 0]


# ---------------------------------------------------------------------------------- partitioned count path
@pytest.mark.parametrize("k,hint", [(21, 0), (21, 5_000_000), (28, 3_000_000), (9, 0), (2, 0)])
def test_partitioned_count_vs_oracle(kq, O, k, hint):
    """P1 -> (P2) -> P3 path forced on: single-level (<= 1024 regions) and two-level splits"""
    batch, genome = H.synth_reads(30000, 150, 80000, seed=50 + k, err=0.01, n_rate=0.003)
    gpu, cpu = kq.KreeqDB(k, 128, capacity_hint=hint), O.OracleDB(k, 128)
    gpu.set_option("count_path", "partitioned")
    third = batch.rfind(b"\n", 0, len(batch) // 3)
    for part in (batch[:third], batch[third + 1:]):            # two batches: P3 must merge into existing regions
        gpu.count_batch(part)
        cpu.count_batch(part, threads=8)
    assert gpu.summary(with_hist=True) == cpu.summary(with_hist=True)
    assert H.entries_equal(gpu.export(), cpu.export())
    info = gpu.info()
    assert info["kmers_counted"] == cpu.summary()["total"] and info["slots_used"] == cpu.summary()["distinct"]
    cg, _ = gpu.lookup_sequence(genome)
    cc, _ = cpu.validate_sequence(genome, threads=8)
    assert np.array_equal(cg, cc)


def test_partitioned_high_copy_and_mixed_paths(kq, O):
    """hot k-mers (one region gets most records), then more batches through the direct path"""
    rng = np.random.default_rng(3)
    reads = [b"A" * 700, b"T" * 900, b"ACGT" * 200, b"CA" * 300]
    batch1 = b"\n".join(reads[i] for i in rng.integers(0, 4, 400))
    batch2, _ = H.synth_reads(5000, 120, 30000, seed=8, err=0.02, n_rate=0.01)
    gpu, cpu = kq.KreeqDB(21, 128, capacity_hint=6_000_000), O.OracleDB(21, 128)
    for path, b in (("partitioned", batch1), ("direct", batch2), ("partitioned", batch2), ("partitioned", batch1), ("direct", batch1)):
        gpu.set_option("count_path", path)
        gpu.count_batch(b)
        cpu.count_batch(b, threads=8)
    assert gpu.summary(with_hist=True) == cpu.summary(with_hist=True)
    assert H.entries_equal(gpu.export(), cpu.export())


@pytest.mark.parametrize("k,hint", [(29, 0), (31, 5_000_000), (32, 0), (32, 5_870_000), (29, 40_000_000)])   # large hints: 8-byte hash-remainder records
def test_partitioned_wide_records(kq, O, k, hint):
    """k = 29..32: the key fills the u64, edges travel in a parallel byte array through the splits"""
    batch, genome = H.synth_reads(25000, 150, 70000, seed=60 + k, err=0.01, n_rate=0.003)
    gpu, cpu = kq.KreeqDB(k, 128, capacity_hint=hint), O.OracleDB(k, 128)
    gpu.set_option("count_path", "partitioned")
    half = batch.rfind(b"\n", 0, len(batch) // 2)
    for part in (batch[:half], batch[half + 1:]):
        gpu.count_batch(part)
        cpu.count_batch(part, threads=8)
    assert gpu.summary(with_hist=True) == cpu.summary(with_hist=True)
    assert H.entries_equal(gpu.export(), cpu.export())
    # explicit records (reference edge bytes) through the partitioned insert as well
    keys, edges = O.emit_records(k, batch)
    g2 = kq.KreeqDB(k, 128, capacity_hint=hint)
    g2.set_option("count_path", "partitioned")
    g2.insert_records(keys, edges)
    assert H.entries_equal(g2.export(), cpu.export())
    cg, _ = g2.lookup_sequence(genome)
    cc, _ = cpu.validate_sequence(genome, threads=8)
    assert np.array_equal(cg, cc)


def _unpack_records(recs, k):
    """packed 8-byte records (include/kreeq_amd.h: kq_emit_packed_dev) -> (key, reference edge byte)"""
    recs = recs.astype(np.uint64)
    # bits 0..55 = top 56 bits of the left-aligned invertible mix of the key (kq_device.h table_hash): undo it
    from kreeq_amd.dist import key_of_hash
    key = key_of_hash(recs << np.uint64(8), k)
    f = ((recs >> np.uint64(56)) & np.uint64(7)).astype(np.int64)
    b = ((recs >> np.uint64(59)) & np.uint64(7)).astype(np.int64)
    edge = np.where(f < 4, 1 << (7 - np.minimum(f, 3)), 0) | np.where(b < 4, 1 << (7 - (4 + np.minimum(b, 3))), 0)
    return key, edge.astype(np.uint8)


@pytest.mark.parametrize("k,hint", [(21, 0), (27, 4_000_000), (21, 5_000_000), (17, 3_100_000)])    # the last two: receive side converts to 5-byte records
def test_packed_emit_exchange_insert(kq, O, k, hint):
    """multi-GPU staging with packed records on one GPU: owner split -> per-part insert (partitioned)"""
    import torch

    batch, genome = H.synth_reads(12000, 150, 60000, seed=70 + k, err=0.01, n_rate=0.003)
    cpu = O.OracleDB(k, 128)
    cpu.count_batch(batch, threads=8)
    ok, oe = O.emit_records(k, batch)
    t = torch.frombuffer(bytearray(batch), dtype=torch.uint8).cuda()
    for n_parts in (1, 2, 8, 5):
        src = kq.KreeqDB(k, 128)
        recs = torch.empty(len(batch), dtype=torch.int64, device="cuda")
        counts = src.emit_packed_dev(t.data_ptr(), len(batch), n_parts, recs.data_ptr(), len(batch))
        assert int(counts.sum()) == len(ok)
        host = recs[:len(ok)].cpu().numpy().astype(np.uint64)
        key, edge = _unpack_records(host, k)
        # same multiset of (key, edge) records as the reference loop 1
        a = np.sort(key.astype(np.uint64) * np.uint64(256) + edge)[:0]  # (overflow-safe compare below)
        got = np.stack([key, edge.astype(np.uint64)], axis=1)
        exp = np.stack([ok, oe.astype(np.uint64)], axis=1)
        assert np.array_equal(got[np.lexsort((got[:, 1], got[:, 0]))], exp[np.lexsort((exp[:, 1], exp[:, 0]))])
        dst = kq.KreeqDB(k, 128, capacity_hint=hint)
        off = 0
        for p in range(n_parts):
            n = int(counts[p])
            owner = (key[off:off + n] % np.uint64(128)) * np.uint64(n_parts) // np.uint64(128)
            assert np.all(owner == p)
            dst.insert_packed_dev(recs.data_ptr() + 8 * off, n)
            off += n
        dst.sync()
        assert H.entries_equal(dst.export(), cpu.export()), n_parts
        assert dst.summary(with_hist=True) == cpu.summary(with_hist=True)
    cg, _ = dst.lookup_sequence(genome)
    cc, _ = cpu.validate_sequence(genome)
    assert np.array_equal(cg, cc)


def test_sharded_counter_single_rank(kq, O):
    """kreeq_amd.dist.ShardedCounter with world == 1 through the emit -> insert path"""
    import torch

    from kreeq_amd.dist import GpuEngine, ShardedCounter

    batch, genome = H.synth_reads(9000, 150, 50000, seed=91, err=0.01, n_rate=0.002)
    cpu = O.OracleDB(21, 128)
    cpu.count_batch(batch, threads=8)
    eng = GpuEngine(21, 128, 0)
    sc = ShardedCounter(eng, 21, 128, sharded_path=True)
    sc.count_batch(torch.frombuffer(bytearray(batch), dtype=torch.uint8).cuda())
    eng.sync()
    assert sc.summary() == cpu.summary()
    ctr = sc.validate(torch.frombuffer(bytearray(genome), dtype=torch.uint8).cuda())
    cc, _ = cpu.validate_sequence(genome)
    assert ctr.tolist() == cc.tolist()
    # coverage histogram and the database written from the shard(s)
    import tempfile

    from kreeq_amd import hostdb

    assert sc.histogram() == dict(sorted(cpu.summary(with_hist=True)["hist"].items()))
    with tempfile.TemporaryDirectory() as d:
        db = os.path.join(d, "shards.kreeq")
        assert sc.export_db(db) == cpu.summary()["distinct"]
        got, gk, gm = hostdb.read_db(db)
        assert (gk, gm) == (21, 128) and H.entries_equal(got, cpu.export())


def test_sharded_counter_rccl_world1(kq, O):
    """the exchange code (all_to_all_single with split sizes, all_reduce) over the nccl (= RCCL) backend,
    world size 1 -- the only RCCL configuration a one-GPU box can run"""
    import socket

    import torch
    import torch.distributed as dist

    from kreeq_amd.dist import GpuEngine, ShardedCounter

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        batch, genome = H.synth_reads(9000, 150, 50000, seed=92, err=0.01, n_rate=0.002)
        cpu = O.OracleDB(21, 128)
        cpu.count_batch(batch, threads=8)
        eng = GpuEngine(21, 128, 0)
        sc = ShardedCounter(eng, 21, 128, sharded_path=True)
        sc.force_exchange = True
        sc.count_batch(torch.frombuffer(bytearray(batch), dtype=torch.uint8).cuda())
        eng.sync()
        assert sc.summary() == cpu.summary()
        ctr = sc.validate(torch.frombuffer(bytearray(genome), dtype=torch.uint8).cuda())
        cc, _ = cpu.validate_sequence(genome)
        assert ctr.tolist() == cc.tolist()
        assert sc.histogram() == dict(sorted(cpu.summary(with_hist=True)["hist"].items()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("follow", ["partitioned_sparse", "partitioned", "direct", "insert_records", "import", "lookup", "summary", "export",
                                    "merge_dst", "merge_src", "grow"])
def test_clear_then_every_operation(kq, O, follow):
    """kq_clear defers the slot-array clear into the next partitioned count (k_count_regions writes every
    region, also the ones that receive no record); every other table user must materialise it first.
    The table is filled with unrelated k-mers before the clear, so stale slots would show."""
    junk, _ = H.synth_reads(20000, 150, 70000, seed=901, err=0.01)
    batch, genome = H.synth_reads(6000, 120, 20000, seed=902, err=0.01, n_rate=0.003)
    tiny = batch[:batch.find(b"\n", 2000)]
    gpu, cpu = kq.KreeqDB(21, 128, capacity_hint=3_000_000), O.OracleDB(21, 128)
    gpu.set_option("count_path", "partitioned")
    gpu.count_batch(junk)
    gpu.clear()
    if follow in ("partitioned", "partitioned_sparse", "direct"):
        b = tiny if follow == "partitioned_sparse" else batch          # sparse: most regions get no record at all
        gpu.set_option("count_path", "direct" if follow == "direct" else "partitioned")
        gpu.count_batch(b)
        cpu.count_batch(b, threads=8)
    elif follow == "insert_records":
        keys, edges = kq.KreeqDB(21, 128).emit_records(batch)
        gpu.insert_records(keys, edges)
        cpu.count_batch(batch, threads=8)
    elif follow == "import":
        cpu.count_batch(batch, threads=8)
        gpu.import_entries(cpu.export())
    elif follow == "lookup":
        cg, _ = gpu.lookup_sequence(genome)
        cc, _ = cpu.validate_sequence(genome, threads=8)
        assert np.array_equal(cg, cc)
    elif follow in ("merge_dst", "merge_src"):
        other = kq.KreeqDB(21, 128, capacity_hint=3_000_000)
        other.count_batch(batch)
        cpu.count_batch(batch, threads=8)
        if follow == "merge_dst":
            gpu.merge(other)
        else:
            other.merge(gpu)                                            # a cleared source adds nothing
            gpu = other
    elif follow == "grow":
        gpu.set_option("count_path", "direct")
        big, _ = H.synth_reads(60000, 150, 4_000_000, seed=903, err=0.0)   # more distinct k-mers than the hint: rehash of a lazily cleared table
        gpu.count_batch(big)
        cpu.count_batch(big, threads=8)
    assert gpu.summary(with_hist=True) == cpu.summary(with_hist=True)
    assert H.entries_equal(gpu.export(), cpu.export())
    assert gpu.info()["slots_used"] == cpu.summary()["distinct"]


@pytest.mark.parametrize("path,k", [("partitioned", 21), ("direct", 21), ("partitioned", 31)])
def test_sliced_resident_batch(kq, O, path, k):
    """a resident batch is cut into slices at arbitrary positions (inside reads): every k-mer and
    every edge across a cut must be counted exactly once"""
    batch, _ = H.synth_reads(20000, 150, 60000, seed=77, err=0.01, n_rate=0.003)
    cpu = O.OracleDB(k, 128)
    cpu.count_batch(batch, threads=8)
    for slice_kmers in (7, 1000, 123457, len(batch)):
        if slice_kmers < 1000 and path == "partitioned":
            continue                                        # thousands of tiny partition passes: covered by the direct path
        gpu = kq.KreeqDB(k, 128)
        gpu.set_option("count_path", path)
        gpu.set_option("slice_kmers", slice_kmers)
        n = 30000 if slice_kmers == 7 else len(batch)
        sub = batch[:n]
        ref = cpu
        if n != len(batch):
            ref = O.OracleDB(k, 128)
            ref.count_batch(sub)
        gpu.count_batch(sub)
        assert H.entries_equal(gpu.export(), ref.export()), slice_kmers


@pytest.mark.parametrize("path", ["partitioned", "direct"])
def test_trusted_capacity_overflow_fails_loudly(kq, path):
    """KQ_OPT_TRUST_CAPACITY with a hint that is far too small: a table region fills up and the next
    synchronising call reports KQ_ERR_TABLE_FULL instead of dropping k-mers silently"""
    batch, _ = H.synth_reads(40000, 150, 4_000_000, seed=5, err=0.0)       # ~4 M distinct k-mers
    db = kq.KreeqDB(21, 128, capacity_hint=20000)                            # 16 regions x 2048 slots
    db.set_option("trust_capacity", 1)
    db.set_option("count_path", path)
    with pytest.raises(kq.KqError) as e:
        db.count_batch(batch)
        db.summary()
    assert e.value.code == -5 and "overflow" in str(e.value)
    # the same data without the promise grows the table instead
    ok = kq.KreeqDB(21, 128, capacity_hint=20000)
    ok.set_option("count_path", path)
    ok.count_batch(batch)
    assert ok.summary()["total"] == 40000 * 130


@pytest.mark.parametrize("path", ["partitioned", "direct"])
def test_count_map_range_filter(kq, O, path):
    """memory-bounded mode: count the same reads once per map range; the pieces are disjoint and
    their union is the full database (reference map-range loop, src/kreeq.cpp:59-74)"""
    batch, genome = H.synth_reads(15000, 150, 50000, seed=88, err=0.01, n_rate=0.003)
    cpu = O.OracleDB(21, 128)
    cpu.count_batch(batch, threads=8)
    full = cpu.export()
    whole = kq.KreeqDB(21, 128)
    ctr = np.zeros(3, dtype=np.uint64)
    for lo, hi in ((0, 1), (1, 70), (70, 128)):
        piece = kq.KreeqDB(21, 128)
        piece.set_option("count_path", path)
        piece.set_option("count_map_range", (lo, hi))
        piece.count_batch(batch)
        e = piece.export()
        m = full["key"] % 128
        assert H.entries_equal(e, full[(m >= lo) & (m < hi)])
        c, _ = piece.lookup_sequence(genome, map_lo=lo, map_hi=hi)
        ctr += c
        whole.merge(piece)
    assert H.entries_equal(whole.export(), full)
    cc, _ = cpu.validate_sequence(genome)
    assert np.array_equal(ctr, cc)
    with pytest.raises(kq.KqError):
        whole.set_option("count_map_range", (5, 5))


# ---------------------------------------------------------------------------------- 5-byte (narrow) records
@pytest.mark.parametrize("k,hint", [(21, 5_000_000), (20, 3_100_000), (13, 3_000_000), (21, 40_000_000), (22, 5_000_000)])
def test_narrow_record_path_vs_oracle(kq, O, k, hint):
    """tables of >= 2048 regions are rounded to a multiple of 256 regions, and k <= 21 then splits on the top
    8 hash bits with 5-byte records (k = 22 is the first k on the 8-byte path with the same geometry).
    Two batches (k_count_regions merges into existing regions), N runs, a map-range filtered handle."""
    batch, genome = H.synth_reads(40000, 150, 200000, seed=300 + k, err=0.01, n_rate=0.003)
    gpu, cpu = kq.KreeqDB(k, 128, capacity_hint=hint), O.OracleDB(k, 128)
    gpu.set_option("count_path", "partitioned")
    info = gpu.info()
    assert info["slots_total"] % (256 * 2048) == 0                  # the geometry the narrow path needs
    cut = batch.rfind(b"\n", 0, len(batch) // 2)
    for part in (batch[:cut], batch[cut + 1:]):
        gpu.count_batch(part)
        cpu.count_batch(part, threads=8)
    assert gpu.summary(with_hist=True) == cpu.summary(with_hist=True)
    full = cpu.export()
    assert H.entries_equal(gpu.export(), full)
    cg, _ = gpu.lookup_sequence(genome)
    cc, _ = cpu.validate_sequence(genome, threads=8)
    assert np.array_equal(cg, cc)
    piece = kq.KreeqDB(k, 128, capacity_hint=hint)
    piece.set_option("count_path", "partitioned")
    piece.set_option("count_map_range", (17, 90))
    piece.count_batch(batch)
    m = full["key"] % 128
    assert H.entries_equal(piece.export(), full[(m >= 17) & (m < 90)])


def test_narrow_records_hot_kmers_and_growth(kq, O):
    """skewed input (homopolymers: one region takes most records -> hot-region launch) on the narrow path,
    then a batch that makes the table double (the doubled table keeps the 256-region granularity)"""
    rng = np.random.default_rng(5)
    reads = [b"A" * 900, b"T" * 700, b"AC" * 300, b"ACGT" * 200]
    hot = b"\n".join(reads[i] for i in rng.integers(0, 4, 600))
    mixed, _ = H.synth_reads(30000, 150, 100000, seed=77, err=0.02, n_rate=0.005)
    big, _ = H.synth_reads(50000, 150, 5_000_000, seed=78, err=0.0)
    gpu, cpu = kq.KreeqDB(21, 128, capacity_hint=3_000_000), O.OracleDB(21, 128)
    gpu.set_option("count_path", "partitioned")
    before = gpu.info()["slots_total"]
    for b in (hot, mixed, hot + b"\n" + mixed, big):
        gpu.count_batch(b)
        cpu.count_batch(b, threads=8)
    assert gpu.info()["slots_total"] > before and gpu.info()["slots_total"] % (256 * 2048) == 0
    assert gpu.summary(with_hist=True) == cpu.summary(with_hist=True)
    assert H.entries_equal(gpu.export(), cpu.export())


# ---------------------------------------------------------------------------------- partitioned lookup (K3)
@pytest.mark.parametrize("k,hint", [(21, 5_000_000), (21, 0), (27, 4_000_000), (31, 5_000_000), (13, 3_000_000)])
def test_partitioned_lookup_vs_oracle_and_direct(kq, O, k, hint):
    """kq_lookup_sequence without per-base output through P1 -> level -> k_lookup_regions (every record format),
    against the oracle's evaluateSegment restatement and the direct kernel: map ranges, coverage cut-off,
    N runs, k-mers absent from the table, high-copy k-mers (edge counters in the side table)"""
    reads, genome = H.synth_reads(30000, 150, 120000, seed=400 + k, err=0.01, n_rate=0.002)
    rep = b"\n".join([b"ACGTTGCA" * 100] * 300)                      # cov and edges far beyond 255
    gpu, cpu = kq.KreeqDB(k, 128, capacity_hint=hint), O.OracleDB(k, 128)
    for b in (reads, rep):
        gpu.count_batch(b)
        cpu.count_batch(b, threads=8)
    rng = np.random.default_rng(k)
    asm = bytearray(genome + b"N" + b"ACGTTGCA" * 50 + b"NNN" + bytes(rng.choice(list(b"ACGT"), 30000).tolist()))
    for pos in rng.integers(0, len(genome), 200):
        asm[pos] = b"ACGT"[rng.integers(0, 4)]                       # assembly errors -> missing k-mers / edges
    asm = bytes(asm)
    for cutoff, lo, hi in ((0, 0, 128), (3, 0, 128), (0, 17, 90), (2, 100, 128)):
        cc, _ = cpu.validate_sequence(asm, cov_cutoff=cutoff, map_lo=lo, map_hi=hi, threads=8)
        gpu.set_option("lookup_path", "direct")
        cd, _ = gpu.lookup_sequence(asm, cov_cutoff=cutoff, map_lo=lo, map_hi=hi)
        gpu.set_option("lookup_path", "partitioned")
        cp, _ = gpu.lookup_sequence(asm, cov_cutoff=cutoff, map_lo=lo, map_hi=hi)
        assert np.array_equal(cd, cc), (cutoff, lo, hi)
        assert np.array_equal(cp, cc), (cutoff, lo, hi)
    # sliced sequence: k-mers and neighbours across a cut are evaluated exactly once
    gpu.set_option("slice_kmers", 50021)
    cp, _ = gpu.lookup_sequence(asm)
    cc, _ = cpu.validate_sequence(asm, threads=8)
    assert np.array_equal(cp, cc)


# ---------------------------------------------------------------------------------- union by regions (K4)
@pytest.mark.parametrize("k,hint_dst,hint_src", [(21, 3_000_000, 3_000_000), (21, 9_000_000, 2_000_000), (21, 0, 4_000_000), (31, 2_500_000, 5_000_000)])
def test_merge_by_regions(kq, O, k, hint_dst, hint_src):
    """kq_merge region by region (tables of equal and of different geometry, destination empty / lazily cleared /
    filled / growing, high-copy entries on either side) == per-entry atomic merge == oracle union"""
    a, _ = H.synth_reads(25000, 150, 150000, seed=500 + k, err=0.01, n_rate=0.002)
    b, _ = H.synth_reads(25000, 150, 150000, seed=501 + k, err=0.01)       # different genome: mostly new keys
    rep = b"\n".join([b"GATTACA" * 120] * 200)
    srcs = []
    for batch in (a + b"\n" + rep, b, a + b"\n" + rep):                     # third: every key already present, cov sums pass 255
        g = kq.KreeqDB(k, 128, capacity_hint=hint_src); g.count_batch(batch)
        c = O.OracleDB(k, 128); c.count_batch(batch, threads=8)
        srcs.append((g, c))
    for path in ("partitioned", "direct"):
        dst, ref = kq.KreeqDB(k, 128, capacity_hint=hint_dst), O.OracleDB(k, 128)
        dst.set_option("merge_path", path)
        junk, _ = H.synth_reads(2000, 100, 5000, seed=9)
        dst.count_batch(junk); dst.clear()                                  # lazily cleared destination
        for g, c in srcs:
            dst.merge(g)
            ref.merge(c)
            assert dst.summary(with_hist=True) == ref.summary(with_hist=True), path
        assert H.entries_equal(dst.export(), ref.export()), path
        assert dst.info()["slots_used"] == ref.summary()["distinct"]


@pytest.mark.parametrize("hint,mid,k", [(5_000_000, 2, 21), (5_870_000, 4, 21), (40_000_000, 16, 21), (5_870_000, 2048, 21),
                                          (5_870_000, 4, 31), (5_000_000, 2, 32)])
def test_narrow_middle_level(kq, O, hint, mid, k):
    """very large tables (>= 2048 regions per hash-prefix bucket) split bucket -> sub-bucket -> region; the
    threshold is lowered here so that small tables take the same three-pass route (1, 2 and 3 sub-bucket bits),
    through count, packed insert and the partitioned lookup"""
    import torch

    batch, genome = H.synth_reads(40000, 150, 200000, seed=611, err=0.01, n_rate=0.003)
    gpu, cpu = kq.KreeqDB(k, 128, capacity_hint=hint), O.OracleDB(k, 128)
    gpu.set_option("count_path", "partitioned")
    gpu.set_option("narrow_mid", mid)
    cut = batch.rfind(b"\n", 0, len(batch) // 2)
    gpu.count_batch(batch[:cut])
    cpu.count_batch(batch, threads=8)
    rest = batch[cut + 1:]
    if k <= 28:
        # second half through the multi-GPU staging format (packed records -> converted -> narrow levels)
        t = torch.frombuffer(bytearray(rest), dtype=torch.uint8).cuda()
        recs = torch.empty(len(rest), dtype=torch.int64, device="cuda")
        src = kq.KreeqDB(k, 128)
        counts = src.emit_packed_dev(t.data_ptr(), len(rest), 1, recs.data_ptr(), len(rest))
        gpu.insert_packed_dev(recs.data_ptr(), int(counts[0]))
    else:
        gpu.count_batch(rest)                                       # 8-byte hash-remainder records through the same levels
    gpu.sync()
    assert gpu.summary(with_hist=True) == cpu.summary(with_hist=True)
    assert H.entries_equal(gpu.export(), cpu.export())
    gpu.set_option("lookup_path", "partitioned")
    cg, _ = gpu.lookup_sequence(genome)
    cc, _ = cpu.validate_sequence(genome, threads=8)
    assert np.array_equal(cg, cc)


def test_partitioned_paths_degenerate_inputs(kq, O):
    """forced region-wise paths on inputs that bring nothing: sequences shorter than k / all N / empty,
    an empty source or destination in kq_merge, an all-N read batch on the 5-byte count path"""
    batch, genome = H.synth_reads(20000, 150, 80000, seed=700, err=0.01)
    gpu, cpu = kq.KreeqDB(21, 128, capacity_hint=5_000_000), O.OracleDB(21, 128)
    gpu.set_option("count_path", "partitioned")
    gpu.set_option("lookup_path", "partitioned")
    gpu.set_option("merge_path", "partitioned")
    gpu.count_batch(b"N" * 500000)                                   # no valid k-mer at all
    assert gpu.summary()["total"] == 0 and gpu.info()["slots_used"] == 0
    for seq in (b"", b"ACGT", b"N" * 100000, b"ACGTACGTACGTACGTACGTA", genome[:20]):
        c, _ = gpu.lookup_sequence(seq)
        cc, _ = cpu.validate_sequence(seq)
        assert np.array_equal(c, cc), seq[:30]
    empty = kq.KreeqDB(21, 128, capacity_hint=5_000_000)
    gpu.merge(empty)                                                 # empty into empty
    assert gpu.summary()["total"] == 0
    gpu.count_batch(batch)
    cpu.count_batch(batch, threads=8)
    gpu.merge(empty)                                                 # empty into filled
    empty.set_option("merge_path", "partitioned")
    empty.merge(gpu)                                                 # filled into empty
    assert gpu.summary(with_hist=True) == cpu.summary(with_hist=True)
    assert H.entries_equal(empty.export(), cpu.export())
    c, _ = empty.lookup_sequence(genome)
    cc, _ = cpu.validate_sequence(genome, threads=8)
    assert np.array_equal(c, cc)


def test_lookup_keys_and_branch_scan(kq, O):
    """kq_lookup_keys == map->find per key (present and absent keys, a high-copy k-mer included); kq_branch_scan flags
    = (present, searchVariants has a candidate at depth 0) per position, checked against the oracle's table"""
    from oracle import variants as V

    k = 21
    batch, genome = H.synth_reads(3000, 150, 40000, seed=61, err=0.01)
    hot = b"\n".join([b"ACGTTGCA" * 19] * 300)
    gpu, cpu = kq.KreeqDB(k, 128), O.OracleDB(k, 128)
    for b in (batch, hot):
        gpu.count_batch(b)
        cpu.count_batch(b, threads=4)
    want = cpu.export()
    rng = np.random.default_rng(3)
    absent = rng.integers(0, 1 << 42, 500, dtype=np.uint64)
    absent = absent[~np.isin(absent, want["key"])]
    keys = np.concatenate([want["key"], absent])
    perm = rng.permutation(len(keys))
    got = gpu.lookup_keys(keys[perm])
    inv = np.argsort(perm)
    got = got[inv]
    assert H.entries_equal(got[:len(want)], want)
    assert want["cov"].max() > 255
    assert (got[len(want):]["cov"] == 0).all() and (got[len(want):]["key"] == absent).all()
    # branch scan over the genome (lower-case bases and an N included)
    seq = bytearray(genome[:20000])
    seq[5000] = ord("N")
    seq = bytes(seq)
    flags = gpu.branch_scan(seq)
    g = V.Graph(want, k)
    codes = [V.CTOI.get(chr(c), 4) for c in seq]
    for c in range(len(seq)):
        window = codes[c:c + k]
        f = 0
        if len(window) == k and 4 not in window:
            key, fw = V.hash_kmer(window, k)
            if key in g.nodes:
                f = 1
                nxt = codes[c + k] if c + k < len(seq) else 4
                fwc, bwc, _ = g.nodes[key]
                for i in range(4):
                    edge = fwc[i] != 0 if fw else bwc[i] > 0
                    if edge and (i if fw else 3 - i) != nxt:
                        f |= 2
        assert flags[c] == f, (c, flags[c], f)


@pytest.mark.parametrize("k,hint,n_parts,n_peers", [(21, 5_000_000, 3, 2), (21, 40_000_000, 8, 3), (17, 5_000_000, 2, 1), (21, 5_870_000, 5, 4),
                                                    (21, 100_000_000, 2, 3),      # >= 2^16 regions per receiver: FMT_TIGHT pending sets
                                                    (21, 5_000_000, 1, 2)])       # one part: an ordinary (unwindowed) table
def test_sharded5_emit_exchange_insert(kq, O, k, hint, n_parts, n_peers):
    """multi-GPU exchange with 5-byte records, emulated in one process: n_peers senders split their reads by hash-prefix
    bucket (kq_emit_sharded_dev: the parts are bucket ranges), each of n_parts receivers -- a table that is the WINDOW of
    its buckets (KQ_OPT_BUCKET_WINDOW) -- gets its run from every peer plus the per-bucket counts and inserts them
    (kq_insert_sharded_dev).  Every receiver must end up with exactly the oracle's k-mers of its buckets, and the QV
    counters of the shards must add up to the oracle's."""
    import torch

    from kreeq_amd.dist import bucket_of, bucket_range

    batches = [H.synth_reads(9000 + 700 * q, 150, 300_000, seed=500 + q, err=0.01, n_rate=0.002)[0] for q in range(n_peers)]
    _, genome = H.synth_reads(10, 150, 300_000, seed=500)
    cpu = O.OracleDB(k, 128)
    for b in batches:
        cpu.count_batch(b, threads=8)
        cpu.count_batch(b, threads=8)             # the receivers insert every run twice
    want = cpu.export()
    want_bucket = bucket_of(want["key"], k)
    dev = torch.device("cuda", 0)
    sender = kq.KreeqDB(k, 128)                   # a sender needs no particular table
    runs, metas = [], []
    for b in batches:
        t = torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
        recs = torch.empty(t.numel(), dtype=torch.int32, device=dev)
        aux = torch.empty(t.numel(), dtype=torch.uint8, device=dev)
        meta = torch.empty((n_parts, 256), dtype=torch.int64, device=dev)
        counts = sender.emit_sharded_dev(t.data_ptr(), t.numel(), n_parts, recs.data_ptr(), aux.data_ptr(), recs.numel(), meta.data_ptr())
        assert meta.sum(dim=1).cpu().tolist() == counts.tolist()
        for p in range(n_parts):                  # a part's counts lie in its bucket range only
            lo, hi = bucket_range(p, n_parts)
            assert int(meta[p, :lo].sum()) == 0 and int(meta[p, hi:].sum()) == 0
        off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        runs.append([(recs[off[p]:off[p + 1]].clone(), aux[off[p]:off[p + 1]].clone()) for p in range(n_parts)])
        metas.append(meta.clone())
    total = 0
    ctr_direct, ctr_part = np.zeros(3, dtype=np.uint64), np.zeros(3, dtype=np.uint64)
    summ = {"total": 0, "unique": 0, "distinct": 0, "edges": 0}
    for p in range(n_parts):
        recv = kq.KreeqDB(k, 128, capacity_hint=hint)
        recv.set_option("trust_capacity", 1)
        lo, hi = bucket_range(p, n_parts)
        if n_parts > 1:
            before = recv.info()["slots_total"]
            recv.set_option("bucket_window", lo | (hi << 16))
            assert before <= recv.info()["slots_total"] <= 2 * before + (1 << 22)      # the window keeps the memory kq_create sized (rounded up)
        r = torch.cat([runs[q][p][0] for q in range(n_peers)])
        a = torch.cat([runs[q][p][1] for q in range(n_peers)])
        m = torch.stack([metas[q][p] for q in range(n_peers)]).contiguous()
        recv.insert_sharded_dev(r.data_ptr(), a.data_ptr(), r.numel(), n_peers, m.data_ptr())
        # a second, pending-set round on the filled table
        recv.insert_sharded_dev(r.data_ptr(), a.data_ptr(), r.numel(), n_peers, m.data_ptr())
        got = recv.export()
        mine = want[(want_bucket >= lo) & (want_bucket < hi)]
        assert H.entries_equal(got, mine)
        total += len(got)
        s = recv.summary()
        for f in summ:
            summ[f] += s[f]
        # the shard answers for its buckets only: direct probes and the region-wise path
        recv.set_option("lookup_path", "direct")
        ctr_direct += recv.lookup_sequence(genome)[0]
        recv.set_option("lookup_path", "partitioned")
        ctr_part += recv.lookup_sequence(genome)[0]
        # map-range export of a shard (what the database writer of the multi-GPU driver asks for)
        sub = recv.export(32, 96)
        mm = mine["key"] % np.uint64(128)
        assert H.entries_equal(sub, mine[(mm >= 32) & (mm < 96)])
    assert total == len(want)
    c_cpu, _ = cpu.validate_sequence(genome)
    assert np.array_equal(ctr_direct, c_cpu) and np.array_equal(ctr_part, c_cpu)
    ref = cpu.summary()
    assert all(summ[f] == ref[f] for f in summ)


def test_bucket_window_rules(kq, O):
    """KQ_OPT_BUCKET_WINDOW: only on an empty handle with k <= 21; a windowed handle drops foreign k-mers on every count path"""
    from kreeq_amd.dist import bucket_of

    k = 21
    b, _ = H.synth_reads(9000, 150, 200_000, seed=41, err=0.01, n_rate=0.002)
    db = kq.KreeqDB(31, 128, capacity_hint=5_000_000)
    with pytest.raises(kq.KqError):
        db.set_option("bucket_window", 0 | (128 << 16))          # k > 21
    db = kq.KreeqDB(k, 128, capacity_hint=5_000_000)
    with pytest.raises(kq.KqError):
        db.set_option("bucket_window", 7 | (7 << 16))            # empty range
    db.count_batch(b)
    with pytest.raises(kq.KqError):
        db.set_option("bucket_window", 0 | (128 << 16))          # not empty any more
    cpu = O.OracleDB(k, 128)
    cpu.count_batch(b, threads=8)
    want = cpu.export()
    wb = bucket_of(want["key"], k)
    for path in ("direct", "partitioned"):
        win = kq.KreeqDB(k, 128, capacity_hint=5_000_000)
        win.set_option("bucket_window", 100 | (171 << 16))
        win.set_option("count_path", path)
        win.count_batch(b)
        assert H.entries_equal(win.export(), want[(wb >= 100) & (wb < 171)])
        # growth keeps the window (no trust_capacity: the worst-case reservation of the later batches forces a rehash)
        small = kq.KreeqDB(k, 128, capacity_hint=1_000_000)
        small.set_option("bucket_window", 100 | (171 << 16))
        small.set_option("count_path", path)
        slots0 = small.info()["slots_total"]
        cpu2 = O.OracleDB(k, 128)
        for i in range(3):
            bi, _ = H.synth_reads(9000, 150, 2_000_000, seed=60 + i, err=0.01)
            small.count_batch(bi)
            cpu2.count_batch(bi, threads=8)
        w2 = cpu2.export()
        w2b = bucket_of(w2["key"], k)
        assert H.entries_equal(small.export(), w2[(w2b >= 100) & (w2b < 171)])
        assert small.info()["slots_total"] > slots0


def test_pipelined_host_ingest(kq, O):
    """kq_count_batch_async + kq_host_alloc / kq_host_wait (the CLI's ingest path for large inputs): batches in pinned
    buffers, copies on the copy stream into the ring of device staging buffers, counts behind them; two buffers are reused
    round-robin after waiting for their tickets.  The table must equal the oracle's."""
    import ctypes as C

    from kreeq_amd import capi

    L = capi.load()
    k = 21
    batches = [H.synth_reads(9000 + 400 * i, 150, 200_000, seed=700 + i, err=0.01, n_rate=0.002)[0] for i in range(7)]
    cpu = O.OracleDB(k, 128)
    for b in batches:
        cpu.count_batch(b, threads=8)
    gpu = kq.KreeqDB(k, 128, capacity_hint=5_000_000)
    cap = max(len(b) for b in batches)
    bufs = [L.kq_host_alloc(cap) for _ in range(2)]
    assert all(bufs)
    tickets = [None, None]
    try:
        for i, b in enumerate(batches):
            j = i % 2
            if tickets[j] is not None:
                gpu.host_wait(tickets[j])                # the copy out of this buffer has finished: it may be refilled
            C.memmove(bufs[j], b, len(b))
            tickets[j] = gpu.count_batch_async(bufs[j], len(b))
        # a pageable buffer works too (staged copy)
        extra, _ = H.synth_reads(5000, 150, 200_000, seed=790, err=0.01)
        arr = np.frombuffer(extra, dtype=np.uint8).copy()
        t = gpu.count_batch_async(arr.ctypes.data, len(arr))
        gpu.host_wait(t)
        cpu.count_batch(extra, threads=8)
        gpu.sync()
        assert gpu.summary() == cpu.summary()
        assert H.entries_equal(gpu.export(), cpu.export())
        with pytest.raises(kq.KqError):
            gpu.host_wait(10 ** 9)                       # unknown ticket
    finally:
        for p in bufs:
            L.kq_host_free(p)


def test_concurrent_host_ingest(kq, O):
    """kq_count_batch_async from several threads at once (the CLI's parser threads submit their own buffers without a lock
    of their own): pageable buffers, 16 threads over 8 staging slots, batches of different sizes incl. empty ones"""
    import threading

    k = 21
    batches = [H.synth_reads(300 + 211 * i, 150, 300_000, seed=900 + i, err=0.01, n_rate=0.002)[0] for i in range(48)]
    batches[5] = b""
    batches[17] = b"ACGT"                                 # shorter than k
    cpu = O.OracleDB(k, 128)
    for b in batches:
        if b:
            cpu.count_batch(b, threads=8)
    gpu = kq.KreeqDB(k, 128, capacity_hint=8_000_000)
    arrs = [np.frombuffer(b, dtype=np.uint8).copy() if b else np.zeros(1, dtype=np.uint8) for b in batches]
    errors = []

    def worker(t):
        try:
            for i in range(t, len(batches), 16):
                tk = gpu.count_batch_async(arrs[i].ctypes.data, len(batches[i]))
                gpu.host_wait(tk)
        except Exception as e:                           # noqa: BLE001 - reported below
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(16)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    gpu.sync()
    assert gpu.summary() == cpu.summary()
    assert H.entries_equal(gpu.export(), cpu.export())


@pytest.mark.parametrize("k,path", [(21, "partitioned"), (21, "direct"), (31, "partitioned"), (11, "direct")])
def test_packed_input(kq, O, k, path):
    """2-bit packed batches (kq_pack_bases -> kq_count_packed_dev / _async) count exactly like their ASCII form: read
    separators, N runs, lower case, lengths that are no multiple of 16, slices cutting inside a unit"""
    import torch

    from kreeq_amd import capi

    rng = np.random.default_rng(3)
    batches = [H.synth_reads(9000 + 37 * i, 150 - i, 300_000, seed=40 + i, err=0.01, n_rate=0.003)[0] for i in range(3)]
    batches.append(b"acgtnACGT" * 7 + b"\n" + b"A" * (k - 1) + b"\n" + bytes(rng.choice(list(b"ACGTacgtN\n"), size=100_003).tolist()))
    batches.append(b"ACGTAC")                               # shorter than k
    # pack on the host, check the format against a plain restatement
    codes, inv = capi.pack_bases(batches[3])
    raw = np.frombuffer(batches[3], dtype=np.uint8)
    lut = np.full(256, 4, dtype=np.uint8)
    for ch, v in zip(b"ACGTacgt", [0, 1, 2, 3, 0, 1, 2, 3]):
        lut[ch] = v
    c = lut[raw]
    pad = (-len(c)) % 16
    cc = np.concatenate([c, np.full(pad, 4, dtype=np.uint8)]).reshape(-1, 16)
    want_codes = ((cc & 3).astype(np.uint32) << (2 * np.arange(16, dtype=np.uint32))).sum(axis=1, dtype=np.uint64).astype(np.uint32)
    want_inv = (((cc >> 2) & 1).astype(np.uint32) << np.arange(16, dtype=np.uint32)).sum(axis=1).astype(np.uint16)
    assert np.array_equal(codes, want_codes) and np.array_equal(inv, want_inv)
    cpu = O.OracleDB(k, 128)
    gpu = kq.KreeqDB(k, 128, capacity_hint=6_000_000)
    gpu.set_option("count_path", path)
    gpu.set_option("slice_kmers", 300_001)                 # several slices per batch, cut inside 16-base units
    dev = torch.device("cuda", 0)
    for i, b in enumerate(batches):
        cpu.count_batch(b, threads=8)
        codes, inv = capi.pack_bases(b)
        if i % 2 == 0:                                      # device-resident arrays
            dc = torch.from_numpy(codes.view(np.int32).copy()).to(dev)
            di = torch.from_numpy(inv.view(np.int16).copy()).to(dev)
            torch.cuda.synchronize()
            gpu.count_packed_dev(dc.data_ptr(), di.data_ptr(), len(b))
            gpu.sync()
        else:                                               # host arrays through the staging ring
            tk = gpu.count_packed_async(codes.ctypes.data, inv.ctypes.data, len(b))
            gpu.host_wait(tk)
    gpu.sync()
    assert gpu.summary() == cpu.summary()
    assert H.entries_equal(gpu.export(), cpu.export())


"""world_size-2 (and 3) gloo tests of the bucket-sharded count path (kreeq_amd/dist.py) on CPU.
The routing code under test is the product's; the per-rank compute engine is a host stand-in built
on the oracle (test infrastructure), exactly the interface GpuEngine implements."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import helpers as H


class HostEngine:
    def __init__(self, k, map_count):
        from oracle import oracle as O

        self.O, self.k, self.map_count = O, k, map_count
        self.db = O.OracleDB(k, map_count)

    def emit_partitioned(self, bases, n_parts, slot=0):
        from kreeq_amd.dist import owner_of

        keys, edges = self.O.emit_records(self.k, bases.numpy().tobytes())
        owner = owner_of(keys, n_parts, self.map_count)
        order = np.argsort(owner, kind="stable")
        counts = np.bincount(owner, minlength=n_parts).astype(np.int64)
        return [torch.from_numpy(keys[order].astype(np.int64)), torch.from_numpy(edges[order])], counts

    def insert(self, payload):
        keys, edges = payload
        self.db.insert_records(keys.numpy().astype(np.uint64), edges.numpy())

    def lookup(self, bases, map_lo, map_hi, cov_cutoff=0):
        c, _ = self.db.validate_sequence(bases.numpy().tobytes(), cov_cutoff=cov_cutoff, map_lo=map_lo, map_hi=map_hi)
        return torch.from_numpy(c.astype(np.int64))

    def summary_vector(self):
        s = self.db.summary()
        return torch.tensor([s["total"], s["unique"], s["distinct"], s["edges"]], dtype=torch.int64)

    def histogram(self):
        return self.db.summary(with_hist=True)["hist"]

    def export(self, map_lo, map_hi):
        ent = self.db.export()
        m = ent["key"] % np.uint64(self.map_count)
        return ent[(m >= map_lo) & (m < map_hi)]


class HostBucketEngine(HostEngine):
    """the k <= 21 path of GpuEngine (ownership by hash-prefix bucket range, windowed tables, 5-byte records + per-bucket
    counts) on the host: records travel as (key, edge byte) -- the routing, not the record format, is under test"""
    sharded5 = True

    def __init__(self, k, map_count):
        super().__init__(k, map_count)
        self.window = (0, 256)

    def set_window(self, lo, hi):
        self.window = (lo, hi)

    def emit_partitioned(self, bases, n_parts, slot=0):
        from kreeq_amd.dist import bucket_of, bucket_range

        keys, edges = self.O.emit_records(self.k, bases.numpy().tobytes())
        b = bucket_of(keys, self.k)
        order = np.argsort(b, kind="stable")                                  # bucket-sorted = grouped by owner
        firsts = [bucket_range(p, n_parts)[0] for p in range(n_parts)] + [256]
        per_bucket = np.bincount(b, minlength=256).astype(np.int64)
        counts = np.array([per_bucket[firsts[p]:firsts[p + 1]].sum() for p in range(n_parts)], dtype=np.int64)
        meta = np.zeros((n_parts, 256), dtype=np.int64)
        for p in range(n_parts):
            meta[p, firsts[p]:firsts[p + 1]] = per_bucket[firsts[p]:firsts[p + 1]]
        return [torch.from_numpy(keys[order].astype(np.int64)), torch.from_numpy(edges[order])], counts, torch.from_numpy(meta)

    def insert(self, payload, meta=None):
        from kreeq_amd.dist import bucket_of

        keys = payload[0].numpy().astype(np.uint64)
        assert meta is not None and int(meta.sum()) == len(keys)
        lo, hi = self.window
        assert np.all(meta.numpy()[:, :lo] == 0) and np.all(meta.numpy()[:, hi:] == 0)
        b = bucket_of(keys, self.k)
        assert np.all((b >= lo) & (b < hi))                                   # only k-mers of this rank's buckets arrive
        self.db.insert_records(keys, payload[1].numpy())

    def lookup(self, bases, map_lo, map_hi, cov_cutoff=0):
        from kreeq_amd.dist import bucket_of

        assert (map_lo, map_hi) == (0, self.map_count)
        raw = bases.numpy().tobytes()
        c, _ = self.db.validate_sequence(raw, cov_cutoff=cov_cutoff)
        # a window does not evaluate foreign k-mers; here they were looked up, found absent and counted as missing
        keys, _ = self.O.emit_records(self.k, raw)
        b = bucket_of(keys, self.k)
        foreign = int(((b < self.window[0]) | (b >= self.window[1])).sum())
        c = c.astype(np.int64)
        c[0] -= foreign
        c[1] -= foreign
        return torch.from_numpy(c)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, k, out_dir, buckets=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kreeq_amd.dist import ShardedCounter, bucket_of, bucket_range, owner_range

        sc = ShardedCounter((HostBucketEngine if buckets else HostEngine)(k, 128), k, 128)
        assert (sc.map_lo, sc.map_hi) == owner_range(rank, world, 128)
        assert sc.bucket_mode == buckets and (not buckets or sc.engine.window == bucket_range(rank, world))
        # two batches per rank, different reads on every rank (seed depends on rank)
        for b in range(2):
            n_reads = 1500 if rank != 1 else 3            # rank 1 brings a tiny batch: fewer chunks than its peers
            batch, _ = H.synth_reads(n_reads, 100, 30000, seed=1000 + 10 * rank + b, err=0.01, n_rate=0.003)
            sc.count_batch(torch.frombuffer(bytearray(batch), dtype=torch.uint8))
        _, genome = H.synth_reads(10, 100, 30000, seed=1000)          # same genome (seed of rank 0, batch 0)
        ctr = sc.validate(torch.frombuffer(bytearray(genome), dtype=torch.uint8))
        summ = sc.summary()
        hist = sc.histogram()
        n_written = sc.export_db(os.path.join(out_dir, "sharded.kreeq"))
        ent = sc.engine.db.export()
        m = ent["key"] % 128
        if buckets:                                                       # a rank holds only the buckets it owns, and wrote the maps it owns
            b = bucket_of(ent["key"], k)
            assert np.all((b >= sc.bucket_lo) & (b < sc.bucket_hi))
        else:
            assert n_written == len(ent)
            assert np.all((m >= sc.map_lo) & (m < sc.map_hi))            # a rank holds only the maps it owns
        np.save(os.path.join(out_dir, f"entries_{rank}.npy"), ent)
        if rank == 0:
            np.save(os.path.join(out_dir, "hist.npy"), np.array(sorted(hist.items()), dtype=np.uint64))
            np.save(os.path.join(out_dir, "ctr.npy"), ctr)
            np.save(os.path.join(out_dir, "summ.npy"), np.array([summ[f] for f in ("total", "unique", "distinct", "missing", "edges")], dtype=np.uint64))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,k,buckets", [(2, 21, False), (3, 31, False), (2, 21, True), (3, 19, True)])
def test_sharded_count_matches_single(tmp_path, world, k, buckets):
    from oracle import oracle as O

    port = _free_port()
    mp.spawn(_worker, args=(world, port, k, str(tmp_path), buckets), nprocs=world, join=True)
    ref = O.OracleDB(k, 128)
    for rank in range(world):
        for b in range(2):
            batch, _ = H.synth_reads(1500 if rank != 1 else 3, 100, 30000, seed=1000 + 10 * rank + b, err=0.01, n_rate=0.003)
            ref.count_batch(batch)
    _, genome = H.synth_reads(10, 100, 30000, seed=1000)
    parts = [np.load(os.path.join(tmp_path, f"entries_{r}.npy")) for r in range(world)]
    merged = np.concatenate(parts)
    merged = merged[np.argsort(merged["key"])]
    assert H.entries_equal(merged, ref.export())
    c, _ = ref.validate_sequence(genome)
    assert np.load(os.path.join(tmp_path, "ctr.npy")).tolist() == c.tolist()
    s = ref.summary(with_hist=True)
    assert np.load(os.path.join(tmp_path, "summ.npy")).tolist() == [s[f] for f in ("total", "unique", "distinct", "missing", "edges")]
    # the all-reduced coverage histogram is the single-process one
    assert [tuple(x) for x in np.load(os.path.join(tmp_path, "hist.npy")).tolist()] == sorted(s["hist"].items())
    # the ranks together wrote ONE database: identical content to a single-process one (read back with the host reader)
    from kreeq_amd import hostdb

    got, gk, gm = hostdb.read_db(os.path.join(tmp_path, "sharded.kreeq"))
    assert (gk, gm) == (k, 128)
    assert H.entries_equal(got, ref.export())
    single = os.path.join(tmp_path, "single.kreeq")
    e = ref.export()
    hc = hostdb.write_maps(single, 128, 0, 128, e)
    hostdb.write_finish(single, k, 128, hc)
    for name in [".index", ".map.hc.bin"] + [f".map.{m}.bin" for m in range(128)]:
        a, b = os.path.join(tmp_path, "sharded.kreeq", name), os.path.join(single, name)
        assert os.path.getsize(a) == os.path.getsize(b), name


def test_bucket_ranges_are_a_partition():
    from kreeq_amd.dist import bucket_range

    for world in (1, 2, 3, 5, 8, 256):
        seen = np.zeros(256, dtype=int)
        for r in range(world):
            lo, hi = bucket_range(r, world)
            assert lo < hi
            seen[lo:hi] += 1
        assert np.all(seen == 1)


def test_owner_mapping_is_a_partition():
    from kreeq_amd.dist import owner_of, owner_range

    for world in (1, 2, 3, 5, 8, 128):
        keys = np.arange(0, 5000, dtype=np.uint64) * np.uint64(2654435761)
        own = owner_of(keys, world, 128)
        seen = np.zeros(128, dtype=int)
        for r in range(world):
            lo, hi = owner_range(r, world, 128)
            seen[lo:hi] += 1
            m = keys[own == r] % np.uint64(128)
            assert np.all((m >= lo) & (m < hi))
        assert np.all(seen == 1)

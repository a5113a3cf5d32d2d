"""The multi-GPU driver with the REAL device engine at world size 2 and 3 -- on one GPU: every rank is a process with its
own handle on cuda:0, the collectives run over gloo (payloads staged through the host: RCCL refuses two ranks on one
device).  What is under test is everything above the C ABI that the single-process emulation in test_gpu_parity.py does
not see: GpuEngine, windowed tables set up by ShardedCounter, the chunked exchange, validate / summary / histogram
reductions and the database written from the shards."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import helpers as H

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _batch(rank, b, n=7000):
    return H.synth_reads(n if rank != 1 else 900, 150, 250_000, seed=2000 + 10 * rank + b, err=0.01, n_rate=0.003)[0]


def _worker(rank, world, port, k, hint, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from kreeq_amd.dist import GpuEngine, ShardedCounter, bucket_of, bucket_range

        eng = GpuEngine(k, 128, 0, capacity_hint=hint)
        sc = ShardedCounter(eng, k, 128)
        assert sc.stage_host
        assert sc.bucket_mode == (k <= 21 and hint >= 3_000_000)
        dev = torch.device("cuda", 0)
        for b in range(2):
            t = torch.frombuffer(bytearray(_batch(rank, b)), dtype=torch.uint8).to(dev)
            sc.count_batch(t)
        _, genome = H.synth_reads(10, 150, 250_000, seed=2000)
        ctr = sc.validate(torch.frombuffer(bytearray(genome), dtype=torch.uint8).to(dev))
        summ = sc.summary()
        hist = sc.histogram()
        sc.export_db(os.path.join(out_dir, "sharded.kreeq"))
        ent = eng.db.export()
        if sc.bucket_mode:
            bk = bucket_of(ent["key"], k)
            lo, hi = bucket_range(rank, world)
            assert np.all((bk >= lo) & (bk < hi))
        else:
            m = ent["key"] % np.uint64(128)
            assert np.all((m >= sc.map_lo) & (m < sc.map_hi))
        np.save(os.path.join(out_dir, f"entries_{rank}.npy"), ent)
        if rank == 0:
            np.save(os.path.join(out_dir, "hist.npy"), np.array(sorted(hist.items()), dtype=np.uint64))
            np.save(os.path.join(out_dir, "ctr.npy"), ctr)
            np.save(os.path.join(out_dir, "summ.npy"), np.array([summ[f] for f in ("total", "unique", "distinct", "missing", "edges")], dtype=np.uint64))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,k,hint", [(2, 21, 5_000_000), (3, 21, 5_000_000), (2, 17, 4_000_000), (2, 27, 4_000_000), (2, 21, 500_000)])
def test_sharded_counter_on_device(tmp_path, world, k, hint):
    from oracle import oracle as O

    O.build()
    mp.spawn(_worker, args=(world, _free_port(), k, hint, str(tmp_path)), nprocs=world, join=True)
    ref = O.OracleDB(k, 128)
    for rank in range(world):
        for b in range(2):
            ref.count_batch(_batch(rank, b), threads=8)
    _, genome = H.synth_reads(10, 150, 250_000, seed=2000)
    merged = np.concatenate([np.load(os.path.join(tmp_path, f"entries_{r}.npy")) for r in range(world)])
    merged = merged[np.argsort(merged["key"])]
    assert H.entries_equal(merged, ref.export())
    c, _ = ref.validate_sequence(genome)
    assert np.load(os.path.join(tmp_path, "ctr.npy")).tolist() == c.tolist()
    s = ref.summary(with_hist=True)
    assert np.load(os.path.join(tmp_path, "summ.npy")).tolist() == [s[f] for f in ("total", "unique", "distinct", "missing", "edges")]
    assert [tuple(x) for x in np.load(os.path.join(tmp_path, "hist.npy")).tolist()] == sorted(s["hist"].items())
    from kreeq_amd import hostdb

    got, gk, gm = hostdb.read_db(os.path.join(tmp_path, "sharded.kreeq"))
    assert (gk, gm) == (k, 128)
    assert H.entries_equal(got, ref.export())


def _worker_nccl(rank, world, port, k, hint, out_dir, chunk_bases):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        from kreeq_amd.dist import GpuEngine, ShardedCounter

        eng = GpuEngine(k, 128, 0, capacity_hint=hint)
        sc = ShardedCounter(eng, k, 128, sharded_path=True)
        sc.force_exchange = True                                # world 1: every chunk still goes through all_to_all_single (to itself)
        sc.MAX_CHUNK_BASES = chunk_bases                        # several chunks per batch: the pipeline with its lazy part sizes
        assert not sc.stage_host and eng.sharded5 == (k <= 21 and hint >= 3_000_000)
        n = 0
        for b in range(3):
            t = torch.frombuffer(bytearray(_batch(0, b, n=9000)), dtype=torch.uint8).to(dev)
            n += sc.count_batch(t)
        summ = sc.summary()
        assert n == summ["total"]                               # what the exchange delivered is what the table counted
        np.save(os.path.join(out_dir, "entries_nccl.npy"), eng.db.export())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("k,hint", [(21, 5_000_000), (27, 4_000_000)])
def test_exchange_over_rccl_world1(tmp_path, k, hint):
    """the NCCL-only branch of the exchange (part sizes taken lazily from the device bucket counts, device tensors
    straight into all_to_all_single) ran in no test -- the gloo tests stage through the host.  World size 1 over RCCL on the one
    GPU: three batches in chunks of 300 kb through emit -> count exchange -> insert of the previous chunk -> payload exchange;
    the table equals the oracle's and the per-batch conservation check (sent == received) stays silent."""
    from oracle import oracle as O

    O.build()
    mp.spawn(_worker_nccl, args=(1, _free_port(), k, hint, str(tmp_path), 300_000), nprocs=1, join=True)
    ref = O.OracleDB(k, 128)
    for b in range(3):
        ref.count_batch(_batch(0, b, n=9000), threads=8)
    assert H.entries_equal(np.load(os.path.join(tmp_path, "entries_nccl.npy")), ref.export())


def _worker_guard(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        from kreeq_amd.dist import GpuEngine, ShardedCounter

        sc = ShardedCounter(GpuEngine(21, 128, 0, capacity_hint=5_000_000), 21, 128, sharded_path=True)
        sc.force_exchange = True
        sc.MAX_MESSAGE_BYTES = 1 << 16                          # stands in for the 1 GiB the collective is verified for
        t = torch.frombuffer(bytearray(_batch(0, 0, n=9000)), dtype=torch.uint8).to(dev)
        # one chunk of 1.3 M records = 5 MB > 64 KiB: must be refused, not sent
        payload, counts, meta = sc._emit(sc.engine, t, 1, slot=0, lazy=True)
        with pytest.raises(RuntimeError, match="exceeds"):
            sc._exchange_start(payload, counts, slot=0, meta=meta)
        open(os.path.join(out_dir, "guard_ok"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_oversized_exchange_message_is_refused(tmp_path):
    """round 3: all_to_all_single of this image's RCCL delivers only the first half of a message of ~2 GiB or more
    (profiles/r03/a2a_message_size.log); the exchange refuses a message above the verified size instead of losing records"""
    mp.spawn(_worker_guard, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    assert os.path.exists(os.path.join(tmp_path, "guard_ok"))

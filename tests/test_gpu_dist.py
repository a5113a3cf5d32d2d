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

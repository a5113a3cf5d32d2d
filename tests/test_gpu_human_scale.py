"""BASELINE configs[2] shape (iid genome, 30x 150 bp reads with 0.5 % substitutions, k = 21, count + validate of the
assembly = genome with 1e-4 substitutions) at 100 Mbp -- 1/30 of the human-scale job -- against the CPU oracle on the
SAME batches: 2.6 x 10^9 read k-mers, ~3.5 x 10^8 distinct, a table with a middle split level, pending record sets.

  * one pass (ASCII batches, the whole table in one handle): every summary number, the whole coverage histogram and the
    three QV counters must be identical to the oracle's;
  * the memory-bounded mode bench.py uses at full size (the reference's map ranges, src/kreeq.cpp:59-74): the same read
    set, resident in the 2-bit packed form, counted in THREE map-range passes into a table a third of the size -- every pass
    rescans all reads, keeps the k-mers of its maps (key % 128), is summarised and validated against the assembly for its
    range, then the table is cleared -- with an arena small enough to force several table passes per range: the sums over the
    ranges must be the oracle's numbers again.

(Larger scales are covered by bench.py's closed-form checks; the oracle needs ~1 min here.)"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G, COV, L, K, ERR, ASM_ERR = 100_000_000, 30, 150, 21, 0.005, 1e-4
BATCH_READS = 5_000_000
RANGES = [(0, 43), (43, 85), (85, 128)]


@pytest.fixture(scope="module")
def job():
    import torch

    import kreeq_amd
    from kreeq_amd import synth
    from oracle import oracle as O

    if not kreeq_amd.device_available():
        pytest.fail("no gfx950 device: the product has no CPU fallback")
    O.build()
    dev = torch.device("cuda", 0)
    cores = max(1, min(128, len(os.sched_getaffinity(0))))
    n_reads = G * COV // L
    genome = synth.genome_dev(G, dev, seed=1)
    asm_codes, n_sub = synth.mutate_dev(genome, ASM_ERR, seed=3)
    assembly = synth.ascii_dev(asm_codes)
    gen = torch.Generator(device=dev)
    gen.manual_seed(2)
    hint = int(1.1 * (G + n_reads * L * ERR * K))
    gpu, cpu = kreeq_amd.KreeqDB(K, 128, capacity_hint=hint), O.OracleDB(K, 128)
    gpu.set_option("trust_capacity", 1)
    stream = torch.cuda.Stream(dev)
    gpu.set_stream(stream.cuda_stream)
    packed = []
    with torch.cuda.stream(stream):
        for lo in range(0, n_reads, BATCH_READS):
            n = min(BATCH_READS, n_reads - lo)
            batch = synth.reads_dev(genome, n, L, gen, err=ERR)
            gpu.count_batch_dev(batch.data_ptr(), batch.numel())          # stays pending: no read of the table in between
            codes, inv = synth.pack_dev(batch)
            packed.append((codes, inv, batch.numel()))
            torch.cuda.synchronize(dev)                                   # (the batch tensor is freed below)
            host = batch.cpu().numpy().tobytes()
            del batch
            cpu.count_batch(host, threads=cores)
            del host
    sc = cpu.summary(with_hist=True)
    c_cpu, _ = cpu.validate_sequence(assembly.cpu().numpy().tobytes(), threads=cores)
    cpu.close()
    return {"dev": dev, "stream": stream, "gpu": gpu, "assembly": assembly, "packed": packed, "n_reads": n_reads, "hint": hint, "sc": sc, "c_cpu": c_cpu}


def test_configs2_shape_100mbp_vs_oracle(job):
    import torch

    from oracle import oracle as O

    gpu, dev, stream, assembly, sc, c_cpu = job["gpu"], job["dev"], job["stream"], job["assembly"], job["sc"], job["c_cpu"]
    with torch.cuda.stream(stream):
        sg = gpu.summary(with_hist=True)
        info = gpu.info()
        ctr = torch.zeros(3, dtype=torch.int64, device=dev)
        gpu.lookup_sequence_dev(assembly.data_ptr(), assembly.numel(), ctr.data_ptr())
        gpu.sync()
        c_gpu = ctr.cpu().numpy().astype(np.uint64)
    assert sg["total"] == job["n_reads"] * (L - K + 1)
    assert {k: v for k, v in sg.items() if k != "hist"} == {k: v for k, v in sc.items() if k != "hist"}
    assert sg["hist"] == sc["hist"]
    assert info["slots_used"] == sc["distinct"] and info["kmers_counted"] == sc["total"]
    assert info["table_passes"] <= 2, info
    assert np.array_equal(c_gpu, c_cpu), (c_gpu, c_cpu)
    assert int(c_gpu[1]) == G - K + 1
    assert "%g" % O.qv(int(c_gpu[0]), int(c_gpu[1]), K) == "%g" % O.qv(int(c_cpu[0]), int(c_cpu[1]), K)
    # region-wise lookup of the same assembly (table passes its records through the split levels) agrees too
    gpu.set_option("lookup_path", "partitioned")
    with torch.cuda.stream(stream):
        ctr.zero_()
        gpu.lookup_sequence_dev(assembly.data_ptr(), assembly.numel(), ctr.data_ptr())
        gpu.sync()
        assert np.array_equal(ctr.cpu().numpy().astype(np.uint64), c_cpu)
    gpu.set_option("lookup_path", "auto")


@pytest.mark.parametrize("ranges,shared_hist", [(RANGES, False), ([(0, 64), (64, 128)], True)])
def test_configs2_shape_100mbp_map_range_passes_vs_oracle(job, ranges, shared_hist):
    """map-range passes over the resident packed read set, several table passes per range: three uneven ranges, and bench.py's
    own configuration -- two equal ranges with one histogram scan for both (KQ_OPT_COUNT_MAP_PASSES)"""
    import torch

    import kreeq_amd

    RANGES = ranges
    dev, stream, assembly, sc, c_cpu = job["dev"], job["stream"], job["assembly"], job["sc"], job["c_cpu"]
    job["gpu"].close()                                                    # the one-pass table is not needed any more
    db = kreeq_amd.KreeqDB(K, 128, capacity_hint=int(1.1 * job["hint"] / len(RANGES)))
    db.set_option("trust_capacity", 1)
    db.set_option("pending_bytes", (3 << 30) * 3 // len(RANGES))          # most, not all, of the records of a range: two table passes
    if shared_hist:
        db.set_option("count_map_passes", len(RANGES))
    db.set_stream(stream.cuda_stream)
    tot = {"total": 0, "unique": 0, "distinct": 0, "edges": 0}
    hist, ctr_sum, passes = {}, np.zeros(3, dtype=np.uint64), []
    with torch.cuda.stream(stream):
        for r, (lo, hi) in enumerate(RANGES):
            db.set_option("count_map_range", (lo, hi))
            before = db.info()["table_passes"]
            for codes, inv, n in job["packed"]:
                db.count_packed_dev(codes.data_ptr(), inv.data_ptr(), n)
            s = db.summary(with_hist=True)
            info = db.info()
            passes.append(info["table_passes"] - before)
            assert info["slots_used"] == s["distinct"] and info["kmers_counted"] == s["total"]
            for f in tot:
                tot[f] += s[f]
            for c, n in s["hist"].items():
                hist[c] = hist.get(c, 0) + n
            ctr = torch.zeros(3, dtype=torch.int64, device=dev)
            db.lookup_sequence_dev(assembly.data_ptr(), assembly.numel(), ctr.data_ptr(), map_lo=lo, map_hi=hi)
            db.sync()
            ctr_sum += ctr.cpu().numpy().astype(np.uint64)
            db.clear()
    assert all(p >= 2 for p in passes), passes
    assert tot == {f: sc[f] for f in tot}, (tot, sc)
    assert dict(sorted(hist.items())) == sc["hist"]
    assert np.array_equal(ctr_sum, c_cpu), (ctr_sum, c_cpu)

"""BASELINE configs[2] shape (iid genome, 30x 150 bp reads with 0.5 % substitutions, k = 21, count + validate of the
assembly = genome with 1e-4 substitutions) at 100 Mbp -- 1/30 of the human-scale job -- against the CPU oracle on the
SAME batches: 2.6 x 10^9 read k-mers, ~3.5 x 10^8 distinct, a 16.7 GB table with a middle split level, pending record
sets applied in one or two table passes.  Every summary number, the whole coverage histogram and the three QV
counters must be identical.  (Larger scales are covered by bench.py's closed-form checks; the oracle needs ~1 min here.)"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G, COV, L, K, ERR, ASM_ERR = 100_000_000, 30, 150, 21, 0.005, 1e-4
BATCH_READS = 5_000_000


def test_configs2_shape_100mbp_vs_oracle():
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
    with torch.cuda.stream(stream):
        for lo in range(0, n_reads, BATCH_READS):
            n = min(BATCH_READS, n_reads - lo)
            batch = synth.reads_dev(genome, n, L, gen, err=ERR)
            gpu.count_batch_dev(batch.data_ptr(), batch.numel())          # stays pending: no read of the table in between
            torch.cuda.synchronize(dev)                                   # (the batch tensor is freed below)
            host = batch.cpu().numpy().tobytes()
            del batch
            cpu.count_batch(host, threads=cores)
            del host
        sg = gpu.summary(with_hist=True)
        info = gpu.info()
        ctr = torch.zeros(3, dtype=torch.int64, device=dev)
        gpu.lookup_sequence_dev(assembly.data_ptr(), assembly.numel(), ctr.data_ptr())
        gpu.sync()
        c_gpu = ctr.cpu().numpy().astype(np.uint64)
    sc = cpu.summary(with_hist=True)
    assert sg["total"] == n_reads * (L - K + 1)
    assert {k: v for k, v in sg.items() if k != "hist"} == {k: v for k, v in sc.items() if k != "hist"}
    assert sg["hist"] == sc["hist"]
    assert info["slots_used"] == sc["distinct"] and info["kmers_counted"] == sc["total"]
    assert info["table_passes"] <= 2, info
    c_cpu, _ = cpu.validate_sequence(assembly.cpu().numpy().tobytes(), threads=cores)
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

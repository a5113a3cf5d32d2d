#!/usr/bin/env python3
"""BASELINE.json configs[2] (3 Gbp assembly + 30x 150 bp reads, validate) at a scale that fits one
MI355X: a G-Mbp iid genome, 30x reads with 0.5 % substitutions generated ON THE GPU (torch RNG,
fixed seeds), counted batch by batch, then the assembly (genome with 1e-4 substitutions) is validated.
With --oracle the same batches go through the CPU oracle and every printed number is compared.

  python tools/validate_scaled.py --genome-mbp 100 --oracle
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-mbp", type=float, default=100)
    ap.add_argument("--coverage", type=float, default=30)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--err", type=float, default=0.005)
    ap.add_argument("--asm-err", type=float, default=1e-4)
    ap.add_argument("-k", type=int, default=21)
    ap.add_argument("--batch-reads", type=int, default=2_000_000)
    ap.add_argument("--oracle", action="store_true")
    ap.add_argument("--pending-bytes", type=int, default=-1)
    ap.add_argument("--out", default="")
    args = ap.parse_args()

    from kreeq_amd import KreeqDB

    from kreeq_amd import synth

    dev = torch.device("cuda", 0)
    G = int(args.genome_mbp * 1e6)
    L, k = args.read_len, args.k
    n_reads = int(G * args.coverage / L)
    genome = synth.genome_dev(G, dev, seed=1)
    asm_codes, n_sub = synth.mutate_dev(genome, args.asm_err, seed=3)      # assembly = genome with substitutions
    assembly = synth.ascii_dev(asm_codes)
    del asm_codes

    # expected distinct: genome k-mers + ~k novel k-mers per read error
    hint = int(1.1 * (G + n_reads * L * args.err * k))
    db = KreeqDB(k, 128, capacity_hint=hint)
    db.set_option("trust_capacity", 1)
    db.set_option("pending_bytes", args.pending_bytes)
    stream = torch.cuda.Stream(dev)
    db.set_stream(stream.cuda_stream)

    oracle = None
    if args.oracle:
        from oracle import oracle as O
        oracle = O.OracleDB(k, 128)
        cores = min(128, len(os.sched_getaffinity(0)))

    gen = torch.Generator(device=dev)
    gen.manual_seed(2)
    t_count = t_cpu = 0.0
    done = 0
    with torch.cuda.stream(stream):
        # allocation warm-up outside the timed region: a batch of N's of the same size makes the library size its
        # partition scratch (hipMalloc of several GB takes a few hundred ms once) and inserts nothing
        nb = min(args.batch_reads, n_reads)
        dummy = torch.full((nb * (L + 1) - 1,), ord("N"), dtype=torch.uint8, device=dev)
        db.count_batch_dev(dummy.data_ptr(), dummy.numel())
        db.sync()
        del dummy
        while done < n_reads:
            n = min(args.batch_reads, n_reads - done)
            flat = synth.reads_dev(genome, n, L, gen, err=args.err)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            db.count_batch_dev(flat.data_ptr(), flat.numel())
            torch.cuda.synchronize(dev)             # the batch is consumed; its records may stay pending (no table pass yet)
            t_count += time.perf_counter() - t0
            if oracle is not None:
                host = flat.cpu().numpy().tobytes()
                t0 = time.perf_counter()
                oracle.count_batch(host, threads=cores)
                t_cpu += time.perf_counter() - t0
            done += n
            print(f"counted {done}/{n_reads} reads  gpu {t_count:.2f} s" + (f"  cpu {t_cpu:.1f} s" if oracle else ""), flush=True)
        t0 = time.perf_counter()
        db.sync()                                   # applies what is still pending
        t_count += time.perf_counter() - t0
        st = db.summary()
        ctr = torch.zeros(3, dtype=torch.int64, device=dev)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        db.lookup_sequence_dev(assembly.data_ptr(), assembly.numel(), ctr.data_ptr())
        db.sync()
        t_lookup = time.perf_counter() - t0
        c = ctr.cpu().tolist()
    n_kmers = n_reads * (L - k + 1)
    res = {"genome_bp": G, "reads": n_reads, "read_len": L, "k": k, "read_kmers": n_kmers, "assembly_substitutions": n_sub,
           "gpu": {"summary": st, "qv_counters": c, "count_s": t_count, "count_kmers_per_s": n_kmers / t_count,
                   "lookup_s": t_lookup, "lookup_kmers_per_s": c[1] / t_lookup, "info": db.info()}}
    assert st["total"] == n_kmers, (st, n_kmers)
    assert c[1] == G - k + 1
    if oracle is not None:
        so = oracle.summary()
        t0 = time.perf_counter()
        co, _ = oracle.validate_sequence(assembly.cpu().numpy().tobytes(), threads=cores)
        t_cpu_lookup = time.perf_counter() - t0
        res["cpu_oracle"] = {"summary": so, "qv_counters": co.tolist(), "count_s": t_cpu, "count_kmers_per_s": n_kmers / t_cpu,
                             "lookup_s": t_cpu_lookup, "cores": cores}
        res["identical"] = (so == st) and (co.tolist() == c)
        assert res["identical"], (so, st, co.tolist(), c)
    from oracle import oracle as O2
    res["qv_merqury"] = "%g" % O2.qv(c[0], c[1], k)
    res["qv_kreeq"] = "%g" % O2.qv(c[0] + c[2], c[1], k)
    print(json.dumps(res))
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()

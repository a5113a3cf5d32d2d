#!/usr/bin/env python3
"""bench.py -- k-mers/s of the count hot path on synthetic reads.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Default workload (N = 1) = BASELINE.json configs[2] at the largest scale whose table fits one MI355X next to the
resident read set ("human-scale": G-Mbp iid genome, 30x 150 bp reads with 0.5 % substitutions generated on the
device, k = 21, count, then validate the assembly = genome with 1e-4 substitutions).  The read set is cut into K
equal batches; a "step" = one batch through the count hot path (hashSequences + processBuffers).  The timed region
counts the K batches into ONE table that starts empty and ends with everything applied (kq_sync), inputs resident in
HBM.  value = read k-mers / wall time.  The line also carries: `validate` (assembly lookup + QV), `configs1`
(BASELINE configs[1], 1 M x 150 bp: steady state of several batches into one table and the empty-table figure that
was round 1's headline), `lookup` and `union` (driver-timed, with their own roofline fractions), `cpu_baseline`.

N > 1 runs the SAME read set sharded (BASELINE configs[3], strong scaling): every rank generates 1/N of each batch,
routes the k-mer records to the rank that owns their hash-bucket range with an RCCL all-to-all (kreeq_amd/dist.py),
and counts what it receives into its 1/N of the table; the assembly is validated by every rank against its own
buckets and the QV counters are all-reduced.  --workload cfg1 runs BASELINE configs[1] per GPU instead (clear + one
1 M-read batch per step; weak scaling for N > 1).  One JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

K = 21
N_READS = 1_000_000
READ_LEN = 150
GENOME_LEN = 5_000_000
ERR = 0.005
ASM_ERR = 1e-4
COVERAGE = 30
BYTES_PER_KMER = 35          # 1 B base + 17 B entry read + 17 B entry write
BYTES_PER_LOOKUP = 18        # 1 B base + 17 B entry read
BYTES_PER_UNION = 51         # 17 B read x 2 + 17 B write
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s


def qv(missing, total, k):
    if total == 0:
        return None
    err = 1.0 - (1.0 - missing / total) ** (1.0 / k)        # src/kreeq.cpp:36-40
    return float("inf") if err == 0 else -10.0 * math.log10(err)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["human", "cfg1"], default="human")
    ap.add_argument("--genome-mbp", type=float, default=1000.0, help="human workload: genome size (3000 = BASELINE configs[2] in full)")
    ap.add_argument("--pending-bytes", type=int, default=-1, help="KQ_OPT_PENDING_BYTES (-1 auto, 0 = one table pass per slice)")
    ap.add_argument("--slice-kmers", type=int, default=0, help="KQ_OPT_SLICE_KMERS (0 = the library's choice)")
    ap.add_argument("--reads", type=int, default=N_READS, help="cfg1: reads per GPU (default = BASELINE configs[1])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="human workload: skip the configs1 / lookup / union objects")
    ap.add_argument("--capacity", type=int, default=24_000_000,
                    help="cfg1: expected distinct k-mers per GPU (table = capacity/0.7 slots; the workload has 17.7 M)")
    ap.add_argument("--path", choices=["auto", "direct", "partitioned"], default="auto")
    ap.add_argument("-k", type=int, default=K, help="k-mer length (default 21 = BASELINE.json's metric)")
    ap.add_argument("--read-len", type=int, default=READ_LEN)
    ap.add_argument("--sharded", action="store_true", help="cfg1 with one GPU: still run the N>1 code path (owner split -> insert)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    # rehearsal of the N > 1 path on ONE GPU (never the measured configuration): KQ_BENCH_ONE_GPU=1 puts every rank on
    # cuda:0 and KQ_BENCH_BACKEND=gloo replaces RCCL, which refuses two ranks on one device
    backend = os.environ.get("KQ_BENCH_BACKEND", "nccl")
    if os.environ.get("KQ_BENCH_ONE_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or "RANK" in os.environ:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    out = run_human(args, dev, world, rank) if args.workload == "human" else run_cfg1(args, dev, world, rank, local_rank)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------------
def run_human(args, dev, world=1, rank=0):
    import torch
    import torch.distributed as dist

    from kreeq_amd import KreeqDB, synth

    k, L = args.k, args.read_len
    G = int(args.genome_mbp * 1e6)
    steps, warmup = args.steps, args.warmup
    batch_reads = max(world, int(G * COVERAGE / L) // steps) // world * world      # reads per step, all ranks together
    n_reads = batch_reads * steps
    kmers_per_step = batch_reads * (L - k + 1)
    sharded = world > 1 or args.sharded

    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    genome = synth.genome_dev(G, dev, seed=1)
    asm_codes, n_sub = synth.mutate_dev(genome, ASM_ERR, seed=3)
    assembly = synth.ascii_dev(asm_codes)
    del asm_codes
    gen = torch.Generator(device=dev)
    gen.manual_seed(2 + 1000 * rank)                  # every rank draws its own 1/world of each batch from the same genome
    batches = [synth.reads_dev(genome, batch_reads // world, L, gen, err=ERR) for _ in range(steps)]
    del genome
    torch.cuda.synchronize(dev)
    torch.cuda.empty_cache()

    # distinct k-mers: the genome's + ~k novel ones per read error (jellyfish -s style bound, 10 % margin); a rank owns
    # 1/world of the hash buckets, hence of the k-mers (5 % more room for the spread between ranks)
    hint = int(1.1 * (G + n_reads * L * ERR * k))
    counter = None
    if sharded:
        from kreeq_amd.dist import GpuEngine, ShardedCounter

        engine = GpuEngine(k, 128, dev.index, capacity_hint=hint if world == 1 else int(1.05 * hint / world))
        db = engine.db
        counter = ShardedCounter(engine, k, 128, sharded_path=True)
        counter.force_exchange = dist.is_initialized()               # world 1 under torchrun: rehearse the RCCL exchange too
    else:
        db = KreeqDB(k, 128, device=dev.index, capacity_hint=hint)
        db.set_stream(stream.cuda_stream)
    db.set_option("trust_capacity", 1)
    db.set_option("count_path", args.path)
    # pending-set arena: the library's automatic arena starts small and doubles when it fills up (a short job never pays
    # for tens of GB of hipMalloc); a job that knows it is long sizes it once, here to the automatic ceiling: a few times
    # the table, at most the HBM that is free now less 1/8 of the device -- allocated by the first warm-up step, outside the timed region
    pending = args.pending_bytes
    if pending == -1:
        from kreeq_amd.capi import device_memory
        free_b, total_b = device_memory(dev.index)
        pending = int(min(free_b - total_b // 8 if free_b > total_b // 4 else free_b // 2, 4 * db.info()["table_bytes"]))
    db.set_option("pending_bytes", pending)
    if args.slice_kmers:
        db.set_option("slice_kmers", args.slice_kmers)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def count(i):
        t = batches[i % steps]
        if counter is not None:
            counter.count_batch(t)
        else:
            db.count_batch_dev(t.data_ptr(), t.numel())

    for i in range(warmup):              # sizes the scratch and the pending-set arena, warms the code objects
        count(i)
    db.sync()
    db.clear()
    barrier()

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(steps):
        count(i)
    db.sync()                            # applies what is still pending: the table is complete when the clock stops
    ev1.record()
    barrier()
    dt = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if world > 1:
        tt = torch.tensor([dt, dev_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, dev_ms = float(tt[0]), float(tt[1])

    summ = counter.summary() if counter is not None else db.summary()
    assert summ["total"] == kmers_per_step * steps, (summ, kmers_per_step, steps)
    info = db.info()
    if world == 1:
        assert info["slots_used"] == summ["distinct"], (info, summ)

    # validate: the assembly's k-mers against the table (DBG::validateSequences); N > 1: own buckets + all-reduce
    barrier()
    tv = time.perf_counter()
    if counter is not None:
        c = counter.validate(assembly).tolist()
    else:
        ctr = torch.zeros(3, dtype=torch.int64, device=dev)
        db.lookup_sequence_dev(assembly.data_ptr(), assembly.numel(), ctr.data_ptr())
        torch.cuda.synchronize(dev)
        c = ctr.cpu().tolist()
    t_val = time.perf_counter() - tv
    assert c[1] == G - k + 1, c
    if rank != 0:
        return None

    value = kmers_per_step * steps / dt
    ms_kernel = dev_ms / steps
    achieved = kmers_per_step / world * BYTES_PER_KMER / (ms_kernel * 1e-3) / 1e9        # per GPU
    traffic = None
    tp = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tp):
        tj = json.load(open(tp))
        # measured on the same workload (possibly cut into a different number of steps: the traffic is per k-mer, the
        # table passes are the same -- one for the whole read set)
        if tj.get("genome_mbp") == int(args.genome_mbp) and tj.get("hbm_bytes_per_kmer") and world == 1:
            traffic = round(tj["hbm_bytes_per_kmer"] * kmers_per_step)
    out = {
        "metric": f"distinct+total k-mers/sec at k={k} (count path)", "value": value, "unit": "k-mers/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
        "higher_is_better": True, "scaling": "strong" if world > 1 else "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"configs[{3 if world > 1 else 2}] shape at {G / 3e9:.3f} scale: {G // 1_000_000} Mbp iid genome, {COVERAGE}x {L} bp reads "
                               f"({n_reads} reads, {ERR * 100:g} % substitutions, generated on the device), k={k}, count in {steps} batches + validate, "
                               + (f"{world} GPUs, hash-bucket sharded with RCCL all-to-all" if world > 1 else "1 GPU"),
                   "genome_bp": G, "reads": n_reads, "reads_per_step": batch_reads, "error_rate": ERR, "table_capacity_kmers": hint,
                   "table_bytes": info["table_bytes"], "table_passes": info["table_passes"], "pending_bytes": pending,
                   "count_path": args.path, "sharding": f"bucket x{world}" if sharded else "none"},
        "total_kmers_per_step": kmers_per_step, "distinct_kmers": summ["distinct"], "distinct_kmers_per_s": summ["distinct"] / dt,
        "summary": summ,
        "validate": {"assembly_kmers": c[1], "substitutions": n_sub, "missing": c[0], "edge_missing": c[2], "ms": t_val * 1e3,
                     "kmers_per_s": c[1] / t_val, "qv_merqury": qv(c[0], c[1], k), "qv_kreeq": qv(c[0] + c[2], c[1], k),
                     "roofline_frac": c[1] * BYTES_PER_LOOKUP / t_val / 1e9 / HBM_PEAK_GBS},
        "roofline": {"bound": "hbm", "kernel": "count launch set per batch: k_p1_hist+k_p1_scatter+k_lv_hist+k_lv_scatter (x levels) per slice, "
                                               "k_count_regions per table pass (+scans)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "bytes_per_kmer": BYTES_PER_KMER, "kernel_ms": ms_kernel},
    }
    if not args.no_cpu_baseline and world == 1:
        sample = min(batch_reads, 1_000_000)
        out["cpu_baseline"] = cpu_baseline(batches[0][:sample * (L + 1) - 1].cpu().numpy(), k, L)
    del db, batches, assembly
    torch.cuda.synchronize(dev)
    torch.cuda.empty_cache()
    if not args.no_extras and world == 1 and not sharded:
        out.update(extras_cfg1(dev, stream))
    return out


def extras_cfg1(dev, stream):
    """BASELINE configs[1] (1 M x 150 bp, k = 21) on the same GPU: count (steady state and empty table), lookup, union --
    each timed with HIP events on the stream the library launches on."""
    import torch

    from kreeq_amd import KreeqDB, synth

    g = synth.genome_codes(GENOME_LEN, seed=1)
    batches = [torch.from_numpy(synth.reads_batch(g, N_READS, READ_LEN, seed=s, err=ERR)).to(dev) for s in (2, 3, 4, 5)]
    asm = torch.from_numpy(synth.codes_to_ascii(synth.mutate(g, ASM_ERR, seed=3))).to(dev)
    kmers = N_READS * (READ_LEN - K + 1)

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize(dev)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize(dev)
        return a.elapsed_time(b) / reps

    res = {}
    # count, steady state: four different batches into one table, then clear
    db = KreeqDB(K, 128, device=dev.index, capacity_hint=70_000_000)
    db.set_option("trust_capacity", 1)
    db.set_stream(stream.cuda_stream)

    def four():
        db.clear()
        for t in batches:
            db.count_batch_dev(t.data_ptr(), t.numel())
        db.sync()
    ms4 = timed(four, 5)
    s4 = db.summary()
    assert s4["total"] == 4 * kmers
    # the same with one table pass per batch (what a caller that reads the table after every batch gets)
    db.set_option("pending_bytes", 0)
    ms4_each = timed(four, 5)
    del db
    # count, empty table: clear + one batch (round 1's headline configuration)
    db = KreeqDB(K, 128, device=dev.index, capacity_hint=24_000_000)
    db.set_option("trust_capacity", 1)
    db.set_stream(stream.cuda_stream)

    def one():
        db.clear()
        db.count_batch_dev(batches[0].data_ptr(), batches[0].numel())
        db.sync()
    ms1 = timed(one, 10)
    s1 = db.summary()
    assert s1["total"] == kmers
    res["configs1"] = {"workload": "synthetic 1000000 x 150 bp reads, k=21, count-only (configs[1])",
                       "steady_state": {"what": "4 different batches into one table (70 M-entry hint), one table pass", "ms_per_batch": ms4 / 4,
                                        "kmers_per_s": kmers / (ms4 / 4 * 1e-3), "roofline_frac": kmers * BYTES_PER_KMER / (ms4 / 4 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        "distinct": s4["distinct"]},
                       "steady_state_pass_per_batch": {"ms_per_batch": ms4_each / 4, "kmers_per_s": kmers / (ms4_each / 4 * 1e-3)},
                       "empty_table": {"what": "clear + one batch (round 1's headline)", "ms_per_batch": ms1, "kmers_per_s": kmers / (ms1 * 1e-3),
                                       "roofline_frac": kmers * BYTES_PER_KMER / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBS, "distinct": s1["distinct"]}}
    # lookup: the 130 M read k-mers and the 5 Mbp assembly against the 17.7 M-entry table (counters only)
    ctr = torch.zeros(3, dtype=torch.int64, device=dev)
    ms_lr = timed(lambda: db.lookup_sequence_dev(batches[0].data_ptr(), batches[0].numel(), ctr.data_ptr()), 5)
    ctr.zero_()
    db.lookup_sequence_dev(asm.data_ptr(), asm.numel(), ctr.data_ptr())
    torch.cuda.synchronize(dev)
    ca = ctr.cpu().tolist()
    pb = torch.zeros(batches[0].numel() * 16, dtype=torch.uint8, device=dev)
    ms_pb = timed(lambda: db.lookup_sequence_dev(batches[0].data_ptr(), batches[0].numel(), ctr.data_ptr(), per_base_ptr=pb.data_ptr()), 3)
    del pb
    res["lookup"] = {"workload": "130 M read k-mers against the configs[1] table (17.7 M entries), QV counters only", "ms": ms_lr,
                     "kmers_per_s": kmers / (ms_lr * 1e-3), "bytes_per_kmer": BYTES_PER_LOOKUP,
                     "roofline_frac": kmers * BYTES_PER_LOOKUP / (ms_lr * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "per_base": {"ms": ms_pb, "kmers_per_s": kmers / (ms_pb * 1e-3), "bytes_per_kmer": BYTES_PER_LOOKUP + 16,
                                  "roofline_frac": kmers * (BYTES_PER_LOOKUP + 16) / (ms_pb * 1e-3) / 1e9 / HBM_PEAK_GBS},
                     "assembly_5mbp": {"missing": ca[0], "total": ca[1], "edge_missing": ca[2], "qv_merqury": qv(ca[0], ca[1], K)}}
    # union: a second database of the same size merged into a copy of the first
    other = KreeqDB(K, 128, device=dev.index, capacity_hint=24_000_000)
    other.set_stream(stream.cuda_stream)
    other.count_batch_dev(batches[1].data_ptr(), batches[1].numel())
    other.sync()
    n_other = other.summary()["distinct"]
    times = []
    for _ in range(3):
        dst = KreeqDB(K, 128, device=dev.index, capacity_hint=40_000_000)
        dst.set_stream(stream.cuda_stream)
        dst.merge(db)
        torch.cuda.synchronize(dev)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        dst.merge(other)
        b.record()
        torch.cuda.synchronize(dev)
        times.append(a.elapsed_time(b))
        n_union = dst.summary()["distinct"]
        del dst
    ms_u = min(times[1:])
    res["union"] = {"workload": "kq_merge of a 17.7 M-entry database into a filled 17.7 M-entry one (40 M-entry table)", "ms": ms_u,
                    "entries": n_other, "entries_per_s": n_other / (ms_u * 1e-3), "bytes_per_entry": BYTES_PER_UNION,
                    "roofline_frac": n_other * BYTES_PER_UNION / (ms_u * 1e-3) / 1e9 / HBM_PEAK_GBS, "union_distinct": n_union}
    return res


# ---------------------------------------------------------------------------------------------------------------------
def run_cfg1(args, dev, world, rank, local_rank):
    """BASELINE configs[1] per GPU: clear + one batch per step (weak scaling for N > 1)."""
    import torch
    import torch.distributed as dist

    from kreeq_amd import synth
    from kreeq_amd.dist import GpuEngine, ShardedCounter

    K_, RL = args.k, args.read_len
    # synthetic input: same genome everywhere, a different read shard per rank
    genome = synth.genome_codes(GENOME_LEN, seed=1)
    reads_np = synth.reads_batch(genome, args.reads, RL, seed=2 + 1000 * rank, err=ERR)
    reads = torch.from_numpy(reads_np).to(dev)
    kmers_per_rank = args.reads * (RL - K_ + 1)

    # table sized for the records this rank will own (weak scaling: ~ one batch worth)
    # run on an explicit (non-null) stream: the handle launches on it and the HIP events below are
    # recorded on the same stream, so they bracket exactly the kernels of one step
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    engine = GpuEngine(K_, 128, local_rank, capacity_hint=args.capacity)
    engine.db.set_option("trust_capacity", 1)     # the hint is an upper bound of the distinct k-mers (jellyfish -s style)
    engine.db.set_option("count_path", args.path)
    counter = ShardedCounter(engine, K_, 128, sharded_path=args.sharded)
    counter.force_exchange = args.sharded and dist.is_initialized()      # rehearse the RCCL exchange even at world size 1

    def step():
        engine.clear()
        counter.count_batch(reads)
        engine.flush()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()

    # per-step kernel time via HIP events on the stream the kernels are launched on
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        engine.clear()
        ev[i][0].record()
        counter.count_batch(reads)
        engine.flush()
        ev[i][1].record()
    barrier()
    dt = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    # per-stage HIP-event times of one extra (untimed) step, for the roofline breakdown
    stages = None
    if world == 1 and not args.sharded and args.path != "direct":
        engine.db.set_option("profile", 1)
        step()
        stages = engine.db.profile()
        engine.db.set_option("profile", 0)

    # correctness guard inside the bench: totals must match the closed form
    summ = counter.summary()
    assert summ["total"] == kmers_per_rank * world or os.environ.get("KQ_BENCH_NOCHECK"), (summ, kmers_per_rank, world)
    if rank != 0:
        return None
    total_kmers = kmers_per_rank * world * args.steps
    value = total_kmers / dt
    achieved = kmers_per_rank * BYTES_PER_KMER / (kern_ms * 1e-3) / 1e9
    out = {
        "metric": f"distinct+total k-mers/sec at k={K_} (count path)", "value": value, "unit": "k-mers/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"synthetic {args.reads} x {RL} bp reads per GPU, k={K_}, count-only, clear + one batch per step" + (" (configs[1])" if (K_, RL, args.reads) == (K, READ_LEN, N_READS) else ""),
                   "genome_bp": GENOME_LEN, "error_rate": ERR, "table_capacity_kmers": args.capacity, "count_path": args.path, "sharding": f"bucket x{world}" if world > 1 else "none"},
        "total_kmers_per_step": kmers_per_rank * world, "distinct_kmers": summ["distinct"],
        "distinct_kmers_per_s": summ["distinct"] * args.steps / dt,
        "roofline": {"bound": "hbm", "kernel": ("count_batch launch set: k_p1_hist+k_p1_scatter+k_lv_hist+k_lv_scatter+k_count_regions (+scans)" if args.path != "direct"
                                                else "k_count_direct") if world == 1 and not args.sharded else "owner split (k_p1_*) + all_to_all + k_lv_* + k_count_regions",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": None, "bytes_per_kmer": BYTES_PER_KMER, "kernel_ms": kern_ms, "stage_ms": stages},
    }
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(reads_np, K_, RL)
    return out


def cpu_baseline(reads_np, k=K, read_len=READ_LEN):
    """The CPU restatement of the reference algorithm (oracle/, kind "port": the reference itself
    cannot be built -- gfalibs is absent) timed on this host's cores on a bounded sample of the same reads."""
    from oracle import oracle as O

    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 128))      # one job per map in loop 2: more than map_count threads cannot help
    sample_reads = min(N_READS, (len(reads_np) + 1) // (read_len + 1))
    buf = reads_np[:sample_reads * (read_len + 1) - 1].tobytes()
    db = O.OracleDB(k, 128)
    t0 = time.perf_counter()
    db.count_batch(buf, threads=cores)
    dt = time.perf_counter() - t0
    n = sample_reads * (read_len - k + 1)
    return {"value": n / dt, "unit": "k-mers/s", "cores": cores, "kind": "port",
            "sample": f"{sample_reads} of the same reads ({n} k-mers), {dt:.1f} s wall, input in memory"}


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- k-mers/s of the count hot path on synthetic reads (BASELINE.json configs[1]).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one whole count job on one batch per GPU: clear the table, then hashSequences +
processBuffers (K1+K2) over 1 M x 150 bp reads already resident in HBM (k=21).  With N>1 every
rank counts its OWN 1 M reads (weak scaling) and routes records to the owning rank with one RCCL
all-to-all (kreeq_amd/dist.py).  value = k-mer instances processed by all ranks / max-over-ranks
wall time of the K timed steps.  One JSON line on rank 0.
"""
import argparse
import json
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
BYTES_PER_KMER = 35          # 1 B base + 17 B entry read + 17 B entry write
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=N_READS, help="reads per GPU (default = BASELINE configs[1])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--capacity", type=int, default=24_000_000,
                    help="expected distinct k-mers per GPU (table = capacity/0.6 slots; the workload has 17.7 M)")
    ap.add_argument("--path", choices=["auto", "direct", "partitioned"], default="auto")
    ap.add_argument("-k", type=int, default=K, help="k-mer length (default 21 = BASELINE.json's metric)")
    ap.add_argument("--read-len", type=int, default=READ_LEN)
    ap.add_argument("--sharded", action="store_true", help="with one GPU: still run the N>1 code path (owner split -> insert)")
    args = ap.parse_args()
    K_, RL = args.k, args.read_len

    import torch
    import torch.distributed as dist

    from kreeq_amd import synth
    from kreeq_amd.dist import GpuEngine, ShardedCounter

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or "RANK" in os.environ:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

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

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()

    # per-step kernel time of the dominant kernel via HIP events on the stream it is launched on
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        engine.clear()
        ev[i][0].record()
        counter.count_batch(reads)
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
    assert summ["total"] == kmers_per_rank * world, (summ, kmers_per_rank, world)

    if rank == 0:
        total_kmers = kmers_per_rank * world * args.steps
        value = total_kmers / dt
        launch_kmers = kmers_per_rank            # k-mers one launch of the dominant kernel processes
        achieved = launch_kmers * BYTES_PER_KMER / (kern_ms * 1e-3) / 1e9
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp) and world == 1 and args.reads == N_READS:
            traffic = json.load(open(tp)).get("hbm_bytes_per_launch")
        out = {
            "metric": f"distinct+total k-mers/sec at k={K_} (count path)", "value": value, "unit": "k-mers/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"synthetic {args.reads} x {RL} bp reads per GPU, k={K_}, count-only" + (" (configs[1])" if (K_, RL, args.reads) == (K, READ_LEN, N_READS) else ""),
                       "genome_bp": GENOME_LEN, "error_rate": ERR, "table_capacity_kmers": args.capacity, "count_path": args.path, "sharding": f"bucket x{world}" if world > 1 else "none"},
            "total_kmers_per_step": kmers_per_rank * world, "distinct_kmers": summ["distinct"],
            "distinct_kmers_per_s": summ["distinct"] * args.steps / dt,
            "roofline": {"bound": "hbm", "kernel": ("count_batch launch set: k_p1_hist+k_p1_scatter+k_lv_hist+k_lv_scatter+k_count_regions (+scans)" if args.path != "direct"
                                                    else "k_count_direct") if world == 1 and not args.sharded else "owner split (k_p1_*) + all_to_all + k_lv_* + k_count_regions",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "bytes_per_kmer": BYTES_PER_KMER, "kernel_ms": kern_ms, "stage_ms": stages},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(reads_np, K_, RL)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(reads_np, k=K, read_len=READ_LEN):
    """The CPU restatement of the reference algorithm (oracle/, kind "port": the reference itself
    cannot be built -- gfalibs is absent) timed on this host's cores on the same batch."""
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

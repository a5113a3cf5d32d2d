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
TABLE_MARGIN = 1.04          # table capacity hint over the expected distinct k-mers (the table is sized hint / 0.7 slots)
TABLE_LOAD = 0.8             # load factor the human workload's table is sized for: expected distinct k-mers / slots (0.67 -> 0.8: -3 % at 3 Gbp; 0.85 another -4 %, but a region then overflows at 7 sigma)
CPU_SAMPLE_READS = 5_000_000 # reads of the cpu_baseline sample


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
    ap.add_argument("--genome-mbp", type=float, default=3000.0, help="human workload: genome size (3000 = BASELINE configs[2] in full)")
    ap.add_argument("--ranges", type=int, default=0, help="human workload on one GPU: map-range passes (0 = planned from the free HBM)")
    ap.add_argument("--ascii", action="store_true", help="human workload on one GPU: keep the reads as ASCII instead of the 2-bit packed form")
    ap.add_argument("--no-third", action="store_true", help="skip the 1000 Mbp (round-2 headline) extra of a larger run")
    ap.add_argument("--pending-bytes", type=int, default=-1, help="KQ_OPT_PENDING_BYTES (-1 auto, 0 = one table pass per slice)")
    ap.add_argument("--slice-kmers", type=int, default=0, help="KQ_OPT_SLICE_KMERS (0 = the library's choice)")
    ap.add_argument("--slice-cap", type=int, default=1 << 31, help="human workload: a batch is cut into equal slices of at most this many k-mer starts")
    ap.add_argument("--no-overlap", action="store_true", help="KQ_OPT_OVERLAP = 0")
    ap.add_argument("--alt-kernels", default="", help="measurement only: comma-separated KQ_OPT_KERNEL_SET masks, step i runs with mask[i %% n] "
                    "(previous and shipped kernels timed in one process on the same buffers: read the kernel trace, not `value`)")
    ap.add_argument("--range-by", choices=["map", "bucket"], default="map",
                    help="human workload in several passes: ranges of the reference's maps (key %% 128) or of the table's 256 hash-prefix buckets")
    ap.add_argument("--narrow-mid", type=int, default=0, help="KQ_OPT_NARROW_MID (tuning: regions per bucket from which the record split gets a middle level)")
    ap.add_argument("--no-map-pass-cache", action="store_true", help="map-range passes: every pass runs its own histogram scan (KQ_OPT_COUNT_MAP_PASSES off)")
    ap.add_argument("--table-load", type=float, default=TABLE_LOAD, help="human workload: load factor the table is sized for (expected distinct k-mers / slots)")
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

    # stdout carries ONE line, the JSON: native libraries that print to file descriptor 1 (RCCL's version banner at
    # communicator creation) are sent to stderr, the line itself goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

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
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------------
def map_ranges(n, map_count=128):
    """n contiguous ranges of the reference's maps (key % mapCount): the units of its memory-bounded mode, src/kreeq.cpp:59-74"""
    return [(map_count * i // n, map_count * (i + 1) // n) for i in range(n)]


def plan_ranges(free_b, est_distinct, n_records, scratch_b, scratch_ranged, margin=TABLE_MARGIN, max_ranges=16):
    """How many map-range passes the count job is cut into on ONE GPU (the reference picks its ranges from the free memory the
    same way, src/kreeq.cpp:59-63).  A range pass rescans every read (P1 scan + hash of all k-mers, records of 1/n of them)
    into a table of 1/n of the k-mers; what is left of the HBM is the pending-record arena, and every arena-full costs one
    stream of the table.  Cost model (constants measured on MI355X): 3.9 ps per scanned k-mer and range
    pass, table streams at 2.5 TB/s (the first pass of a range writes the table, later ones read and write it).
    -> (n_ranges, table_bytes, arena_bytes, table_passes_per_range)"""
    best = None
    for n in range(1, max_ranges + 1):
        table = int(est_distinct / n * margin / 0.7) * 16
        arena = free_b - table - (scratch_b if n == 1 else scratch_ranged) - (6 << 30)
        if arena < (4 << 30):
            continue
        passes = max(1, -(-(n_records // n * 4) // arena))
        cost = n * n_records * 3.9e-12 + n * table * (2 * passes - 1) / 2.5e12
        if best is None or cost < best[0]:
            best = (cost, n, table, arena, passes)
    if best is None:
        sys.exit("the table does not fit this GPU even in 16 map-range passes")
    return best[1:]


def run_human(args, dev, world=1, rank=0):
    import torch
    import torch.distributed as dist

    from kreeq_amd import KreeqDB, synth
    from kreeq_amd.capi import device_memory

    k, L = args.k, args.read_len
    G = int(args.genome_mbp * 1e6)
    steps, warmup = args.steps, args.warmup
    batch_reads = max(world, int(G * COVERAGE / L) // steps) // world * world      # reads per step, all ranks together
    n_reads = batch_reads * steps
    kmers_per_step = batch_reads * (L - k + 1)
    n_kmers = kmers_per_step * steps
    sharded = world > 1 or args.sharded
    packed = not sharded and not args.ascii          # single GPU: the read set is resident in the 2-bit packed form (kq_pack_bases layout)

    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    genome = synth.genome_dev(G, dev, seed=1)
    asm_codes, n_sub = synth.mutate_dev(genome, ASM_ERR, seed=3)
    assembly = synth.ascii_dev(asm_codes)
    del asm_codes
    gen = torch.Generator(device=dev)
    gen.manual_seed(2 + 1000 * rank)                  # every rank draws its own 1/world of each batch from the same genome
    batches, cpu_sample = [], None
    for i in range(steps):
        t = synth.reads_dev(genome, batch_reads // world, L, gen, err=ERR)
        if i == 0 and rank == 0 and world == 1 and not args.no_cpu_baseline:
            cpu_sample = t[:min(batch_reads, CPU_SAMPLE_READS) * (L + 1) - 1].cpu().numpy()
        if packed:
            codes, inv = synth.pack_dev(t)
            batches.append((codes, inv, t.numel()))
            del t, codes, inv
        else:
            batches.append(t)
    del genome
    torch.cuda.synchronize(dev)
    torch.cuda.empty_cache()

    # distinct k-mers: the genome's + one novel k-mer per read window with at least one substitution (jellyfish -s style
    # bound; TABLE_MARGIN on top); a rank owns 1/world of the hash buckets, hence of the k-mers (5 % more room for the spread)
    est = G + n_kmers * (1.0 - (1.0 - ERR) ** k)
    margin = 0.7 / args.table_load                    # the library sizes the table capacity_hint / 0.7 slots
    starts_per_batch = (batch_reads // world) * (L + 1)
    n_slices = max(1, -(-starts_per_batch // args.slice_cap))     # two or more slices per call: their partition stages overlap (KQ_OPT_OVERLAP), a scratch set each
    slice_kmers = args.slice_kmers or (-(-starts_per_batch // n_slices) + 64)
    # partition scratch of a slice (two sets when the slices of a call overlap: KQ_OPT_OVERLAP, not in map-range passes) +
    # high-copy side table (2.7 GB at this scale) + small buffers
    scratch1 = int(10.6 * min(slice_kmers, starts_per_batch))
    scratch_b = (2 if (n_slices > 1 or args.slice_kmers) and not args.no_overlap else 1) * scratch1 + (5 << 30)
    free_b, total_b = device_memory(dev.index)
    if sharded:
        n_ranges, passes_planned = 1, None
        hint = int(margin * est) if world == 1 else int(1.05 * margin * est / world)
        pending = args.pending_bytes
        if pending == -1:
            pending = int(max(1 << 30, free_b - hint / 0.7 * 16 - scratch_b - (8 << 30) - (2 * starts_per_batch * 6 if world > 1 else 0)))
    else:
        if args.ranges:
            n_ranges = args.ranges
            table_b = int(est / n_ranges * margin / 0.7) * 16
            arena_b = free_b - table_b - (scratch_b if n_ranges == 1 else scratch1 + (5 << 30)) - (6 << 30)
            passes_planned = max(1, -(-(n_kmers // n_ranges * 4) // max(arena_b, 1)))
        else:
            n_ranges, table_b, arena_b, passes_planned = plan_ranges(free_b, est, n_kmers, scratch_b, scratch1 + (5 << 30), margin)
        hint = int(margin * est / n_ranges)
        pending = args.pending_bytes if args.pending_bytes != -1 else int(arena_b)
    by_bucket = args.range_by == "bucket" and n_ranges > 1 and not sharded
    ranges = map_ranges(n_ranges, 256) if by_bucket else map_ranges(n_ranges)

    counter = None
    if sharded:
        from kreeq_amd.dist import GpuEngine, ShardedCounter

        engine = GpuEngine(k, 128, dev.index, capacity_hint=hint)
        db = engine.db
        counter = ShardedCounter(engine, k, 128, sharded_path=True)
        counter.force_exchange = dist.is_initialized()               # world 1 under torchrun: rehearse the RCCL exchange too
    else:
        db = KreeqDB(k, 128, device=dev.index, capacity_hint=hint)
        db.set_stream(stream.cuda_stream)
    db.set_option("trust_capacity", 1)
    db.set_option("count_path", args.path)
    # pending-set arena: sized once, here (the library's automatic arena starts small and doubles as it fills up, so that a
    # short job never pays for tens of GB of hipMalloc); allocated by the first warm-up step, outside the timed region
    db.set_option("pending_bytes", pending)
    if args.no_overlap:
        db.set_option("overlap", 0)
    if args.narrow_mid:
        db.set_option("narrow_mid", args.narrow_mid)
    if args.slice_kmers or n_slices > 1:
        db.set_option("slice_kmers", slice_kmers)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def count(i):
        t = batches[i % steps]
        if counter is not None:
            counter.count_batch(t)
        elif packed:
            db.count_packed_dev(t[0].data_ptr(), t[1].data_ptr(), t[2])
        else:
            db.count_batch_dev(t.data_ptr(), t.numel())

    def set_range(rg):
        if by_bucket:
            db.set_option("bucket_window", rg[0] | (rg[1] << 16))
        else:
            db.set_option("count_map_range", rg)

    if n_ranges > 1:
        set_range(ranges[0])
        if n_ranges in (2, 4, 8) and not args.no_map_pass_cache:
            # the batches are resident and unchanged for the whole job: the first range pass counts every slice for all ranges,
            # the later ones skip the histogram scan (KQ_OPT_COUNT_MAP_PASSES)
            db.set_option("count_map_passes", n_ranges)
    for i in range(warmup):              # sizes the scratch and the pending-set arena, warms the code objects
        count(i)
    db.sync()
    db.clear()
    if n_ranges in (2, 4, 8) and not sharded and not args.no_map_pass_cache:
        db.set_option("count_map_passes", n_ranges)      # drops the count matrices the warm-up made: the timed job makes its own
    barrier()

    # The timed region: the K batches go through the count hot path once per map range (each pass ends with everything applied:
    # kq_sync).  Between two ranges -- outside the clock, like the final validation -- the finished table is summarised and the
    # assembly is validated against it (counters accumulate over the ranges, src/kreeq.cpp:59-74), then it is cleared.
    dt = dev_ms = t_val = 0.0
    alt_kernels = [int(x) for x in args.alt_kernels.split(",") if x != ""]
    summ = {"total": 0, "unique": 0, "distinct": 0, "edges": 0}
    c = [0, 0, 0]
    table_passes, passes_before = [], db.info()["table_passes"]        # (the warm-up's pass is not the job's)
    info = None
    for r, (mlo, mhi) in enumerate(ranges):
        if n_ranges > 1:
            set_range((mlo, mhi))
        barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for i in range(steps):
            if alt_kernels:
                db.set_option("kernel_set", alt_kernels[i % len(alt_kernels)])
            count(i)
        db.sync()                        # applies what is still pending: the table is complete when the clock stops
        ev1.record()
        barrier()
        dt += time.perf_counter() - t0
        dev_ms += ev0.elapsed_time(ev1)
        s_r = counter.summary() if counter is not None else db.summary()
        info = db.info()
        if world == 1:
            assert info["slots_used"] == s_r["distinct"], (info, s_r)
        for f in summ:
            summ[f] += s_r[f]
        table_passes.append(info["table_passes"] - passes_before)
        passes_before = info["table_passes"]
        # validate: the assembly's k-mers against the table (DBG::validateSequences); N > 1: own buckets + all-reduce
        barrier()
        tv = time.perf_counter()
        if counter is not None:
            c_r = counter.validate(assembly).tolist()
        else:
            ctr = torch.zeros(3, dtype=torch.int64, device=dev)
            if by_bucket:                # the window answers for the k-mers of its buckets only
                db.lookup_sequence_dev(assembly.data_ptr(), assembly.numel(), ctr.data_ptr())
            else:
                db.lookup_sequence_dev(assembly.data_ptr(), assembly.numel(), ctr.data_ptr(), map_lo=mlo, map_hi=mhi)
            torch.cuda.synchronize(dev)
            c_r = ctr.cpu().tolist()
        t_val += time.perf_counter() - tv
        c = [a + b for a, b in zip(c, c_r)]
        if r + 1 < n_ranges:
            db.clear()
    if world > 1:
        tt = torch.tensor([dt, dev_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, dev_ms = float(tt[0]), float(tt[1])
    summ["missing"] = (4 ** k - summ["distinct"]) % (1 << 64)
    assert summ["total"] == n_kmers, (summ, kmers_per_step, steps)
    assert c[1] == G - k + 1, c
    if rank != 0:
        return None

    value = n_kmers / dt
    ms_kernel = dev_ms / steps
    achieved = kmers_per_step / world * BYTES_PER_KMER / (ms_kernel * 1e-3) / 1e9        # per GPU
    traffic = None
    tp = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tp):
        tj = json.load(open(tp))
        # measured on the same workload (possibly cut into a different number of steps: the traffic is per k-mer, the
        # range and table passes are the same)
        if tj.get("genome_mbp") == int(args.genome_mbp) and tj.get("hbm_bytes_per_kmer") and world == 1 and tj.get("ranges", 1) == n_ranges:
            traffic = round(tj["hbm_bytes_per_kmer"] * kmers_per_step)
    out = {
        "metric": f"distinct+total k-mers/sec at k={k} (count path)", "value": value, "unit": "k-mers/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
        "higher_is_better": True, "scaling": "strong" if world > 1 else "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"configs[{3 if world > 1 else 2}] shape at {G / 3e9:.3f} scale: {G // 1_000_000} Mbp iid genome, {COVERAGE}x {L} bp reads "
                               f"({n_reads} reads, {ERR * 100:g} % substitutions, generated on the device"
                               + (", resident 2-bit packed" if packed else "") + f"), k={k}, count in {steps} batches"
                               + (f" x {n_ranges} map-range passes (every pass rescans all reads; wall time of all passes counted)" if n_ranges > 1 else "")
                               + " + validate, "
                               + (f"{world} GPUs, hash-bucket sharded with RCCL all-to-all" if world > 1 else "1 GPU"),
                   "genome_bp": G, "reads": n_reads, "reads_per_step": batch_reads, "error_rate": ERR, "table_capacity_kmers": hint,
                   "map_range_passes": n_ranges, "table_passes_per_range": table_passes, "table_passes_planned": passes_planned,
                   "table_bytes": info["table_bytes"], "table_passes": sum(table_passes), "pending_bytes": pending,
                   "slice_kmers": slice_kmers if (args.slice_kmers or n_slices > 1) else None, "input": "packed" if packed else "ascii",
                   "count_path": args.path, "sharding": f"bucket x{world}" if sharded else "none"},
        "total_kmers_per_step": kmers_per_step, "distinct_kmers": summ["distinct"], "distinct_kmers_per_s": summ["distinct"] / dt,
        "summary": summ,
        "validate": {"assembly_kmers": c[1], "substitutions": n_sub, "missing": c[0], "edge_missing": c[2], "ms": t_val * 1e3,
                     "kmers_per_s": c[1] / t_val, "qv_merqury": qv(c[0], c[1], k), "qv_kreeq": qv(c[0] + c[2], c[1], k),
                     "roofline_frac": c[1] * BYTES_PER_LOOKUP / t_val / 1e9 / HBM_PEAK_GBS},
        "roofline": {"bound": "hbm", "kernel": "count launch set per batch: k_p1_hist+k_p1_scatter+k_lv_hist+k_lv_scatter (x levels) per slice"
                                               + (f" and map range (x{n_ranges})" if n_ranges > 1 else "") + ", k_count_regions per table pass (+scans)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "bytes_per_kmer": BYTES_PER_KMER, "kernel_ms": ms_kernel},
    }
    del db, batches, assembly
    torch.cuda.synchronize(dev)
    torch.cuda.empty_cache()
    if cpu_sample is not None:
        out["cpu_baseline"] = cpu_baseline(cpu_sample, k, L)
    if not args.no_extras and world == 1 and not sharded:
        out.update(extras_cfg1(dev, stream))
        if args.genome_mbp > 1000 and not args.no_third:
            out["third_scale"] = third_scale(dev, stream, args)
    return out


def third_scale(dev, stream, args):
    """The round-2 headline workload (configs[2] shape at 1/3 scale: 1000 Mbp, one map range, the whole read set pending, ONE
    table pass), kept as an extra key next to the full-size line."""
    import copy

    a = copy.copy(args)
    a.genome_mbp, a.no_extras, a.no_cpu_baseline, a.ranges, a.steps, a.warmup = 1000.0, True, True, 0, 20, 2
    o = run_human(a, dev)
    return {"workload": o["config"]["workload"], "value": o["value"], "ms_per_step": o["ms_per_step"], "kmers_per_step": o["total_kmers_per_step"],
            "roofline_frac": o["roofline"]["frac"], "map_range_passes": o["config"]["map_range_passes"], "table_passes": o["config"]["table_passes"],
            "table_bytes": o["config"]["table_bytes"], "validate_ms": o["validate"]["ms"], "qv_merqury": o["validate"]["qv_merqury"]}


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
    """The CPU restatement of the reference algorithm (oracle/, kind "port": the reference itself cannot be built -- gfalibs
    is absent) timed on this host's cores on a bounded sample of the same reads: up to CPU_SAMPLE_READS reads (a few 1e8
    k-mers, ~10 s per run), best of two runs into a fresh database each, threads = the cores this process may use (at most
    one per map in the insert loop).  What the sample cannot show: on a genome-scale read set nearly every k-mer of the
    sample is new (coverage << 1), the hash maps' slow path; the 5 Mbp genome of configs[1] gives the same code twice the
    rate."""
    from oracle import oracle as O

    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 128))      # one job per map in loop 2: more than map_count threads cannot help
    sample_reads = min(CPU_SAMPLE_READS, (len(reads_np) + 1) // (read_len + 1))
    buf = reads_np[:sample_reads * (read_len + 1) - 1].tobytes()
    n = sample_reads * (read_len - k + 1)
    runs = []
    for _ in range(2):
        db = O.OracleDB(k, 128)
        t0 = time.perf_counter()
        db.count_batch(buf, threads=cores)
        runs.append(time.perf_counter() - t0)
        assert db.summary()["total"] == n
        db.close()
    dt = min(runs)
    return {"value": n / dt, "unit": "k-mers/s", "cores": cores, "threads": cores, "kind": "port",
            "sample": f"the first {sample_reads} reads of the same read set ({n} k-mers), best of 2 runs ({runs[0]:.1f} s, {runs[1]:.1f} s wall), input in memory",
            "runs_s": runs}


if __name__ == "__main__":
    main()

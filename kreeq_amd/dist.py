"""Bucket-sharded multi-GPU counting: one process per GPU, torch.distributed (backend "nccl" is RCCL
on ROCm) over xGMI.

Two ownership rules, one per record format:

* k <= 21 on tables of >= 2048 regions (the default k at any real size) -- HASH-PREFIX BUCKETS: rank r owns the buckets
  [ceil(256 r / world), ceil(256 (r+1) / world)) of the top 8 bits of the table hash, and its table is the window of
  those buckets (KQ_OPT_BUCKET_WINDOW).  The first split of the single-GPU count path (256 buckets) is then the owner
  split as well: a rank scans its reads once, its bucket-sorted 5-byte records are already grouped by destination, one
  all-to-all(v) routes them, and the receiver's split levels take the (bucket, peer) runs as their input segments -- the
  N > 1 path runs the kernels of the single-GPU path and nothing more.  Validation: every rank evaluates the assembly
  k-mers of its buckets; database files: the entries are routed to the rank that writes their map (export_db).
* otherwise -- the reference's own bucket function: rank r owns maps [r*map_count/world, (r+1)*map_count/world) of
  key % map_count (src/graph-builder.cpp:95, src/kreeq.cpp:146), records are grouped by owner in a split level of their
  own (8-byte packed records up to k = 28, key + edge byte above); validation by map range (src/kreeq.cpp:150).

Per read batch every rank emits records from ITS reads, one all-to-all(v) per array routes them, every rank inserts what
it received into its own table.  QV counters, summary numbers and the coverage histogram are all-reduced (sum).

The compute engine is injected (`engine`): the product uses GpuEngine (C ABI, HBM-resident
tensors); the gloo/CPU tests drive the same routing code with a host engine of their own.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def owner_range(rank, world, map_count):
    """maps owned by `rank` -- must match owner_part() in csrc/kreeq_amd.hip"""
    lo = -(-rank * map_count // world)          # ceil(rank*map_count/world)
    hi = -(-(rank + 1) * map_count // world)
    return lo, hi


def owner_of(keys, world, map_count):
    m = keys % np.uint64(map_count)
    return (m * np.uint64(world) // np.uint64(map_count)).astype(np.int64)


def bucket_range(rank, world):
    """hash-prefix buckets owned by `rank` -- must match part_first_bucket() in csrc/kreeq_amd.hip"""
    return -(-rank * 256 // world), -(-(rank + 1) * 256 // world)


_MIX_MUL = 0x9E3779B97F4A7C15
_FEI_C = (0x9E3779, 0x85EBCB, 0xC2B2AF)


def table_hash(keys, k):
    """the library's table hash of canonical keys (kq_device.h), vectorised; used by tests and by the host engine of the gloo
    tests.  k <= 24: three Feistel rounds on the k-bit halves, F(R) = bits 8..8+k-1 of the 24 x 24-bit product R * C;
    k >= 25: an invertible xorshift-multiply-xorshift on the 2k key bits.  Left-aligned in 64 bits either way."""
    keys = np.asarray(keys, dtype=np.uint64)
    pad = np.uint64(64 - 2 * k)
    if k <= 24:
        m = np.uint64((1 << k) - 1)
        lo, hi = keys & m, keys >> np.uint64(k)
        for c in _FEI_C:
            t = lo ^ ((((hi * np.uint64(c)) & np.uint64(0xFFFFFFFF)) >> np.uint64(8)) & m)
            lo, hi = hi, t
        return ((lo << np.uint64(k)) | hi) << pad
    x = keys ^ (keys >> np.uint64(k))
    with np.errstate(over="ignore"):
        x = (x * np.uint64(_MIX_MUL)) << pad
    hi_mask = np.uint64((~0 << (64 - 2 * k)) & 0xFFFFFFFFFFFFFFFF)
    return x ^ ((x >> np.uint64(k)) & hi_mask)


def key_of_hash(h, k):
    """inverse of table_hash (kq_device.h key_of_hash): the canonical key of a left-aligned 64-bit table hash"""
    h = np.asarray(h, dtype=np.uint64)
    pad = np.uint64(64 - 2 * k)
    if k <= 24:
        m = np.uint64((1 << k) - 1)
        x = h >> pad
        lo, hi = x >> np.uint64(k), x & m
        for c in reversed(_FEI_C):
            t = hi ^ ((((lo * np.uint64(c)) & np.uint64(0xFFFFFFFF)) >> np.uint64(8)) & m)
            hi, lo = lo, t
        return (hi << np.uint64(k)) | lo
    inv = np.uint64(pow(_MIX_MUL, -1, 1 << 64))
    with np.errstate(over="ignore"):
        x = (h ^ ((h >> np.uint64(k)) & (~np.uint64(0) << pad))) >> pad
        x = ((x * inv) << pad) >> pad
    return x ^ (x >> np.uint64(k))


def bucket_of(keys, k):
    """hash-prefix bucket (top 8 bits of the table hash) of canonical keys"""
    return (table_hash(keys, k) >> np.uint64(56)).astype(np.int64)


class GpuEngine:
    """HBM-resident engine on one MI355X through the C ABI."""

    def __init__(self, k, map_count, device_index, capacity_hint=0):
        from .capi import KreeqDB

        self.k = k
        self.device = torch.device("cuda", device_index)
        self.db = KreeqDB(k, map_count, device=device_index, capacity_hint=capacity_hint)
        # kernels, torch ops on the exchanged tensors and the RCCL collectives must share ONE stream;
        # the legacy null stream cannot be handed to the library, so fall back to a dedicated stream
        cur = torch.cuda.current_stream(self.device)
        self.stream = cur if cur.cuda_stream != 0 else torch.cuda.Stream(self.device)
        self.db.set_stream(self.stream.cuda_stream)
        self._send = {}
        # 5-byte records (u32 + u8, grouped by owner and hash-prefix bucket) when k <= 21 and the table has the 256-bucket
        # geometry (>= 2048 regions of 2048 slots); else 8-byte packed records (k <= 28) or key + edge byte
        self.sharded5 = k <= 21 and self.db.info()["slots_total"] // 2048 >= 2048

    lazy_counts = True      # emit_partitioned(..., lazy=True) may return counts=None: the part sizes are meta.sum(dim=1) on the device

    def emit_partitioned(self, bases: torch.Tensor, n_parts: int, slot: int = 0, lazy: bool = False):
        """-> ([payload tensors grouped by owner part], per-part record counts).  `slot` selects one of
        two send buffers, so a chunk can be scanned while the previous one is still being exchanged.
        lazy (5-byte records only): nothing is read back -- the payload tensors are the whole send buffers and counts is None;
        the caller slices them once it knows the sizes (it exchanges the counts anyway: one host round trip instead of two)."""
        n = bases.numel()
        if self.sharded5:
            buf = self._send.get(("s5", slot))
            if buf is None or buf[0].numel() < n or buf[2].shape[0] != n_parts:
                buf = (torch.empty(n, dtype=torch.int32, device=self.device), torch.empty(n, dtype=torch.uint8, device=self.device),
                       torch.empty((n_parts, 256), dtype=torch.int64, device=self.device))
                self._send[("s5", slot)] = buf
            recs, aux, meta = buf
            counts = self.db.emit_sharded_dev(bases.data_ptr(), n, n_parts, recs.data_ptr(), aux.data_ptr(), recs.numel(), meta.data_ptr(), sync=not lazy)
            if lazy:
                return [recs, aux], None, meta
            tot = int(counts.sum())
            return [recs[:tot], aux[:tot]], counts.astype(np.int64), meta
        buf = self._send.get(slot)
        if buf is None or buf[0].numel() < n:
            buf = (torch.empty(n, dtype=torch.int64, device=self.device),
                   torch.empty(n, dtype=torch.uint8, device=self.device) if self.k > 28 else None)
            self._send[slot] = buf
        keys, edges = buf
        if self.k <= 28:        # packed 8-byte records: one array to exchange, atomic-free receive side
            counts = self.db.emit_packed_dev(bases.data_ptr(), n, n_parts, keys.data_ptr(), keys.numel())
            tot = int(counts.sum())
            return [keys[:tot]], counts.astype(np.int64)
        counts = self.db.emit_partitioned_dev(bases.data_ptr(), n, n_parts, keys.data_ptr(), edges.data_ptr(), keys.numel())
        tot = int(counts.sum())
        return [keys[:tot], edges[:tot]], counts.astype(np.int64)

    def set_window(self, bucket_lo, bucket_hi):
        """this rank's table holds the hash-prefix buckets [bucket_lo, bucket_hi) only (KQ_OPT_BUCKET_WINDOW)"""
        self.db.set_option("bucket_window", bucket_lo | (bucket_hi << 16))

    def count(self, bases: torch.Tensor):
        """fused K1+K2 (no record materialisation): the single-GPU path"""
        self.db.count_batch_dev(bases.data_ptr(), bases.numel())

    def insert(self, payload, meta=None):
        if meta is not None:        # 5-byte records + [n_peers, 256] bucket counts
            self.db.insert_sharded_dev(payload[0].data_ptr(), payload[1].data_ptr(), payload[0].numel(), meta.shape[0], meta.data_ptr())
        elif len(payload) == 1:
            self.db.insert_packed_dev(payload[0].data_ptr(), payload[0].numel())
        else:
            self.db.insert_records_dev(payload[0].data_ptr(), payload[1].data_ptr(), payload[0].numel())

    def lookup(self, bases: torch.Tensor, map_lo, map_hi, cov_cutoff=0):
        ctr = torch.zeros(3, dtype=torch.int64, device=self.device)
        self.db.lookup_sequence_dev(bases.data_ptr(), bases.numel(), ctr.data_ptr(), cov_cutoff, map_lo, map_hi)
        return ctr

    def summary_vector(self):
        s = self.db.summary()
        return torch.tensor([s["total"], s["unique"], s["distinct"], s["edges"]], dtype=torch.int64, device=self.device)

    def histogram(self):
        """{cov: count} of this rank's k-mers (finalHistogram, src/graph-builder.cpp:274-278)"""
        return self.db.summary(with_hist=True)["hist"]

    def export(self, map_lo, map_hi):
        return self.db.export(map_lo, map_hi)

    def sync(self):
        self.db.sync()

    def flush(self):
        self.db.flush()

    def clear(self):
        self.db.clear()


class ShardedCounter:
    def __init__(self, engine, k, map_count=128, group=None, sharded_path=False):
        """sharded_path=True runs emit -> (exchange) -> insert even on one rank (rehearsal of the N>1 code)"""
        self.engine, self.k, self.map_count, self.group = engine, k, map_count, group
        self.sharded_path = sharded_path
        self.force_exchange = False
        self.n_chunks = int(os.environ.get("KQ_EXCHANGE_CHUNKS", "2"))   # pipeline depth of the exchange
        self._recv = {}
        self.check_conservation = os.environ.get("KQ_EXCHANGE_CHECK", "1") != "0"     # per batch: all-reduced records sent == received
        self._sent = self._received = 0
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if self.world > map_count:
            raise ValueError("more ranks than maps")
        self.map_lo, self.map_hi = owner_range(self.rank, self.world, map_count)
        # gloo has no all-to-all on device tensors: payloads of a device engine are staged through the host (tests that run
        # several ranks on ONE GPU; the product's backend is nccl = RCCL, which moves device memory directly)
        self.stage_host = (dist.is_initialized() and dist.get_backend(group) == "gloo"
                           and getattr(getattr(engine, "device", None), "type", "cpu") != "cpu")
        # every rank must emit the record format every rank can insert: 5-byte records only if all tables allow them
        if self.world > 1 and hasattr(engine, "sharded5"):
            flag = torch.tensor([1 if engine.sharded5 else 0], dtype=torch.int64, device=getattr(engine, "device", torch.device("cpu")))
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            engine.sharded5 = bool(flag.item())
        # ownership by hash-prefix bucket range (module docstring): this rank's table becomes the window of its buckets
        self.bucket_mode = bool(getattr(engine, "sharded5", False))
        self.bucket_lo, self.bucket_hi = bucket_range(self.rank, self.world)
        if self.bucket_mode and self.world > 1:
            engine.set_window(self.bucket_lo, self.bucket_hi)

    def _stream_ctx(self):
        import contextlib

        st = getattr(self.engine, "stream", None)
        return torch.cuda.stream(st) if st is not None else contextlib.nullcontext()

    def count_batch(self, bases: torch.Tensor):
        """bases: this rank's read batch (uint8 tensor on the engine's device). Returns #records received."""
        with self._stream_ctx():
            return self._count_batch(bases)

    def _cut_points(self, bases: torch.Tensor, n_chunks: int):
        """chunk boundaries at read separators (a non-ACGT byte), so that no k-mer is cut"""
        n = bases.numel()
        cuts = [0]
        for i in range(1, n_chunks):
            lo = max(cuts[-1], n * i // n_chunks)
            win = bases[lo:min(n, lo + (1 << 16))]
            is_base = torch.zeros_like(win, dtype=torch.bool)
            for c in b"ACGTacgt":
                is_base |= win == c
            sep = (~is_base).nonzero()
            # no separator nearby: this chunk stays empty and the next one is longer.  The number of chunks
            # is ALWAYS n_chunks, so every rank issues the same number of collectives without having to agree
            cuts.append(lo + int(sep[0]) + 1 if sep.numel() else cuts[-1])
        cuts.append(n)
        return list(zip(cuts[:-1], cuts[1:]))

    def _recv_buffer(self, slot, j, n, like):
        """persistent receive arrays (two slots: chunk i is received while chunk i-1 is inserted), grown geometrically"""
        key = (slot, j)
        buf = self._recv.get(key)
        if buf is None or buf.numel() < n or buf.dtype != like.dtype:
            buf = torch.empty(max(n + n // 8, 1), dtype=like.dtype, device=like.device)
            self._recv[key] = buf
        return buf[:n]

    @staticmethod
    def _emit(engine, bases, world, slot=0, lazy=False):
        """-> (payload tensors, per-part counts, meta): meta = per-(part, bucket) counts of the 5-byte format, else None"""
        if lazy and getattr(engine, "lazy_counts", False) and getattr(engine, "sharded5", False):
            return engine.emit_partitioned(bases, world, slot=slot, lazy=True)
        res = engine.emit_partitioned(bases, world, slot=slot)
        return res if len(res) == 3 else (res[0], res[1], None)

    def _counts_begin(self, send_counts, meta, slot):
        """enqueue the exchange of the part sizes (tiny all-to-all) and their copy to page-locked host memory; nothing waits
        here -- the caller enqueues more GPU work (the insert of the previous chunk) before it reads them (_counts_end), so
        that the host round trip all_to_all_single's host-side split sizes require never leaves the GPU idle"""
        dev = meta.device if send_counts is None else None
        if send_counts is None:
            sc = meta.sum(dim=1)                                         # lazy emit: the part sizes are the row sums of the bucket counts, on the device
        else:
            dev = getattr(self.engine, "device", torch.device("cpu"))
            sc = torch.from_numpy(send_counts).to(dev, non_blocking=True)
        rc = torch.empty_like(sc)
        dist.all_to_all_single(rc, sc, group=self.group)                 # how many records each peer sends me
        key = ("counts", slot)
        host = self._recv.get(key)
        if host is None or host.shape[1] != sc.numel():
            host = torch.empty((2, sc.numel()), dtype=torch.int64).pin_memory() if dev.type == "cuda" else torch.empty((2, sc.numel()), dtype=torch.int64)
            self._recv[key] = host
        host.copy_(torch.stack([sc, rc]), non_blocking=True)
        ev = None
        if dev.type == "cuda":
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
        return host, ev

    @staticmethod
    def _counts_end(token):
        host, ev = token
        if ev is not None:
            ev.synchronize()
        both = host.numpy()
        return both[0].copy(), both[1].copy()

    def _exchange_start(self, payload, send_counts, slot=0, meta=None, before_wait=None):
        """counts first (tiny; its result is needed on the host because all_to_all_single takes host split sizes), then
        one asynchronous all-to-all(v) per payload array into persistent receive buffers.  `before_wait()` runs between the
        enqueue of the count exchange and the host's read of its result."""
        if send_counts is None and self.stage_host:                      # lazy emit: the part sizes are still on the device
            send_counts = meta.sum(dim=1).cpu().numpy()
            payload = [t[:int(send_counts.sum())] for t in payload]
        if self.stage_host:
            if before_wait is not None:
                before_wait()
            return self._exchange_via_host(payload, send_counts, slot, meta)
        token = self._counts_begin(send_counts, meta, slot)
        if before_wait is not None:
            before_wait()
        send_counts, recv_counts = self._counts_end(token)
        payload = [t[:int(send_counts.sum())] for t in payload]
        # conservation of the exchange: what the ranks send is what they receive
        if self.check_conservation:
            self._sent += int(send_counts.sum())
            self._received += int(recv_counts.sum())
        esz = max(t.element_size() for t in payload)
        for c in list(send_counts) + list(recv_counts):                  # one message = one (source, destination) part of one array
            if int(c) * esz > self.MAX_MESSAGE_BYTES:
                raise RuntimeError(f"exchange message of {int(c)} records exceeds {self.MAX_MESSAGE_BYTES} bytes: lower KQ_EXCHANGE_MAX_BASES")
        n_recv = int(recv_counts.sum())
        received, works = [], []
        for j, t in enumerate(payload):
            r = self._recv_buffer(slot, j, n_recv, t)
            works.append(dist.all_to_all_single(r, t, output_split_sizes=recv_counts.tolist(), input_split_sizes=send_counts.tolist(),
                                                group=self.group, async_op=True))
            received.append(r)
        rmeta = None
        if meta is not None:                                             # row p of the result = peer p's 256 bucket counts for me
            rmeta = self._recv_buffer(slot, "meta", meta.numel(), meta).view_as(meta)
            works.append(dist.all_to_all_single(rmeta, meta, group=self.group, async_op=True))
        return received, works, n_recv, rmeta

    def _exchange_via_host(self, payload, send_counts, slot, meta):
        """the same exchange with every array staged through host memory (gloo with a device engine, see __init__)"""
        dev = payload[0].device
        torch.cuda.current_stream(dev).synchronize()
        sc = torch.from_numpy(send_counts)
        rc = torch.empty_like(sc)
        dist.all_to_all_single(rc, sc, group=self.group)
        recv_counts = rc.numpy()
        n_recv = int(recv_counts.sum())
        if self.check_conservation:
            self._sent += int(send_counts.sum())
            self._received += n_recv
        received = []
        for j, t in enumerate(payload):
            r = torch.empty(n_recv, dtype=t.dtype)
            dist.all_to_all_single(r, t.cpu(), output_split_sizes=recv_counts.tolist(), input_split_sizes=send_counts.tolist(), group=self.group)
            rd = self._recv_buffer(slot, j, n_recv, t)
            rd.copy_(r)
            received.append(rd)
        rmeta = None
        if meta is not None:
            rm = torch.empty(meta.shape, dtype=meta.dtype)
            dist.all_to_all_single(rm, meta.cpu(), group=self.group)
            rmeta = self._recv_buffer(slot, "meta", meta.numel(), meta).view_as(meta)
            rmeta.copy_(rm)
        return received, [], n_recv, rmeta

    def _count_batch(self, bases: torch.Tensor):
        if self.world == 1 and hasattr(self.engine, "count") and not self.sharded_path:
            self.engine.count(bases)
            return None
        if self.world == 1 and not self.force_exchange:
            payload, send_counts, meta = self._emit(self.engine, bases, self.world)
            if meta is not None:
                self.engine.insert(payload, meta)
            else:
                self.engine.insert(payload)
            return int(send_counts.sum())
        # pipeline over chunks: the all-to-all of chunk i runs while chunk i-1 is inserted and chunk
        # i+1 is scanned (xGMI is point-to-point: the exchange costs about as much as the compute).
        # A chunk stays below 2^28 bases (1 GB of u32 records per all-to-all: messages beyond 4 GB lost records; on one GPU
        # 2^28 measured 31.4 ms per 1.3e9 k-mers, 2^27 37.5, 2^29 51.2); every rank must issue the same number of
        # collectives, so the chunk count is the maximum over the ranks (one small all-reduce per batch)
        n_total, pending = 0, None
        rec_bytes = 4 if getattr(self.engine, "sharded5", False) else 8       # widest array of a record: u32 (+ a byte array), or u64
        n_chunks = max(self.n_chunks, -(-bases.numel() // min(self.MAX_CHUNK_BASES, self.MAX_MESSAGE_BYTES // rec_bytes)))
        if self.world > 1:
            nc = torch.tensor([n_chunks], dtype=torch.int64, device=bases.device)
            dist.all_reduce(nc, op=dist.ReduceOp.MAX, group=self.group)
            n_chunks = int(nc.item())
        chunks = self._cut_points(bases, n_chunks)               # exactly n_chunks (possibly empty) chunks on every rank
        for i, (lo, hi) in enumerate(chunks):
            # send buffer i % 2: the exchange of chunk i-2 was waited for before chunk i-1 was started.
            # Order of the enqueues: scan of chunk i, exchange of its part sizes, INSERT of chunk i-1 -- and only then does the
            # host read the part sizes (the split sizes of the payload exchange): while it waits, the GPU has the scan and the
            # insert to run, so the round trip costs no GPU time (round 2: it sat between the scan and everything else)
            payload, send_counts, meta = self._emit(self.engine, bases[lo:hi], self.world, slot=i % 2, lazy=True)
            prev, pending = pending, None
            started = self._exchange_start(payload, send_counts, slot=i % 2, meta=meta,
                                           before_wait=(lambda p=prev: self._insert_received(p)) if prev is not None else None)
            pending = started
            n_total += started[2]
        if pending is not None:
            self._insert_received(pending)
        if self.check_conservation and self.world > 1:
            t = torch.tensor([self._sent, self._received], dtype=torch.int64, device=bases.device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            sent, received = (int(x) for x in t.cpu())
            if sent != received:
                raise RuntimeError(f"record exchange lost records: {sent} sent, {received} received")
        return n_total

    def _insert_received(self, pending):
        received, works, _, rmeta = pending
        for w in works:
            w.wait()
        if rmeta is not None:
            self.engine.insert(received, rmeta)
        else:
            self.engine.insert(received)

    def validate(self, bases: torch.Tensor, cov_cutoff=0):
        """every rank passes the SAME assembly sequence; returns the global (missing, total, edgeMissing)"""
        with self._stream_ctx():
            if self.bucket_mode:          # the window answers for the k-mers of its buckets only
                ctr = self.engine.lookup(bases, 0, self.map_count, cov_cutoff)
            else:
                ctr = self.engine.lookup(bases, self.map_lo, self.map_hi, cov_cutoff)
            if self.world > 1:
                dist.all_reduce(ctr, op=dist.ReduceOp.SUM, group=self.group)
            return ctr.cpu().numpy().astype(np.uint64)

    # A chunk holds at most 2^28 bases WHATEVER the environment says, and at most 2^30 bytes of its widest record
    # array: one (source, destination) message then stays within 1 GiB.  Round 2 saw records lost "beyond 4 GB" and never
    # found out why.  Round 3 measured it (tools/bench_extra/a2a_4gb.py, profiles/r03/a2a_message_size.log): with this image's
    # RCCL (2.26.6, the one torch bundles) all_to_all_single delivers a message of 1 GiB intact and only the FIRST HALF of a
    # message of 2 GiB - 4 KiB or more (int32, uint8 and int64 alike; the rest of the receive buffer is left untouched) --
    # a defect below this code, not in its offsets.  So _exchange_start refuses any message above MAX_MESSAGE_BYTES (the largest
    # size verified) instead of trusting the collective, and the per-batch conservation check would catch a loss anyway.
    MAX_CHUNK_BASES = min(1 << 28, max(1 << 16, int(os.environ.get("KQ_EXCHANGE_MAX_BASES", str(1 << 28)))))
    MAX_MESSAGE_BYTES = 1 << 30
    HIST_DENSE = 4096        # coverages below this travel as one dense all-reduce; the few above are gathered as pairs

    def histogram(self):
        """the coverage histogram of the whole database {cov: count}: all-reduce(sum) of the per-rank histograms
        (the shards hold disjoint k-mers): the all-reduce for the final histogram that BASELINE.json's north_star names"""
        with self._stream_ctx():
            local = self.engine.histogram()
            if self.world == 1:
                return dict(sorted(local.items()))
            dev = getattr(self.engine, "device", torch.device("cpu"))
            dense = torch.zeros(self.HIST_DENSE, dtype=torch.int64, device=dev)
            small = [(c, n) for c, n in local.items() if c < self.HIST_DENSE]
            if small:
                idx = torch.tensor([c for c, _ in small], dtype=torch.int64, device=dev)
                dense[idx] = torch.tensor([n for _, n in small], dtype=torch.int64, device=dev)
            dist.all_reduce(dense, op=dist.ReduceOp.SUM, group=self.group)
            big = [(c, n) for c, n in local.items() if c >= self.HIST_DENSE]
            gathered = [None] * self.world
            dist.all_gather_object(gathered, big, group=self.group)
            out = {int(c): int(n) for c, n in enumerate(dense.cpu().tolist()) if n}
            for part in gathered:
                for c, n in part:
                    out[int(c)] = out.get(int(c), 0) + int(n)
            return dict(sorted(out.items()))

    def export_db(self, db_dir):
        """ONE .kreeq database from the shards: every rank writes the map files it owns (.map.<m>.bin, m in its
        range), the high-copy k-mers are gathered on rank 0, which writes .map.hc.bin and .index -- the reference's HPC
        flow (separate databases, then `union`; README.md:31-39) without the union: the shards are bucket-disjoint.
        All ranks must see the same directory (one node)."""
        from . import hostdb

        if self.bucket_mode and self.world > 1:
            ent = self._route_entries_to_map_owners()
        else:
            ent = self.engine.export(self.map_lo, self.map_hi)
        hc = hostdb.write_maps(db_dir, self.map_count, self.map_lo, self.map_hi, ent)
        if self.world > 1:
            gathered = [None] * self.world if self.rank == 0 else None
            dist.gather_object(hc, gathered, dst=0, group=self.group)
            if self.rank == 0:
                hc = np.concatenate(gathered) if gathered else hc
        if self.rank == 0:
            hostdb.write_finish(db_dir, self.k, self.map_count, hc)
        if self.world > 1:
            dist.barrier(group=self.group)
        return len(ent)

    def _route_entries_to_map_owners(self):
        """bucket ownership: a shard holds k-mers of every map.  Every rank exports, per destination, the entries of the
        maps that destination writes; one all-to-all(v) of the raw entries (48 B each) brings every map to its writer."""
        from .capi import ENTRY_DTYPE

        parts = [np.ascontiguousarray(self.engine.export(*owner_range(d, self.world, self.map_count)), dtype=ENTRY_DTYPE) for d in range(self.world)]
        dev = torch.device("cpu") if self.stage_host else getattr(self.engine, "device", torch.device("cpu"))
        send_counts = torch.tensor([p.nbytes for p in parts], dtype=torch.int64, device=dev)
        recv_counts = torch.empty_like(send_counts)
        dist.all_to_all_single(recv_counts, send_counts, group=self.group)
        rc = recv_counts.cpu().tolist()
        raw = np.concatenate([p.view(np.uint8).reshape(-1) for p in parts]) if parts else np.zeros(0, np.uint8)
        send = torch.from_numpy(raw).to(dev)
        recv = torch.empty(int(sum(rc)), dtype=torch.uint8, device=dev)
        dist.all_to_all_single(recv, send, output_split_sizes=rc, input_split_sizes=[p.nbytes for p in parts], group=self.group)
        return np.frombuffer(recv.cpu().numpy().tobytes(), dtype=ENTRY_DTYPE)

    def summary(self):
        with self._stream_ctx():
            v = self.engine.summary_vector()
            if self.world > 1:
                dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.group)
            t = v.cpu().tolist()
        space = (1 << (2 * self.k)) if self.k < 32 else 0
        return {"total": t[0], "unique": t[1], "distinct": t[2], "missing": (space - t[2]) % (1 << 64), "edges": t[3]}

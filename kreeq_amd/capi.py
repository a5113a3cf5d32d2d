"""ctypes binding of the C ABI (include/kreeq_amd.h).  Thin: argument marshalling and error
translation only -- all compute happens in libkreeq_amd.so on the GPU.  There is no fallback:
if the library is missing or no gfx950 device is visible, calls raise."""
import ctypes as C
import os

import numpy as np

from . import build as _build

ENTRY_DTYPE = np.dtype([("key", "<u8"), ("fw", "<u4", 4), ("bw", "<u4", 4), ("cov", "<u4"), ("hc", "<u4")])
DBGBASE_DTYPE = np.dtype([("fw", "<u4"), ("bw", "<u4"), ("cov", "<u4"), ("isFw", "u1"), ("pad", "u1", 3)])

# every symbol include/kreeq_amd.h declares
SYMBOLS = ["kq_create", "kq_destroy", "kq_clear", "kq_set_option", "kq_get_profile", "kq_set_stream", "kq_get_stream", "kq_sync", "kq_flush", "kq_get_info", "kq_last_error",
           "kq_abi_version", "kq_device_available", "kq_device_memory", "kq_count_batch", "kq_count_batch_dev", "kq_host_alloc", "kq_host_free", "kq_count_batch_async", "kq_host_wait", "kq_pack_bases", "kq_count_packed_dev", "kq_count_packed_async", "kq_emit_records",
           "kq_emit_partitioned_dev", "kq_emit_packed_dev", "kq_insert_packed_dev", "kq_emit_sharded_dev", "kq_insert_sharded_dev", "kq_insert_records", "kq_insert_records_dev", "kq_summary", "kq_histogram",
           "kq_lookup_sequence", "kq_lookup_sequence_dev", "kq_lookup_keys", "kq_branch_scan", "kq_merge", "kq_import", "kq_export"]


class KqError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"kreeq_amd error {code}: {msg}")
        self.code = code


class Stats(C.Structure):
    _fields_ = [("total", C.c_uint64), ("unique", C.c_uint64), ("distinct", C.c_uint64), ("missing", C.c_uint64),
                ("edges", C.c_uint64)]


class Info(C.Structure):
    _fields_ = [("kmers_counted", C.c_uint64), ("slots_used", C.c_uint64), ("slots_total", C.c_uint64), ("hc_used", C.c_uint64),
                ("hc_total", C.c_uint64), ("table_bytes", C.c_uint64), ("table_passes", C.c_uint64)]


_lib = None


def lib_path():
    return _build.LIB


def _preload_process_hip_runtime():
    """One process must hold ONE HIP/HSA runtime.  PyTorch-ROCm wheels bundle their own
    libamdhip64.so.7; if /opt/rocm's copy gets loaded first (through this library's NEEDED entry)
    and torch's afterwards, the second runtime sees no GPU.  So when torch is installed, load its
    copy first: this library's NEEDED soname then resolves to it, and torch later reuses it."""
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if not spec or not spec.origin:
        return
    d = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        p = os.path.join(d, name)
        if os.path.exists(p):
            try:
                C.CDLL(p, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def load():
    """Loads libkreeq_amd.so (must have been built: __graft_entry__.build() / kreeq_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("KQ_LIB") or _build.LIB          # KQ_LIB: a tuning variant built by kreeq_amd.build.build_lib(out=...)
    if not os.path.exists(path):
        raise KqError(-2, f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(the product has no CPU fallback)")
    _preload_process_hip_runtime()
    L = C.CDLL(path)
    vp, u64, u32, u16, ci = C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint16, C.c_int
    L.kq_create.argtypes = [C.POINTER(vp), ci, ci, ci, u64]
    L.kq_destroy.argtypes = [vp]
    L.kq_destroy.restype = None
    L.kq_clear.argtypes = [vp]
    L.kq_set_stream.argtypes = [vp, vp]
    L.kq_set_option.argtypes = [vp, ci, C.c_int64]
    L.kq_get_profile.argtypes = [vp, C.c_char_p, u64]
    L.kq_get_stream.argtypes = [vp]
    L.kq_get_stream.restype = vp
    L.kq_sync.argtypes = [vp]
    L.kq_flush.argtypes = [vp]
    L.kq_get_info.argtypes = [vp, C.POINTER(Info)]
    L.kq_last_error.restype = C.c_char_p
    L.kq_device_memory.argtypes = [ci, C.POINTER(u64), C.POINTER(u64)]
    L.kq_count_batch.argtypes = [vp, vp, u64]
    L.kq_count_batch_dev.argtypes = [vp, vp, u64]
    L.kq_host_alloc.argtypes = [u64]
    L.kq_host_alloc.restype = vp
    L.kq_host_free.argtypes = [vp]
    L.kq_host_free.restype = None
    L.kq_count_batch_async.argtypes = [vp, vp, u64, C.POINTER(u64)]
    L.kq_host_wait.argtypes = [vp, u64]
    L.kq_pack_bases.argtypes = [vp, u64, vp, vp]
    L.kq_pack_bases.restype = None
    L.kq_count_packed_dev.argtypes = [vp, vp, vp, u64]
    L.kq_count_packed_async.argtypes = [vp, vp, vp, u64, C.POINTER(u64)]
    L.kq_emit_records.argtypes = [vp, vp, u64, vp, vp, u64, C.POINTER(u64)]
    L.kq_emit_partitioned_dev.argtypes = [vp, vp, u64, ci, vp, vp, u64, vp]
    L.kq_emit_packed_dev.argtypes = [vp, vp, u64, ci, vp, u64, vp]
    L.kq_insert_packed_dev.argtypes = [vp, vp, u64]
    L.kq_emit_sharded_dev.argtypes = [vp, vp, u64, ci, vp, vp, u64, vp, vp]
    L.kq_insert_sharded_dev.argtypes = [vp, vp, vp, u64, ci, vp]
    L.kq_insert_records.argtypes = [vp, vp, vp, u64]
    L.kq_insert_records_dev.argtypes = [vp, vp, vp, u64]
    L.kq_summary.argtypes = [vp, C.POINTER(Stats)]
    L.kq_histogram.argtypes = [vp, vp, vp, u64, C.POINTER(u64)]
    L.kq_lookup_sequence.argtypes = [vp, vp, u64, u32, u16, u16, vp, vp]
    L.kq_lookup_sequence_dev.argtypes = [vp, vp, u64, u32, u16, u16, vp, vp]
    L.kq_lookup_keys.argtypes = [vp, vp, u64, vp]
    L.kq_branch_scan.argtypes = [vp, vp, u64, u32, vp]
    L.kq_merge.argtypes = [vp, vp]
    L.kq_import.argtypes = [vp, vp, u64]
    L.kq_export.argtypes = [vp, u16, u16, vp, u64, C.POINTER(u64)]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise KqError(rc, load().kq_last_error().decode(errors="replace"))


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def pack_bases(bases: bytes):
    """-> (codes u32[ceil(n/16)], inv u16[ceil(n/16)]): the 2-bit packed form of a batch (kq_pack_bases; host only)"""
    buf = np.frombuffer(bases, dtype=np.uint8)
    units = (len(buf) + 15) // 16
    codes, inv = np.zeros(max(units, 1), dtype=np.uint32), np.zeros(max(units, 1), dtype=np.uint16)
    load().kq_pack_bases(_p(buf) if len(buf) else None, len(buf), _p(codes), _p(inv))
    return codes[:units], inv[:units]


def device_available():
    return bool(load().kq_device_available())


def device_memory(device=0):
    """(free, total) HBM bytes"""
    f, t = C.c_uint64(0), C.c_uint64(0)
    _check(load().kq_device_memory(device, C.byref(f), C.byref(t)))
    return f.value, t.value


class KreeqDB:
    """One k-mer database resident on one GPU (mirrors the reference's DBG object for the hot path)."""

    def __init__(self, k=21, map_count=128, device=0, capacity_hint=0):
        self.k, self.map_count, self.device = k, map_count, device
        self._h = C.c_void_p()
        _check(load().kq_create(C.byref(self._h), device, k, map_count, capacity_hint))

    def close(self, _load=load):                 # bound at definition: module globals may be gone at interpreter exit
        if getattr(self, "_h", None):
            _load().kq_destroy(self._h)
            self._h = None

    __del__ = close

    @property
    def handle(self):
        return self._h

    # -- stream / state
    def set_stream(self, hip_stream_ptr):
        _check(load().kq_set_stream(self._h, C.c_void_p(hip_stream_ptr)))

    def sync(self):
        _check(load().kq_sync(self._h))

    def flush(self):
        """apply pending record sets now (asynchronous)"""
        _check(load().kq_flush(self._h))

    def set_option(self, option, value):
        """option: 'trust_capacity' | 'count_path' ('auto'|'direct'|'partitioned') | 'slice_kmers' | 'count_map_range' ((lo, hi))"""
        opt = {"trust_capacity": 1, "count_path": 2, "slice_kmers": 3, "count_map_range": 4, "profile": 5, "lookup_path": 6, "merge_path": 7, "narrow_mid": 8, "pending_bytes": 9, "bucket_window": 10, "overlap": 11, "count_map_passes": 12, "kernel_set": 13, "test_fail_plan": 100}[option]
        if option == "count_map_range":
            value = int(value[0]) | (int(value[1]) << 16)
        if option in ("count_path", "lookup_path", "merge_path"):
            value = {"auto": 0, "direct": 1, "partitioned": 2}[value]
        _check(load().kq_set_option(self._h, opt, int(value)))

    def clear(self):
        _check(load().kq_clear(self._h))

    def profile(self):
        """{stage: ms} of the last partitioned count (needs set_option('profile', 1))"""
        buf = C.create_string_buffer(1024)
        _check(load().kq_get_profile(self._h, buf, 1024))
        return {k: float(v) for k, v in (kv.split("=") for kv in buf.value.decode().split(";") if kv)}

    def info(self):
        i = Info()
        _check(load().kq_get_info(self._h, C.byref(i)))
        return {f: getattr(i, f) for f, _ in Info._fields_}

    # -- count
    def count_batch(self, bases: bytes):
        buf = np.frombuffer(bases, dtype=np.uint8)
        _check(load().kq_count_batch(self._h, _p(buf), len(buf)))

    def count_batch_async(self, host_ptr, n):
        """pipelined ingest: enqueue copy + count of a host batch (pinned memory from host_alloc); returns the ticket to wait
        on before the buffer is refilled"""
        t = C.c_uint64(0)
        _check(load().kq_count_batch_async(self._h, C.c_void_p(host_ptr), n, C.byref(t)))
        return t.value

    def count_packed_dev(self, codes_ptr, inv_ptr, n_bases):
        _check(load().kq_count_packed_dev(self._h, C.c_void_p(codes_ptr), C.c_void_p(inv_ptr), n_bases))

    def count_packed_async(self, codes_ptr, inv_ptr, n_bases):
        """pipelined ingest of a 2-bit packed batch (pack_bases); returns the ticket to wait on before the arrays are refilled"""
        t = C.c_uint64(0)
        _check(load().kq_count_packed_async(self._h, C.c_void_p(codes_ptr), C.c_void_p(inv_ptr), n_bases, C.byref(t)))
        return t.value

    def host_wait(self, ticket):
        _check(load().kq_host_wait(self._h, ticket))

    def count_batch_dev(self, ptr, n):
        _check(load().kq_count_batch_dev(self._h, C.c_void_p(ptr), n))

    def emit_records(self, bases: bytes):
        buf = np.frombuffer(bases, dtype=np.uint8)
        n = C.c_uint64(0)
        _check(load().kq_emit_records(self._h, _p(buf), len(buf), None, None, 0, C.byref(n)))
        keys = np.empty(n.value, dtype=np.uint64)
        edges = np.empty(n.value, dtype=np.uint8)
        if n.value:
            _check(load().kq_emit_records(self._h, _p(buf), len(buf), _p(keys), _p(edges), n.value, C.byref(n)))
        return keys, edges

    def emit_partitioned_dev(self, bases_ptr, n, n_parts, keys_ptr, edges_ptr, cap):
        counts = np.zeros(n_parts, dtype=np.uint64)
        _check(load().kq_emit_partitioned_dev(self._h, C.c_void_p(bases_ptr), n, n_parts, C.c_void_p(keys_ptr), C.c_void_p(edges_ptr),
                                              cap, _p(counts)))
        return counts

    def emit_packed_dev(self, bases_ptr, n, n_parts, recs_ptr, cap):
        counts = np.zeros(n_parts, dtype=np.uint64)
        _check(load().kq_emit_packed_dev(self._h, C.c_void_p(bases_ptr), n, n_parts, C.c_void_p(recs_ptr), cap, _p(counts)))
        return counts

    def insert_packed_dev(self, recs_ptr, n):
        _check(load().kq_insert_packed_dev(self._h, C.c_void_p(recs_ptr), n))

    def emit_sharded_dev(self, bases_ptr, n, n_parts, recs_ptr, aux_ptr, cap, bucket_counts_ptr, sync=True):
        """sync=False: only enqueues and returns None (the part sizes are the row sums of the device bucket counts)"""
        counts = np.zeros(n_parts, dtype=np.uint64) if sync else None
        _check(load().kq_emit_sharded_dev(self._h, C.c_void_p(bases_ptr), n, n_parts, C.c_void_p(recs_ptr), C.c_void_p(aux_ptr), cap,
                                          C.c_void_p(bucket_counts_ptr), _p(counts)))
        return counts

    def insert_sharded_dev(self, recs_ptr, aux_ptr, n, n_peers, bucket_counts_ptr):
        _check(load().kq_insert_sharded_dev(self._h, C.c_void_p(recs_ptr), C.c_void_p(aux_ptr), n, n_peers, C.c_void_p(bucket_counts_ptr)))

    def insert_records(self, keys, edges):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        edges = np.ascontiguousarray(edges, dtype=np.uint8)
        assert len(keys) == len(edges)
        _check(load().kq_insert_records(self._h, _p(keys), _p(edges), len(keys)))

    def insert_records_dev(self, keys_ptr, edges_ptr, n):
        _check(load().kq_insert_records_dev(self._h, C.c_void_p(keys_ptr), C.c_void_p(edges_ptr), n))

    # -- summary
    def summary(self, with_hist=False):
        st = Stats()
        _check(load().kq_summary(self._h, C.byref(st)))
        d = {f: getattr(st, f) for f, _ in Stats._fields_}
        if with_hist:
            n = C.c_uint64(0)
            _check(load().kq_histogram(self._h, None, None, 0, C.byref(n)))
            cov = np.zeros(n.value, dtype=np.uint64)
            cnt = np.zeros(n.value, dtype=np.uint64)
            _check(load().kq_histogram(self._h, _p(cov), _p(cnt), n.value, C.byref(n)))
            d["hist"] = dict(zip(cov.tolist(), cnt.tolist()))
        return d

    # -- lookup
    def lookup_sequence(self, bases: bytes, cov_cutoff=0, map_lo=0, map_hi=None, per_base=False, per_base_buf=None):
        if map_hi is None:
            map_hi = self.map_count
        buf = np.frombuffer(bases, dtype=np.uint8)
        ctr = np.zeros(3, dtype=np.uint64)
        pb = per_base_buf if per_base_buf is not None else (np.zeros(len(buf), dtype=DBGBASE_DTYPE) if per_base else None)
        _check(load().kq_lookup_sequence(self._h, _p(buf), len(buf), cov_cutoff, map_lo, map_hi, _p(pb), _p(ctr)))
        return ctr, pb

    def lookup_sequence_dev(self, bases_ptr, n, counters_ptr, cov_cutoff=0, map_lo=0, map_hi=None, per_base_ptr=None):
        if map_hi is None:
            map_hi = self.map_count
        _check(load().kq_lookup_sequence_dev(self._h, C.c_void_p(bases_ptr), n, cov_cutoff, map_lo, map_hi,
                                             C.c_void_p(per_base_ptr) if per_base_ptr else None, C.c_void_p(counters_ptr)))

    def lookup_keys(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        out = np.zeros(len(keys), dtype=ENTRY_DTYPE)
        _check(load().kq_lookup_keys(self._h, _p(keys), len(keys), _p(out)))
        return out

    def branch_scan(self, bases: bytes, cov_cutoff=0):
        buf = np.frombuffer(bases, dtype=np.uint8)
        flags = np.zeros(len(buf), dtype=np.uint8)
        _check(load().kq_branch_scan(self._h, _p(buf), len(buf), cov_cutoff, _p(flags)))
        return flags

    # -- union / io
    def merge(self, other):
        _check(load().kq_merge(self._h, other._h))

    def import_entries(self, entries):
        entries = np.ascontiguousarray(entries, dtype=ENTRY_DTYPE)
        _check(load().kq_import(self._h, _p(entries), len(entries)))

    def export(self, map_lo=0, map_hi=None):
        if map_hi is None:
            map_hi = self.map_count
        n = C.c_uint64(0)
        _check(load().kq_export(self._h, map_lo, map_hi, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=ENTRY_DTYPE)
        if n.value:
            _check(load().kq_export(self._h, map_lo, map_hi, _p(out), n.value, C.byref(n)))
        return out

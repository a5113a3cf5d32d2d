"""ctypes wrapper of the CPU oracle (oracle/kreeq_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg -- never by kreeq_amd/ (the product).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libkreeq_oracle.so")

ENTRY_DTYPE = np.dtype([("key", "<u8"), ("fw", "<u4", 4), ("bw", "<u4", 4), ("cov", "<u4"), ("hc", "<u4")])
DBGBASE_DTYPE = np.dtype([("fw", "<u4"), ("bw", "<u4"), ("cov", "<u4"), ("isFw", "u1"), ("pad", "u1", 3)])
KMER8_DTYPE = np.dtype([("fw", "u1", 4), ("bw", "u1", 4), ("cov", "u1")])


class Stats(C.Structure):
    _fields_ = [("total", C.c_uint64), ("unique", C.c_uint64), ("distinct", C.c_uint64),
                ("missing", C.c_uint64), ("edges", C.c_uint64)]


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("kreeq_oracle.c", "kreeq_oracle.h", "Makefile")]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libkreeq_oracle.so"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        L = _lib
        L.kqo_create.restype = C.c_void_p
        L.kqo_create.argtypes = [C.c_int, C.c_int]
        L.kqo_destroy.argtypes = [C.c_void_p]
        L.kqo_hash.restype = C.c_uint64
        L.kqo_hash.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.kqo_emit_records.restype = C.c_uint64
        L.kqo_emit_records.argtypes = [C.c_int, C.c_char_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.kqo_insert_records.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        L.kqo_count_batch.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_int]
        L.kqo_summary.argtypes = [C.c_void_p, C.POINTER(Stats), C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.kqo_lookup_segment.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint16, C.c_uint16,
                                         C.c_void_p, C.c_void_p, C.c_int]
        L.kqo_validate_sequence.argtypes = L.kqo_lookup_segment.argtypes
        L.kqo_merge.argtypes = [C.c_void_p, C.c_void_p]
        L.kqo_export.restype = C.c_uint64
        L.kqo_export.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64]
        L.kqo_export_raw8.restype = C.c_uint64
        L.kqo_export_raw8.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64]
        L.kqo_import.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.kqo_error_rate.restype = C.c_double
        L.kqo_error_rate.argtypes = [C.c_uint64, C.c_uint64, C.c_int]
        L.kqo_qv.restype = C.c_double
        L.kqo_qv.argtypes = [C.c_uint64, C.c_uint64, C.c_int]
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def hash_kmer(codes, k):
    """codes: iterable of k base codes 0..3 -> (key, is_fw)"""
    a = np.ascontiguousarray(np.array(codes, dtype=np.uint8))
    fw = C.c_int(0)
    key = lib().kqo_hash(_ptr(a), k, C.byref(fw))
    return key, bool(fw.value)


def emit_records(k, bases: bytes):
    n = lib().kqo_emit_records(k, bases, len(bases), None, None)
    keys = np.empty(n, dtype=np.uint64)
    edges = np.empty(n, dtype=np.uint8)
    if n:
        lib().kqo_emit_records(k, bases, len(bases), _ptr(keys), _ptr(edges))
    return keys, edges


def error_rate(missing, total, k):
    return lib().kqo_error_rate(missing, total, k)


def qv(missing, total, k):
    return lib().kqo_qv(missing, total, k)


class OracleDB:
    def __init__(self, k=21, map_count=128):
        self.k, self.map_count = k, map_count
        self.h = lib().kqo_create(k, map_count)
        if not self.h:
            raise ValueError("bad k / map_count")

    def close(self):
        if self.h:
            lib().kqo_destroy(self.h)
            self.h = None

    __del__ = close

    def count_batch(self, bases: bytes, threads=1):
        rc = lib().kqo_count_batch(self.h, bases, len(bases), threads)
        assert rc == 0

    def insert_records(self, keys, edges):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        edges = np.ascontiguousarray(edges, dtype=np.uint8)
        assert len(keys) == len(edges)
        lib().kqo_insert_records(self.h, _ptr(keys), _ptr(edges), len(keys))

    def summary(self, with_hist=False):
        st = Stats()
        n = C.c_uint64(0)
        lib().kqo_summary(self.h, C.byref(st), None, None, 0, C.byref(n))
        d = {f: getattr(st, f) for f, _ in Stats._fields_}
        if with_hist:
            cov = np.zeros(n.value, dtype=np.uint64)
            cnt = np.zeros(n.value, dtype=np.uint64)
            lib().kqo_summary(self.h, C.byref(st), _ptr(cov), _ptr(cnt), n.value, C.byref(n))
            d["hist"] = dict(zip(cov.tolist(), cnt.tolist()))
        return d

    def _lookup(self, fn, bases, cov_cutoff, map_lo, map_hi, per_base, threads):
        if map_hi is None:
            map_hi = self.map_count
        ctr = np.zeros(3, dtype=np.uint64)
        pb = np.zeros(len(bases), dtype=DBGBASE_DTYPE) if per_base else None
        rc = fn(self.h, bases, len(bases), cov_cutoff, map_lo, map_hi, _ptr(pb), _ptr(ctr), threads)
        if rc != 0:
            raise ValueError("non-ACGT byte inside a segment")
        return ctr, pb

    def lookup_segment(self, bases: bytes, cov_cutoff=0, map_lo=0, map_hi=None, per_base=False):
        return self._lookup(lib().kqo_lookup_segment, bases, cov_cutoff, map_lo, map_hi, per_base, 1)

    def validate_sequence(self, bases: bytes, cov_cutoff=0, map_lo=0, map_hi=None, per_base=False, threads=1):
        return self._lookup(lib().kqo_validate_sequence, bases, cov_cutoff, map_lo, map_hi, per_base, threads)

    def merge(self, other):
        rc = lib().kqo_merge(self.h, other.h)
        if rc != 0:
            raise ValueError("k / map_count mismatch")

    def export(self, m=-1):
        n = lib().kqo_export(self.h, m, None, 0)
        out = np.zeros(n, dtype=ENTRY_DTYPE)
        if n:
            lib().kqo_export(self.h, m, _ptr(out), n)
        return out

    def export_raw8(self, m):
        n = lib().kqo_export_raw8(self.h, m, None, None, 0)
        keys = np.zeros(n, dtype=np.uint64)
        vals = np.zeros(n, dtype=KMER8_DTYPE)
        if n:
            lib().kqo_export_raw8(self.h, m, _ptr(keys), _ptr(vals), n)
        return keys, vals

    def import_entries(self, entries):
        entries = np.ascontiguousarray(entries, dtype=ENTRY_DTYPE)
        rc = lib().kqo_import(self.h, _ptr(entries), len(entries))
        assert rc == 0

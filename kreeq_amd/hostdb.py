"""ctypes binding of the host-side .kreeq database (de)serialiser (kreeq_amd/host/kreeq_db.cpp through hostlib.cpp).
Host code only: phmap dump files, no GPU.  Used by the multi-GPU driver to write one database from bucket-disjoint shards."""
import ctypes as C
import os

import numpy as np

from . import build as _build
from .capi import ENTRY_DTYPE

_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_build.HOSTLIB):
            raise RuntimeError(f"{_build.HOSTLIB} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(_build.HOSTLIB)
        L.kqh_last_error.restype = C.c_char_p
        L.kqh_write_maps.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64)]
        L.kqh_write_finish.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_uint64]
        L.kqh_read_db.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise RuntimeError(_load().kqh_last_error().decode(errors="replace"))


def write_maps(db_dir, map_count, map_lo, map_hi, entries):
    """writes <db>/.map.<m>.bin for m in [map_lo, map_hi) from the logical entries of those maps; returns the high-copy ones"""
    entries = np.ascontiguousarray(entries, dtype=ENTRY_DTYPE)
    os.makedirs(db_dir, exist_ok=True)
    hc = np.zeros(len(entries), dtype=ENTRY_DTYPE)
    n_hc = C.c_uint64(0)
    _check(_load().kqh_write_maps(db_dir.encode(), map_count, map_lo, map_hi, entries.ctypes.data_as(C.c_void_p), len(entries),
                                  hc.ctypes.data_as(C.c_void_p), C.byref(n_hc)))
    return hc[:n_hc.value].copy()


def write_finish(db_dir, k, map_count, hc_entries):
    """writes <db>/.map.hc.bin (ALL high-copy k-mers of the database) and <db>/.index"""
    hc = np.ascontiguousarray(hc_entries, dtype=ENTRY_DTYPE)
    _check(_load().kqh_write_finish(db_dir.encode(), k, map_count, hc.ctypes.data_as(C.c_void_p), len(hc)))


def read_db(db_dir):
    """-> (entries sorted by key, k, map_count)"""
    n, k, mc = C.c_uint64(0), C.c_int(0), C.c_int(0)
    _check(_load().kqh_read_db(db_dir.encode(), None, 0, C.byref(n), C.byref(k), C.byref(mc)))
    out = np.zeros(n.value, dtype=ENTRY_DTYPE)
    _check(_load().kqh_read_db(db_dir.encode(), out.ctypes.data_as(C.c_void_p), n.value, C.byref(n), C.byref(k), C.byref(mc)))
    return out[np.argsort(out["key"], kind="stable")], k.value, mc.value

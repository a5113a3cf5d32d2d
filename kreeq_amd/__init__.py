"""kreeq_amd -- MI355X-native k-mer count / QV engine behind kreeq's hot-path boundary.

Layout: csrc/ (HIP kernels + C ABI), host/ (C++ CLI clone of `kreeq validate|union`), capi.py
(ctypes binding of include/kreeq_amd.h), dist.py (bucket-sharded multi-GPU driver over RCCL),
synth.py (deterministic synthetic reads/assemblies for bench and tests).
"""
from .capi import KreeqDB, KqError, device_available, ENTRY_DTYPE, DBGBASE_DTYPE  # noqa: F401

__all__ = ["KreeqDB", "KqError", "device_available", "ENTRY_DTYPE", "DBGBASE_DTYPE"]

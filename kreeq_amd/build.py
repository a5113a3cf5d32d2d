"""In-tree build of the native pieces (no JIT cache: the .so/.bin travel with the repo snapshot).

  kreeq_amd/lib/libkreeq_amd.so   HIP kernels + C ABI (include/kreeq_amd.h), gfx950 only
  kreeq_amd/bin/kreeq             host CLI clone (validate / union), links the library above
"""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
LIB = os.path.join(PKG, "lib", "libkreeq_amd.so")
CLI = os.path.join(PKG, "bin", "kreeq")
HOSTLIB = os.path.join(PKG, "lib", "libkreeq_host.so")       # .kreeq database files for the multi-GPU driver (no GPU code)

HIP_SOURCES = [os.path.join(PKG, "csrc", "kreeq_amd.hip")]
HIP_DEPS = HIP_SOURCES + [os.path.join(PKG, "csrc", f) for f in ("kq_device.h", "kq_partition.h", "kq_kernels.h")] + [os.path.join(ROOT, "include", "kreeq_amd.h")]
HOST_DIR = os.path.join(PKG, "host")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP library cannot be built (there is no CPU fallback)")


def build_lib(force=False, verbose=False, out=None, defines=()):
    """out / defines: a tuning variant of the library (A/B measurements: KQ_LIB=<path> selects it at load time)"""
    target = out or LIB
    if force or _stale(target, HIP_DEPS):
        os.makedirs(os.path.dirname(target), exist_ok=True)
        cmd = [hipcc(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-Wall", "-Wextra",
               "-o", target] + ["-D" + d for d in defines] + HIP_SOURCES
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return target


def host_sources():
    if not os.path.isdir(HOST_DIR):
        return []
    return sorted(os.path.join(HOST_DIR, f) for f in os.listdir(HOST_DIR) if f.endswith(".cpp") and f != "hostlib.cpp")


def build_hostlib(force=False, verbose=False):
    srcs = [os.path.join(HOST_DIR, f) for f in ("hostlib.cpp", "kreeq_db.cpp")]
    deps = srcs + [os.path.join(HOST_DIR, "kreeq_db.h"), os.path.join(ROOT, "include", "kreeq_amd.h")]
    if force or _stale(HOSTLIB, deps):
        os.makedirs(os.path.dirname(HOSTLIB), exist_ok=True)
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-fPIC", "-shared", "-pthread", "-I", os.path.join(ROOT, "include"), "-o", HOSTLIB] + srcs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return HOSTLIB


def build_cli(force=False, verbose=False):
    srcs = host_sources()
    if not srcs:
        return None
    deps = srcs + [os.path.join(HOST_DIR, f) for f in os.listdir(HOST_DIR) if f.endswith(".h")] + [LIB]
    if force or _stale(CLI, deps):
        os.makedirs(os.path.dirname(CLI), exist_ok=True)
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-pthread", "-I", os.path.join(ROOT, "include"), "-o", CLI] + srcs + \
              ["-L", os.path.dirname(LIB), "-lkreeq_amd", "-lz", "-Wl,-rpath,$ORIGIN/../lib"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return CLI


def build_all(force=False, verbose=False):
    build_lib(force, verbose)
    build_cli(force, verbose)
    build_hostlib(force, verbose)

"""Build libasr_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python build.py [--force]

Warp / SR kernels are compiled with -ffp-contract=off: they restate TensorFlow CPU kernels
whose multiplies and adds round separately (see asr_warp_device.h).
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
LIB = os.path.join(PKG, "libasr_hip.so")

# (source, extra flags)
SOURCES = [
    ("core.cpp", []),
    ("warp.hip", ["-ffp-contract=off"]),
    ("sr.hip", ["-ffp-contract=off"]),
    ("reduce.hip", ["-ffp-contract=off"]),
    ("gemm.hip", []),
    ("dwconv.hip", []),
    ("layers.hip", []),
    ("sepconv.hip", []),
]
HEADERS = ["asr_common.h", "asr_warp_device.h", os.path.join("..", "..", "include", "asr_hip.h")]
COMMON = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
          "-x", "hip"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """Serialised by an exclusive lock on build/.lock: the ranks of one launch may all find the library missing, but only
    one runs hipcc at a time and the others then find every object up to date."""
    import fcntl
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    with open(os.path.join(objdir, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(objdir, force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(objdir, force, verbose):
    hipcc = _hipcc()
    hdrs = [os.path.join(HERE, h) for h in HEADERS] + [os.path.abspath(__file__)]
    objs = []
    for src, extra in SOURCES:
        s = os.path.join(HERE, src)
        o = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc] + COMMON + extra + os.environ.get("ASR_EXTRA_HIPFLAGS", "").split() + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)

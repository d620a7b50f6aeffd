"""Build libssdhip.so (the hand-written gfx950 kernels + C ABI) in-tree with hipcc.

    python -m ssd_keras_amd.build            # build if sources are newer than the library
    python -m ssd_keras_amd.build --force

hipcc cross-compiles for gfx950 without a GPU.  `-ffp-contract=off` is part of the
contract: the kernels promise the reference's operation order, one IEEE rounding per
operation, so the compiler must not fuse a*b+c.
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libssdhip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-I", os.path.join(ROOT, "include"), "-I", CSRC, "-o", LIB] + sources()
    if verbose:
        print("[ssd_keras_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)

"""Build libssdhip.so (the hand-written gfx950 kernels + C ABI) in-tree with hipcc.

    python -m ssd_keras_amd.build            # rebuild what is older than its sources
    python -m ssd_keras_amd.build --force

hipcc cross-compiles for gfx950 without a GPU.  `-ffp-contract=off` is part of the
contract: the kernels promise the reference's operation order, one IEEE rounding per
operation, so the compiler must not fuse a*b+c.

Every `csrc/*.hip` is compiled to its own object under `build/` (in parallel, only when it or a header changed) and the
objects are linked into `ssd_keras_amd/libssdhip.so`.
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(ROOT, "build", "obj")
LIB = os.path.join(PKG, "libssdhip.so")
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fno-gpu-rdc", "-Wall",
          "-Wno-unused-function"]


# Per-file additions.  ssdhip_decode.hip: the 512-thread NMS kernel lives on 80 VGPRs (six waves per SIMD); LLVM's machine-LICM hoists
# a dozen loop-invariant LDS addresses and constants out of its round loop and then SPILLS them at the top of every workgroup
# (32-48 bytes per lane = 10-15 MB of scratch stores per launch, profiles/r04zz_decode_pmc_traffic.json).  Without that pass the
# kernel has no scratch at all; the re-materialised address arithmetic is a handful of VALU instructions per round.
EXTRA_CFLAGS = {"ssdhip_decode.hip": ["-mllvm", "-disable-machine-licm"]}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))


def _obj_of(src):
    return os.path.join(OBJ, os.path.splitext(os.path.basename(src))[0] + ".o")


def _newer(deps, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    hdrs = _headers() + [os.path.abspath(__file__)]
    todo = [s for s in sources() if force or _newer([s] + hdrs, _obj_of(s))]

    def compile_one(src):
        cmd = ([hipcc] + CFLAGS + EXTRA_CFLAGS.get(os.path.basename(src), []) +
               ["-I", os.path.join(ROOT, "include"), "-I", CSRC, "-c", src, "-o", _obj_of(src)])
        if verbose:
            print("[ssd_keras_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if todo:
        with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 4)) as pool:
            list(pool.map(compile_one, todo))
    objs = [_obj_of(s) for s in sources()]
    stale_objs = [o for o in glob.glob(os.path.join(OBJ, "*.o")) if o not in objs]
    for o in stale_objs:                                  # a deleted source must not stay linked in
        os.remove(o)
    if todo or stale_objs or _newer(objs, LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-o", LIB] + objs
        if verbose:
            print("[ssd_keras_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)

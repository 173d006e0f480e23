"""Builds libgsv_hip.so (the C-ABI HIP library, include/gsv.h) in-tree for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
.so travels to the GPU box with the snapshot.  No CPU fallback exists: if the library is
missing every gsv op raises.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
INCLUDE = os.path.join(os.path.dirname(os.path.dirname(HERE)), "include")
LIB = os.path.join(HERE, "libgsv_hip.so")
OBJ = os.path.join(CSRC, "_obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed", "-Wno-unused-value"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _newer(dst, srcs):
    if not os.path.exists(dst):
        return False
    t = os.path.getmtime(dst)
    return all(os.path.getmtime(s) <= t for s in srcs)


def build(force: bool = False, verbose: bool = True) -> str:
    srcs = sources()
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
        [os.path.join(INCLUDE, "gsv.h")]
    if not force and _newer(LIB, deps):
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [d for d in deps if d.endswith(".h")]

    def cc(src):
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        if not force and _newer(obj, [src] + hdrs):
            return obj
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print("[gsv.build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(cc, srcs))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print("[gsv.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)

"""Builds libcslam_hip.so (the C-ABI engine of include/cslam.h) for gfx950 with hipcc.

In-tree build: the shared object lands in conan_slam_amd/lib/ (git-ignored, but it travels with the
working tree to the GPU box).  hipcc cross-compiles without a GPU.  Every translation unit is compiled to its own
object (in parallel) and re-used while none of its dependencies -- every header under csrc/ plus the public header --
has changed.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB_PATH = os.path.join(LIB_DIR, "libcslam_hip.so")
PUBLIC_HEADER = os.path.join(_HERE, "..", "include", "cslam.h")
ARCH = "gfx950"
# RCCL (cslam_pf_resample_sharded) is bound at run time with dlopen, so that the process keeps ONE copy of it
# (PyTorch wheels bundle their own librccl, exactly as they bundle libamdhip64: see _capi.py)
LINK_LIBS = ["-ldl"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X engine cannot be built (there is no CPU fallback)")


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h", ".inc"))]
    return hs + [PUBLIC_HEADER]


def _newer_than(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(p) and os.path.getmtime(p) > t for p in deps)


def is_stale() -> bool:
    deps = [os.path.join(CSRC, s) for s in sources()] + _headers()
    return _newer_than(LIB_PATH, deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc, hdrs = _hipcc(), _headers()
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result"]
    jobs = []
    for s in sources():
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ_DIR, s[:-4] + ".o")
        if force or _newer_than(obj, [src] + hdrs):
            jobs.append([hipcc] + flags + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), 4))) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ_DIR, s[:-4] + ".o") for s in sources()]
    run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH] + objs + LINK_LIBS)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

"""ctypes binding of libcslam_hip.so -- exactly the entry points include/cslam.h declares.

The library is loaded from conan_slam_amd/lib/ (in-tree).  If it is missing or cannot be loaded this
module raises: the engine has no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcslam_hip.so")
HEADER_PATH = os.path.join(_HERE, "..", "include", "cslam.h")

OK, ERR_BAD_ARG, ERR_CAPACITY, ERR_HIP, ERR_NO_DEVICE, ERR_ALLOC = 0, 1, 2, 3, 4, 5
FACTOR_OK, FACTOR_FALLBACK, FACTOR_ZEROED, FACTOR_SKIPPED, FACTOR_BAD_IDF, FACTOR_INTERNAL = 0, 1, 2, 4, 8, 16
F32, F64 = 0, 1
Q_LOWER_CHOL_GAIN, Q_PREDICT_NM4, Q_REF_EXACT, Q_TEXTBOOK = 1, 2, 3, 0
STAGE_GATHER, STAGE_FACTOR, STAGE_GAIN, STAGE_DOWNDATE, N_STAGES = 0, 1, 2, 3, 4
STAGE_NAMES = ["gather", "factor", "gain", "downdate"]


class CslamError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"cslam status {code}: {msg}")
        self.code = code


def declared_symbols(header_path: str = HEADER_PATH):
    """Names of every function include/cslam.h declares (used by the export test)."""
    text = open(header_path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cslam_[a-z0-9_]+)\s*\(", text)))


_lib = None
hip_runtime_path = None  # which libamdhip64 the engine ended up bound to (diagnostics)


def _preload_shared_hip_runtime():
    """One process must hold ONE HIP runtime.  PyTorch-ROCm wheels bundle their own libamdhip64.so
    (SONAME libamdhip64.so.7, the same SONAME libcslam_hip.so is linked against); if the engine bound the
    system copy under /opt/rocm and torch later loaded its own, the second runtime would see no GPU.  So when
    torch is installed its bundled runtime is loaded first (without importing torch) and the engine's
    DT_NEEDED resolves to it by SONAME.  CSLAM_HIP_RUNTIME=system keeps the /opt/rocm runtime (for processes
    that never touch torch's GPU side)."""
    global hip_runtime_path
    if os.environ.get("CSLAM_HIP_RUNTIME", "torch") == "system":
        return
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
        cand = os.path.join(libdir, "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
            hip_runtime_path = cand
            # (RCCL, used by cslam_pf_resample_sharded, is bound by the engine with dlopen("librccl.so.1"): a process
            # that has imported torch gets torch's copy by SONAME, any other process the system one.  It must NOT be
            # preloaded here: loading torch's librccl ahead of `import torch` ends in a double free at exit.)
    except Exception:
        pass  # fall back to the system runtime


def lib() -> C.CDLL:
    """Load the engine; raises if the HIP library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -m conan_slam_amd.build` (hipcc, gfx950). "
                "There is no CPU fallback for the engine.")
        _preload_shared_hip_runtime()
        _lib = C.CDLL(LIB_PATH)
        _lib.cslam_last_error.restype = C.c_char_p
    return _lib


def check(rc: int):
    if rc != OK:
        raise CslamError(rc, lib().cslam_last_error().decode("utf-8", "replace"))


def device_count() -> int:
    c = C.c_int(0)
    rc = lib().cslam_device_count(C.byref(c))
    return c.value if rc == OK else 0

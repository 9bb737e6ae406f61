"""CPU tests of the drop-in boundary: libcslam_hip.so loads, exports every entry point include/cslam.h declares,
and refuses loudly (no CPU fallback) when there is no GPU.  No compute calls here."""
import ctypes
import os

import pytest

from conan_slam_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_declares_the_hot_path_surface():
    names = _capi.declared_symbols()
    for need in ("cslam_ekf_create", "cslam_ekf_predict", "cslam_ekf_update", "cslam_ekf_update_device",
                 "cslam_ekf_augment", "cslam_ekf_observe_heading", "cslam_ekf_get_state", "cslam_ekf_set_state",
                 "cslam_pf_create", "cslam_pf_predict", "cslam_pf_sample_proposal", "cslam_pf_feature_update",
                 "cslam_pf_add_features", "cslam_pf_weight_sums", "cslam_pf_pack", "cslam_pf_unpack",
                 "cslam_pf_resample_sharded", "cslam_comm_create", "cslam_comm_unique_id", "cslam_ekf_run_many",
                 "cslam_ekf_get_streams", "cslam_last_error"):
        assert need in names, need
    assert len(names) >= 40


def test_library_loads_and_exports_every_declared_symbol():
    assert os.path.exists(_capi.LIB_PATH), "build the engine first: python -m conan_slam_amd.build"
    lib = ctypes.CDLL(_capi.LIB_PATH)
    missing = [s for s in _capi.declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    lib.cslam_version.restype = ctypes.c_int
    assert lib.cslam_version() == 100


def test_every_declaration_cites_the_reference():
    """Each hot-path entry point names the reference interface it replaces (file:line)."""
    text = open(os.path.join(ROOT, "include", "cslam.h")).read()
    for cite in ("slam.h:841-847", "slam.h:938-943", "slam.h:190-191", "slam.h:788", "EKF.cpp:406-455",
                 "EKF.cpp:481-496", "EKF.cpp:9-26", "EKF.cpp:328-352", "PF.cpp:419-471", "PF.cpp:502-544",
                 "PF.cpp:222-277", "PF.cpp:9-60", "PF.cpp:473-500"):
        assert cite in text, cite


def test_adapter_header_compiles_and_links_against_the_library(tmp_path):
    """include/cslam_adapter.hpp (HipEKF / HipPF : public EKF / PF, the reference-side binding) compiled with g++ against
    an Eigen-free stand-in of the reference's types (tests/adapter/adapter_standin.hpp) and linked against
    libcslam_hip.so: every forwarding call names an exported entry point with matching argument types."""
    import shutil
    import subprocess

    gxx = shutil.which("g++")
    assert gxx, "g++ is part of the image"
    src = os.path.join(ROOT, "tests", "adapter", "adapter_compile_check.cpp")
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tests", "adapter")]
    subprocess.run([gxx, "-std=c++17", "-Wall", "-Werror", "-fsyntax-only"] + inc + [src], check=True)
    obj = str(tmp_path / "adapter_check.o")
    subprocess.run([gxx, "-std=c++17", "-c"] + inc + [src, "-o", obj], check=True)
    exe = str(tmp_path / "adapter_check")
    libdir = os.path.dirname(_capi.LIB_PATH)
    r = subprocess.run([gxx, obj, "-L" + libdir, "-lcslam_hip", "-L/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib",
                        "-Wl,--allow-shlib-undefined", "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # the replay driver that tests/test_adapter_gpu.py builds and RUNS on the GPU box: same compile + link check here
    rsrc = os.path.join(ROOT, "tests", "adapter", "adapter_replay.cpp")
    rexe = str(tmp_path / "adapter_replay")
    r = subprocess.run([gxx, "-std=c++17", "-Wall", "-Werror"] + inc + [rsrc, "-L" + libdir, "-lcslam_hip", "-L/opt/rocm/lib",
                        "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,--allow-shlib-undefined", "-o", rexe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # without the stand-in and without Eigen the header must compile to nothing (it is guarded by __has_include)
    empty = tmp_path / "empty.cpp"
    empty.write_text('#include "cslam_adapter.hpp"\nint main() { return 0; }\n')
    subprocess.run([gxx, "-std=c++17", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), str(empty)], check=True)


def test_header_is_plain_c():
    """include/cslam.h is the C ABI: it must compile as C99 on its own."""
    import shutil
    import subprocess

    gcc = shutil.which("gcc")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", "cslam.h")],
                   check=True)


def test_no_cpu_fallback_without_a_gpu():
    import conan_slam_amd

    if conan_slam_amd.device_count() > 0:
        pytest.skip("a GPU is visible here; the refusal path is exercised in the CPU container")
    with pytest.raises(conan_slam_amd.CslamError) as ei:
        conan_slam_amd.EKF(10)
    assert ei.value.code == _capi.ERR_NO_DEVICE and "no CPU fallback" in str(ei.value)
    from conan_slam_amd.pf import ParticleShard

    with pytest.raises(conan_slam_amd.CslamError) as ei:
        ParticleShard(8, 4)
    assert ei.value.code == _capi.ERR_NO_DEVICE


def test_product_code_never_touches_the_oracle():
    """The engine (package + csrc + include) must not import, link or mention anything under oracle/."""
    bad = []
    for base in ("conan_slam_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h")):
                    txt = open(os.path.join(dirpath, f), errors="replace").read()
                    for needle in ("pyoracle", "slam_oracle", "np_restatement", "sim_driver", "orc_"):
                        if needle in txt:
                            bad.append((f, needle))
    assert not bad, bad

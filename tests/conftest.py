"""pytest configuration: the `gpu` marker, import paths, and shared fixtures."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


def _have_gpu() -> bool:
    try:
        import conan_slam_amd

        return conan_slam_amd.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_required():
    """GPU tests FAIL (not skip) when the HIP engine cannot be used: there is no CPU fallback to hide behind."""
    import conan_slam_amd

    n = conan_slam_amd.device_count()
    assert n > 0, "no HIP device visible: -m gpu tests must run on an MI355X box"
    return n


@pytest.fixture(scope="session")
def rng():
    return np.random.default_rng(12345)

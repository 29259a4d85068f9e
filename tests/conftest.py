"""Shared fixtures. GPU tests are marked `gpu`; everything else runs on CPU."""
import importlib.util
import os
import sys

import pytest

try:  # torch first: its bundled HIP runtime and the one libiile_gpu.so links must be the same copy
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_binding():
    """The package directory is named after the reference repo (hyphens), so it
    is loaded by path."""
    name = "iile_binding"
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(REPO, "pbrt-v3-iile_amd", "binding.py")
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def binding():
    import __graft_entry__ as ge
    ge.build_if_needed()
    return load_binding()


@pytest.fixture(scope="session")
def oracle(binding):
    import oracle_binding
    return oracle_binding.Oracle()


@pytest.fixture(scope="session")
def scene_c1(binding):
    """BASELINE config 0: killeroo-simple, 400x400, 8 spp."""
    return binding.HostScene(xres=400, yres=400, spp=8)


@pytest.fixture(scope="session")
def scene_small(binding):
    """A small frame for whole-image parity checks."""
    return binding.HostScene(xres=160, yres=120, spp=4)


@pytest.fixture(scope="session")
def gpu_c1(binding, scene_c1):
    return binding.GpuScene(scene_c1)


@pytest.fixture(scope="session")
def gpu_small(binding, scene_small):
    return binding.GpuScene(scene_small)

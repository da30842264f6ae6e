import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cpu_library():
    """csrc/libdrm_cpu.so — the host build of the C ABI behind device="cpu" models — built if it is missing or older than its
    sources (g++, ~8 s), as `__graft_entry__.build()` does."""
    import __graft_entry__ as entry
    entry.build_cpu_library()
    import importlib
    return importlib.import_module("differentiable_robot_model_amd.backend").load_library(kind="cpu")


@pytest.fixture(scope="session")
def hostcall_module():
    """csrc/drm_hostcall.so — the per-call host work of the hot public methods as a torch C++ extension — built if it is missing or
    older than its source (~30 s), as `__graft_entry__.build()` does."""
    if os.environ.get("DRM_NO_HOSTCALL") == "1":
        pytest.skip("DRM_NO_HOSTCALL=1: this run exercises the Python host path")
    import __graft_entry__ as entry
    entry.build_hostcall()
    import importlib
    backend = importlib.import_module("differentiable_robot_model_amd.backend")
    backend._hostcall = False          # (a process that looked before the build cached "unavailable")
    mod = backend.hostcall()
    assert mod is not None
    return mod

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def pkg():
    return entry.load_package()


@pytest.fixture(scope="session")
def abi(pkg):
    return pkg.abi


@pytest.fixture(scope="session")
def scenes(pkg):
    return pkg.scenes


@pytest.fixture(scope="session")
def ob():
    """The CPU oracle (test infrastructure): built on demand."""
    binding = entry.load_oracle()
    binding.build(native=False)
    return binding


@pytest.fixture(scope="session")
def native_lib(pkg):
    """libdrmlt_amd.so: built in-tree by hipcc (cross-compiles without a GPU)."""
    pkg.build_native()
    return pkg.binding.load_library()

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def repo_root():
    return ROOT


def pytest_collection_modifyitems(config, items):
    """tests/test_gpu_dist.py starts child processes (one per rank).  A child is a fresh interpreter, i.e. fork + exec --
    which must not happen from a process that has already initialised the GPU on this pool.  The pytest process itself
    never touches the GPU in those tests, so they only have to run BEFORE every other GPU test: move them to the front."""
    first = [it for it in items if "test_gpu_dist" in it.nodeid]
    rest = [it for it in items if "test_gpu_dist" not in it.nodeid]
    items[:] = first + rest

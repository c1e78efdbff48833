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


class _AttnSelect:
    """fvta_attn_kernel_select for the duration of a test: exact() / fast(wave16) switch the focal-attention forward main
    kernel inside the process (the library reads FVTA_ATTN_EXACT / FVTA_ATTN_WAVE16 once); the default returns afterwards."""

    def __init__(self):
        from fvta_memexqa_amd import _lib
        self.lib = _lib.load()

    def exact(self):
        self.lib.fvta_attn_kernel_select(1, 0)

    def fast(self, wave16=-1):
        self.lib.fvta_attn_kernel_select(0, int(wave16))

    def default(self):
        self.lib.fvta_attn_kernel_select(-1, -1)


@pytest.fixture
def attn_select():
    sel = _AttnSelect()
    yield sel
    sel.default()


@pytest.fixture(autouse=True)
def _collector_back_on():
    """Trainer.step_device freezes and disables CPython's collector (Trainer.own_host) until the trainer goes away; a
    failed test's traceback can keep its trainer alive: give every test the collector back."""
    import gc
    yield
    if not gc.isenabled():
        gc.enable()
        gc.unfreeze()

"""CPU-side checks of the C ABI: the library loads and exports every symbol include/fvta_hip.h
declares (no compute calls: there is no GPU here), descriptors size queries answer, bad arguments
come back as status codes with a message (nothing throws across the ABI)."""
import ctypes
import os
import re

import pytest

from fvta_memexqa_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "fvta_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fvta_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()


def test_header_and_binding_agree():
    assert _declared() == _lib.exported_symbols()


def test_library_exports_every_declared_symbol(lib):
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.fvta_version() >= 100


def test_size_queries_and_error_reporting(lib):
    d = _lib.AttnDesc(64, 6, 1200, 30, 1024, 2, 0, 1)
    assert lib.fvta_attn_saved_bytes(ctypes.byref(d)) > 64 * 6 * 1200 * 4
    assert lib.fvta_attn_workspace_bytes(ctypes.byref(d)) > 0
    bad = _lib.AttnDesc(64, 6, 1200, 30, 1000, 2, 0, 1)          # unsupported width
    assert lib.fvta_attn_saved_bytes(ctypes.byref(bad)) == 0
    assert b"unsupported" in lib.fvta_last_error()
    bad5 = _lib.AttnDesc(1, 1, 8, 4, 64, 5, 0, 0)                 # model_v2.py:255-257
    assert lib.fvta_attn_saved_bytes(ctypes.byref(bad5)) == 0
    assert b"similarity matrix not implemented" in lib.fvta_last_error()
    ld = _lib.LstmDesc(12800, 30, 200, 512, 1, 0, 1, 0)
    assert lib.fvta_lstm_plan_bytes(ctypes.byref(ld)) > 2 * 2 * 30 * 12800 * 8
    assert lib.fvta_lstm_saved_bytes(ctypes.byref(ld)) >= 2 * 30 * 12800 * 5 * 512 * 4
    ldb = _lib.LstmDesc(12800, 30, 200, 512, 1, 1, 1, 0)
    assert lib.fvta_lstm_workspace_bytes(ctypes.byref(ldb)) > lib.fvta_lstm_workspace_bytes(ctypes.byref(ld))
    # invalid arguments -> negative status + message, never an exception / crash
    st = lib.fvta_lstm_plan(ctypes.byref(_lib.LstmDesc(4, 3, 6, 32, 1, 0, 0, 0)), None, None, None, None, 64, None, None)
    assert st == -1 and b"multiple of 4" in lib.fvta_last_error()
    st = lib.fvta_adam_step(None, None, None, None, 0, 0.1, 0.9, 0.999, 1e-8, 1, 1.0, None)
    assert st == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.FvtaError, match="no CPU fallback"):
        _lib.load()


def test_ops_refuse_to_run_without_gpu():
    import torch
    from fvta_memexqa_amd import ops
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.FvtaError, match="no CPU fallback"):
        ops.require_gpu()
    from fvta_memexqa_amd.model_v2 import Model
    with pytest.raises(_lib.FvtaError):
        Model(dict(hidden_size=32))


def test_descriptor_struct_sizes_match_the_library():
    """The ctypes mirrors of the descriptor structs against the library's own sizeof (fvta_abi_struct_bytes): _lib.load()
    refuses a mismatch; here the table itself is checked, including an unknown id."""
    import ctypes
    from fvta_memexqa_amd import _lib
    lib = _lib.load()
    for which, cls in enumerate((_lib.AttnDesc, _lib.LstmDesc, _lib.ScorerDesc, _lib.TimewarpDesc, _lib.EmbedDesc, _lib.ImgTransDesc)):
        assert lib.fvta_abi_struct_bytes(which) == ctypes.sizeof(cls), cls.__name__
    assert lib.fvta_abi_struct_bytes(99) == -1
    assert ctypes.sizeof(_lib.EmbedDesc) == 48 and ctypes.sizeof(_lib.AttnDesc) == 40

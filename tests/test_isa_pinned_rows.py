"""Static guard for the 16-row attention kernel's untracked row loads: see fvta_memexqa_amd/_isa_guard.py (the same
check __graft_entry__.build() runs, so a compiler that breaks the pinned-register contract fails the build)."""
import os
import shutil

import pytest

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="hipcc not available")
def test_nothing_else_writes_the_pinned_row_registers():
    from fvta_memexqa_amd import _isa_guard
    assert _isa_guard.check(HIPCC) >= 12

"""Static guard for the 16-row attention kernel's untracked row loads (csrc/attn_fwd.hip, rows_issue / rows_wait): the
three row buffers are pinned to v160..v255 and loaded through inline asm the compiler's waitcnt pass does not see, so
NOTHING else may write those registers while loads can be in flight.  Compiles the file to gfx950 assembly (no GPU
needed) and checks every instantiation: after the first row load the only other writers of v160..v255 are the prologue's
zero fills of buffers B / C on the one- and two-tile paths (no load of those buffers has been issued there), all before
the first barrier of the tile loop."""
import os
import re
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "..", "fvta_memexqa_amd", "csrc", "attn_fwd.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="hipcc not available")
def test_nothing_else_writes_the_pinned_row_registers(tmp_path):
    out = tmp_path / "attn_fwd.s"
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "-Wno-unused-variable",
                    "-S", "--cuda-device-only", "-o", str(out), SRC], check=True, stderr=subprocess.DEVNULL)
    txt = out.read_text()
    names = re.findall(r"^(_ZN4fvta15attn_fwd_rows16\w*):", txt, re.M)
    assert len(names) >= 12
    store_like = ("s_", "ds_write", "global_store", "buffer_store", "global_atomic", "scratch_store")
    for name in names:
        a = txt.index(name + ":")
        lines = txt[a:txt.index("s_endpgm", a)].split("\n")
        first = next(i for i, l in enumerate(lines) if "global_load_dwordx4" in l)
        loads = [l for l in lines if "global_load_dwordx4" in l]
        assert all(re.search(r"global_load_dwordx4 v\[(\d+):", l) and 160 <= int(re.search(r"v\[(\d+):", l).group(1)) <= 252
                   for l in loads), name                                    # every row load lands in the pinned range
        assert "scratch_" not in txt[a:txt.index("s_endpgm", a)], name + ": spills"
        first_barrier = next(i for i, l in enumerate(lines) if i > first and "s_barrier" in l)
        for i, l in enumerate(lines[first + 1:], first + 1):
            l = l.strip()
            m = re.match(r"(\S+)\s+(?:v\[(\d+):(\d+)\]|v(\d+))\b", l)
            if not m or "global_load_dwordx4" in l or m.group(1).startswith(store_like):
                continue
            lo = int(m.group(2) or m.group(4))
            hi = int(m.group(3) or m.group(4))
            if hi < 160 or lo > 255:
                continue
            # the only tolerated writers: zero fills of buffers B / C (v192..v255) ahead of the tile loop, on the paths
            # where the stream is too short for those buffers ever to be loaded
            ok = i < first_barrier and re.match(r"v_mov_b32_e32 v(19[2-9]|2[0-4]\d|25[0-5]), (0|v\d+)$", l)
            assert ok, "%s line %d: %s" % (name, i, l)

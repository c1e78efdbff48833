"""Static guard for the 16-row attention kernel's untracked row loads (csrc/attn_fwd.hip, rows_issue / rows_wait): the
three row buffers are pinned to v160..v255 and loaded through inline asm the compiler's waitcnt pass does not see, so
NOTHING else may write those registers while loads can be in flight.  `check()` compiles the file to gfx950 assembly (no
GPU needed) and verifies every instantiation: after the first row load the only other writers of v160..v255 are the
prologue's zero fills of buffers B / C on the one- and two-tile paths (no load of those buffers has been issued there),
all before the first barrier of the tile loop; no spills.  Run by __graft_entry__.build() (a compiler that breaks the
assumption fails the build) and by tests/test_isa_pinned_rows.py."""
import os
import re
import subprocess
import tempfile

SRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "attn_fwd.hip")


class IsaGuardError(RuntimeError):
    pass


def hipcc_version(hipcc):
    try:
        out = subprocess.run([hipcc, "--version"], check=True, capture_output=True, text=True).stdout
        return " | ".join(l.strip() for l in out.splitlines()[:2])
    except Exception as exc:                                   # noqa: BLE001
        return "unknown (%r)" % (exc,)


def check(hipcc="/opt/rocm/bin/hipcc", src=SRC):
    """raises IsaGuardError when the compiled kernels do not honour the pinned-register contract; returns the number of
    instantiations checked"""
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "attn_fwd.s")
        subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "-Wno-unused-variable",
                        "-S", "--cuda-device-only", "-o", out, src], check=True, stderr=subprocess.DEVNULL)
        txt = open(out).read()
    names = re.findall(r"^(_ZN4fvta15attn_fwd_rows16\w*):", txt, re.M)
    if len(names) < 12:
        raise IsaGuardError("expected >= 12 attn_fwd_rows16 instantiations, found %d" % len(names))
    store_like = ("s_", "ds_write", "global_store", "buffer_store", "global_atomic", "scratch_store")

    def need(cond, msg):
        if not cond:
            raise IsaGuardError("attn_fwd_rows16 pinned-register contract broken (%s); compiler: %s.  Work-around: "
                                "FVTA_ATTN_EXACT=1 selects the exact-fp32 kernel." % (msg, hipcc_version(hipcc)))

    for name in names:
        a = txt.index(name + ":")
        body = txt[a:txt.index("s_endpgm", a)]
        lines = body.split("\n")
        first = next(i for i, l in enumerate(lines) if "global_load_dwordx4" in l)
        loads = [l for l in lines if "global_load_dwordx4" in l]
        need(all(re.search(r"global_load_dwordx4 v\[(\d+):", l) and 160 <= int(re.search(r"v\[(\d+):", l).group(1)) <= 252
                 for l in loads), name + ": a row load lands outside v160..v255")
        need("scratch_" not in body, name + ": spills")
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
            need(ok, "%s line %d: %s" % (name, i, l))
    # The two-waves-per-tile kernel (attn_fwd_pair16, the default for w >= 512) holds half a tile in registers at two
    # waves per SIMD: compiler-allocated, but only worth running while the allocation stays spill-free (the same loop
    # body wrapped in a generic lambda once spilled 1.1 KB per lane and ran 4x slower).
    pair = re.findall(r"^(_ZN4fvta15attn_fwd_pair16\w*):", txt, re.M)
    if len(pair) < 6:
        raise IsaGuardError("expected >= 6 attn_fwd_pair16 instantiations, found %d" % len(pair))
    for name in pair:
        a = txt.index(name + ":")
        body = txt[a:txt.index("s_endpgm", a)]
        if "scratch_" in body:
            raise IsaGuardError("%s spills to scratch; compiler: %s.  Work-around: FVTA_ATTN_WAVE16=0 selects "
                                "attn_fwd_rows16." % (name, hipcc_version(hipcc)))
    return len(names)

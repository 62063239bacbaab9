"""The two kernels that issue their MFMAs as inline asm get no hazard padding from the compiler: check the
ISA hipcc emits for them -- no VALU write of an MFMA operand within two wait states of the MFMA, no accumulator copies in the
MFMA loops.  Cross-compiles for gfx950; needs no GPU."""
import os
import shutil
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_inline_asm_mfma_kernels_have_no_unpadded_hazards(tmp_path):
    import isa_hazard_check as chk
    for path in chk.build(str(tmp_path)):
        nmfma, hazards, copies = chk.check(path)
        assert nmfma >= 256 and hazards == 0 and copies == 0, (path, nmfma, hazards, copies)


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_inline_asm_lds_stores_are_waited_for_before_barriers(tmp_path):
    """conv3x3_wgrad_winoz.hip: its image stores are inline asm, which the compiler's s_waitcnt bookkeeping does not see --
    every barrier must have an explicit lgkmcnt(0) wait behind the last store in front of it (round 4 found the prologue's
    barrier without one)."""
    import isa_hazard_check as chk
    for path in chk.build_lds(str(tmp_path)):
        seen, bad = chk.check_lds_stores_before_barriers(path)
        assert seen >= 12 and bad == 0, (path, seen, bad)


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_two_tile_kernel_keeps_its_k_loops_free_of_spills(tmp_path):
    """conv3x3_wino_t2.hip instantiates the wide kernel's K loop twice behind force-inlined lambdas; how hipcc allocates its
    512 registers around the two epilogues decides whether it pays (a loop-form of the same source spilled 24 registers into
    the loops and ran SLOWER than one tile per workgroup).  Checked on the emitted ISA: both kernels hold two
    64-MFMA loops each with no scratch access inside, and at most a handful of scratch accesses anywhere."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "t2.s")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "-S", "--cuda-device-only",
                           "-I", os.path.join(root, "scm_gan_amd", "csrc"), "-I", os.path.join(root, "include"),
                           os.path.join(root, "scm_gan_amd", "csrc", "conv3x3_wino_t2.hip"), "-o", out],
                          stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\d+conv3x3_wino_t2_kernel\w*:", l)]
    assert len(starts) == 2, starts
    for s in starts:
        e = next(i for i in range(s, len(lines)) if lines[i].strip().startswith("s_endpgm"))
        body = lines[s:e]
        labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        loops = []
        for i, l in enumerate(body):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                seg = body[labels[m.group(1)]:i + 1]
                if sum(1 for x in seg if x.strip().startswith("v_mfma")) == 64:
                    loops.append(seg)
        assert len(loops) == 2, len(loops)
        for seg in loops:
            assert not any("scratch_" in x for x in seg)
            assert sum(1 for x in seg if x.strip() and not x.strip().startswith((";", "."))) <= 280      # (the one-tile loop: 252, 268 with the zero-padding selects)
        assert sum(1 for x in body if "scratch_" in x) <= 8

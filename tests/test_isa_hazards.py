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

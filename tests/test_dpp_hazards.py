"""CPU tier: the DPP data hazards of the row-oriented Riccati recursion, checked statically in the ISA hipcc generates
(tools/check_dpp_hazards.py). The recursion's cross-lane reads are inline asm (tsat_riccati_dpp.inc), which the compiler's hazard
recogniser does not look into: a DPP read of a VGPR less than two wait states after a VALU write of it returns the OLD value on the
hardware — silently, and only in the lanes that read across — so the generated code is held to the rule here, where no GPU is
needed to see it."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_dpp_hazards as chk  # noqa: E402


def _listing(tmp_path, body):
    p = tmp_path / "t.s"
    p.write_text("_Z4testv:\n" + body)
    return str(p)


def test_checker_sees_a_seeded_hazard(tmp_path):
    dpp = "\tv_fmac_f64_dpp v[10:11], v[2:3], v[4:5] row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
    # the DPP source written by the instruction right in front: 0 wait states
    n, bad = chk.check_listing(_listing(tmp_path, "\tv_mul_f64 v[2:3], v[6:7], v[8:9]\n" + dpp))
    assert n == 1 and len(bad) == 1 and "[2, 3]" in bad[0]
    # one independent instruction in between: 1 wait state, still a hazard; two: fine; s_nop 1: fine; s_nop 0: not enough
    mid = "\tv_add_f64 v[20:21], v[22:23], v[24:25]\n"
    assert len(chk.check_listing(_listing(tmp_path, "\tv_mul_f64 v[2:3], v[6:7], v[8:9]\n" + mid + dpp))[1]) == 1
    assert chk.check_listing(_listing(tmp_path, "\tv_mul_f64 v[2:3], v[6:7], v[8:9]\n" + mid + mid + dpp))[1] == []
    assert chk.check_listing(_listing(tmp_path, "\tv_mul_f64 v[2:3], v[6:7], v[8:9]\n\ts_nop 1\n" + dpp))[1] == []
    assert len(chk.check_listing(_listing(tmp_path, "\tv_mul_f64 v[2:3], v[6:7], v[8:9]\n\ts_nop 0\n" + dpp))[1]) == 1
    # the non-DPP operand (src1) and the accumulator may be fresh: no hazard
    assert chk.check_listing(_listing(tmp_path, "\tv_mul_f64 v[4:5], v[6:7], v[8:9]\n" + dpp))[1] == []
    assert chk.check_listing(_listing(tmp_path, "\tv_mul_f64 v[10:11], v[6:7], v[8:9]\n" + dpp))[1] == []
    # a VALU write of EXEC needs five wait states in front of a DPP instruction
    assert len(chk.check_listing(_listing(tmp_path, "\tv_cmpx_lt_f64_e64 v[6:7], v[8:9]\n\ts_nop 3\n" + dpp))[1]) == 1
    assert chk.check_listing(_listing(tmp_path, "\tv_cmpx_lt_f64_e64 v[6:7], v[8:9]\n\ts_nop 4\n" + dpp))[1] == []


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc (cross-compiles without a GPU)")
def test_generated_isa_of_the_riccati_rows_has_no_dpp_hazard():
    assert chk.main() == 0

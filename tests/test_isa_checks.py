"""Static checks on the generated gfx950 code (hipcc cross-compiles without a GPU)."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "scripts", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_overlap_finder_on_a_known_listing():
    chk = _load("check_load_overlap")
    bad = """
_Zkernel_a:
\tglobal_load_dwordx4 v[14:17], v18, s[20:21] offset:48
\tv_cmp_gt_f32_e32 vcc, 0, v48
\tglobal_load_dwordx4 v[16:19], v18, s[20:21] offset:32
\ts_waitcnt vmcnt(0)
\tglobal_load_dwordx4 v[14:17], v18, s[20:21]
"""
    found = chk.find_overlaps(bad)
    assert len(found) == 1 and found[0][0] == "_Zkernel_a" and "offset:32" in found[0][2]
    good = bad.replace("v[16:19]", "v[20:23]")
    assert chk.find_overlaps(good) == []


def test_packet_step_budget_on_a_known_listing():
    chk = _load("check_load_overlap")
    step = "_Zkernel_p:\n.LBB0_1:\n\ts_load_dwordx16 s[0:15], s[42:43], s51 offset:0x20\n" + "\tv_pk_fma_f32 v[0:1], s[0:1], v[2:3], v[4:5]\n" * 36
    assert chk.packet_step_budget(step + "\ts_cselect_b32 s1, s2, s3\n" * 45 + ".Lfunc_end0:\n") == []
    over = chk.packet_step_budget(step + "\ts_cselect_b32 s1, s2, s3\n" * 60 + ".Lfunc_end0:\n")
    assert len(over) == 1 and over[0][2:] == (36, 61)


@pytest.mark.skipif(not shutil.which("hipcc") and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_no_loads_in_flight_with_overlapping_destinations():
    """Two loads in flight whose destination registers overlap serialise in hardware: 12 % on the 10 M-triangle soup when the
    register allocator produced that in the BVH4q node step (profiles/r02s_c4_regression.txt).  The node steps keep the dead
    half of their last 16-byte load live to prevent it; this checks the code the Makefile's flags generate.  The same
    compile checks that the packet walk's node step stays within its instruction budget."""
    chk = _load("check_load_overlap")
    assert chk.main() == 0

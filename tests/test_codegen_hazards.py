"""Gate on a hipcc code-generation hazard in large kernels (tools/check_long_branch_hazard.py): an
expanded long branch whose scavenged SGPR pair still has a scalar load in flight makes the wave jump to a data address --
an intermittent "Memory access fault by GPU".  The product build must contain none."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import check_long_branch_hazard as chk  # noqa: E402

HAZARD = """
k_demo:                                 ; @k_demo
	s_load_dwordx2 s[0:1], s[0:1], 0x8
	s_cmp_lg_u32 s2, 0
	s_cbranch_scc1 .LBB0_1
	s_getpc_b64 s[2:3]
.Lpost_getpc0:
	s_add_u32 s2, s2, (.LBB0_2-.Lpost_getpc0)&4294967295
	s_addc_u32 s3, s3, (.LBB0_2-.Lpost_getpc0)>>32
	s_setpc_b64 s[2:3]
.LBB0_1:
	s_waitcnt lgkmcnt(0)
	s_endpgm
.LBB0_2:
	%s
	s_getpc_b64 s[0:1]
.Lpost_getpc1:
	s_add_u32 s0, s0, (.LBB0_1-.Lpost_getpc1)&4294967295
	s_addc_u32 s1, s1, (.LBB0_1-.Lpost_getpc1)>>32
	s_setpc_b64 s[0:1]
.Lfunc_end0:
"""


def _hazards(text):
    return [h for name, lines in chk.functions(text) for h in chk.analyse(name, lines)]


def test_checker_sees_the_hazard_and_its_absence():
    bad = _hazards(HAZARD % "s_nop 0")
    assert len(bad) == 1 and bad[0][0] == "k_demo" and bad[0][2] == [0, 1]
    assert _hazards(HAZARD % "s_waitcnt lgkmcnt(0)") == []
    assert _hazards(HAZARD % "s_waitcnt vmcnt(0) lgkmcnt(1)") != []  # scalar loads return out of order: only lgkmcnt(0) settles


def test_product_device_code_has_no_long_branch_over_a_pending_scalar_load():
    paths = chk.compile_to_asm([])  # one assembly file per translation unit (every curve)
    assert len(paths) >= 4
    for path in paths:
        long_branches, found, live = chk.check_file(path)
        assert found == [], (path, found)
        assert live == [], (path, live)  # ... and no expanded branch whose scratch pair is read at its target before it is written
        assert long_branches >= 0


LIVE = """
k_demo2:                                ; @k_demo2
	s_and_saveexec_b64 s[4:5], vcc
	s_cbranch_execnz .LBB1_1
	s_getpc_b64 s[%s]
.Lpost_getpc7:
	s_add_u32 s%d, s%d, (.LBB1_2-.Lpost_getpc7)&4294967295
	s_addc_u32 s%d, s%d, (.LBB1_2-.Lpost_getpc7)>>32
	s_setpc_b64 s[%s]
.LBB1_1:
	v_mov_b32_e32 v0, 1
.LBB1_2:
	s_or_b64 exec, exec, s[4:5]
	s_endpgm
.Lfunc_end1:
"""


def test_checker_sees_a_long_branch_through_a_live_register_pair():
    def live(pair, lo, hi):
        text = LIVE % (pair, lo, lo, hi, hi, pair)
        return [h for name, lines in chk.functions(text) for h in chk.analyse_liveness(name, lines)]

    bad = live("4:5", 4, 5)  # the saved exec mask is read at the target
    assert len(bad) == 1 and bad[0][0] == "k_demo2" and bad[0][3] == [4, 5]
    assert live("6:7", 6, 7) == []


def test_machine_verifier_report_is_parsed():
    """the third gate's parser: LLVM's machine-verifier report -> (function, message)"""
    import check_machine_verifier as mv

    class P:
        returncode = 1
        stdout = ("# After Register Coalescer\n*** Bad machine code: Live range continues after dead def flag ***\n"
                  "- function:    k_demo\n- basic block: %bb.34\n- instruction: 34736B\tdead undef %4299.sub0:vreg_128_align2 = V_ADD3_U32_e64\n"
                  "*** Bad machine code: A Subrange is not covered by the main range ***\n- function:    k_demo\nLLVM ERROR: Found 2 machine code errors.\n")

    real = mv.subprocess.run
    mv.subprocess.run = lambda *a, **k: P
    try:
        found = mv.verify_unit("msm_hip.hip", [])
    finally:
        mv.subprocess.run = real
    assert found == [("k_demo", "Live range continues after dead def flag"), ("k_demo", "A Subrange is not covered by the main range")]


def test_product_device_code_passes_the_machine_verifier():
    """LLVM's machine verifier over every translation unit, compiled with the build's own flags (tools/check_machine_verifier.py): the
    register-coalescer miscompile that made the all-assembly diagnostic build return wrong bucket sums (round 3: "cause not isolated";
    profiles/r04_asm_everywhere_rootcause.txt) is reported by it, and the product must be free of it.  Cached under build/ by the sources'
    hash: about three minutes after a change of csrc/, nothing otherwise."""
    import check_machine_verifier as mv

    reports = mv.check()
    assert len(reports) >= 4
    for unit, found in reports.items():
        assert found == [], (unit, found)


def test_machine_verifier_fires_on_the_miscompiled_variant():
    """... and the gate does fire on the build that was wrong on the GPU: the assembly multipliers in every kernel (since round 4 the
    product's shape) compiled WITH the SLP vectoriser (MSM_HIP_SLP=1: the compiler's default, which rounds 1 - 3 used) -- the 14-limb
    unit's k_smvp_stitch, exactly the kernel whose P + P path returned garbage (profiles/r03_ab_asm_everywhere.txt)."""
    import check_machine_verifier as mv

    old = os.environ.get("MSM_HIP_SLP")
    os.environ["MSM_HIP_SLP"] = "1"
    try:
        found = mv.check(units=["curve_bls12_381.hip"])["curve_bls12_381.hip"]
    finally:
        if old is None:
            del os.environ["MSM_HIP_SLP"]
        else:
            os.environ["MSM_HIP_SLP"] = old
    assert found and all("k_smvp_stitch" in fn for fn, _ in found), found
    assert any("dead def" in msg for _, msg in found), found


def test_library_on_disk_was_built_with_the_gated_flags(built, monkeypatch):
    """The gates above inspect what the CURRENT flags and sources compile to; the stamp next to libmsm_hip.so says the library that is
    loaded was built from exactly those (a library built once with MSM_HIP_SLP=1 or MSM_HIP_EXTRA_FLAGS must not be reused as the product)."""
    import importlib

    import msm_webgpu_amd  # noqa: F401

    b = importlib.import_module("msm_webgpu_amd.build")  # (the package re-exports the FUNCTION build under the same name: take the module)

    assert b.stamp_is_current() and not b.needs_build()
    plain = b.build_stamp()
    monkeypatch.setenv("MSM_HIP_SLP", "1")
    assert b.build_stamp() != plain and b.needs_build()  # variant flags: another build ...
    with pytest.raises(RuntimeError):                    # ... which must name its own file: the product library is never overwritten by one
        b.build()
    monkeypatch.delenv("MSM_HIP_SLP")
    monkeypatch.setenv("MSM_HIP_EXTRA_FLAGS", "-DX=1")
    assert b.build_stamp() != plain

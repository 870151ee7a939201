"""Compile-time checks of the gfx950 code of ffgpu_kernels.hip (hipcc cross-compiles here, no GPU needed): tools/isa_lint.py -- the same lint the library
build runs (ffcnn_amd/csrc/Makefile) -- on a fresh listing, plus unit checks of the lint itself.

Round 5 found the round-4 version of this test blind: its operand regex never reached the THIRD source of a packed FMA, so the very form it was written for
(`v_pk_fma_f32 d, a, v[0:1], v[0:1] op_sel:[0,0,1] op_sel_hi:[1,0,1]`) passed -- and 16 such instructions each sat in k_irbw2<2,3,false>, k_irbw2<4,3,true>
(both on yolo-fastest's path) and k_conv_first.  test_lint_sees_the_third_source pins the parser."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_lint  # noqa: E402

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def test_lint_sees_the_third_source():
    bad = "\tv_pk_fma_f32 v[6:7], v[2:3], v[0:1], v[0:1] op_sel:[0,0,1] op_sel_hi:[1,0,1]"
    ok = "\tv_pk_fma_f32 v[6:7], v[2:3], v[0:1], v[4:5] op_sel_hi:[1,0,1]"
    assert isa_lint.pk_signature(bad) == ("v_pk_fma_f32", "vvv", True, "op_sel:[0,0,1] op_sel_hi:[1,0,1]")
    assert isa_lint.pk_signature(ok) == ("v_pk_fma_f32", "vvv", False, "op_sel_hi:[1,0,1]")
    assert isa_lint.pk_signature("\tv_pk_mul_f32 v[0:1], s[2:3], v[4:5] op_sel:[1,0]")[1:3] == ("sv", False)
    assert isa_lint.pk_signature("\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], 1.0 op_sel_hi:[1,1,0]")[1] == "vvc"
    forms, flagged, n = isa_lint.census(bad + "\n" + ok + "\n")
    assert n == 2 and len(flagged) == 1 and len(forms) == 2
    errs, _ = isa_lint.lint(bad + "\n")
    assert errs and "one register pair in two slots" in errs[0]
    # an unknown modifier form is an error too (a compiler upgrade that starts emitting one must be looked at)
    errs, _ = isa_lint.lint("\tv_pk_fma_f32 v[6:7], v[2:3], v[0:1], v[4:5] op_sel:[1,1,1] op_sel_hi:[0,0,0]\n")
    assert errs and "NEW packed-fp32 modifier form" in errs[0]


def test_lint_barrier_rule():
    k = "_Z1kv:\n\tbuffer_load_dwordx4 v1, s[0:3], s4 offen lds\n\tbuffer_load_dwordx4 v[2:5], v1, s[0:3], s5 offen\n\t%s\n\ts_barrier\n\ts_endpgm\n"
    assert isa_lint.check_lds_dma(k % "s_waitcnt vmcnt(1) lgkmcnt(0)")[0] == []              # the plain load may stay in flight: loads return in order
    assert isa_lint.check_lds_dma(k % "s_waitcnt vmcnt(0)")[0] == []
    assert isa_lint.check_lds_dma(k % "s_waitcnt vmcnt(2)")[0]                                # would leave the LDS-DMA piece in flight
    assert isa_lint.check_lds_dma(k % "s_waitcnt lgkmcnt(0)")[0]                              # no vmcnt wait at all
    k2 = "_Z1kv:\n\tglobal_load_lds_dwordx4 v[0:1], off\n\ts_waitcnt vmcnt(0)\n\tglobal_load_dword v2, v[0:1], off\n\ts_barrier\n\ts_endpgm\n"
    assert isa_lint.check_lds_dma(k2)[0]                                                      # (conservative: any vector-memory instruction between the wait and the barrier is refused)


def test_lint_reads_past_an_early_endpgm():
    """a kernel with an early return has several s_endpgm; the body must run to .Lfunc_end, or every barrier behind the first one goes unchecked"""
    k = ("_Z1kv:\n\tbuffer_load_dwordx4 v1, s[0:3], s4 offen lds\n\ts_cbranch_execz .LBB0_2\n\ts_endpgm\n.LBB0_2:\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier\n\ts_endpgm\n"
         ".Lfunc_end0:\n\t.size _Z1kv, .Lfunc_end0-_Z1kv\n_Z2k2v:\n\ts_barrier\n\ts_endpgm\n.Lfunc_end1:\n")
    bodies = isa_lint.kernels(k)
    assert list(bodies) == ["_Z1kv", "_Z2k2v"] and bodies["_Z1kv"].count("s_endpgm") == 2 and "_Z2k2v" not in bodies["_Z1kv"]
    errs, seen = isa_lint.check_lds_dma(k)
    assert seen == 1 and len(errs) == 1 and "_Z1kv" in errs[0]


def test_census_rule_is_a_warning_outside_strict_builds():
    new = "\tv_pk_fma_f32 v[6:7], v[2:3], v[0:1], v[4:5] op_sel:[1,1,1] op_sel_hi:[0,0,0]\n"
    errs, facts = isa_lint.lint(new, strict=False)
    assert errs == [] and facts["warnings"] and "NEW packed-fp32 modifier form" in facts["warnings"][0]
    bad = "\tv_pk_fma_f32 v[6:7], v[2:3], v[0:1], v[0:1] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n"
    assert isa_lint.lint(bad, strict=False)[0]                                                  # the hazard rules stay errors


def test_lint_buffer_store_rule():
    """the pair found in round 5 (k_pw_x3t with buffer stores: element 1 of sporadic 16-byte stores wrong), and what must NOT fire"""
    st = "\tbuffer_store_dwordx4 v[2:5], v158, s[28:31], s6 offen nt\n"
    assert isa_lint.check_buffer_store(st + "\tv_fma_f32 v3, v115, v6, v7\n")[0]                       # the original
    assert isa_lint.check_buffer_store(st + "\tv_pk_mul_f32 v[4:5], v[8:9], v[10:11]\n")[0]             # a register pair that overlaps the data
    assert isa_lint.check_buffer_store(st + "\t;;#ASMSTART\n\t;;#ASMEND\n\tv_max_f32_e32 v2, v1, v2\n")[0]    # comments are not wait states
    assert isa_lint.check_buffer_store(st + "\tv_fma_f32 v1, v131, v6, v7\n")[0] == []                  # writes something else
    assert isa_lint.check_buffer_store(st + "\ts_nop 0\n\tv_fma_f32 v3, v115, v6, v7\n")[0] == []       # one wait state is enough
    assert isa_lint.check_buffer_store("\tbuffer_store_dwordx4 v[2:5], v158, s[28:31], 0 offen\n\tv_fma_f32 v3, v115, v6, v7\n")[0] == []   # immediate soffset: hipcc's business
    assert isa_lint.check_buffer_store("\tbuffer_store_dwordx2 v[2:3], v158, s[28:31], s6 offen\n\tv_fma_f32 v3, v115, v6, v7\n")[0] == []  # 8 bytes: no hazard
    errs, facts = isa_lint.lint(st + "\tv_fma_f32 v3, v115, v6, v7\n")
    assert errs and "SGPR soffset" in errs[0] and facts["buffer_store_hazards"] == 1


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    src = os.path.join(ROOT, "ffcnn_amd", "csrc")
    out = str(tmp_path_factory.mktemp("isa") / "k.s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + src, "-S", "--cuda-device-only",
                    os.path.join(src, "ffgpu_kernels.hip"), "-o", out], check=True, cwd=src, timeout=900)
    return open(out).read()


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_kernel_listing_is_clean(isa):
    errs, facts = isa_lint.lint(isa)
    assert facts["npk"] > 1000, facts["npk"]            # the listing really is the kernels' code
    assert facts["lds_dma_kernels"] >= 10, facts        # k_conv_x3 (12 instantiations) + k_pw_gemm32 (2) + k_pw_x3t + ...
    assert not errs, errs[:5]

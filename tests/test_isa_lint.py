"""Compile-time checks of the gfx950 code of ffgpu_kernels.hip (hipcc cross-compiles here, no GPU needed).

(1) No packed fp32 instruction may take ONE register pair in two source slots under op_sel / op_sel_hi modifiers, e.g.
    v_pk_fma_f32 v[6:7], v[2:3], v[0:1], v[0:1] op_sel:[0,0,1] op_sel_hi:[1,0,1]      (hipcc's form of acc * sc + bi with sc, bi in one pair).
While a bf16 MFMA is in flight on the SIMD -- the wave's own or another kernel's -- that instruction sporadically drops its addend on lanes 48-63
(found in k_pw_x3, then in k_pw_mfma under the split-bf16 kernels of neighbouring chains: tools/x3s_exec_race.py).

(2) The kernels that stage weights by LDS-DMA (global_load_lds_dwordx4): every s_barrier must be preceded by an
s_waitcnt vmcnt(0) with no vector-memory instruction in between.  hipcc does not model that these loads write LDS, so a __syncthreads() alone
does not make a wave wait for its pieces; k_conv_x3 lost that race about once in 400 forwards with two chains in flight until the explicit wait
went in (ffgpu_conv_x3.inc, group body).  Runs here (hipcc cross-compiles gfx950 without a GPU)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    src = os.path.join(ROOT, "ffcnn_amd", "csrc")
    out = str(tmp_path_factory.mktemp("isa") / "k.s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + src, "-S", "--cuda-device-only",
                    os.path.join(src, "ffgpu_kernels.hip"), "-o", out], check=True, cwd=src, timeout=900)
    return open(out).read()


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_no_packed_fp32_op_with_one_register_pair_in_two_slots_under_op_sel(isa):
    bad, npk = [], 0
    for line in isa.split("\n"):
        m = re.match(r"\s+(v_pk_\w+_f32)\s+(\S+),\s*(\S+),\s*(\S+)(?:,\s*(\S+))?(.*)", line)
        if not m:
            continue
        npk += 1
        if "op_sel" not in line:
            continue
        regs = [x.rstrip(",") for x in (m.group(3), m.group(4), m.group(5)) if x and x[0] in "vs" and "[" in x]
        if len(regs) != len(set(regs)):
            bad.append(line.strip())
    assert npk > 1000, npk               # the listing really is the kernels' code
    assert not bad, bad[:5]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_every_barrier_of_an_lds_dma_kernel_waits_for_its_pieces(isa):
    s = isa
    seen = 0
    for name in re.findall(r"^(_Z\w+):", s, re.M):
        m = re.search(r"^" + name + r":(.*?)^\s*s_endpgm", s, re.S | re.M)
        if not m or "global_load_lds" not in m.group(1):
            continue
        seen += 1
        lines = [l.strip() for l in m.group(1).split("\n") if l.strip() and not l.strip().startswith(";")]
        for i, l in enumerate(lines):
            if not l.startswith("s_barrier"):
                continue
            j, good = i - 1, False
            while j >= 0:
                if "vmcnt(0)" in lines[j]:
                    good = True
                    break
                if lines[j].startswith(("global_load", "buffer_load", "global_store", "buffer_store")):
                    break
                j -= 1
            assert good, "%s: s_barrier without a preceding vmcnt(0): %s" % (name, " | ".join(lines[max(0, i - 6):i + 1]))
    assert seen >= 10, seen            # k_conv_x3 (12 instantiations) + k_pw_gemm32 (2)

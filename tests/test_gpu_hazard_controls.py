"""POSITIVE controls for the two gfx950 hazards the product works around (tools/lab/hazard_controls.hip, both instruction pairs in inline assembly):
  (A) buffer_store_dwordx4 with an SGPR soffset + a vector write to one of its data registers in the NEXT instruction (tools/isa_lint.py rule 4);
  (B) v_pk_fma_f32 with one register pair in two source slots under op_sel, next to bf16 MFMAs (rule 1).
The controls are EXPECTED to compute wrong results on this hardware: they are marked xfail (non-strict) and print their counts, so a box / ROCm release on which a hazard stops
reproducing shows up as XPASS -- the day the workaround can go -- and "hardware, not compiler" is a file anybody can run.  The NEGATIVE controls (the same pair with the wait
state / with plain v_fma_f32: what the product does) must always be clean."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
LIB = os.path.join(ROOT, "ffcnn_amd", "lib", "libffcnn_hazard_lab.so")


@pytest.fixture(scope="module")
def hz():
    import torch
    if not os.path.exists(LIB):
        pytest.fail("libffcnn_hazard_lab.so is not built (make -C ffcnn_amd/csrc)")
    L = C.CDLL(LIB)
    L.ffhz_store.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.ffhz_pkfma.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    return L, torch


def _store_counts(hz, mode, launches=20):
    L, torch = hz
    blocks, rows = 2048, 32
    out = torch.zeros((blocks * rows, 256, 4), dtype=torch.int32, device="cuda")
    s = torch.cuda.Stream()
    row = np.arange(blocks * rows, dtype=np.uint32)[:, None, None]
    lane = np.arange(256, dtype=np.uint32)[None, :, None]
    e = np.arange(4, dtype=np.uint32)[None, None, :]
    want = torch.from_numpy((((row * np.uint32(2654435761)) ^ (lane * np.uint32(40503)) ^ (e * np.uint32(0x9E3779B9))) | np.uint32(1)).view(np.int32)).cuda()
    poison_v = int(np.array(0xDEADBEEF, np.uint32).view(np.int32))
    poison = other = 0
    lanes = np.zeros(16, np.int64)
    for _ in range(launches):
        out.zero_()
        torch.cuda.synchronize()
        assert L.ffhz_store(out.data_ptr(), blocks, rows, mode, 1, s.cuda_stream) == 0
        bad = out != want
        nb = int(bad.sum())
        if nb:
            np_ = int((bad & (out == poison_v)).sum())
            poison += np_
            other += nb - np_
            lanes += np.bincount((torch.nonzero(bad)[:, 1] % 16).cpu().numpy(), minlength=16)
    return poison, other, launches * blocks * rows * 256, lanes


@pytest.mark.xfail(strict=False, reason="positive control: the gfx950 buffer-store hazard is EXPECTED to corrupt stores here (XPASS = it no longer reproduces on this box / ROCm)")
def test_buffer_store_sgpr_soffset_hazard_reproduces(hz):
    poison, other, total, lanes = _store_counts(hz, 0)
    print("\nhazard (A) buffer_store_dwordx4 + SGPR soffset, data register overwritten in the next instruction: %d of %d stores carry the NEW value of v3 (+ %d other wrong dwords); by lane %% 16: %s"
          % (poison, total, other, lanes.tolist()))
    assert poison == 0 and other == 0, "hazard reproduced: %d corrupted stores of %d" % (poison, total)


def test_buffer_store_with_the_wait_state_is_clean(hz):
    poison, other, total, _ = _store_counts(hz, 1, launches=6)
    assert poison == 0 and other == 0, "%d / %d wrong dwords of %d stores WITH the wait state" % (poison, other, total)


def _pk_counts(hz, mode, launches):
    L, torch = hz
    bad = torch.zeros(16, dtype=torch.int32, device="cuda")
    s = torch.cuda.Stream()
    assert L.ffhz_pkfma(bad.data_ptr(), 1024, 2000, mode, launches, s.cuda_stream) == 0
    return bad.cpu().numpy().view(np.uint32)[:8].reshape(4, 2).astype(np.int64), launches * 1024 * 256 * 2000


@pytest.mark.xfail(strict=False, reason="positive control: v_pk_fma_f32 with one register pair in two op_sel slots next to bf16 MFMAs is EXPECTED to drop its addend sporadically (XPASS = not reproduced in this run)")
def test_pk_fma_op_sel_next_to_bf16_mfma_hazard_reproduces(hz):
    tot = 0
    for mode, what in ((0, "behind 8 bf16 MFMAs of its own wave"), (2, "between the bf16 MFMAs of its own wave"), (3, "in waves whose SIMD neighbours issue the bf16 MFMAs")):
        bad, total = _pk_counts(hz, mode, 200)
        print("\nhazard (B) v_pk_fma_f32 d, a, v[n:n+1], v[n:n+1] op_sel:[0,0,1] op_sel_hi:[1,0,1] %s: wrong results by lane group of 16 [low half, high half]: %s of %d per half"
              % (what, bad.tolist(), total))
        tot += int(bad.sum())
    assert tot == 0, "hazard reproduced"


@pytest.mark.parametrize("mode", [4, 6, 7])
def test_plain_fma_next_to_bf16_mfma_is_clean(hz, mode):
    bad, total = _pk_counts(hz, mode, 100)
    assert int(bad.sum()) == 0, bad.tolist()

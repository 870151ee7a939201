"""The slot tables of the split-bf16 ("X3") expand GEMM (ffcnn_amd/csrc/ffgpu_x3_terms.h, read through the diagnostics library; host
code, no GPU): an fp32 product w * x becomes the partial products w_i * x_j of the operands' three exact bf16 parts, and the kernels
keep the six with i + j <= 2.  For every supported channel count each (i, j, channel pair) with i + j <= 2 must occupy exactly ONE
dword slot of the MFMAs, nothing else may, and the 48-channel form's operand windows must hold the input parts its MFMAs pair with."""
import ctypes as C
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def diag():
    L = C.CDLL(os.path.join(ROOT, "ffcnn_amd", "lib", "libffcnn_hip_diag.so"))
    L.ffgpu_diag_x3_term.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    return L


@pytest.mark.parametrize("ks1", [2, 4, 6, 12])
def test_every_partial_product_once(diag, ks1):
    out = (C.c_int * 3)()
    nm = diag.ffgpu_diag_x3_term(ks1, 0, 0, out)
    assert nm == (ks1 * 6 + 7) // 8
    seen = {}
    for m in range(nm):
        for d in range(4):
            assert diag.ffgpu_diag_x3_term(ks1, m, d, out) == nm
            wp, xp, pair = out[0], out[1], out[2]
            if wp < 0:
                continue
            assert 0 <= wp <= 2 and 0 <= xp <= 2 and wp + xp <= 2 and 0 <= pair < ks1 // 2, (m, d, wp, xp, pair)
            assert (wp, xp, pair) not in seen, "partial product twice: %r at %r and %r" % ((wp, xp, pair), seen[(wp, xp, pair)], (m, d))
            seen[(wp, xp, pair)] = (m, d)
    want = {(i, j, p) for i in range(3) for j in range(3) if i + j <= 2 for p in range(ks1 // 2)}
    assert set(seen) == want
    assert diag.ffgpu_diag_x3_term(ks1, nm, 0, out) == -1 and diag.ffgpu_diag_x3_term(5, 0, 0, out) == -1


def test_xl_operand_windows(diag):
    """48 input channels: the lane's LDS image is x0 a-f | x1 a-f | x2 a-f | x0 e f (20 dwords); MFMA m multiplies the 16-byte window
    ffgpu_diag_xl_op(m) -- its four dwords must be exactly the (input part, pair) the table pairs with the MFMA's weight dwords"""
    image = [(0, p) for p in range(6)] + [(1, p) for p in range(6)] + [(2, p) for p in range(6)] + [(0, 4), (0, 5)]
    out = (C.c_int * 3)()
    for m in range(9):
        op = diag.ffgpu_diag_xl_op(m)
        assert 0 <= op <= 4
        for d in range(4):
            diag.ffgpu_diag_x3_term(12, m, d, out)
            assert image[4 * op + d] == (out[1], out[2]), (m, d)

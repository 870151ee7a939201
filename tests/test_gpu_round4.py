"""Round 4: fp32-equivalent products on the bf16 matrix cores ("X3": each operand = three exact bf16 parts, six partial products, fp32
accumulation; ffcnn_amd/csrc/ffgpu_irb_wave.inc, ffgpu_pw_x3.inc).  The claim under test is that the split form is NOT a reduced
precision: it must sit as close to the oracle (conv-v0.c restated, oracle/) as the fp32-MFMA form of the same kernel does, launch after
launch -- the second half matters because the one defect of the round was timing dependent (a packed FMA with op_sel broadcasts right
behind bf16 MFMAs dropped its addend on a quarter of a wave about once in 10^4 tiles; only back-to-back launches showed it)."""
import os

import numpy as np
import pytest

from test_gpu_kernels import make_filter


@pytest.fixture(scope="module")
def env():
    import torch
    from ffcnn_amd import capi
    from oracle import orc
    orc.build()
    capi.lib()
    return capi, torch, orc


# (ic, ec, oc, N, H, W): the three families that run the split expand GEMM by default + the two that can (FFGPU_IRBW_X3 bit 0)
X3_BLOCKS = [(16, 96, 16, 24, 40, 40), (24, 136, 24, 24, 20, 20), (48, 224, 48, 24, 10, 10), (8, 48, 8, 24, 40, 40), (8, 32, 8, 8, 80, 80)]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", X3_BLOCKS)
def test_x3_fused_block_is_an_fp32_reorder(env, shape, monkeypatch):
    capi, torch, orc = env
    ic, ec, oc, N, H, W = shape
    rng = np.random.default_rng(hash(shape) & 0xffff)
    x = rng.uniform(-1, 1, (ic * N, H, W)).astype(np.float32)
    f1, fd, f2 = make_filter(rng, ec, ic), make_filter(rng, ec, 9), make_filter(rng, oc, ec)
    res = rng.uniform(-1, 1, (oc * N, H, W)).astype(np.float32)
    t = [torch.from_numpy(a).cuda() for a in (x, f1, fd, f2, res)]
    outs = {}
    for mode in ("0", "15"):
        monkeypatch.setenv("FFGPU_IRBW_X3", mode)
        out = torch.full((oc * N, H, W), float("nan"), device="cuda")
        # 40 launches back to back, the last one's output is checked (a timing-dependent slip shows as a handful of wrong tiles)
        capi.irb_dev(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), t[4].data_ptr(), out.data_ptr(), N, W, H, ic, ec, oc, 1, warmup=0, iters=40)
        torch.cuda.synchronize()
        outs[mode] = out.cpu().numpy().reshape(oc, N, H, W)
    assert not np.isnan(outs["15"]).any()
    xf, rf = x.reshape(ic, N, H, W), res.reshape(oc, N, H, W)
    worst = {"0": 0.0, "15": 0.0}
    for n in range(0, N, 5):
        o1 = orc.groupconv(np.ascontiguousarray(xf[:, n]), f1, 1, 0, 1, 1, 2)
        o2 = orc.groupconv(o1, fd, ec, 1, 1, 3, 2)
        o3 = orc.shortcut(orc.groupconv(o2, f2, 1, 0, 1, 1, 0), np.ascontiguousarray(rf[:, n]), 0)
        for k in outs:
            worst[k] = max(worst[k], float(np.abs(outs[k][:, n] - o3).max()))
    scale = float(np.abs(outs["0"]).max())
    # both forms within a few fp32 ulps of the block's largest value of the oracle, the split form no further than twice the fp32 form
    assert worst["0"] <= 8e-7 * scale + 1e-6 and worst["15"] <= 8e-7 * scale + 1e-6, (worst, scale)
    assert worst["15"] <= 2.0 * worst["0"] + 1e-6, worst
    assert float(np.abs(outs["15"] - outs["0"]).max()) <= 8e-7 * scale + 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(256, 512, 64, 20, 20), (120, 255, 32, 20, 20), (96, 255, 32, 10, 10), (72, 300, 21, 20, 20)])
@pytest.mark.parametrize("mt", [4, 2, 1])
def test_pw_x3_back_to_back(env, shape, mt, monkeypatch):
    """k_pw_x3 with leaky activation, random scale' / bias', 30 launches back to back: the regression test of the dropped-addend defect"""
    capi, torch, orc = env
    monkeypatch.setenv("FFGPU_PWX3_MT", str(mt))
    ic, oc, N, H, W = shape
    rng = np.random.default_rng(hash(shape) & 0xffff)
    x = rng.uniform(-1, 1, (ic * N, H, W)).astype(np.float32)
    f = make_filter(rng, oc, ic)
    dx, df = torch.from_numpy(x).cuda(), torch.from_numpy(f).cuda()
    ref = torch.empty((oc * N, H, W), device="cuda")
    capi.groupconv_dev(dx.data_ptr(), df.data_ptr(), ref.data_ptr(), N, W, H, ic, 1, 0, 1, 1, oc, 2, 0, capi.FFGPU.K_PW_MFMA, None)
    for rep in range(3):
        y = torch.full((oc * N, H, W), float("nan"), device="cuda")
        capi.groupconv_time_dev(dx.data_ptr(), df.data_ptr(), y.data_ptr(), N, W, H, ic, 1, 0, 1, 1, oc, act=2, variant=capi.FFGPU.K_PW_X3, warmup=0, iters=30)
        torch.cuda.synchronize()
        d = (y - ref).abs()
        assert not torch.isnan(y).any() and float(d.max()) <= 1e-4, "rep %d: %d outputs off, max |d| %.3g" % (rep, int((d > 1e-4).sum()), float(d.max()))

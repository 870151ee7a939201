#!/usr/bin/env python3
"""fused blocks with the expand GEMM as exact split-bf16 products (FFGPU_IRBW_X3=1, default) against the fp32-MFMA form (=0) and
the oracle: max |d| and max |d| / (1e-3 + 1e-3 |ref|), per shape, + launch times (single launch, one chain)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
from oracle import orc
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_kernels import make_filter

SHAPES = [(8, 32, 8, 1, 64, 80, 80, True), (8, 48, 8, 1, 64, 40, 40, True), (8, 48, 16, 1, 64, 40, 40, False), (16, 96, 16, 1, 64, 40, 40, True),
          (24, 136, 24, 1, 64, 20, 20, True), (48, 224, 48, 1, 64, 10, 10, True), (4, 24, 8, 2, 64, 160, 160, False)]
for shape in SHAPES:
    ic, ec, oc, stride, N, H, W, use_res = shape
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, (ic * N, H, W)).astype(np.float32)
    f1, fd, f2 = make_filter(rng, ec, ic), make_filter(rng, ec, 9), make_filter(rng, oc, ec)
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    res = rng.uniform(-1, 1, (oc * N, OH, OW)).astype(np.float32)
    t = [torch.from_numpy(a).cuda() for a in (x, f1, fd, f2, res)]
    outs, us = {}, {}
    for x3 in ("0", "1"):
        os.environ["FFGPU_IRBW_X3"] = "15" if x3 == "1" else "0"
        out = torch.full((oc * N, OH, OW), float("nan"), device="cuda")
        capi.irb_dev(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), t[4].data_ptr() if use_res else None, out.data_ptr(), N, W, H, ic, ec, oc, stride)
        us[x3] = capi.irb_dev(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), t[4].data_ptr() if use_res else None, out.data_ptr(), N, W, H, ic, ec, oc, stride, warmup=5, iters=50)
        torch.cuda.synchronize()
        outs[x3] = out.cpu().numpy().reshape(oc, N, OH, OW)
    xf, rf = x.reshape(ic, N, H, W), res.reshape(oc, N, OH, OW)
    worst = {"0": (0, 0), "1": (0, 0)}
    for n in range(0, N, 16):
        o1 = orc.groupconv(np.ascontiguousarray(xf[:, n]), f1, 1, 0, 1, 1, 2)
        o2 = orc.groupconv(o1, fd, ec, 1, stride, 3, 2)
        o3 = orc.groupconv(o2, f2, 1, 0, 1, 1, 0)
        if use_res:
            o3 = orc.shortcut(o3, np.ascontiguousarray(rf[:, n]), 0)
        for k in outs:
            d = np.abs(outs[k][:, n] - o3)
            worst[k] = (max(worst[k][0], float(d.max())), max(worst[k][1], float((d / (1e-3 + 1e-3 * np.abs(o3))).max())))
    dd = float(np.abs(outs["0"] - outs["1"]).max())
    print("%-34s fp32 MFMA: %.1f us, |d| %.2e (tol ratio %.4f)   split bf16: %.1f us, |d| %.2e (tol ratio %.4f)   between them %.2e   |ref| max %.1f" %
          (shape, us["0"], worst["0"][0], worst["0"][1], us["1"], worst["1"][0], worst["1"][1], dd, float(np.abs(o3).max())))

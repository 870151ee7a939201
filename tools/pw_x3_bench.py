#!/usr/bin/env python3
"""pointwise layers: the split-bf16 kernel (k_pw_x3) against the fp32-MFMA kernels (k_pw_gemm32 / k_pw_mfma), us per launch alone,
BASELINE config[2] and the head layers of yolo-fastest at batch 64; error against a float64 reference on a sample"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
SHAPES = [(256, 512, 256, 20, 20), (120, 120, 64, 20, 20), (120, 255, 64, 20, 20), (192, 96, 64, 10, 10), (96, 96, 64, 10, 10), (96, 255, 64, 10, 10),
          (136, 24, 64, 20, 20), (224, 48, 64, 10, 10), (96, 16, 64, 40, 40)]
mts = [int(a) for a in sys.argv[1:]] or [4, 2, 1]
for ic, oc, N, H, W in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(1235)
    x = torch.rand((ic * N, H, W), device="cuda", generator=g) * 2 - 1
    y = torch.empty((oc * N, H, W), device="cuda")
    k4 = (ic + 3) & ~3
    filt = torch.zeros((oc, k4 + 4), device="cuda")
    filt[:, :ic] = torch.rand((oc, ic), device="cuda", generator=g) - 0.5
    filt[:, k4] = torch.rand((oc,), device="cuda", generator=g) + 0.5
    filt[:, k4 + 1] = torch.rand((oc,), device="cuda", generator=g) * 0.2 - 0.1
    torch.cuda.synchronize()
    flops = 2.0 * oc * ic * N * H * W
    os.environ["FFGPU_PW_X3"] = "0"
    name = capi.kernel_name(N, W, H, ic, 1, 0, 1, 1, oc)
    us0 = capi.groupconv_time_dev(x.data_ptr(), filt.data_ptr(), y.data_ptr(), N, W, H, ic, 1, 0, 1, 1, oc, act=2, warmup=10, iters=50)
    y0 = y.clone()
    os.environ["FFGPU_PW_X3"] = "1"
    # float64 reference on the first 256 pixels of frame 0
    xs = x.view(ic, N * H * W)[:, :256].double(); ref = (filt[:, :ic].double() @ xs) * filt[:, k4:k4 + 1].double() + filt[:, k4 + 1:k4 + 2].double()
    ref = torch.where(ref > 0, ref, 0.1 * ref)
    e0 = float((y0.view(oc, -1)[:, :256].double() - ref).abs().max())
    out = "%3d -> %3d on %2dx%2d x %3d: %-8s %7.1f us (%5.1f TF, |d| %.1e)" % (ic, oc, H, W, N, name, us0, flops / us0 / 1e6, e0)
    for mt in mts:
        os.environ["FFGPU_PWX3_MT"] = str(mt)
        us = capi.groupconv_time_dev(x.data_ptr(), filt.data_ptr(), y.data_ptr(), N, W, H, ic, 1, 0, 1, 1, oc, act=2, variant=capi.FFGPU.K_PW_X3, warmup=10, iters=50)
        e = float((y.view(oc, -1)[:, :256].double() - ref).abs().max())
        d = float((y - y0).abs().max())
        if d > 1e-3:                                            # where? (output-channel block of 16, pixel tile of 64)
            bad = ((y - y0).abs().view(oc, -1) > 1e-3)
            rows = sorted(set((bad.any(1).nonzero().flatten() // 16).tolist()))[:12]
            cols = sorted(set((bad.any(0).nonzero().flatten() // 64).tolist()))
            out += " BAD row blocks %s, %d pixel tiles (first %s)" % (rows, len(cols), cols[:8])
        out += " | x3 MT=%d %7.1f us (%5.1f TF, |d| %.1e, vs fp32 kernel %.1e)" % (mt, us, flops / us / 1e6, e, d)
    print(out, flush=True)

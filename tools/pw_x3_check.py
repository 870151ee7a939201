#!/usr/bin/env python3
"""k_pw_x3 run repeatedly on one input: how many (row block, pixel tile) cells differ from the fp32 kernel's output by more than 1e-3,
run by run (a correct kernel: 0 every time; a timing-dependent hazard: a few cells, different ones each run)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
shapes = [(256, 512, 256, 20, 20), (120, 255, 64, 20, 20), (96, 96, 64, 10, 10)]
for ic, oc, N, H, W in shapes:
    g = torch.Generator(device="cuda").manual_seed(1235)
    x = torch.rand((ic * N, H, W), device="cuda", generator=g) * 2 - 1
    y = torch.empty((oc * N, H, W), device="cuda")
    k4 = (ic + 3) & ~3
    filt = torch.zeros((oc, k4 + 4), device="cuda")
    filt[:, :ic] = torch.rand((oc, ic), device="cuda", generator=g) - 0.5
    filt[:, k4] = 1.0
    os.environ["FFGPU_PW_X3"] = "0"
    capi.groupconv_dev(x.data_ptr(), filt.data_ptr(), y.data_ptr(), N, W, H, ic, 1, 0, 1, 1, oc, 0, 0, capi.FFGPU.K_AUTO, None)
    torch.cuda.synchronize()
    y0 = y.clone()
    for mt in (4, 2, 1):
        os.environ["FFGPU_PWX3_MT"] = str(mt)
        counts = []
        for rep in range(12):
            y.fill_(float("nan"))
            capi.groupconv_dev(x.data_ptr(), filt.data_ptr(), y.data_ptr(), N, W, H, ic, 1, 0, 1, 1, oc, 0, 0, capi.FFGPU.K_PW_X3, None)
            torch.cuda.synchronize()
            bad = ~((y - y0).abs().view(oc, -1) <= 1e-3)
            counts.append(int(bad.sum()))
        print("%d -> %d x %d  MT=%d: wrong outputs per run %s" % (ic, oc, N, mt, counts), flush=True)

#!/usr/bin/env python3
"""Sweep planes-per-item / band height of k_dw_lds on the 5x5 depthwise layers of the heads (batch 64)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
N = 64
for (C, HW) in ((120, 20), (96, 10)):
    x = torch.rand((C * N, HW, HW), device="cuda") - 0.5
    y = torch.empty_like(x)
    f = torch.rand((C, 32), device="cuda") - 0.5
    res = []
    for npl in (0, 1, 2, 4, 8, 16):
        for bh in (0, 5, 10):
            os.environ.update(FFGPU_DWL_NPL=str(npl), FFGPU_DWL_BH=str(bh))
            try:
                us = capi.groupconv_time_dev(x.data_ptr(), f.data_ptr(), y.data_ptr(), N, HW, HW, C, C, 2, 1, 5, C, act=2, warmup=3, iters=20)
            except RuntimeError as e:
                continue
            res.append((us, npl, bh))
    res.sort()
    print("dw5 %dx%dx%d: " % (HW, HW, C) + "  ".join("NPL%d/BH%d %.1f" % (n, b, u) for u, n, b in res))

#!/usr/bin/env python3
"""The heads' depthwise 5x5 + pointwise pairs: fused kernel (k_dwpw) vs the two separate launches, batch 64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
s = torch.cuda.Stream()
for (C, OC, N, H, W, fs) in ((120, 120, 64, 20, 20, 5), (96, 96, 64, 10, 10, 5), (120, 120, 32, 20, 20, 5)):
    x = torch.rand((C * N, H, W), device="cuda") - 0.5
    mid = torch.empty_like(x)
    y = torch.empty((OC * N, H, W), device="cuda")
    k4d, k4p = (fs * fs + 3) & ~3, (C + 3) & ~3
    fd = torch.zeros((C, k4d + 4), device="cuda"); fd[:, :fs * fs] = torch.rand((C, fs * fs), device="cuda") - 0.5; fd[:, k4d] = 1.0
    fp = torch.zeros((OC, k4p + 4), device="cuda"); fp[:, :C] = torch.rand((OC, C), device="cuda") - 0.5; fp[:, k4p] = 1.0
    fused = capi.dwpw_dev(x.data_ptr(), fd.data_ptr(), fp.data_ptr(), y.data_ptr(), N, W, H, C, OC, fs, warmup=5, iters=50, stream=s.cuda_stream)
    t1 = capi.groupconv_time_dev(x.data_ptr(), fd.data_ptr(), mid.data_ptr(), N, W, H, C, C, fs // 2, 1, fs, C, act=2, warmup=5, iters=50, stream=s.cuda_stream)
    t2 = capi.groupconv_time_dev(mid.data_ptr(), fp.data_ptr(), y.data_ptr(), N, W, H, C, 1, 0, 1, 1, OC, act=0, warmup=5, iters=50, stream=s.cuda_stream)
    print("dw%dx%d + pw %d->%d on %dx%d, batch %d: fused %.1f us; separate %.1f + %.1f = %.1f us" % (fs, fs, C, OC, W, H, N, fused, t1, t2, t1 + t2))

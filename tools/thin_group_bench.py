#!/usr/bin/env python3
"""Grouped convolutions with thin groups (2..7 input channels per group): k_conv_thin against the one-thread-per-output generic kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
s = torch.cuda.Stream()
for ic, oc, groups, N, H, W, fs in [(64, 64, 16, 16, 52, 52, 3), (128, 128, 32, 16, 26, 26, 3), (48, 48, 8, 64, 40, 40, 3), (96, 96, 24, 64, 20, 20, 5), (6, 32, 1, 16, 104, 104, 1)]:
    K = fs * fs * ic // groups
    x = torch.rand((ic * N, H, W), device="cuda") * 2 - 1
    y = torch.empty((oc * N, H, W), device="cuda")
    filt = torch.zeros((oc, ((K + 3) & ~3) + 4), device="cuda")
    filt[:, :K] = torch.rand((oc, K), device="cuda") - 0.5
    filt[:, (K + 3) & ~3] = 1.0
    t = {}
    for name, var in (("thin", capi.FFGPU.K_AUTO), ("generic", capi.FFGPU.K_GENERIC)):
        t[name] = capi.groupconv_time_dev(x.data_ptr(), filt.data_ptr(), y.data_ptr(), N, W, H, ic, groups, fs // 2, 1, fs, oc, act=2, variant=var, warmup=3, iters=20, stream=s.cuda_stream)
    by = 4.0 * (ic + oc) * N * H * W
    print("%3d->%3d groups %2d %3dx%3d N=%2d %dx%d: %s %8.1f us (%5.0f GB/s algorithmic)  generic %8.1f us  x%.1f" % (
        ic, oc, groups, H, W, N, fs, fs, capi.kernel_name(N, W, H, ic, groups, fs // 2, 1, fs, oc), t["thin"], by / t["thin"] / 1e3, t["generic"], t["generic"] / t["thin"]))

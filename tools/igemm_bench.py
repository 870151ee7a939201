#!/usr/bin/env python3
"""Dense KxK layers of a yolov3-tiny-style net: implicit-GEMM MFMA kernel vs the generic one-thread-per-output kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
SHAPES = [(16, 32, 16, 208, 208, 3, 1, 1), (32, 64, 16, 104, 104, 3, 1, 1), (64, 128, 16, 52, 52, 3, 1, 1), (128, 256, 16, 26, 26, 3, 1, 1),
          (256, 512, 16, 13, 13, 3, 1, 1), (512, 1024, 16, 13, 13, 3, 1, 1), (256, 512, 64, 13, 13, 3, 1, 1)]
s = torch.cuda.Stream()
for (ic, oc, N, H, W, fs, st, pad) in SHAPES:
    x = torch.rand((ic * N, H, W), device="cuda") - 0.5
    K = fs * fs * ic
    filt = torch.zeros((oc, ((K + 3) & ~3) + 4), device="cuda")
    filt[:, :K] = (torch.rand((oc, K), device="cuda") - 0.5) / K ** 0.5
    filt[:, (K + 3) & ~3] = 1.0
    OH, OW = (H + 2 * pad - fs) // st + 1, (W + 2 * pad - fs) // st + 1
    y = torch.empty((oc * N, OH, OW), device="cuda")
    fl = 2.0 * K * oc * N * OH * OW
    t = {}
    for name, v, it in (("igemm", capi.FFGPU.K_IGEMM, 20), ("generic", capi.FFGPU.K_GENERIC, 3)):
        t[name] = capi.groupconv_time_dev(x.data_ptr(), filt.data_ptr(), y.data_ptr(), N, W, H, ic, 1, pad, st, fs, oc, act=2, variant=v,
                                          warmup=2, iters=it, stream=s.cuda_stream)
    print("%4d->%4d %3dx%3d N=%2d %dx%d: igemm %8.1f us (%5.1f TFLOP/s, %.2f of the fp32 matrix peak)  generic %9.1f us  x%.1f" %
          (ic, oc, W, H, N, fs, fs, t["igemm"], fl / t["igemm"] / 1e6, fl / t["igemm"] / 1e6 / 157.3, t["generic"], t["generic"] / t["igemm"]))

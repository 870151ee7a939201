#!/usr/bin/env python3
"""A pointwise shape on both MFMA kernels: python tools/pw_variants.py N ic oc H W [...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
args = [int(v) for v in sys.argv[1:]] or [64, 120, 255, 20, 20]
s = torch.cuda.Stream()
for k in range(0, len(args), 5):
    N, ic, oc, H, W = args[k:k + 5]
    x = torch.rand((ic * N, H, W), device="cuda") - 0.5
    y = torch.empty((oc * N, H, W), device="cuda")
    k4 = (ic + 3) & ~3
    filt = torch.zeros((oc, k4 + 4), device="cuda"); filt[:, :ic] = torch.rand((oc, ic), device="cuda") - 0.5; filt[:, k4] = 1.0
    out = []
    for name, v in (("pw_mfma", capi.FFGPU.K_PW_MFMA), ("pw_gemm", capi.FFGPU.K_PW_GEMM)):
        try:
            for _ in range(2):
                us = capi.groupconv_time_dev(x.data_ptr(), filt.data_ptr(), y.data_ptr(), N, W, H, ic, 1, 0, 1, 1, oc, act=0, variant=v, warmup=5, iters=40, stream=s.cuda_stream)
            out.append("%s %.1f us" % (name, us))
        except RuntimeError as e:
            out.append("%s n/a" % name)
    print("%d->%d %dx%d N=%d: %s (auto: %s)" % (ic, oc, W, H, N, ", ".join(out), capi.kernel_name(N, W, H, ic, 1, 0, 1, 1, oc)))

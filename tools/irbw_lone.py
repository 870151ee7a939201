#!/usr/bin/env python3
"""Run fused blocks with so few frames that every wave has a SIMD to itself: the kernel time is one wave's serial chain."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
os.environ["FFGPU_IRBW_G"] = "1"
for (ic, ec, oc, s, HW, res) in [(8, 32, 8, 1, 80, True), (8, 48, 8, 1, 40, True), (16, 96, 16, 1, 40, True), (24, 136, 24, 1, 20, True), (48, 224, 48, 1, 10, True)]:
    for N in ((1,) if os.environ.get("FFGPU_IRB_TRACE") else (1, 4, 16, 64)):
        OH = (HW - 1) // s + 1
        x = torch.rand((ic * N, HW, HW), device="cuda") - 0.5
        f1 = torch.rand((ec, ((ic + 3) & ~3) + 4), device="cuda") - 0.5
        fd = torch.rand((ec, 16), device="cuda") - 0.5
        f2 = torch.rand((oc, ((ec + 3) & ~3) + 4), device="cuda") - 0.5
        r = torch.rand((oc * N, OH, OH), device="cuda")
        out = torch.empty((oc * N, OH, OH), device="cuda")
        us = capi.irb_dev(x.data_ptr(), f1.data_ptr(), fd.data_ptr(), f2.data_ptr(), r.data_ptr() if res else None, out.data_ptr(),
                          N, HW, HW, ic, ec, oc, s, warmup=1 if os.environ.get("FFGPU_IRB_TRACE") else 3, iters=1 if os.environ.get("FFGPU_IRB_TRACE") else 20)
        print("%dx%d %d->%d->%d N=%d: %.1f us (%d groups)" % (HW, HW, ic, ec, oc, N, us, (ec + 15) // 16))

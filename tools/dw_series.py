#!/usr/bin/env python3
"""Per-launch durations of the roofline depthwise kernel over a long back-to-back series (rocprofv3 --kernel-trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
N, C, H, W = 64, 64, 320, 320
x = torch.rand((C * N, H, W), device="cuda") * 2 - 1
y = torch.empty_like(x)
f = torch.zeros((C, 16), device="cuda")
f[:, :9] = torch.rand((C, 9), device="cuda") - 0.5
f[:, 12] = 1.0
torch.cuda.synchronize()
s = torch.cuda.Stream()
us = capi.groupconv_time_dev(x.data_ptr(), f.data_ptr(), y.data_ptr(), N, W, H, C, C, 1, 1, 3, C, act=2, warmup=10, iters=100, stream=s.cuda_stream)
print("mean us", us)

#!/usr/bin/env python3
"""Launch ONLY the BASELINE config[1] depthwise kernel plus a bare copy of the same bytes (calibration),
for the rocprofv3 --pmc passes (tools/pmc_traffic.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
N, C, H, W = 64, 64, 320, 320
g = torch.Generator(device="cuda").manual_seed(1234)
x = torch.rand((C * N, H, W), device="cuda", generator=g) * 2 - 1
y = torch.empty_like(x)
f = torch.zeros((C, 16), device="cuda")
f[:, :9] = torch.rand((C, 9), device="cuda", generator=g) - 0.5
f[:, 12] = 1.0
torch.cuda.synchronize()
s = torch.cuda.Stream()
nbytes = x.numel() * 4
for _ in range(3):
    capi.diag().ffgpu_membench(y.data_ptr(), x.data_ptr(), nbytes, 0, 1024, 1, s.cuda_stream)       # 3 launches each (2 warm + 1)
    capi.groupconv_dev(x.data_ptr(), f.data_ptr(), y.data_ptr(), N, W, H, C, C, 1, 1, 3, C, act=2, stream=s.cuda_stream)
torch.cuda.synchronize()
print("bytes per launch (read) =", nbytes, "(written) =", nbytes)

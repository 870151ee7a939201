#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
N, ic, oc, H, W = 256, 256, 512, 20, 20
x = torch.rand((ic * N, H, W), device="cuda") * 2 - 1
y = torch.empty((oc * N, H, W), device="cuda")
f = torch.zeros((oc, ic + 4), device="cuda"); f[:, :ic] = torch.rand((oc, ic), device="cuda") - 0.5; f[:, ic] = 1.0
s = torch.cuda.Stream()
us = capi.groupconv_time_dev(x.data_ptr(), f.data_ptr(), y.data_ptr(), N, W, H, ic, 1, 0, 1, 1, oc, act=2, warmup=1, iters=3, stream=s.cuda_stream)
print(us)

#!/usr/bin/env python3
"""In-kernel timeline of every fused block of the net (FFGPU_IRB_TRACE; eager executor)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
B = 64
net = capi.Net()
x = torch.rand((B, 3, 320, 320), device="cuda")
ex = net.executor(B, capi.FFGPU.NO_GRAPH)
ex.forward_dev(x.data_ptr()); torch.cuda.synchronize()
os.environ["FFGPU_IRB_TRACE"] = "1"
os.environ["FFGPU_VERBOSE_IRB"] = "1"
ex.forward_dev(x.data_ptr()); torch.cuda.synchronize()

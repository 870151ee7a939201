#!/usr/bin/env python3
"""In-kernel timeline of every fused block of the net (FFGPU_IRB_TRACE; eager executor)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
B = 64
net = capi.Net()
x = torch.rand((B, 3, 320, 320), device="cuda")
# IRB_TRACE_CONCURRENT=1: the plans the bench runs (four chains in flight: FFGPU_CONCURRENT), still one chain, eager
ex = net.executor(B, capi.FFGPU.NO_GRAPH | (capi.FFGPU.CONCURRENT if os.environ.get("IRB_TRACE_CONCURRENT") == "1" else 0))
ex.forward_dev(x.data_ptr()); torch.cuda.synchronize()
os.environ["FFGPU_IRB_TRACE"] = "1"
os.environ["FFGPU_VERBOSE_IRB"] = "1"
ex.forward_dev(x.data_ptr()); torch.cuda.synchronize()

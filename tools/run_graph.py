#!/usr/bin/env python3
"""A few graph replays of the batched executor (for rocprofv3 --kernel-trace): python tools/run_graph.py [batch] [reps] [flags]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
net = capi.Net()
ex = net.executor(B, flags)
x = torch.rand((B, 3, 320, 320), device="cuda")
torch.cuda.synchronize()
for _ in range(reps):
    ex.forward_dev(x.data_ptr())
torch.cuda.synchronize()

#!/usr/bin/env python3
"""Launch one fused-block shape a few times (for rocprofv3 --pmc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
ic, ec, oc, s, HW, res = [int(a) for a in sys.argv[1:7]]
N = 64
OH = (HW - 1) // s + 1
x = torch.rand((ic * N, HW, HW), device="cuda") - 0.5
f1 = torch.rand((ec, ((ic + 3) & ~3) + 4), device="cuda") - 0.5
fd = torch.rand((ec, 16), device="cuda") - 0.5
f2 = torch.rand((oc, ((ec + 3) & ~3) + 4), device="cuda") - 0.5
r = torch.rand((oc * N, OH, OH), device="cuda")
out = torch.empty((oc * N, OH, OH), device="cuda")
us = capi.irb_dev(x.data_ptr(), f1.data_ptr(), fd.data_ptr(), f2.data_ptr(), r.data_ptr() if res else None, out.data_ptr(),
                  N, HW, HW, ic, ec, oc, s, warmup=1, iters=3)
print("us", us)

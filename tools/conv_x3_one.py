#!/usr/bin/env python3
"""One dense 3x3 layer through AUTO, `iters` launches (for rocprofv3 --pmc passes: tools/pmc.sh conv_x3 "<counters>" -- python tools/conv_x3_one.py ic oc N H W)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
ic, oc, N, H, W = (int(a) for a in sys.argv[1:6])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 20
os.environ.setdefault("FFGPU_IGX3_MIN_WGS", "1")
x = torch.rand((ic * N, H, W), device="cuda") - 0.5
K = 9 * ic
filt = torch.zeros((oc, ((K + 3) & ~3) + 4), device="cuda")
filt[:, :K] = (torch.rand((oc, K), device="cuda") - 0.5) / K ** 0.5
filt[:, (K + 3) & ~3] = 1.0
y = torch.empty((oc * N, H, W), device="cuda")
s = torch.cuda.Stream()
us = capi.groupconv_time_dev(x.data_ptr(), filt.data_ptr(), y.data_ptr(), N, W, H, ic, 1, 1, 1, 3, oc, act=2, variant=0, warmup=2, iters=iters, stream=s.cuda_stream)
fl = 2.0 * K * oc * N * H * W
print("%s %d->%d %dx%d N=%d: %.1f us  %.1f TFLOP/s  %.2f of 157.3" % (capi.kernel_name(N, W, H, ic, 1, 1, 1, 3, oc), ic, oc, W, H, N, us, fl / us / 1e6, fl / us / 1e6 / 157.3))

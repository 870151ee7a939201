#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
N, HW = 64, 160
for (ic, ec, oc, res) in [(8, 8, 4, False), (4, 8, 4, True)]:
    x = torch.rand((ic * N, HW, HW), device="cuda") - 0.5
    f1 = torch.rand((ec, ((ic + 3) & ~3) + 4), device="cuda") - 0.5
    fd = torch.rand((ec, 16), device="cuda") - 0.5
    f2 = torch.rand((oc, ((ec + 3) & ~3) + 4), device="cuda") - 0.5
    r = torch.rand((oc * N, HW, HW), device="cuda")
    out = torch.empty((oc * N, HW, HW), device="cuda")
    for band in (4, 8, 16, 32):
        os.environ["FFGPU_THIN_BAND"] = str(band)
        us = min(capi.irb_dev(x.data_ptr(), f1.data_ptr(), fd.data_ptr(), f2.data_ptr(), r.data_ptr() if res else None, out.data_ptr(),
                              N, HW, HW, ic, ec, oc, 1, warmup=2, iters=10) for _ in range(3))
        print("thin %d->%d->%d res=%d band=%2d: %.1f us" % (ic, ec, oc, res, band, us))

#!/usr/bin/env python3
"""Sweep the tile shape of the fused block kernel for every distinct block of yolo-fastest (batch 64)."""
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi

SHAPES = [(8, 8, 4, 1, 160, False), (4, 8, 4, 1, 160, True), (4, 24, 8, 2, 160, False), (8, 32, 8, 1, 80, True), (8, 32, 8, 2, 80, False),
          (8, 48, 8, 1, 40, True), (16, 96, 16, 1, 40, True), (16, 96, 24, 2, 40, False), (24, 136, 24, 1, 20, True),
          (24, 136, 48, 2, 20, False), (48, 224, 48, 1, 10, True)]
N = 64
only = [int(a) for a in sys.argv[1:]]
for si, (ic, ec, oc, s, HW, res) in enumerate(SHAPES):
    if only and si not in only:
        continue
    OH = (HW - 1) // s + 1
    x = torch.rand((ic * N, HW, HW), device="cuda") - 0.5
    f1 = torch.rand((ec, ((ic + 3) & ~3) + 4), device="cuda") - 0.5
    fd = torch.rand((ec, 16), device="cuda") - 0.5
    f2 = torch.rand((oc, ((ec + 3) & ~3) + 4), device="cuda") - 0.5
    r = torch.rand((oc * N, OH, OH), device="cuda")
    out = torch.empty((oc * N, OH, OH), device="cuda")
    results = []
    tws = sorted({8, 16, 32, OH} if OH <= 24 else {8, 16, 32})
    ths = sorted({2, 4, 5, 8, 10, 16, OH} if OH <= 32 else {2, 4, 5, 8, 10, 16})
    for tw, th, nf in itertools.product(tws, ths, (1, 2, 4)):
        if tw > OH or th > OH or (nf > 1 and (tw < OH or th < OH)):
            continue
        os.environ.update(FFGPU_IRB_TW=str(tw), FFGPU_IRB_TH=str(th), FFGPU_IRB_NF=str(nf))
        try:
            us = capi.irb_dev(x.data_ptr(), f1.data_ptr(), fd.data_ptr(), f2.data_ptr(), r.data_ptr() if res else None, out.data_ptr(),
                              N, HW, HW, ic, ec, oc, s, warmup=2, iters=8)
        except RuntimeError:
            continue
        results.append((us, tw, th, nf))
    for k in ("FFGPU_IRB_TW", "FFGPU_IRB_TH", "FFGPU_IRB_NF"):
        os.environ.pop(k, None)
    auto = capi.irb_dev(x.data_ptr(), f1.data_ptr(), fd.data_ptr(), f2.data_ptr(), r.data_ptr() if res else None, out.data_ptr(),
                        N, HW, HW, ic, ec, oc, s, warmup=2, iters=8)
    results.sort()
    print("block %2d  %3dx%-3d %2d->%3d->%2d s%d: auto %.1f us; best: %s" %
          (si, HW, HW, ic, ec, oc, s, auto, "  ".join("%dx%dx%d %.1f" % (tw, th, nf, us) for us, tw, th, nf in results[:6])))

#!/usr/bin/env python3
"""A/B the fused block kernel under environment knobs (interleaved rounds), all block shapes, batch 64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
SHAPES = [(4, 24, 8, 2, 160, False), (8, 32, 8, 1, 80, True), (8, 32, 8, 2, 80, False),
          (8, 48, 8, 1, 40, True), (16, 96, 16, 1, 40, True), (16, 96, 24, 2, 40, False), (24, 136, 24, 1, 20, True),
          (24, 136, 48, 2, 20, False), (48, 224, 48, 1, 10, True)]
COUNT = [1, 2, 1, 3, 4, 1, 4, 1, 5]          # how many times each block occurs in yolo-fastest-1.1
variants = [dict(e.split("=") for e in v.split(",")) if v else {} for v in sys.argv[1:]] or [{}]
N = 64
tot = [0.0] * len(variants)
for si, (ic, ec, oc, s, HW, res) in enumerate(SHAPES):
    OH = (HW - 1) // s + 1
    x = torch.rand((ic * N, HW, HW), device="cuda") - 0.5
    f1 = torch.rand((ec, ((ic + 3) & ~3) + 4), device="cuda") - 0.5
    fd = torch.rand((ec, 16), device="cuda") - 0.5
    f2 = torch.rand((oc, ((ec + 3) & ~3) + 4), device="cuda") - 0.5
    r = torch.rand((oc * N, OH, OH), device="cuda")
    out = torch.empty((oc * N, OH, OH), device="cuda")
    best = [1e9] * len(variants)
    for rnd in range(3):
        for vi, v in enumerate(variants):
            for k in list(os.environ):
                if k.startswith("FFGPU_IRB_"):
                    del os.environ[k]
            os.environ.update(v)
            us = capi.irb_dev(x.data_ptr(), f1.data_ptr(), fd.data_ptr(), f2.data_ptr(), r.data_ptr() if res else None, out.data_ptr(),
                              N, HW, HW, ic, ec, oc, s, warmup=2, iters=10)
            best[vi] = min(best[vi], us)
    for vi in range(len(variants)):
        tot[vi] += best[vi] * COUNT[si]
    print("%3dx%-3d %2d->%3d->%2d s%d x%d: " % (HW, HW, ic, ec, oc, s, COUNT[si]) + "  ".join("%7.1f" % b for b in best))
print("net total (us): " + "  ".join("%7.1f" % t for t in tot) + "   variants: " + " | ".join(str(v) for v in variants))

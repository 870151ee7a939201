#!/usr/bin/env python3
"""Time the tail layers (layer >= 109) of the net per step under env variants."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
variants = [dict(e.split("=") for e in v.split(",")) if v else {} for v in sys.argv[1:]] or [{}]
B = 64
net = capi.Net()
x = torch.rand((B, 3, 320, 320), device="cuda")
res = []
for v in variants:
    for k in list(os.environ):
        if k.startswith("FFGPU_"):
            del os.environ[k]
    os.environ.update(v)
    ex = net.executor(B, capi.FFGPU.NO_GRAPH)
    steps = ex.profile_steps(x.data_ptr())
    res.append(steps)
    ex.close()
print("layer " + " ".join("%9s" % ("v%d" % i) for i in range(len(variants))))
for i, (lay, _) in enumerate(res[0]):
    if lay >= 109 or lay == 0:
        a, b = net.layer(lay), net.layer(lay + 1)
        print("%5d " % lay + " ".join("%9.1f" % r[i][1] for r in res) + "   type %d %dx%dx%d->%d k%d" % (a.type, a.w, a.h, a.c, b.c, a.fs))
print("total " + " ".join("%9.1f" % sum(u for _, u in r) for r in res), variants)

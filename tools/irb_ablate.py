#!/usr/bin/env python3
"""Ablate phases of the fused IRB kernel on the whole net (tuning aid, GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
B = 64
net = capi.Net()
x = torch.rand((B, 3, 320, 320), device="cuda")
os.environ["FFGPU_VERBOSE_IRB"] = "1"
ex = net.executor(B, capi.FFGPU.NO_GRAPH)
ex.forward_dev(x.data_ptr()); torch.cuda.synchronize()
os.environ["FFGPU_VERBOSE_IRB"] = "0"
rows = {}
for skip in (0, 1, 2, 4, 8, 6, 14, 15):
    os.environ["FFGPU_IRB_SKIP"] = str(skip)
    rows[skip] = ex.profile_steps(x.data_ptr())
os.environ["FFGPU_IRB_SKIP"] = "0"
print("%5s " % "layer" + " ".join("skip=%-3d" % k for k in rows))
for i, (lay, _) in enumerate(rows[0]):
    if lay < 0 or net.layer(lay).type != 0 or lay < 3 or lay > 108:
        continue
    print("%5d " % lay + " ".join("%8.1f" % rows[k][i][1] for k in rows))

#!/usr/bin/env python3
"""config[2] on k_pw_x3t only, a few environment variants in one process: tools/pw_x3t_quick.py "label:VAR=1,VAR2=0" ...  (us per launch, best of 3 x 60)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
ic, oc, N, H, W = 256, 512, 256, 20, 20
s = torch.cuda.Stream()
x = torch.rand((ic * N, H, W), device="cuda") - 0.5
filt = torch.zeros((oc, ic + 4), device="cuda")
filt[:, :ic] = (torch.rand((oc, ic), device="cuda") - 0.5) / ic ** 0.5
filt[:, ic] = 1.0
y = torch.empty((oc * N, H, W), device="cuda")
specs = sys.argv[1:] or ["default:"]
keys = set()
out = []
for rep in range(2):
    for spec in specs:
        label, _, envs = spec.partition(":")
        for k in keys:
            os.environ.pop(k, None)
        for kv in filter(None, envs.split(",")):
            k, _, v = kv.partition("=")
            os.environ[k] = v
            keys.add(k)
        kn = capi.kernel_name(N, W, H, ic, 1, 0, 1, 1, oc)
        us = min(capi.groupconv_time_dev(x.data_ptr(), filt.data_ptr(), y.data_ptr(), N, W, H, ic, 1, 0, 1, 1, oc, act=2, variant=0, warmup=60, iters=60, stream=s.cuda_stream) for _ in range(3))
        out.append("%s %s %.1f us" % (label, kn, us))
print(" | ".join(out))

#!/usr/bin/env python3
"""BASELINE config[1] (depthwise 3x3, 320x320x64, batch 64): k_dw3_stream under its tuning switches, alternating, same box:
band length (halo re-reads: 2 rows per band), non-temporal stores, rows in flight"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
N, C, H, W = 64, 64, 320, 320
g = torch.Generator(device="cuda").manual_seed(1234)
x = torch.rand((C * N, H, W), device="cuda", generator=g) * 2 - 1
y = torch.empty_like(x)
f = torch.zeros((C, 16), device="cuda")
f[:, :9] = torch.rand((C, 9), device="cuda", generator=g) - 0.5
f[:, 12] = 1.0
torch.cuda.synchronize()
s = torch.cuda.Stream(priority=-1)
nbytes = 2 * x.numel() * 4
capi.diag().ffgpu_membench(y.data_ptr(), x.data_ptr(), nbytes // 2, 0, 1024, 64, s.cuda_stream)
variants = [{}, {"FFGPU_DW_NT": "0"}, {"FFGPU_DW_BAND": "8"}, {"FFGPU_DW_BAND": "16"}, {"FFGPU_DW_BAND": "8", "FFGPU_DW_NT": "0"}, {"FFGPU_DW_U": "2"}, {"FFGPU_DW_BAND": "8", "FFGPU_DW_U": "2"},
            {"FFGPU_DW_XCD": "1"}, {"FFGPU_DW_BAND": "8", "FFGPU_DW_XCD": "1"}, {"FFGPU_DW_BAND": "20"}, {"FFGPU_DW_BAND": "40"}]
res = {i: [] for i in range(len(variants))}
for rnd in range(4):
    for i, v in enumerate(variants):
        for k in ("FFGPU_DW_BAND", "FFGPU_DW_NT", "FFGPU_DW_U", "FFGPU_DW_XCD"):
            os.environ.pop(k, None)
        os.environ.update(v)
        us = capi.groupconv_time_dev(x.data_ptr(), f.data_ptr(), y.data_ptr(), N, W, H, C, C, 1, 1, 3, C, act=2, warmup=4, iters=40, stream=s.cuda_stream)
        res[i].append(us)
for i, v in enumerate(variants):
    r = sorted(res[i])
    print("%-48s median %.1f us  (%.3f of 8 TB/s)  runs %s" % (v or "default (band 4, U 4, streamed stores)", r[len(r) // 2], nbytes / r[len(r) // 2] / 1e3 / 8000, ["%.1f" % u for u in res[i]]))

#!/usr/bin/env python3
"""MFMA-only floor of BASELINE config[2] on k_pw_x3t's tiling: the launch's 4 915 200 v_mfma_f32_32x32x16_bf16 and nothing else
(ffgpu_mfma_floor in libffcnn_hip_diag.so: no loads, no split, no LDS, no stores -- wrong results by construction), on operands split from the same
distributions the bench uses, on zeros, and with one workgroup per CU; beside it the kernel itself, same process, alternating."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ffcnn_amd import capi
D = capi.diag()
s = torch.cuda.Stream()
rng = np.random.default_rng(1)

def parts(x):
    x = x.astype(np.float32)
    out = []
    for _ in range(3):
        t = (x.view(np.uint32) & 0xffff0000).view(np.float32)
        out.append((t.view(np.uint32) >> 16).astype(np.uint32))
        x = x - t
    return out

def frags(zero=False):
    f = np.zeros((18, 256, 4), np.uint32)
    if not zero:
        w = parts(rng.uniform(-0.5, 0.5, (2, 256, 8)) / 16.0)          # weights of a 256-channel contraction
        x = parts(rng.uniform(-0.5, 0.5, (4, 256, 8)))
        for r in range(2):
            for pt in range(3):
                f[r * 3 + pt] = w[pt][r, :, 0::2] | (w[pt][r, :, 1::2] << 16)
        for j in range(4):
            for pt in range(3):
                f[6 + j * 3 + pt] = x[pt][j, :, 0::2] | (x[pt][j, :, 1::2] << 16)
    return torch.from_numpy(f.view(np.int32)).cuda()

ic, oc, N, H, W = 256, 512, 256, 20, 20
x = torch.rand((ic * N, H, W), device="cuda") - 0.5
filt = torch.zeros((oc, ic + 4), device="cuda")
filt[:, :ic] = (torch.rand((oc, ic), device="cuda") - 0.5) / ic ** 0.5
filt[:, ic] = 1.0
y = torch.empty((oc * N, H, W), device="cuda")
TOTAL = 4915200                                                        # MFMAs of one config[2] launch (6 x 26.84 GFLOP / 32 768)
fr, fz = frags(), frags(True)
for rep in range(3):
    k = min(capi.groupconv_time_dev(x.data_ptr(), filt.data_ptr(), y.data_ptr(), N, W, H, ic, 1, 0, 1, 1, oc, act=2, variant=0, warmup=60, iters=60, stream=s.cuda_stream) for _ in range(2))
    row = ["k_pw_x3t %.1f us" % k]
    for name, f, blocks in (("floor, two workgroups per CU", fr, 512), ("one per CU", fr, 256), ("four per CU (launch bound 2: two rounds)", fr, 1024), ("zero operands", fz, 512)):
        trips = TOTAL // (blocks * 4 * 48)
        us = D.ffgpu_mfma_floor(f.data_ptr(), trips, blocks, 40, s.cuda_stream)
        ghz = trips * 48 * 32 * (blocks * 4 / 1024.0) / us / 1e3
        row.append("%s %.1f us (%.2f GHz x 100 %% busy)" % (name, us, ghz))
    print(" | ".join(row), flush=True)

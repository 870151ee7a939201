#!/usr/bin/env python3
"""Sweep launch knobs of the roofline kernels on the BASELINE shapes (run on the GPU box)."""
import itertools
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from ffcnn_amd import capi  # noqa: E402


def dw(N=64, C=64, H=320, W=320, iters=20):
    g = torch.Generator(device="cuda").manual_seed(1234)
    x = torch.rand((C * N, H, W), device="cuda", generator=g) * 2 - 1
    y = torch.empty_like(x)
    f = torch.zeros((C, 16), device="cuda")
    f[:, :9] = torch.rand((C, 9), device="cuda", generator=g) - 0.5
    f[:, 12] = 1.0
    s = torch.cuda.Stream()
    res = []
    for U, NT, BAND in itertools.product((2, 4), (0, 1), (0, 40, 80, 160, 320)):
        os.environ.update(FFGPU_DW_U=str(U), FFGPU_DW_NT=str(NT), FFGPU_DW_BAND=str(BAND))
        us = capi.groupconv_time_dev(x.data_ptr(), f.data_ptr(), y.data_ptr(), N, W, H, C, C, 1, 1, 3, C, act=2, warmup=3,
                                     iters=iters, stream=s.cuda_stream)
        gbs = 2 * x.numel() * 4 / us / 1e3
        res.append((gbs, U, NT, BAND, us))
        print("dw3 %dx%dx%dx%d U=%d NT=%d BAND=%3d  %8.1f us  %7.1f GB/s  %.1f%% of 8 TB/s" % (N, C, H, W, U, NT, BAND, us, gbs, gbs / 80))
    # plain copy ceiling for reference
    y.copy_(x); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print("torch copy_ same bytes: %.1f us  %.1f GB/s" % (us, 2 * x.numel() * 4 / us / 1e3))
    return max(res)


def pw(N=256, ic=256, oc=512, H=20, W=20, iters=10):
    g = torch.Generator(device="cuda").manual_seed(1235)
    x = torch.rand((ic * N, H, W), device="cuda", generator=g) * 2 - 1
    y = torch.empty((oc * N, H, W), device="cuda")
    f = torch.zeros((oc, ic + 4), device="cuda")
    f[:, :ic] = torch.rand((oc, ic), device="cuda", generator=g) - 0.5
    f[:, ic] = 1.0
    s = torch.cuda.Stream()
    us = capi.groupconv_time_dev(x.data_ptr(), f.data_ptr(), y.data_ptr(), N, W, H, ic, 1, 0, 1, 1, oc, act=2, warmup=2,
                                 iters=iters, stream=s.cuda_stream)
    tf = 2.0 * oc * ic * N * H * W / us / 1e6
    print("pw %d->%d P=%d: %.1f us  %.2f TFLOP/s (%.1f%% of 157.3)" % (ic, oc, N * H * W, us, tf, tf / 1.573))


if __name__ == "__main__":
    what = sys.argv[1:] or ["dw", "pw"]
    if "dw" in what:
        print("best:", dw())
    if "dw160" in what:
        print("best:", dw(64, 8, 160, 160))
    if "pw" in what:
        pw()

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
    cfgs = [dict(U=u, NT=0, BAND=b, LDS=0, XCD=x) for u in (2, 4) for b in (4, 8, 16, 32, 320) for x in (0, 1)]
    cfgs += [dict(U=4, NT=1, BAND=4, LDS=0, XCD=1), dict(U=2, NT=0, BAND=320, LDS=40960, XCD=0)]
    times = {i: [] for i in range(len(cfgs))}
    for rnd in range(5):                      # interleaved rounds: box/thermal drift hits every config alike
        for i, c in enumerate(cfgs):
            os.environ.update({"FFGPU_DW_" + k: str(v) for k, v in c.items()})
            times[i].append(capi.groupconv_time_dev(x.data_ptr(), f.data_ptr(), y.data_ptr(), N, W, H, C, C, 1, 1, 3, C, act=2,
                                                    warmup=2, iters=iters, stream=s.cuda_stream))
    res = []
    nbytes = 2 * x.numel() * 4
    for i, c in enumerate(cfgs):
        t = sorted(times[i])
        med, best = t[len(t) // 2], t[0]
        res.append((nbytes / med / 1e3, c))
        print("dw3 %dx%dx%dx%d %-48s median %7.1f us %7.1f GB/s (%.1f%%)  best %7.1f us %7.1f GB/s" %
              (N, C, H, W, c, med, nbytes / med / 1e3, nbytes / med / 80e3, best, nbytes / best / 1e3))
    # plain copy ceiling for reference
    y.copy_(x); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print("torch copy_ same bytes: %.1f us  %.1f GB/s" % (us, 2 * x.numel() * 4 / us / 1e3))
    return max(res, key=lambda r: r[0])


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


def mem(nbytes=1677721600):
    a = torch.empty(nbytes // 4, device="cuda").uniform_(-1, 1)
    b = torch.empty_like(a)
    s = torch.cuda.Stream()
    names = {0: "copy", 1: "copy nt", 2: "read", 3: "write", 4: "copy x4", 5: "wave-span", 6: "block-span", 7: "wave-span nt"}
    for mode in (0, 1, 5, 7, 6):
        for blocks in (256, 512, 768, 1024, 1280, 1536, 2048, 4096):
            us = capi.diag().ffgpu_membench(b.data_ptr(), a.data_ptr(), nbytes, mode, blocks, 10, s.cuda_stream)
            moved = nbytes * (1 if mode in (2, 3) else 2)
            print("membench %-8s blocks=%5d  %8.1f us  %7.1f GB/s" % (names[mode], blocks, us, moved / us / 1e3))


if __name__ == "__main__":
    what = sys.argv[1:] or ["mem", "dw", "pw"]
    if "mem" in what:
        mem()
    if "dw" in what:
        print("best:", dw())
    if "dw160" in what:
        print("best:", dw(64, 8, 160, 160))
    if "pw" in what:
        pw()

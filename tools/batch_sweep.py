#!/usr/bin/env python3
"""Frames/s for other batch sizes and chain counts (the bench's metric is 64 frames per step): does a bigger batch per launch beat four 64-frame chains in lockstep?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
net = capi.Net()
for B, S, flags in ((64, 4, capi.FFGPU.CONCURRENT), (256, 1, 0), (256, 1, capi.FFGPU.CONCURRENT), (128, 2, capi.FFGPU.CONCURRENT), (256, 2, capi.FFGPU.CONCURRENT), (512, 1, 0)):
    os.environ["FFGPU_BRANCH"] = "0" if S > 1 else "1"
    exs = [net.executor(B, flags | capi.FFGPU.HOST_DETS) for _ in range(S)]
    sts = [torch.cuda.Stream() for _ in range(S)]
    xs = [torch.rand((B, 3, 320, 320), device="cuda") for _ in range(max(2, 512 // B))]
    def run(k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(k):
            exs[i % S].forward_dev(xs[i % len(xs)].data_ptr(), sts[i % S].cuda_stream)
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / k
    run(8 * S); t = run(64 * 256 // B)
    print("batch %3d x %d chains: %.4f ms per launch = %.0f frames/s" % (B, S, t * 1e3, B / t))
    for e in exs: e.close()
    del xs

#!/usr/bin/env python3
"""Experiment: K executors of 64/K frames each on K streams, launched together, vs one executor of 64 frames."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
net = capi.Net()
x = torch.rand((64, 3, 320, 320), device="cuda")
for K in (1, 2, 4, 1, 2):
    B = 64 // K
    exs = [net.executor(B, capi.FFGPU.HOST_DETS) for _ in range(K)]
    sts = [torch.cuda.Stream() for _ in range(K)]
    xs = [x[k * B:(k + 1) * B].contiguous() for k in range(K)]
    def run(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            for k in range(K):
                exs[k].forward_dev(xs[k].data_ptr(), sts[k].cuda_stream)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    run(20)
    print("K=%d executors x %d frames: %.4f ms per 64 frames" % (K, B, run(200)))
    for e in exs:
        e.close()

#!/usr/bin/env python3
"""Experiments with concurrency across and inside batches (frames/s is what counts: ms per 64 frames).
   A: one plain executor            B: one split executor (two half-batch chains in one graph)
   C: two split executors on two streams, alternating batches (the drain of batch i overlaps the head of batch i+1)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
net = capi.Net()
x = torch.rand((64, 3, 320, 320), device="cuda")
F = capi.FFGPU


def bench(exs, sts, n=300):
    def run(k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(k):
            j = i % len(exs)
            exs[j].forward_dev(x.data_ptr(), sts[j].cuda_stream)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / k * 1e3
    run(20)
    return run(n)


for name, flags, k in (("A plain", F.HOST_DETS, 1), ("B split", F.HOST_DETS | F.SPLIT2, 1), ("C 2 x split, 2 streams", F.HOST_DETS | F.SPLIT2, 2),
                       ("D 2 x plain, 2 streams", F.HOST_DETS, 2), ("E 3 x plain, 3 streams", F.HOST_DETS, 3), ("F 4 x plain, 4 streams", F.HOST_DETS, 4),
                       ("G 3 x split, 3 streams", F.HOST_DETS | F.SPLIT2, 3), ("D 2 x plain, 2 streams", F.HOST_DETS, 2)):
    exs = [net.executor(64, flags) for _ in range(k)]
    sts = [torch.cuda.Stream() for _ in range(k)]
    print("%-28s %.4f ms per 64 frames" % (name, bench(exs, sts)))
    for e in exs:
        e.close()

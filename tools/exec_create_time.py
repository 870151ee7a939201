#!/usr/bin/env python3
"""time to create / first-use / destroy executors (what `bench.py --node` pays per device and slot): feeds bench.node_budget_s"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
net = capi.Net()
x = torch.rand((64, 3, 320, 320), device="cuda")
for batch in (32, 64):
    ts = []
    exs = []
    for i in range(8):
        t0 = time.perf_counter()
        ex = net.executor(batch, capi.FFGPU.CONCURRENT)
        ex.set_scale(640, 320)
        ex.forward_dev(x.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
        exs.append(ex)
    t0 = time.perf_counter()
    for e in exs:
        e.close()
    td = (time.perf_counter() - t0) / len(exs)
    print("batch %d: create + first forward %s s (mean %.3f), destroy %.3f s each" % (batch, " ".join("%.3f" % t for t in ts), sum(ts) / len(ts), td))
t0 = time.perf_counter()
nd = capi.Node(net, 1, 64, exec_flags=capi.FFGPU.CONCURRENT, node_flags=capi.Node.DEPTH(8))
print("node of 1 device x 8 slots: create %.3f s" % (time.perf_counter() - t0))
nd.close()
net.close()

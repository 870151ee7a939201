#!/usr/bin/env python3
"""What each stretch of the net costs when it has the device TO ITSELF: only its launches run (FFGPU_DBG_KEEP, wrong results),
as one chain and as four chains in flight.  Against tools/ablate_layers.py (its marginal cost inside the whole net) this says
whether a stretch is slow by itself or slowed by its neighbours."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
os.environ.setdefault("FFGPU_BRANCH", "0")
net = capi.Net()
x = torch.rand((64, 3, 320, 320), device="cuda")
SEG = [("layer 0", (0, 0)), ("thin blocks 1-8", (1, 8)), ("160->80 s2 9-11", (9, 11)), ("80x80 12-21", (12, 21)),
       ("80->40 s2 22-24", (22, 24)), ("40x40x48 25-37", (25, 37)), ("40x40x96 38-57", (38, 57)), ("40->20 s2 58-60", (58, 60)),
       ("20x20 61-80", (61, 80)), ("20->10 s2 81-83", (81, 83)), ("10x10 84-108", (84, 108)), ("SPP 109-114", (109, 114)),
       ("whole net", None)]
# (the two heads are left out: fed with whatever the arena holds, their decode kernels find thousands of "detections")
tot1 = tot4 = 0.0
for name, rng in SEG:
    if rng:
        os.environ["FFGPU_DBG_KEEP"] = "%d:%d" % rng
    else:
        os.environ.pop("FFGPU_DBG_KEEP", None)
    res = []
    for S in (1, 4):
        exs = [net.executor(64, capi.FFGPU.HOST_DETS | (capi.FFGPU.CONCURRENT if S > 1 else 0)) for _ in range(S)]
        sts = [torch.cuda.Stream() for _ in range(S)]
        def run(k):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(k):
                exs[i % S].forward_dev(x.data_ptr(), sts[i % S].cuda_stream)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / k * 1e6
        run(40)
        res.append(run(400))
        for e in exs:
            e.close()
    if name != "whole net":
        tot1 += res[0]; tot4 += res[1]
    print("%-22s alone: one chain %7.1f us per batch, four chains %7.1f us per batch" % (name, res[0], res[1]))
print("sum of the stretches: %.1f / %.1f us" % (tot1, tot4))

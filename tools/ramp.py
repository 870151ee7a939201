#!/usr/bin/env python3
"""How the four chains fill and drain: completion time of every step of a short run (events on the chains' streams)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
os.environ.setdefault("FFGPU_BRANCH", "0")
S, K = 4, int(sys.argv[1]) if len(sys.argv) > 1 else 24
net = capi.Net()
exs = [net.executor(64, capi.FFGPU.HOST_DETS | capi.FFGPU.CONCURRENT) for _ in range(S)]
sts = [torch.cuda.Stream(priority=int(os.environ["RAMP_PRIO"])) if "RAMP_PRIO" in os.environ else torch.cuda.Stream() for _ in range(S)]       # RAMP_PRIO=-1: bench.py's chain streams
xs = [torch.rand((64, 3, 320, 320), device="cuda") for _ in range(8)]
for rep in range(3):
    for i in range(8):
        exs[i % S].forward_dev(xs[i % 8].data_ptr(), sts[i % S].cuda_stream)
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t0.record(sts[0])
    off = float(os.environ.get("RAMP_OFFSET_MS", "0"))          # stagger the chains: chain j starts j * off ms late
    if off > 0:
        for j in range(1, S):
            with torch.cuda.stream(sts[j]):
                torch.cuda._sleep(int(j * off * 1e-3 * float(os.environ.get("RAMP_HZ", "2.4e9"))))
    evs = []
    for i in range(K):
        exs[i % S].forward_dev(xs[i % 8].data_ptr(), sts[i % S].cuda_stream)
        e = torch.cuda.Event(enable_timing=True); e.record(sts[i % S]); evs.append(e)
    torch.cuda.synchronize()
    t = [t0.elapsed_time(e) for e in evs]
    if rep == 2:
        print("completion (ms) of steps 0..%d: %s" % (K - 1, " ".join("%.2f" % v for v in t)))
        d = sorted(t)
        print("whole run %.3f ms = %.4f ms per step; steps 8..%d: %.4f ms per step" % (d[-1], d[-1] / K, K - 9, (d[K - 9] - d[8]) / (K - 17)))

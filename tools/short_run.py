#!/usr/bin/env python3
"""Where a SHORT run's time goes (the driver times 20 steps): host clock around the enqueue loop and the final
synchronisation next to the device-side completion time of every step (events on the chains' streams)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
os.environ.setdefault("FFGPU_BRANCH", "0")
S, K = 4, int(sys.argv[1]) if len(sys.argv) > 1 else 20
net = capi.Net()
exs = [net.executor(64, capi.FFGPU.HOST_DETS | capi.FFGPU.CONCURRENT) for _ in range(S)]
sts = [torch.cuda.Stream() for _ in range(S)]
xs = [torch.rand((64, 3, 320, 320), device="cuda") for _ in range(8)]
for i in range(8):
    exs[i % S].forward_dev(xs[i % 8].data_ptr(), sts[i % S].cuda_stream)
torch.cuda.synchronize()
for rep in range(4):
    ev = rep == 3 or os.environ.get("EVENTS", "0") == "1"
    torch.cuda.synchronize()
    if os.environ.get("IDLE_MS"):
        time.sleep(float(os.environ["IDLE_MS"]) * 1e-3)
    t_start = time.perf_counter()
    t0 = torch.cuda.Event(enable_timing=True)
    if ev:
        t0.record(sts[0])
    evs = []
    for i in range(K):
        exs[i % S].forward_dev(xs[i % 8].data_ptr(), sts[i % S].cuda_stream)
        if ev:
            e = torch.cuda.Event(enable_timing=True); e.record(sts[i % S]); evs.append(e)
    t_enq = time.perf_counter()
    torch.cuda.synchronize()
    t_end = time.perf_counter()
    line = "rep %d: enqueue %.3f ms, sync done %.3f ms -> %.1f k frames/s" % (rep, (t_enq - t_start) * 1e3, (t_end - t_start) * 1e3, 64 * K / (t_end - t_start) / 1e3)
    if ev:
        t = [t0.elapsed_time(e) for e in evs]
        line += " | device completion: " + " ".join("%.2f" % v for v in t)
    print(line)

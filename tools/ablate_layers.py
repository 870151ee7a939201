#!/usr/bin/env python3
"""What each stretch of the net costs WITH SEVERAL BATCHES IN FLIGHT: drop its launches (FFGPU_DBG_SKIP, wrong results) and
see how much time per batch goes away.  4 executors on 4 streams, 64 frames each."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
os.environ.setdefault("FFGPU_BRANCH", "0")
net = capi.Net()
x = torch.rand((64, 3, 320, 320), device="cuda")
SEG = [("none", None), ("layer 0", (0, 0)), ("thin blocks 1-8", (1, 8)), ("160->80 s2 9-11", (9, 11)), ("80x80 12-21", (12, 21)),
       ("80->40 s2 22-24", (22, 24)), ("40x40x48 25-37", (25, 37)), ("40x40x96 38-57", (38, 57)), ("40->20 s2 58-60", (58, 60)),
       ("20x20 61-80", (61, 80)), ("20->10 s2 81-83", (81, 83)), ("10x10 84-108", (84, 108)), ("SPP 109-114", (109, 114)),
       ("head 10x10 115-121", (115, 121)), ("head 20x20 122-130", (122, 130)), ("none", None)]
if len(sys.argv) > 1:                      # custom stretches: python tools/ablate_layers.py 123:123 125:125 ...
    SEG = [("none", None)] + [("layers " + a, tuple(int(v) for v in a.split(":"))) for a in sys.argv[1:]] + [("none", None)]
base = None
for name, rng in SEG:
    if rng:
        os.environ["FFGPU_DBG_SKIP"] = "%d:%d" % rng
    else:
        os.environ.pop("FFGPU_DBG_SKIP", None)
    exs = [net.executor(64, capi.FFGPU.HOST_DETS | capi.FFGPU.CONCURRENT) for _ in range(4)]
    sts = [torch.cuda.Stream(priority=-1) for _ in range(4)]
    def run(k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(k):
            exs[i % 4].forward_dev(x.data_ptr(), sts[i % 4].cuda_stream)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / k * 1e3
    run(20)
    ms = run(300)
    if base is None:
        base = ms
    print("%-22s %.4f ms per batch   (%+.1f us)" % (name, ms, (ms - base) * 1e3))
    for e in exs:
        e.close()

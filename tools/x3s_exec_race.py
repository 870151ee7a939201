#!/usr/bin/env python3
"""Several KEEP_ALL executors of yolo-fastest on their own streams, the same frames over and over: after every round the outputs of the given layers are
compared bit for bit with the first round's (per executor).  A layer that differs while its input layer does not is where a kernel slipped.
python tools/x3s_exec_race.py [batch] [execs] [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ffcnn_amd import capi as F
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ne = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 200
LAYERS = [int(a) for a in os.environ.get("RACE_LAYERS", "").split(",") if a] or [114, 115, 116, 117, 118, 119, 120, 124, 125, 126, 127, 128, 129]
x = torch.rand((batch, 3, 320, 320), device="cuda")
with F.Net(F.CFG, F.WEIGHTS) as net:
    exs = [net.executor(batch, F.FFGPU.KEEP_ALL | F.FFGPU.CONCURRENT) for _ in range(ne)]
    sts = [torch.cuda.Stream() for _ in range(ne)]
    ok = []
    for e in exs[:1]:
        e.forward_dev(x.data_ptr(), sts[0].cuda_stream)
        torch.cuda.synchronize()
        for l in LAYERS:
            try:
                e.read_layer(l, 0)
                ok.append(l)
            except RuntimeError:
                pass                                     # fused away in this plan
    LAYERS = ok
    first = None
    bad = {}
    for r in range(rounds):
        for rep in range(2):
            for e, s in zip(exs, sts):
                e.forward_dev(x.data_ptr(), s.cuda_stream)
        torch.cuda.synchronize()
        cur = [[e.read_layer(l, f).tobytes() for l in LAYERS for f in (0, batch - 1)] for e in exs]
        if first is None:
            first = cur[0]
            for k in range(1, ne):
                for i, b in enumerate(cur[k]):
                    if b != first[i]:
                        print("round 0: executor %d layer %d differs from executor 0" % (k, LAYERS[i // 2]), flush=True)
            continue
        for k in range(ne):
            for i, b in enumerate(cur[k]):
                if b != first[i]:
                    l = LAYERS[i // 2]
                    bad.setdefault(l, 0)
                    bad[l] += 1
                    if sum(bad.values()) <= 6:
                        a = np.frombuffer(b, np.float32); w = np.frombuffer(first[i], np.float32)
                        d = np.abs(a - w); idx = np.nonzero(d)[0]
                        print("round %d executor %d layer %d frame %s: %d values differ, max |d| %.3g, first flat index %d" % (r, k, l, "0" if i % 2 == 0 else "last", idx.size, d.max(), idx[0]), flush=True)
    print("differing (layer: count):", dict(sorted(bad.items())), "in", rounds - 1, "rounds x", ne, "executors")
    for e in exs:
        e.close()

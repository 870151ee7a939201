#!/usr/bin/env python3
"""Executor life cycle: device memory in use after many create / forward / destroy cycles, per flag set."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
net = capi.Net()
x = torch.rand((16, 3, 320, 320), device="cuda")
def used():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 1e6
for name, flags in (("default", 0), ("host dets", capi.FFGPU.HOST_DETS), ("concurrent", capi.FFGPU.CONCURRENT), ("split2", capi.FFGPU.SPLIT2),
                    ("keep all", capi.FFGPU.KEEP_ALL), ("no graph", capi.FFGPU.NO_GRAPH), ("default", 0), ("keep all", capi.FFGPU.KEEP_ALL)):
    marks = []
    for it in range(30):
        with net.executor(16, flags) as ex:
            ex.forward_dev(x.data_ptr())
            ex.read_dets()
        if it in (9, 29): marks.append(used())
    print("%-10s MB in use after 10 / 30 cycles: %.1f / %.1f  (%+.2f MB per cycle)" % (name, marks[0], marks[1], (marks[1] - marks[0]) / 20))

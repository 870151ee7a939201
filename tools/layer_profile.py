#!/usr/bin/env python3
"""Per-layer device time of one batched forward (eager, hipEvents between launches)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from ffcnn_amd import capi  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
net = capi.Net()
ex = net.executor(B, flags)
x = torch.rand((B, 3, 320, 320), device="cuda")
torch.cuda.synchronize()
steps = ex.profile_steps(x.data_ptr())
tot = sum(u for _, u in steps)
kinds = ["conv", "avgpool", "maxpool", "upsample", "dropout", "shortcut", "route", "yolo"]
print("# batch %d flags %d: %d launches, %.1f us device time (eager, sum of per-step gaps), arena %.1f MB" %
      (B, flags, len(steps), tot, ex.arena_bytes / 2**20))
print("%5s %-9s %-14s %-28s %9s %6s %9s" % ("layer", "kind", "kernel", "shape", "us", "pct", "GB/s(alg)"))
for lay, us in steps:
    if lay < 0:
        print("%5d %-9s %-14s %-28s %9.1f %6.2f" % (lay, "-", "-", "-", us, 100 * us / tot))
        continue
    a, b = net.layer(lay), net.layer(lay + 1)
    name, shape, gbs = "-", "", 0.0
    if a.type == 0:
        name = capi.kernel_name(B, a.w, a.h, a.c, a.groups, a.pad, a.stride, a.fs, a.fn)
        shape = "%dx%dx%d->%dx%dx%d k%d s%d g%d" % (a.w, a.h, a.c, b.w, b.h, b.c, a.fs, a.stride, a.groups)
        gbs = 4.0 * B * (a.w * a.h * a.c + b.w * b.h * b.c) / us / 1e3
    else:
        shape = "%dx%dx%d" % (a.w, a.h, a.c)
    print("%5d %-9s %-14s %-28s %9.1f %6.2f %9.0f" % (lay, kinds[a.type], name, shape, us, 100 * us / tot, gbs))

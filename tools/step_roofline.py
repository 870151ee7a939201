#!/usr/bin/env python3
"""Per fused step of one batched forward: algorithmic HBM bytes, flops, the time either bound allows, measured time.

   python tools/step_roofline.py [batch]        (eager, hipEvents between launches: ~2 us overhead per step)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from ffcnn_amd import capi  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
HBM, MFMA = 5.5e12, 157e12          # achievable copy rate on this pool; dense fp32 MFMA peak
net = capi.Net()
ex = net.executor(B, 0)
x = torch.rand((B, 3, 320, 320), device="cuda")
torch.cuda.synchronize()
best = None
for rep in range(5):
    steps = ex.profile_steps(x.data_ptr())
    if best is None:
        best = [list(s) for s in steps]
    else:
        for b, s in zip(best, steps):
            b[1] = min(b[1], s[1])
steps = best
L = net.layer_num
starts = [l for l, _ in steps if l >= 0]
tot_us = tot_ideal = 0.0
print("%5s %-5s %-34s %9s %9s %8s %8s %8s %6s" % ("layer", "span", "shape", "MB", "GFLOP", "hbm_us", "mfma_us", "meas_us", "x"))
for k, (lay, us) in enumerate(steps):
    if lay < 0:
        continue
    nxt = [l for l, _ in steps[k + 1:] if l > lay]
    end = min(nxt) if nxt else L
    a = net.layer(lay)
    byts = 4.0 * B * a.w * a.h * a.c
    fl = 0.0
    last_conv = lay
    for i in range(lay, end):
        li, lo = net.layer(i), net.layer(i + 1)
        if li.type == 0:
            fl += 2.0 * B * lo.w * lo.h * lo.c * li.fs * li.fs * li.c / li.groups
            last_conv = i
        if li.type == 5:
            byts += 4.0 * B * li.w * li.h * li.c      # the shortcut's second operand
    o = net.layer(min(end, L))
    if net.layer(end - 1).type in (4, 6, 7):           # aliases / head: output of the last real layer
        o = net.layer(last_conv + 1)
    byts += 4.0 * B * o.w * o.h * o.c
    hb, mf = byts / HBM * 1e6, fl / MFMA * 1e6
    ideal = max(hb, mf)
    tot_us += us
    tot_ideal += ideal
    print("%5d %-5d %-34s %9.1f %9.2f %8.1f %8.1f %8.1f %6.1f" % (lay, end - lay, "%dx%dx%d -> %dx%dx%d" % (a.w, a.h, a.c, o.w, o.h, o.c),
                                                                  byts / 1e6, fl / 1e9, hb, mf, us, us / max(ideal, 1e-9)))
print("total measured %.1f us, sum of per-step bounds %.1f us" % (tot_us, tot_ideal))

#!/usr/bin/env python3
"""k_conv_x3 under concurrency: the same layer launched on several streams at once, many times, every output compared bit for bit with the first one
(a deterministic kernel must reproduce itself; a race inside it shows as a handful of differing tiles).
python tools/x3s_race.py ic oc N H W fs [streams] [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
ic, oc, N, H, W, fs = (int(a) for a in sys.argv[1:7])
ns = int(sys.argv[7]) if len(sys.argv) > 7 else 4
rounds = int(sys.argv[8]) if len(sys.argv) > 8 else 200
pad = fs // 2
K = fs * fs * ic
k4 = (K + 3) & ~3
g = torch.Generator(device="cuda").manual_seed(7)
x = torch.rand((ic * N, H, W), device="cuda", generator=g) * 2 - 1
filt = torch.zeros((oc, k4 + 4), device="cuda")
filt[:, :K] = (torch.rand((oc, K), device="cuda", generator=g) - 0.5) * 3 / K ** 0.5
filt[:, k4] = torch.rand((oc,), device="cuda", generator=g) + 0.5
filt[:, k4 + 1] = torch.rand((oc,), device="cuda", generator=g) * 0.2 - 0.1
streams = [torch.cuda.Stream() for _ in range(ns)]
outs = [torch.empty((oc * N, H, W), device="cuda") for _ in range(ns)]
ref = torch.empty((oc * N, H, W), device="cuda")
v = capi.FFGPU.K_CONV_X3
print(capi.kernel_name(N, W, H, ic, 1, pad, 1, fs, oc, v), os.environ.get("FFGPU_IGX3_MT", "auto"))
capi.groupconv_dev(x.data_ptr(), filt.data_ptr(), ref.data_ptr(), N, W, H, ic, 1, pad, 1, fs, oc, 2, 0, v, None)
torch.cuda.synchronize()
bad = 0
for r in range(rounds):
    for o in outs:
        o.fill_(float("nan"))
    torch.cuda.synchronize()
    for rep in range(3):
        for s, o in zip(streams, outs):
            capi.groupconv_dev(x.data_ptr(), filt.data_ptr(), o.data_ptr(), N, W, H, ic, 1, pad, 1, fs, oc, 2, 0, v, s.cuda_stream)
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        ne = (o != ref) | torch.isnan(o)
        if bool(ne.any()):
            bad += 1
            idx = ne.nonzero()
            if bad <= 5:
                d = (o - ref).abs()
                print("round %d stream %d: %d outputs differ, max |d| %.3g, first at plane %d (channel %d frame %d) y %d x %d" %
                      (r, i, int(ne.sum()), float(d[ne].max()), int(idx[0, 0]), int(idx[0, 0]) // N, int(idx[0, 0]) % N, int(idx[0, 1]), int(idx[0, 2])))
print("%d of %d outputs differed" % (bad, rounds * ns))

#!/usr/bin/env python3
"""Dense 3x3 layers of a yolov3-tiny-style net: split-bf16 kernel (k_conv_x3, every MT / NW it supports) against the fp32 implicit GEMM (k_conv_igemm).
Both timed through AUTO (weights packed once, as the executor does); the pick is steered with FFGPU_IG_X3 / FFGPU_IGX3_*.  Prints the largest
difference of the two results relative to the output scale."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
SHAPES = [(16, 32, 16, 208, 208), (32, 64, 16, 104, 104), (64, 128, 16, 52, 52), (128, 256, 16, 26, 26), (256, 512, 16, 13, 13), (512, 1024, 16, 13, 13),
          (256, 512, 64, 13, 13), (16, 32, 64, 208, 208), (64, 128, 64, 52, 52), (512, 1024, 64, 13, 13), (384, 256, 64, 26, 26)]
ST = 1
if len(sys.argv) > 1 and sys.argv[1] == "s2":        # the stride-2 form: the downsampling layers of a yolov3-style backbone (input planes given)
    ST = 2
    SHAPES = [(32, 64, 8, 416, 416), (64, 128, 16, 208, 208), (128, 256, 16, 104, 104), (256, 512, 16, 52, 52), (512, 1024, 16, 26, 26), (512, 1024, 64, 26, 26), (32, 64, 16, 208, 208)]
elif len(sys.argv) > 1:
    SHAPES = SHAPES[:int(sys.argv[1])]
s = torch.cuda.Stream()
os.environ["FFGPU_IGX3_MIN_WGS"] = "1"
for (ic, oc, N, H, W) in SHAPES:
    x = torch.rand((ic * N, H, W), device="cuda") - 0.5
    K = 9 * ic
    filt = torch.zeros((oc, ((K + 3) & ~3) + 4), device="cuda")
    filt[:, :K] = (torch.rand((oc, K), device="cuda") - 0.5) / K ** 0.5
    filt[:, (K + 3) & ~3] = 1.0
    OH, OW = H // ST, W // ST
    fl = 2.0 * K * oc * N * OH * OW
    res = {}
    outs = {}
    for name, env in (("igemm", {"FFGPU_IG_X3": "0"}), ("x3 mt4 nw8", {"FFGPU_IGX3_MT": "4", "FFGPU_IGX3_NW": "8"}), ("x3 mt2 nw8", {"FFGPU_IGX3_MT": "2", "FFGPU_IGX3_NW": "8"}),
                      ("x3 mt4 nw4", {"FFGPU_IGX3_MT": "4", "FFGPU_IGX3_NW": "4"}), ("x3 mt2 nw4", {"FFGPU_IGX3_MT": "2", "FFGPU_IGX3_NW": "4"})):
        for k in ("FFGPU_IG_X3", "FFGPU_IGX3_MT", "FFGPU_IGX3_NW"):
            os.environ.pop(k, None)
        os.environ.update(env)
        want = "conv_igemm" if name == "igemm" else "conv_x3"
        assert capi.kernel_name(N, W, H, ic, 1, 1, ST, 3, oc) == want, (name, capi.kernel_name(N, W, H, ic, 1, 1, ST, 3, oc))
        y = torch.full((oc * N, OH, OW), float("nan"), device="cuda")
        res[name] = capi.groupconv_time_dev(x.data_ptr(), filt.data_ptr(), y.data_ptr(), N, W, H, ic, 1, 1, ST, 3, oc, act=2, variant=0,
                                            warmup=2, iters=20, stream=s.cuda_stream)
        torch.cuda.synchronize()
        outs[name] = y
    d = max(float((outs[k] - outs["igemm"]).abs().max()) for k in outs) / float(outs["igemm"].abs().max())
    best = min((v, k) for k, v in res.items() if k != "igemm")
    print("%4d->%4d %3dx%3d N=%2d: igemm %8.1f us (%.2f of fp32 peak) | " % (ic, oc, W, H, N, res["igemm"], fl / res["igemm"] / 1e6 / 157.3) +
          "  ".join("%s %7.1f" % (k[3:], v) for k, v in res.items() if k != "igemm") +
          " | best %.2f of fp32 peak, x%.2f, max rel diff %.1e" % (fl / best[0] / 1e6 / 157.3, res["igemm"] / best[0], d), flush=True)

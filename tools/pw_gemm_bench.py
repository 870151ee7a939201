#!/usr/bin/env python3
"""BASELINE config[2] (1x1, 256 -> 512, 20x20, batch 256) on the pointwise GEMM kernels: microseconds, TFLOP/s, fraction of
the fp32 matrix peak; FFGPU_PWG_DBG (1 no loads in the loop, 8 no epilogue: wrong results) and FFGPU_PWG_NARROW=0 are the ablations; BF16=1 times the
opt-in bf16 kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
N, ic, oc, H, W = 256, 256, 512, 20, 20
if len(sys.argv) > 1:
    N, ic, oc, H, W = (int(v) for v in sys.argv[1:6])
g = torch.Generator(device="cuda").manual_seed(1235)
x = torch.rand((ic * N, H, W), device="cuda", generator=g) * 2 - 1
y = torch.empty((oc * N, H, W), device="cuda")
filt = torch.zeros((oc, ((ic + 3) & ~3) + 4), device="cuda")
filt[:, :ic] = torch.rand((oc, ic), device="cuda", generator=g) - 0.5
filt[:, (ic + 3) & ~3] = 1.0
s = torch.cuda.Stream()
for rep in range(3):
    us = capi.groupconv_time_dev(x.data_ptr(), filt.data_ptr(), y.data_ptr(), N, W, H, ic, 1, 0, 1, 1, oc, act=2, warmup=5, iters=50, stream=s.cuda_stream)
if os.environ.get("BF16"):
    for rep in range(3):
        us = capi.groupconv_time_dev(x.data_ptr(), filt.data_ptr(), y.data_ptr(), N, W, H, ic, 1, 0, 1, 1, oc, act=2, flags=capi.FFGPU.BF16_PW,
                                     variant=capi.FFGPU.K_AUTO, warmup=5, iters=50, stream=s.cuda_stream)
    by = 4.0 * (ic + oc) * N * H * W
    print("pw_bf16 (opt-in, FFGPU_BF16_PW): %.1f us  %.0f GB/s algorithmic (%.2f of 8 TB/s)  %.0f TFLOP/s" % (us, by / us / 1e3, by / us / 1e3 / 8000, 2.0 * oc * ic * N * H * W / us / 1e6))
    sys.exit(0)
fl = 2.0 * oc * ic * N * H * W
print("%s narrow=%s: %.1f us  %.1f TFLOP/s  %.3f of 157.3" % (capi.kernel_name(N, W, H, ic, 1, 0, 1, 1, oc), os.environ.get("FFGPU_PWG_NARROW", "1"),
      us, fl / us / 1e6, fl / us / 1e6 / 157.3))

#!/usr/bin/env python3
"""The small pointwise layers of yolo-fastest's heads at batch 64: the fp32-MFMA kernel AUTO picks against the pointwise form of k_conv_x3 (FFGPU_PW_X3S=1), us alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
SHAPES = [(256, 512, 256, 20, 20), (1024, 256, 16, 13, 13), (512, 256, 16, 26, 26), (256, 128, 16, 26, 26), (256, 512, 64, 13, 13), (128, 64, 16, 52, 52), (64, 32, 16, 104, 104), (120, 120, 64, 20, 20), (120, 255, 64, 20, 20), (192, 96, 64, 10, 10), (96, 96, 64, 10, 10), (96, 255, 64, 10, 10), (136, 24, 64, 20, 20), (224, 48, 64, 10, 10)]
s = torch.cuda.Stream()
os.environ["FFGPU_PWX3S_MIN_OC"] = "8"; os.environ["FFGPU_PWX3S_MIN_IC"] = "8"; os.environ["FFGPU_PWX3S_MIN_WGS"] = "1"
for ic, oc, N, H, W in SHAPES:
    x = torch.rand((ic * N, H, W), device="cuda") - 0.5
    k4 = (ic + 3) & ~3
    filt = torch.zeros((oc, k4 + 4), device="cuda")
    filt[:, :ic] = (torch.rand((oc, ic), device="cuda") - 0.5) / ic ** 0.5
    filt[:, k4] = 1.0
    y = torch.empty((oc * N, H, W), device="cuda")
    out = []
    for name, env in (("auto without x3s", {"FFGPU_PW_X3S": "0"}), ("fp32", {"FFGPU_PW_X3S": "0", "FFGPU_PW_X3": "0"}), ("x3s", {"FFGPU_PW_X3S": "1", "FFGPU_PW_X3": "0"}), ("x3s mt2", {"FFGPU_PW_X3S": "1", "FFGPU_PW_X3": "0", "FFGPU_IGX3_MT": "2"}), ("x3s mt1", {"FFGPU_PW_X3S": "1", "FFGPU_PW_X3": "0", "FFGPU_IGX3_MT": "1"})):
        os.environ.pop("FFGPU_IGX3_MT", None); os.environ.pop("FFGPU_PW_X3", None)
        os.environ.update(env)
        kn = capi.kernel_name(N, W, H, ic, 1, 0, 1, 1, oc)
        us = capi.groupconv_time_dev(x.data_ptr(), filt.data_ptr(), y.data_ptr(), N, W, H, ic, 1, 0, 1, 1, oc, act=2, variant=0, warmup=5, iters=50, stream=s.cuda_stream)
        out.append("%s %s %.1f us" % (name, kn, us))
    print("%3d -> %3d on %dx%d x %d: " % (ic, oc, W, H, N) + " | ".join(out), flush=True)

#!/usr/bin/env python3
"""BASELINE config[2] (256 -> 512 on 20x20 x 256) and relatives: the tiled split-bf16 GEMM (k_pw_x3t, round 5) against pw_x3s / k_pw_x3 / k_pw_gemm32, us alone,
with the kernel's own ablations (FFGPU_PWXT_DBG=2: no epilogue; FFGPU_PWXT_NARROW=0: no narrow tiles)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
SHAPES = [(256, 512, 256, 20, 20), (256, 512, 64, 20, 20), (512, 256, 16, 26, 26), (1024, 256, 16, 13, 13), (120, 255, 64, 20, 20), (120, 120, 64, 20, 20), (128, 256, 128, 20, 20)]
if len(sys.argv) > 1 and sys.argv[1] in ("one", "pmc"):
    SHAPES = SHAPES[:1]
s = torch.cuda.Stream()
FORCE = {"FFGPU_PWX3T_MIN_IC": "8", "FFGPU_PWX3T_MIN_OC": "8", "FFGPU_PWX3T_MIN_P": "1"}
VARIANTS = [("x3t", dict(FORCE, FFGPU_PW_X3T="1")), ("x3t persistent", dict(FORCE, FFGPU_PW_X3T="1", FFGPU_PWXT_PERSIST="1")), ("x3t no narrow", dict(FORCE, FFGPU_PW_X3T="1", FFGPU_PWXT_NARROW="0")), ("x3t no epilogue", dict(FORCE, FFGPU_PW_X3T="1", FFGPU_PWXT_DBG="2")),
            ("x3s", {"FFGPU_PW_X3T": "0", "FFGPU_PWX3S_MIN_OC": "8", "FFGPU_PWX3S_MIN_IC": "8", "FFGPU_PWX3S_MIN_WGS": "1"}), ("pw_x3", {"FFGPU_PW_X3T": "0", "FFGPU_PW_X3S": "0", "FFGPU_PWX3_MIN_IC": "8", "FFGPU_PWX3_MIN_OC": "8", "FFGPU_PWX3_MIN_P": "1"}),
            ("fp32", {"FFGPU_PW_X3T": "0", "FFGPU_PW_X3S": "0", "FFGPU_PW_X3": "0"})]
if len(sys.argv) > 1 and sys.argv[1] == "pmc":
    VARIANTS = VARIANTS[:1]
KEYS = set(k for _, e in VARIANTS for k in e)
for ic, oc, N, H, W in SHAPES:
    x = torch.rand((ic * N, H, W), device="cuda") - 0.5
    k4 = (ic + 3) & ~3
    filt = torch.zeros((oc, k4 + 4), device="cuda")
    filt[:, :ic] = (torch.rand((oc, ic), device="cuda") - 0.5) / ic ** 0.5
    filt[:, k4] = 1.0
    y = torch.empty((oc * N, H, W), device="cuda")
    out = []
    for name, env in VARIANTS:
        for k in KEYS:
            os.environ.pop(k, None)
        os.environ.update(env)
        kn = capi.kernel_name(N, W, H, ic, 1, 0, 1, 1, oc)
        us = min(capi.groupconv_time_dev(x.data_ptr(), filt.data_ptr(), y.data_ptr(), N, W, H, ic, 1, 0, 1, 1, oc, act=2, variant=0, warmup=60, iters=60, stream=s.cuda_stream) for _ in range(3))
        out.append("%s %s %.1f us" % (name, kn, us))
    gf = 2.0 * ic * oc * N * H * W / 1e9
    print("%3d -> %3d on %dx%d x %d (%.2f GFLOP): " % (ic, oc, W, H, N, gf) + " | ".join(out), flush=True)

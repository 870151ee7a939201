import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ffcnn_amd import capi
N, C, H, W = 64, 64, 320, 320
g = torch.Generator(device="cuda").manual_seed(1234)
x = torch.rand((C * N, H, W), device="cuda", generator=g) * 2 - 1
y = torch.empty_like(x)
f = torch.zeros((C, 16), device="cuda")
f[:, :9] = torch.rand((C, 9), device="cuda", generator=g) - 0.5
f[:, 12] = 1.0
s = torch.cuda.Stream(priority=-1)
out = []
for rnd in range(7):
    for nt in ("0", "1"):
        os.environ["FFGPU_DW_NT"] = nt
        us = capi.groupconv_time_dev(x.data_ptr(), f.data_ptr(), y.data_ptr(), N, W, H, C, C, 1, 1, 3, C, act=2, warmup=8, iters=60, stream=s.cuda_stream)
        out.append("nt%s %.1f" % (nt, us))
print(" | ".join(out))

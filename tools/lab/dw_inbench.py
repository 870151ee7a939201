"""config[1] variants measured the way bench.py's roofline leg does (after the net's executors exist), alternating"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ffcnn_amd import capi
os.environ.setdefault("FFGPU_BRANCH", "0")
net = capi.Net()
exs = [net.executor(64, capi.FFGPU.HOST_DETS | capi.FFGPU.CONCURRENT) for _ in range(4)]
sts = [torch.cuda.Stream(priority=-1) for _ in range(4)]
xs = [torch.rand((64, 3, 320, 320), device="cuda") for _ in range(8)]
for i in range(32):
    exs[i % 4].forward_dev(xs[i % 8].data_ptr(), sts[i % 4].cuda_stream)
torch.cuda.synchronize()
N, C, H, W = 64, 64, 320, 320
g = torch.Generator(device="cuda").manual_seed(1234)
x = torch.rand((C * N, H, W), device="cuda", generator=g) * 2 - 1
y = torch.empty_like(x)
f = torch.zeros((C, 16), device="cuda")
f[:, :9] = torch.rand((C, 9), device="cuda", generator=g) - 0.5
f[:, 12] = 1.0
s = sts[0]
nbytes = 2 * x.numel() * 4
variants = [{}, {"FFGPU_DW_XCD": "1"}, {"FFGPU_DW_BAND": "8", "FFGPU_DW_XCD": "1"}, {"FFGPU_DW_BAND": "8"}]
res = {i: [] for i in range(len(variants))}
for rnd in range(6):
    for i, v in enumerate(variants):
        for k in ("FFGPU_DW_BAND", "FFGPU_DW_XCD"):
            os.environ.pop(k, None)
        os.environ.update(v)
        capi.diag().ffgpu_membench(y.data_ptr(), x.data_ptr(), nbytes // 2, 0, 1024, 64, s.cuda_stream)
        us = capi.groupconv_time_dev(x.data_ptr(), f.data_ptr(), y.data_ptr(), N, W, H, C, C, 1, 1, 3, C, act=2, warmup=4, iters=50, stream=s.cuda_stream)
        res[i].append(us)
for i, v in enumerate(variants):
    r = sorted(res[i])
    print("%-48s median %.1f us (%.3f)  runs %s" % (v or "default", r[len(r) // 2], nbytes / r[len(r) // 2] / 1e3 / 8000, ["%.1f" % u for u in res[i]]))

#!/usr/bin/env python3
"""The shader clock the chip HOLDS while the bench's mix runs (four 64-frame chains on four priority streams), while config[1] / config[2] kernels run, and idle:
a one-wave probe on a fifth stream samples s_memrealtime (100 MHz) and s_memtime (shader clock) every ~50 us (libffcnn_hip_diag.so: ffgpu_clock_probe).
The issue-bound model of DESIGN.md 5.4 prices instructions in cycles; what a cycle lasts is this number, not the 2.4 GHz of the peak table."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ffcnn_amd import capi
D = capi.diag()
os.environ.setdefault("FFGPU_BRANCH", "0")
SAMPLES, GAP = 400, 10


def probe(work, label):
    ps = torch.cuda.Stream(priority=-1)
    buf = torch.zeros(2 * SAMPLES, dtype=torch.int64, device="cuda")
    work(0.02)                                                       # warm
    torch.cuda.synchronize()
    D.ffgpu_clock_probe(buf.data_ptr(), SAMPLES, GAP, ps.cuda_stream)
    t = work(None)
    torch.cuda.synchronize()
    v = buf.cpu().numpy().astype(np.float64).reshape(SAMPLES, 2)
    rt, st = v[:, 0], v[:, 1]
    span_ms = (rt[-1] - rt[0]) / 1e5
    n = min(SAMPLES, max(8, int(SAMPLES * min(1.0, (t or span_ms) / span_ms))))        # samples taken while the work ran
    mhz = (st[1:n] - st[:n - 1]) / (rt[1:n] - rt[:n - 1]) * 100.0
    print("%-46s clock while it ran: median %.0f MHz (p10 %.0f, p90 %.0f) over %.1f ms" % (label, np.median(mhz), np.percentile(mhz, 10), np.percentile(mhz, 90), (rt[n - 1] - rt[0]) / 1e5), flush=True)


def idle(ms):
    torch.cuda._sleep(int(2e6))
    return None


S = 4
net = capi.Net()
exs = [net.executor(64, capi.FFGPU.HOST_DETS | capi.FFGPU.CONCURRENT) for _ in range(S)]
sts = [torch.cuda.Stream(priority=-1) for _ in range(S)]
xs = [torch.randint(0, 256, (64, 320, 960), dtype=torch.uint8, device="cuda") for _ in range(8)]


def mix(seconds):
    steps = 40 if seconds else 900
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(sts[0])
    for i in range(steps):
        exs[i % S].forward_bgr_dev(xs[i % 8].data_ptr(), 320, 320, stream=sts[i % S].cuda_stream)
    for s in sts:
        s.synchronize()
    e1.record(sts[0]); e1.synchronize()
    ms = e0.elapsed_time(e1)
    if not seconds:
        print("   mix: %d steps in %.1f ms = %.0f frames/s" % (steps, ms, steps * 64 / ms * 1e3))
    return ms


def conv_loop(N, W, H, ic, groups, pad, fs, oc, iters):
    x = torch.rand((ic * N, H, W), device="cuda") - 0.5
    K = fs * fs * (ic // groups); k4 = (K + 3) & ~3
    f = torch.zeros((oc, k4 + 4), device="cuda"); f[:, :K] = (torch.rand((oc, K), device="cuda") - 0.5) / K ** 0.5; f[:, k4] = 1.0
    y = torch.empty((oc * N, H, W), device="cuda")
    st = torch.cuda.Stream()

    def run(seconds):
        us = capi.groupconv_time_dev(x.data_ptr(), f.data_ptr(), y.data_ptr(), N, W, H, ic, groups, pad, 1, fs, oc, act=2, variant=0, warmup=0, iters=(20 if seconds else iters), stream=st.cuda_stream)
        return us * (20 if seconds else iters) / 1e3
    return run


probe(idle, "idle (a sleeping kernel)")
probe(mix, "yolo-fastest, 4 chains x 64 u8 frames")
probe(conv_loop(64, 320, 320, 64, 64, 1, 3, 64, 40), "config[1] k_dw3_stream (HBM-bound)")
probe(conv_loop(256, 20, 20, 256, 1, 0, 1, 512, 200), "config[2] %s" % capi.kernel_name(256, 20, 20, 256, 1, 0, 1, 1, 512))
os.environ["FFGPU_PW_X3T"] = "0"; os.environ["FFGPU_PW_X3S"] = "0"; os.environ["FFGPU_PW_X3"] = "0"
probe(conv_loop(256, 20, 20, 256, 1, 0, 1, 512, 150), "config[2] %s (fp32 MFMA)" % capi.kernel_name(256, 20, 20, 256, 1, 0, 1, 1, 512))

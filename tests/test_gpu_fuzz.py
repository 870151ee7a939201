"""Seeded random groupconv geometries through the dispatcher (FFGPU_K_AUTO: whatever kernel the library picks for the shape)
against the CPU oracle, frame by frame: pointwise / depthwise / dense / grouped, filter sizes 1-7, strides 1-3, paddings 0-3,
channel counts around every tile size (4, 16, 32, 64, 128, 256), odd widths and heights, batches 1-5, all four activations
(conv.h:4-7 semantics, conv-v0.c:7-31 is what the oracle restates).  The point is the seams BETWEEN the specialised kernels:
a shape that just misses one kernel's predicate must land on another that computes the same thing."""
import os

import numpy as np
import pytest

from test_gpu_kernels import check, make_filter, run_dev

pytestmark = pytest.mark.gpu


def draw(rng):
    kind = rng.choice(["pw", "dw", "dense", "grouped", "dwbig", "first"], p=[0.3, 0.2, 0.2, 0.12, 0.1, 0.08])
    N = int(rng.integers(1, 6))
    act = int(rng.integers(0, 4))
    edge = [1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 24, 31, 32, 33, 48, 63, 64, 65, 96, 120, 127, 128, 129, 136, 200, 255, 256, 260]
    if kind == "pw":
        ic, oc = int(rng.choice(edge)), int(rng.choice(edge))
        fs, stride, pad, groups = 1, 1, 0, 1
        H, W = int(rng.integers(1, 41)), int(rng.integers(1, 41))
        if rng.random() < 0.3:                                   # enough pixels for the LDS-tiled GEMM to be picked
            H, W, N = 20, 20, int(rng.integers(40, 50))
    elif kind == "dw":
        ic = oc = groups = int(rng.choice([1, 2, 3, 8, 24, 32, 48, 96, 120, 136]))
        fs = int(rng.choice([3, 3, 5, 7]))
        stride = int(rng.choice([1, 1, 2, 3]))
        pad = int(rng.choice([fs // 2, fs // 2, fs // 2, 0, 1]))
        H, W = int(rng.integers(fs, 70)), int(rng.integers(fs, 90))
    elif kind == "dwbig":                                        # the streaming depthwise kernel's domain (W % 4 == 0, 40..512) and just outside it
        ic = oc = groups = int(rng.integers(1, 17))
        fs, stride, pad = 3, 1, 1
        W = int(rng.choice([40, 44, 64, 100, 128, 160, 252, 256, 260, 320, 512, 516, int(rng.integers(36, 530))]))
        H = int(rng.integers(2, 140))
        N = int(rng.integers(1, 4))
        if act == 3 and rng.random() < 0.7:
            act = 2
    elif kind == "first":                                        # a darknet first layer: 3 channels, 3x3 stride 2, up to the four-pixel kernel's batch sizes
        ic, groups = 3, 1
        oc = int(rng.choice([8, 8, 16, 5]))
        fs, stride, pad = 3, 2, 1
        W, H = int(rng.choice([64, 96, 160, 224, 320, 322])), int(rng.choice([64, 128, 192, 320, 318]))
        N = int(rng.choice([1, 2, 5, 11])) if W * H < 320 * 320 else int(rng.choice([1, 3, 11]))
        if act == 3:
            act = 2
    elif kind == "dense":
        ic, oc, groups = int(rng.choice([1, 3, 4, 7, 8, 9, 16, 33, 64])), int(rng.choice([1, 5, 8, 16, 21, 64, 70, 130])), 1
        fs = int(rng.choice([1, 2, 3, 3, 5]))
        stride = int(rng.choice([1, 1, 2, 3]))
        pad = int(rng.choice([0, 1, fs // 2]))
        H, W = int(rng.integers(fs, 30)), int(rng.integers(fs, 30))
    else:
        groups = int(rng.choice([2, 3, 4, 8]))
        ic, oc = groups * int(rng.choice([1, 2, 5, 8, 12])), groups * int(rng.choice([1, 3, 8, 16]))
        fs = int(rng.choice([1, 3, 3, 5]))
        stride = int(rng.choice([1, 2]))
        pad = int(rng.choice([0, fs // 2]))
        H, W = int(rng.integers(fs, 26)), int(rng.integers(fs, 26))
    if (W + 2 * pad - fs) // stride + 1 < 1 or (H + 2 * pad - fs) // stride + 1 < 1:
        pad = fs // 2
    return kind, ic, oc, groups, fs, stride, pad, N, H, W, act


@pytest.mark.parametrize("seed", range(int(os.environ.get("FFCNN_FUZZ_SEED0", "0")), int(os.environ.get("FFCNN_FUZZ_SEED0", "0")) + int(os.environ.get("FFCNN_FUZZ_SHAPES", "8"))))
def test_random_geometries_vs_oracle(orc, seed):
    import torch
    from ffcnn_amd import capi
    capi.lib()
    rng = np.random.default_rng(7000 + seed)
    picked = {}
    for case in range(24):
        kind, ic, oc, groups, fs, stride, pad, N, H, W, act = draw(rng)
        K = fs * fs * (ic // groups)
        x = rng.uniform(-1, 1, (ic * N, H, W)).astype(np.float32)
        f = make_filter(rng, oc, K)
        f[:, :K] *= min(1.0, 4.0 / np.sqrt(K))                 # keep the sums O(1): the tolerance is absolute + relative
        name = capi.kernel_name(N, W, H, ic, groups, pad, stride, fs, oc)
        picked[name] = picked.get(name, 0) + 1
        got = run_dev(capi, torch, x, f, N, W, H, ic, groups, pad, stride, fs, oc, act, capi.FFGPU.K_AUTO)
        oh, ow = got.shape[1], got.shape[2]
        xf = x.reshape(ic, N, H, W)
        what = "%s ic%d oc%d g%d k%d s%d p%d N%d %dx%d act%d -> %s" % (kind, ic, oc, groups, fs, stride, pad, N, H, W, act, name)
        for n in range(N):
            ref = orc.groupconv(np.ascontiguousarray(xf[:, n]), f, groups, pad, stride, fs, act)
            check(got.reshape(oc, N, oh, ow)[:, n], ref.reshape(oc, oh, ow), what + " frame %d" % n)
            if n == 0 and groups == ic == oc and fs == 5 and stride == 1 and pad == 2 and H >= 4 and W >= 4:
                # conv-v6.c's 5x5 depthwise path (row oh - 2 without tap row 0, conv-v6.c:422-441) behind FFGPU_COMPAT_V6
                dx, df = torch.from_numpy(x).cuda(), torch.from_numpy(f).cuda()
                dy = torch.full((oc * N, oh, ow), float("nan"), device="cuda")
                capi.groupconv_dev(dx.data_ptr(), df.data_ptr(), dy.data_ptr(), N, W, H, ic, groups, pad, stride, fs, oc, act, capi.FFGPU.COMPAT_V6, capi.FFGPU.K_AUTO, None)
                torch.cuda.synchronize()
                v6 = orc.groupconv(np.ascontiguousarray(xf[:, 0]), f, groups, pad, stride, fs, act, compat_v6=1)
                check(dy.cpu().numpy().reshape(oc, N, oh, ow)[:, 0], v6.reshape(oc, oh, ow), what + " (COMPAT_V6)")
            if n == 0 and case % 3 == 0:                        # the conv.h drop-in itself (host pointers, one frame: conv.h:4-7)
                check(capi.groupconv(np.ascontiguousarray(xf[:, 0]), f, groups, pad, stride, fs, act), ref.reshape(oc, oh, ow), what + " (conv.h drop-in)")
    assert len(picked) >= 3, picked                              # the draw reaches several kernels per seed

"""Specialised conv kernels (dw3_stream, pw_mfma) against the generic kernel on the GPU and
against the CPU oracle, over shapes that exercise every tail: odd heights, partial lane
groups, batch > 1 (channel = plane / N), K padding, channel counts that do not fill a tile."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ATOL, RTOL = 1e-3, 1e-3


def make_filter(rng, fn, K):
    k4 = (K + 3) & ~3
    f = np.zeros((fn, k4 + 4), np.float32)
    f[:, :K] = rng.uniform(-0.5, 0.5, (fn, K))
    f[:, k4] = rng.uniform(0.5, 1.5, fn)
    f[:, k4 + 1] = rng.uniform(-0.1, 0.1, fn)
    return f


def run_dev(capi, torch, x_cnhw, f, N, iw, ih, ic, groups, pad, stride, fs, fn, act, variant):
    ow, oh = (iw + 2 * pad - fs) // stride + 1, (ih + 2 * pad - fs) // stride + 1
    dx, df = torch.from_numpy(x_cnhw).cuda(), torch.from_numpy(f).cuda()
    dy = torch.full((fn * N, oh, ow), float("nan"), device="cuda")
    capi.groupconv_dev(dx.data_ptr(), df.data_ptr(), dy.data_ptr(), N, iw, ih, ic, groups, pad, stride, fs, fn, act,
                       0, variant, None)
    torch.cuda.synchronize()
    return dy.cpu().numpy()


def check(a, ref, what):
    assert not np.isnan(a).any(), what + ": unwritten outputs"
    err = np.abs(a - ref) - (ATOL + RTOL * np.abs(ref))
    assert err.max() <= 0, "%s: max |d| %.3g" % (what, np.abs(a - ref).max())


@pytest.fixture(scope="module")
def env():
    import torch
    from ffcnn_amd import capi
    capi.lib()
    return capi, torch


DW_SHAPES = [  # (C, N, H, W, act)
    (8, 3, 160, 160, 2), (5, 2, 37, 80, 2), (3, 1, 2, 40, 0), (7, 2, 5, 64, 1), (2, 2, 19, 256, 2),
    (3, 2, 33, 320, 2), (2, 1, 9, 512, 2), (4, 1, 320, 320, 2), (13, 1, 3, 48, 2), (1, 5, 41, 100, 2),
]


@pytest.mark.parametrize("shape", DW_SHAPES)
@pytest.mark.parametrize("U,NT", [(4, 0), (2, 1)])
def test_dw3_stream(env, orc, shape, U, NT, monkeypatch):
    capi, torch = env
    C, N, H, W, act = shape
    monkeypatch.setenv("FFGPU_DW_U", str(U))
    monkeypatch.setenv("FFGPU_DW_NT", str(NT))
    rng = np.random.default_rng(hash(shape) & 0xffff)
    x = rng.uniform(-1, 1, (C * N, H, W)).astype(np.float32)
    f = make_filter(rng, C, 9)
    assert capi.kernel_name(N, W, H, C, C, 1, 1, 3, C, capi.FFGPU.K_DW_STREAM) == "dw3_stream"
    got = run_dev(capi, torch, x, f, N, W, H, C, C, 1, 1, 3, C, act, capi.FFGPU.K_DW_STREAM)
    ref = run_dev(capi, torch, x, f, N, W, H, C, C, 1, 1, 3, C, act, capi.FFGPU.K_GENERIC)
    check(got, ref, "dw3_stream %s vs generic" % (shape,))
    if H * W * C * N <= 300000:       # oracle frame by frame (CNHW -> CHW per frame)
        xf = x.reshape(C, N, H, W)
        for n in range(N):
            o = orc.groupconv(np.ascontiguousarray(xf[:, n]), f, C, 1, 1, 3, act)
            check(got.reshape(C, N, H, W)[:, n], o, "dw3_stream %s frame %d vs oracle" % (shape, n))


@pytest.mark.parametrize("shape", [(3, 8, 2, 64, 96, 2), (3, 8, 1, 8, 8, 2), (3, 5, 3, 16, 40, 0), (3, 12, 2, 24, 16, 1), (3, 8, 1, 320, 320, 2)])
@pytest.mark.parametrize("first", [0, 1])
@pytest.mark.parametrize("stride", [2, 1])
def test_dense_first_layer(env, orc, shape, first, stride, monkeypatch):
    """3x3 dense conv from 3 channels, stride 2 (yolo-fastest) and stride 1 (the other darknet cfgs): k_conv_dense8 (first=0) and the
    four-pixels-per-thread k_conv_first (first=1, forced on small batches) against the generic kernel, each other (bit for bit) and the oracle"""
    capi, torch = env
    ic, oc, N, H, W, act = shape
    if stride == 1:
        oc = 2 * oc + 3                                   # (16 / 32 filters there: several blocks of 8, a ragged last one)
    monkeypatch.setenv("FFGPU_CONV_FIRST_MIN_PX", "1")
    monkeypatch.setenv("FFGPU_NO_CONV_FIRST", "0" if first else "1")
    rng = np.random.default_rng(hash(shape) & 0xffff)
    x = rng.uniform(-1, 1, (ic * N, H, W)).astype(np.float32)
    f = make_filter(rng, oc, 9 * ic)
    got = run_dev(capi, torch, x, f, N, W, H, ic, 1, 1, stride, 3, oc, act, capi.FFGPU.K_DENSE_SMALL)
    ref = run_dev(capi, torch, x, f, N, W, H, ic, 1, 1, stride, 3, oc, act, capi.FFGPU.K_GENERIC)
    check(got, ref, "dense first %s vs generic" % (shape,))
    monkeypatch.setenv("FFGPU_NO_CONV_FIRST", "1")
    other = run_dev(capi, torch, x, f, N, W, H, ic, 1, 1, stride, 3, oc, act, capi.FFGPU.K_DENSE_SMALL)
    assert np.array_equal(got, other), "k_conv_first and k_conv_dense8 differ"
    if H * W * N <= 20000:
        xf = x.reshape(ic, N, H, W)
        for n in range(N):
            o = orc.groupconv(np.ascontiguousarray(xf[:, n]), f, 1, 1, stride, 3, act)
            check(got.reshape(oc, N, H // stride, W // stride)[:, n], o, "dense first %s frame %d vs oracle" % (shape, n))


PW_SHAPES = [  # (ic, oc, N, H, W, act)
    (8, 8, 2, 160, 160, 2), (8, 4, 1, 16, 16, 0), (4, 24, 3, 20, 20, 2), (24, 136, 2, 20, 20, 2),
    (96, 255, 2, 10, 10, 0), (6, 7, 1, 6, 6, 2), (224, 48, 3, 10, 10, 0), (192, 96, 1, 10, 10, 2),
    (256, 512, 2, 20, 20, 2), (16, 96, 1, 3, 4, 1), (120, 120, 1, 20, 20, 3),
]


@pytest.mark.parametrize("shape", PW_SHAPES)
def test_pw_mfma(env, orc, shape):
    capi, torch = env
    ic, oc, N, H, W, act = shape
    rng = np.random.default_rng(hash(shape) & 0xffff)
    x = rng.uniform(-1, 1, (ic * N, H, W)).astype(np.float32)
    f = make_filter(rng, oc, ic)
    assert capi.kernel_name(N, W, H, ic, 1, 0, 1, 1, oc, capi.FFGPU.K_PW_MFMA) == "pw_mfma"
    got = run_dev(capi, torch, x, f, N, W, H, ic, 1, 0, 1, 1, oc, act, capi.FFGPU.K_PW_MFMA)
    ref = run_dev(capi, torch, x, f, N, W, H, ic, 1, 0, 1, 1, oc, act, capi.FFGPU.K_GENERIC)
    check(got, ref, "pw_mfma %s vs generic" % (shape,))
    if ic * oc * N * H * W <= 4e7:
        xf = x.reshape(ic, N, H, W)
        for n in range(N):
            o = orc.groupconv(np.ascontiguousarray(xf[:, n]), f, 1, 0, 1, 1, act)
            check(got.reshape(oc, N, H, W)[:, n], o, "pw_mfma %s frame %d vs oracle" % (shape, n))


DWL_SHAPES = [  # (C, N, H, W, fs, stride, act, compat)
    (24, 2, 160, 160, 3, 2, 2, 0), (32, 1, 80, 80, 3, 2, 2, 0), (136, 2, 20, 20, 3, 1, 2, 0), (224, 3, 10, 10, 3, 1, 2, 0),
    (96, 2, 10, 10, 5, 1, 2, 0), (96, 2, 10, 10, 5, 1, 2, 1), (120, 1, 20, 20, 5, 1, 2, 0), (120, 2, 20, 20, 5, 1, 2, 1),
    (5, 3, 7, 9, 3, 1, 0, 0), (4, 1, 7, 7, 3, 2, 1, 0), (3, 2, 6, 8, 5, 1, 2, 0), (2, 1, 33, 45, 5, 2, 2, 0), (3, 1, 1, 1, 3, 1, 2, 0),
    (136, 3, 45, 68, 3, 1, 3, 0), (40, 5, 37, 50, 5, 1, 2, 0),   # several planes per workgroup AND a partial last band (found by tests/test_gpu_fuzz.py)
]


@pytest.mark.parametrize("shape", DWL_SHAPES)
def test_dw_lds(env, orc, shape):
    capi, torch = env
    C, N, H, W, fs, stride, act, compat = shape
    pad = fs // 2
    rng = np.random.default_rng(hash(shape) & 0xffff)
    x = rng.uniform(-1, 1, (C * N, H, W)).astype(np.float32)
    f = make_filter(rng, C, fs * fs)
    flags = capi.FFGPU.COMPAT_V6 if compat else 0
    assert capi.kernel_name(N, W, H, C, C, pad, stride, fs, C, capi.FFGPU.K_DW_LDS) == "dw_lds"
    OH, OW = (H + 2 * pad - fs) // stride + 1, (W + 2 * pad - fs) // stride + 1
    dx, df = torch.from_numpy(x).cuda(), torch.from_numpy(f).cuda()
    outs = []
    for variant in (capi.FFGPU.K_DW_LDS, capi.FFGPU.K_GENERIC):
        dy = torch.full((C * N, OH, OW), float("nan"), device="cuda")
        capi.groupconv_dev(dx.data_ptr(), df.data_ptr(), dy.data_ptr(), N, W, H, C, C, pad, stride, fs, C, act, flags, variant, None)
        torch.cuda.synchronize()
        outs.append(dy.cpu().numpy())
    check(outs[0], outs[1], "dw_lds %s vs generic" % (shape,))
    xf = x.reshape(C, N, H, W)
    for n in range(N):
        o = orc.groupconv(np.ascontiguousarray(xf[:, n]), f, C, pad, stride, fs, act, compat)
        check(outs[0].reshape(C, N, OH, OW)[:, n], o, "dw_lds %s frame %d vs oracle" % (shape, n))


@pytest.mark.parametrize("pair", [1, 0])
@pytest.mark.parametrize("shape", [(120, 3, 20, 20, 5, 1, 2, 0), (96, 5, 10, 10, 5, 1, 2, 0), (6, 70, 20, 20, 5, 1, 0, 0), (4, 2, 13, 6, 5, 1, 1, 0),
                                   (2, 1, 7, 38, 5, 1, 2, 0), (8, 64, 40, 40, 5, 1, 2, 0), (10, 3, 22, 18, 5, 1, 2, 0)])
def test_dw5_channel_pairs(env, orc, shape, pair, monkeypatch):
    """depthwise 5x5 on small planes: the channel-pair kernel (k_dw_pair: frames split over several workgroups, ragged
    bands / quads, W % 4 != 0) and, forced, the k_dw_lds path it replaces there -- against generic and the oracle"""
    monkeypatch.setenv("FFGPU_NO_DW_PAIR", "0" if pair else "1")
    test_dw_lds(env, orc, shape)


# the last three fill the chip more than once (main tiles + narrow tiles behind them): ragged K (72 = 4.5 chunks), ragged channel
# tiles (300, 255), sigmoid
@pytest.mark.parametrize("shape", [(256, 512, 2, 20, 20, 2), (128, 255, 1, 20, 20, 0), (64, 130, 3, 10, 10, 2), (100, 200, 1, 12, 12, 1),
                                   (72, 300, 83, 20, 20, 2), (64, 256, 164, 20, 20, 0), (80, 255, 170, 20, 20, 3)])
@pytest.mark.parametrize("wm", [0, 4, 8])
def test_pw_gemm(env, orc, shape, wm, monkeypatch):
    """wm: the workgroup form -- 0 = the launcher's choice (8 waves / 256-channel tiles from 248 filters up, else 4 waves / 128), or forced"""
    capi, torch = env
    if wm:
        monkeypatch.setenv("FFGPU_PWG_WM", str(wm))
    ic, oc, N, H, W, act = shape
    rng = np.random.default_rng(hash(shape) & 0xffff)
    x = rng.uniform(-1, 1, (ic * N, H, W)).astype(np.float32)
    f = make_filter(rng, oc, ic)
    assert capi.kernel_name(N, W, H, ic, 1, 0, 1, 1, oc, capi.FFGPU.K_PW_GEMM) == "pw_gemm"
    got = run_dev(capi, torch, x, f, N, W, H, ic, 1, 0, 1, 1, oc, act, capi.FFGPU.K_PW_GEMM)
    ref = run_dev(capi, torch, x, f, N, W, H, ic, 1, 0, 1, 1, oc, act, capi.FFGPU.K_GENERIC)
    check(got, ref, "pw_gemm %s vs generic" % (shape,))
    if ic * oc * N * H * W <= 4e7:
        xf = x.reshape(ic, N, H, W)
        o = orc.groupconv(np.ascontiguousarray(xf[:, 0]), f, 1, 0, 1, 1, act)
        check(got.reshape(oc, N, H, W)[:, 0], o, "pw_gemm %s frame 0 vs oracle" % (shape,))


# ragged K (72, 100, 120: partial k-steps of 32), ragged channel blocks (255, 130, 21), one pixel tile and many, residual-free; sigmoid
@pytest.mark.parametrize("shape", [(256, 512, 2, 20, 20, 2), (120, 120, 3, 20, 20, 2), (120, 255, 2, 20, 20, 0), (192, 96, 4, 10, 10, 2), (96, 255, 2, 10, 10, 0),
                                   (64, 130, 3, 10, 10, 2), (100, 200, 1, 12, 12, 1), (72, 300, 21, 20, 20, 2), (33, 21, 1, 6, 6, 3), (8, 16, 2, 4, 4, 2)])
@pytest.mark.parametrize("mt", [0, 1, 2])
def test_pw_x3(env, orc, shape, mt, monkeypatch):
    """pointwise layer as SPLIT-bf16 products on the bf16 matrix cores (ffgpu_pw_x3.inc: three exact bf16 parts per operand, six partial
    products, fp32 accumulation): same tolerance as every fp32 kernel against generic and the oracle, and -- the claim of the kernel --
    as close to the oracle as the fp32 MFMA kernel is (a summation order, not a precision): |d| <= 2^-20 * scale' * sum|w x| + 1 ulp"""
    capi, torch = env
    if mt:
        monkeypatch.setenv("FFGPU_PWX3_MT", str(mt))
    ic, oc, N, H, W, act = shape
    rng = np.random.default_rng(hash(shape) & 0xffff)
    x = rng.uniform(-1, 1, (ic * N, H, W)).astype(np.float32)
    f = make_filter(rng, oc, ic)
    assert capi.kernel_name(N, W, H, ic, 1, 0, 1, 1, oc, capi.FFGPU.K_PW_X3) == "pw_x3"
    got = run_dev(capi, torch, x, f, N, W, H, ic, 1, 0, 1, 1, oc, act, capi.FFGPU.K_PW_X3)
    ref = run_dev(capi, torch, x, f, N, W, H, ic, 1, 0, 1, 1, oc, act, capi.FFGPU.K_GENERIC)
    check(got, ref, "pw_x3 %s vs generic" % (shape,))
    xf = x.reshape(ic, N, H, W)
    o = orc.groupconv(np.ascontiguousarray(xf[:, 0]), f, 1, 0, 1, 1, act)
    g0 = got.reshape(oc, N, H, W)[:, 0]
    check(g0, o, "pw_x3 %s frame 0 vs oracle" % (shape,))
    if act != 3:
        k4 = (ic + 3) & ~3
        bound = 2.0 ** -20 * np.abs(f[:, k4])[:, None, None] * np.einsum("ok,khw->ohw", np.abs(f[:, :ic]).astype(np.float64), np.abs(xf[:, 0]).astype(np.float64))
        assert np.all(np.abs(g0 - o) <= bound + 2.0 ** -22 * np.abs(o) + 1e-30), float(np.max(np.abs(g0 - o) / (bound + 1e-30)))


IGEMM_SHAPES = [  # (ic, oc, N, H, W, fs, stride, pad, act)
    (16, 32, 2, 24, 20, 3, 1, 1, 2), (32, 64, 1, 13, 13, 3, 1, 1, 2), (64, 130, 3, 7, 9, 3, 1, 1, 0), (8, 21, 2, 12, 8, 5, 1, 2, 0),
    (24, 48, 2, 17, 15, 3, 2, 1, 2), (12, 8, 1, 11, 11, 5, 2, 1, 1), (20, 10, 2, 9, 9, 3, 1, 0, 1), (128, 256, 1, 6, 4, 3, 1, 1, 2),
    (40, 70, 1, 5, 5, 1, 1, 0, 3), (9, 5, 3, 3, 3, 2, 1, 0, 2), (256, 160, 4, 6, 4, 3, 1, 1, 2), (16, 16, 1, 1, 1, 3, 1, 1, 2),
    # round 3: the quad-gather form (3x3 s1 p1, ow % 4 == 0): both tile shapes, ragged K (10 and 13 channels: 90 / 117 = 3.75 / 4.9 chunks
    # of 24), one-row and four-column planes (every quad touches both borders), ragged channel tiles, a tile count beyond one round
    (10, 24, 2, 8, 8, 3, 1, 1, 2), (13, 100, 1, 5, 12, 3, 1, 1, 0), (8, 64, 3, 1, 4, 3, 1, 1, 2), (32, 33, 2, 16, 4, 3, 1, 1, 1),
    (16, 32, 3, 52, 52, 3, 1, 1, 2), (64, 128, 2, 26, 28, 3, 1, 1, 3), (24, 200, 5, 20, 20, 3, 1, 1, 2),
    # ... and widths that are not multiples of 4 (rows padded to quads; the last quad of a row shifts the row's last four columns)
    (16, 40, 2, 13, 13, 3, 1, 1, 2), (32, 64, 3, 26, 26, 3, 1, 1, 2), (8, 16, 1, 5, 5, 3, 1, 1, 0), (12, 70, 2, 3, 7, 3, 1, 1, 1),
    (8, 9, 2, 9, 6, 3, 1, 1, 2), (64, 96, 1, 13, 13, 3, 1, 1, 2), (9, 130, 4, 4, 4, 3, 1, 1, 2),
]


@pytest.mark.parametrize("split", [0, 3])
@pytest.mark.parametrize("shape", IGEMM_SHAPES)
def test_conv_igemm(env, orc, shape, split, monkeypatch):
    """dense KxK as implicit GEMM (k_conv_igemm: both tile shapes, ragged K / channels / pixels, odd pixel counts, stride 2,
    pad 0, 5x5, 2x2, 1x1, sigmoid) against the generic kernel and, frame by frame, the oracle"""
    capi, torch = env
    if split:                                                       # split-K forced (3 parts, or one per chunk): partial sums + the reduction kernel
        monkeypatch.setenv("FFGPU_IGEMM_SPLIT", str(split))
    ic, oc, N, H, W, fs, stride, pad, act = shape
    rng = np.random.default_rng(hash(shape) & 0xffff)
    x = rng.uniform(-1, 1, (ic * N, H, W)).astype(np.float32)
    f = make_filter(rng, oc, fs * fs * ic)
    f[:, :fs * fs * ic] *= 3.0 / np.sqrt(fs * fs * ic)
    assert capi.kernel_name(N, W, H, ic, 1, pad, stride, fs, oc, capi.FFGPU.K_IGEMM) == "conv_igemm"
    got = run_dev(capi, torch, x, f, N, W, H, ic, 1, pad, stride, fs, oc, act, capi.FFGPU.K_IGEMM)
    ref = run_dev(capi, torch, x, f, N, W, H, ic, 1, pad, stride, fs, oc, act, capi.FFGPU.K_GENERIC)
    check(got, ref, "conv_igemm %s vs generic" % (shape,))
    OH, OW = (H + 2 * pad - fs) // stride + 1, (W + 2 * pad - fs) // stride + 1
    xf = x.reshape(ic, N, H, W)
    for n in range(N):
        o = orc.groupconv(np.ascontiguousarray(xf[:, n]), f, 1, pad, stride, fs, act)
        check(got.reshape(oc, N, OH, OW)[:, n], o, "conv_igemm %s frame %d vs oracle" % (shape, n))


CONV_X3_SHAPES = [s for s in IGEMM_SHAPES if s[5:8] == (3, 1, 1) and s[4] >= 4 and s[0] % 8 == 0] + [
    # several pixel ranges x several channel tiles; a ragged width with many frames; 16 channels (half of every k-step is padding); one group only
    (48, 72, 4, 40, 44, 3, 1, 1, 2), (96, 40, 9, 13, 13, 3, 1, 1, 2), (16, 16, 2, 64, 64, 3, 1, 1, 0), (8, 20, 1, 9, 12, 3, 1, 1, 1),
    (40, 48, 2, 30, 7, 3, 1, 1, 2), (56, 17, 3, 6, 10, 3, 1, 1, 3), (24, 100, 1, 5, 12, 3, 1, 1, 0), (8, 130, 4, 4, 4, 3, 1, 1, 2), (72, 24, 2, 8, 9, 3, 1, 1, 2),
]


CONV_X3_S2_SHAPES = [  # stride 2 (even planes, iw = 2 ow): the downsampling layers of yolov3 / yolov4-tiny; ragged output widths 5, 6, 13
    (16, 32, 2, 24, 20, 3, 2, 1, 2), (32, 64, 1, 26, 26, 3, 2, 1, 2), (64, 130, 3, 8, 10, 3, 2, 1, 0), (8, 16, 1, 8, 8, 3, 2, 1, 2), (24, 40, 2, 16, 8, 3, 2, 1, 1),
    (128, 96, 1, 52, 52, 3, 2, 1, 2), (40, 24, 5, 10, 12, 3, 2, 1, 2), (64, 64, 2, 28, 26, 3, 2, 1, 3), (32, 17, 4, 104, 104, 3, 2, 1, 2),
]


@pytest.mark.parametrize("mt,nw", [(0, 8), (4, 4), (2, 8), (1, 4)])
@pytest.mark.parametrize("shape", CONV_X3_SHAPES + CONV_X3_S2_SHAPES)
def test_conv_x3(env, orc, shape, mt, nw, monkeypatch):
    """dense 3x3 / pad 1 (stride 1, and stride 2 on even planes) as SPLIT-bf16 products on the bf16 matrix cores (ffgpu_conv_x3.inc): against the generic kernel and the
    oracle with the tolerance of every fp32 kernel, and -- the claim of the kernel -- as close to the oracle as an fp32 reorder is:
    |d| <= 2^-20 * scale' * sum|w x| + 1 ulp.  All channel-tile heights (MT) and both workgroup sizes; widths that are not multiples of 4."""
    capi, torch = env
    if mt:
        monkeypatch.setenv("FFGPU_IGX3_MT", str(mt))
    monkeypatch.setenv("FFGPU_IGX3_NW", str(nw))
    ic, oc, N, H, W, fs, stride, pad, act = shape
    OH, OW = H // stride, W // stride
    K = 9 * ic
    rng = np.random.default_rng(hash(shape) & 0xffff)
    x = rng.uniform(-1, 1, (ic * N, H, W)).astype(np.float32)
    f = make_filter(rng, oc, K)
    f[:, :K] *= 3.0 / np.sqrt(K)
    assert capi.kernel_name(N, W, H, ic, 1, 1, stride, 3, oc, capi.FFGPU.K_CONV_X3) == "conv_x3"
    got = run_dev(capi, torch, x, f, N, W, H, ic, 1, 1, stride, 3, oc, act, capi.FFGPU.K_CONV_X3)
    ref = run_dev(capi, torch, x, f, N, W, H, ic, 1, 1, stride, 3, oc, act, capi.FFGPU.K_GENERIC)
    check(got, ref, "conv_x3 %s vs generic" % (shape,))
    xf = x.reshape(ic, N, H, W)
    k4 = (K + 3) & ~3
    fa = np.abs(f)
    fa[:, k4], fa[:, k4 + 1] = 1.0, 0.0
    for n in range(N):
        o = orc.groupconv(np.ascontiguousarray(xf[:, n]), f, 1, 1, stride, 3, act)
        g = got.reshape(oc, N, OH, OW)[:, n]
        check(g, o, "conv_x3 %s frame %d vs oracle" % (shape, n))
        if act != 3 and n == 0:
            sabs = orc.groupconv(np.ascontiguousarray(np.abs(xf[:, n])), fa, 1, 1, stride, 3, 0).astype(np.float64)
            bound = 2.0 ** -20 * np.abs(f[:, k4])[:, None, None] * sabs
            assert np.all(np.abs(g - o) <= bound + 2.0 ** -22 * np.abs(o) + 1e-30), float(np.max(np.abs(g - o) / (bound + 1e-30)))


PW_X3S_SHAPES = [  # (ic, oc, N, H, W, act): the heads' layers of yolo-fastest and ragged relatives (channels in blocks of 8, pixels in quads)
    (120, 255, 64, 20, 20, 0), (120, 120, 16, 20, 20, 2), (96, 255, 64, 10, 10, 0), (192, 96, 8, 10, 10, 2), (8, 16, 1, 2, 2, 2), (40, 33, 3, 6, 10, 3),
    (136, 24, 5, 20, 20, 0), (224, 48, 2, 10, 10, 1), (16, 70, 7, 4, 9, 2), (256, 130, 1, 13, 12, 2),
]


@pytest.mark.parametrize("mt,nw", [(0, 4), (4, 8), (2, 4), (1, 4)])
@pytest.mark.parametrize("shape", PW_X3S_SHAPES)
def test_pw_x3s(env, orc, shape, mt, nw, monkeypatch):
    """the pointwise form of k_conv_x3 (FS = 1: one k-step per group, weights streamed by LDS-DMA, four waves per workgroup) -- the split-bf16 kernel for the
    SMALL 1x1 layers; same checks as test_conv_x3"""
    capi, torch = env
    if mt:
        monkeypatch.setenv("FFGPU_IGX3_MT", str(mt))
    monkeypatch.setenv("FFGPU_IGX3_NW", str(nw))
    ic, oc, N, H, W, act = shape
    rng = np.random.default_rng(hash(shape) & 0xffff)
    x = rng.uniform(-1, 1, (ic * N, H, W)).astype(np.float32)
    f = make_filter(rng, oc, ic)
    f[:, :ic] *= 3.0 / np.sqrt(ic)
    assert capi.kernel_name(N, W, H, ic, 1, 0, 1, 1, oc, capi.FFGPU.K_CONV_X3) == "pw_x3s"
    got = run_dev(capi, torch, x, f, N, W, H, ic, 1, 0, 1, 1, oc, act, capi.FFGPU.K_CONV_X3)
    ref = run_dev(capi, torch, x, f, N, W, H, ic, 1, 0, 1, 1, oc, act, capi.FFGPU.K_GENERIC)
    check(got, ref, "pw_x3s %s vs generic" % (shape,))
    xf = x.reshape(ic, N, H, W)
    k4 = (ic + 3) & ~3
    for n in range(0, N, 7):
        o = orc.groupconv(np.ascontiguousarray(xf[:, n]), f, 1, 0, 1, 1, act)
        g = got.reshape(oc, N, H, W)[:, n]
        check(g, o, "pw_x3s %s frame %d vs oracle" % (shape, n))
        if act != 3:
            bound = 2.0 ** -20 * np.abs(f[:, k4])[:, None, None] * np.einsum("ok,khw->ohw", np.abs(f[:, :ic]).astype(np.float64), np.abs(xf[:, n]).astype(np.float64))
            assert np.all(np.abs(g - o) <= bound + 2.0 ** -22 * np.abs(o) + 1e-30), float(np.max(np.abs(g - o) / (bound + 1e-30)))


PW_X3T_SHAPES = [  # (ic, oc, N, H, W, act): BASELINE config[2]'s layer at a small batch, ragged K (72, 100, 120: partial chunks of 16; 8: half a chunk),
    # ragged channel tiles (255, 300, 130, 21), one tile / main + narrow tiles / more than one round of workgroups, ragged last pixel tile, sigmoid, relu
    (256, 512, 2, 20, 20, 2), (120, 255, 64, 20, 20, 0), (120, 120, 3, 20, 20, 2), (192, 96, 4, 10, 10, 2), (72, 300, 83, 20, 20, 2), (64, 256, 164, 20, 20, 0),
    (80, 255, 170, 20, 20, 3), (100, 200, 1, 12, 12, 1), (33, 21, 1, 6, 6, 3), (8, 16, 2, 4, 4, 2), (16, 130, 5, 7, 12, 2), (256, 512, 40, 20, 20, 2),
]


@pytest.mark.parametrize("narrow", [1, 0, "nt"])
@pytest.mark.parametrize("shape", PW_X3T_SHAPES)
def test_pw_x3t(env, orc, shape, narrow, monkeypatch):
    """the TILED split-bf16 pointwise GEMM (ffgpu_pw_x3t.inc, round 5: every input value split once per 256 output channels by its workgroup, weights by
    LDS-DMA, two LDS buffers, v_mfma_f32_32x32x16_bf16): the checks of test_pw_x3 -- every fp32 kernel's tolerance against the generic kernel and the oracle,
    and the fp32-reorder bound |d| <= 2^-20 * scale' * sum|w x| + 1 ulp; with and without the narrow tiles of the partial last round, and with the
    streamed (non-temporal) output stores that outputs of 128 MB and more get by default forced on ("nt")"""
    capi, torch = env
    monkeypatch.setenv("FFGPU_PWXT_NARROW", "0" if narrow == 0 else "1")
    monkeypatch.setenv("FFGPU_PWXT_NT", "1" if narrow == "nt" else "0")
    ic, oc, N, H, W, act = shape
    rng = np.random.default_rng(hash(shape) & 0xffff)
    x = rng.uniform(-1, 1, (ic * N, H, W)).astype(np.float32)
    f = make_filter(rng, oc, ic)
    f[:, :ic] *= 3.0 / np.sqrt(ic)
    assert capi.kernel_name(N, W, H, ic, 1, 0, 1, 1, oc, capi.FFGPU.K_PW_X3T) == "pw_x3t"
    got = run_dev(capi, torch, x, f, N, W, H, ic, 1, 0, 1, 1, oc, act, capi.FFGPU.K_PW_X3T)
    ref = run_dev(capi, torch, x, f, N, W, H, ic, 1, 0, 1, 1, oc, act, capi.FFGPU.K_GENERIC)
    check(got, ref, "pw_x3t %s vs generic" % (shape,))
    xf = x.reshape(ic, N, H, W)
    k4 = (ic + 3) & ~3
    for n in sorted({0, N // 2, N - 1}):
        o = orc.groupconv(np.ascontiguousarray(xf[:, n]), f, 1, 0, 1, 1, act)
        g = got.reshape(oc, N, H, W)[:, n]
        check(g, o, "pw_x3t %s frame %d vs oracle" % (shape, n))
        if act != 3:
            bound = 2.0 ** -20 * np.abs(f[:, k4])[:, None, None] * np.einsum("ok,khw->ohw", np.abs(f[:, :ic]).astype(np.float64), np.abs(xf[:, n]).astype(np.float64))
            assert np.all(np.abs(g - o) <= bound + 2.0 ** -22 * np.abs(o) + 1e-30), float(np.max(np.abs(g - o) / (bound + 1e-30)))


def test_conv_x3_auto_pick(env, monkeypatch):
    """AUTO gives the big 3x3 layers of a darknet backbone to conv_x3, keeps k_conv_igemm (split-K) for launches that would not fill the chip,
    and FFGPU_IG_X3=0 switches it off"""
    capi, torch = env
    assert capi.kernel_name(16, 104, 104, 32, 1, 1, 1, 3, 64) == "conv_x3"
    assert capi.kernel_name(1, 13, 13, 256, 1, 1, 1, 3, 512) == "conv_igemm"
    assert capi.kernel_name(16, 104, 104, 32, 1, 1, 2, 3, 64) == "conv_x3"             # stride 2 on an even plane
    assert capi.kernel_name(16, 13, 13, 32, 1, 1, 2, 3, 64) == "conv_igemm"            # ... an odd one
    assert capi.kernel_name(16, 104, 104, 36, 1, 1, 1, 3, 64) == "conv_igemm"          # input channels not in whole blocks of 8
    monkeypatch.setenv("FFGPU_IG_X3", "0")
    assert capi.kernel_name(16, 104, 104, 32, 1, 1, 1, 3, 64) == "conv_igemm"


THIN_SHAPES = [  # (ic, oc, groups, N, H, W, fs, stride, pad, act): 2..7 input channels per group
    (8, 8, 4, 2, 9, 11, 3, 1, 1, 2), (12, 24, 4, 1, 13, 13, 3, 1, 1, 2), (64, 64, 16, 3, 20, 20, 3, 1, 1, 2), (28, 16, 4, 2, 7, 5, 5, 1, 2, 0),
    (6, 9, 3, 2, 17, 16, 3, 2, 1, 1), (10, 30, 2, 1, 8, 8, 1, 1, 0, 3), (21, 35, 7, 2, 6, 6, 3, 1, 0, 2), (4, 5, 1, 2, 10, 9, 2, 1, 0, 2),
    (7, 8, 1, 1, 5, 5, 7, 1, 3, 2), (32, 8, 8, 2, 1, 1, 3, 1, 1, 0), (48, 48, 8, 1, 40, 40, 3, 1, 1, 2), (6, 2, 2, 3, 300, 1, 3, 1, 1, 2),
]


@pytest.mark.parametrize("shape", THIN_SHAPES)
def test_conv_thin_groups(env, orc, shape):
    """grouped (and thin dense) convolutions with 2..7 input channels per group on k_conv_thin (scalar filter taps, several outputs
    per lane; conv-v0.c:7-31, 46-51): against the generic kernel -- the same k-ordered fmaf chain, so equal to the last bit where no
    product is a signed zero -- and, frame by frame, the oracle"""
    capi, torch = env
    ic, oc, groups, N, H, W, fs, stride, pad, act = shape
    rng = np.random.default_rng(hash(shape) & 0xffff)
    x = rng.uniform(-1, 1, (ic * N, H, W)).astype(np.float32)
    f = make_filter(rng, oc, fs * fs * ic // groups)
    assert capi.kernel_name(N, W, H, ic, groups, pad, stride, fs, oc) == "conv_thin"
    got = run_dev(capi, torch, x, f, N, W, H, ic, groups, pad, stride, fs, oc, act, capi.FFGPU.K_AUTO)
    ref = run_dev(capi, torch, x, f, N, W, H, ic, groups, pad, stride, fs, oc, act, capi.FFGPU.K_GENERIC)
    assert np.array_equal(got, ref), "conv_thin %s differs from the generic kernel (max |d| %.3g)" % (shape, np.abs(got - ref).max())
    OH, OW = (H + 2 * pad - fs) // stride + 1, (W + 2 * pad - fs) // stride + 1
    xf = x.reshape(ic, N, H, W)
    for n in range(N):
        o = orc.groupconv(np.ascontiguousarray(xf[:, n]), f, groups, pad, stride, fs, act)
        check(got.reshape(oc, N, OH, OW)[:, n], o, "conv_thin %s frame %d vs oracle" % (shape, n))


@pytest.mark.parametrize("seed", range(4))
def test_conv_thin_groups_random(env, seed):
    """seeded random thin-group geometries (2..7 channels per group, 1..12 filters per group, filter sizes 1-7, strides 1-3, any
    padding up to the filter size, planes from 1x1 to 70x40, residual on or off): k_conv_thin == k_conv_generic to the last bit"""
    capi, torch = env
    rng = np.random.default_rng(7100 + seed)
    done = 0
    while done < 16:
        gic, goc, groups = int(rng.integers(2, 8)), int(rng.integers(1, 13)), int(rng.integers(1, 7))
        fs, stride = int(rng.choice([1, 2, 3, 3, 5, 7])), int(rng.integers(1, 4))
        pad = int(rng.integers(0, fs + 1))
        N, H, W = int(rng.integers(1, 4)), int(rng.integers(1, 41)), int(rng.integers(1, 71))
        if H + 2 * pad < fs or W + 2 * pad < fs:
            continue
        ic, oc, act = gic * groups, goc * groups, int(rng.integers(0, 4))
        if capi.kernel_name(N, W, H, ic, groups, pad, stride, fs, oc) != "conv_thin":
            continue                                                 # (dense 3x3 / 5x5 with <= 8 channels belong to k_conv_dense8)
        x = rng.uniform(-1, 1, (ic * N, H, W)).astype(np.float32)
        f = make_filter(rng, oc, fs * fs * gic)
        got = run_dev(capi, torch, x, f, N, W, H, ic, groups, pad, stride, fs, oc, act, capi.FFGPU.K_AUTO)
        ref = run_dev(capi, torch, x, f, N, W, H, ic, groups, pad, stride, fs, oc, act, capi.FFGPU.K_GENERIC)
        assert np.array_equal(got, ref), "ic %d oc %d groups %d N %d %dx%d fs %d s %d p %d act %d: max |d| %.3g" % (
            ic, oc, groups, N, H, W, fs, stride, pad, act, np.abs(got - ref).max())
        done += 1


@pytest.mark.parametrize("shape", [(32, 48, 2, 2, 11, 9, 3, 1, 1, 2), (64, 64, 4, 1, 8, 8, 3, 2, 1, 2), (48, 24, 3, 3, 7, 7, 1, 1, 0, 0), (32, 160, 2, 1, 6, 10, 5, 1, 2, 1)])
def test_conv_igemm_grouped(env, orc, shape):
    """grouped convolutions with >= 8 channels per group (conv-v0.c:46-51: group g uses its slice of input channels, filter
    rows and output channels): one implicit-GEMM launch per group, against the oracle"""
    capi, torch = env
    ic, oc, groups, N, H, W, fs, stride, pad, act = shape
    rng = np.random.default_rng(hash(shape) & 0xffff)
    x = rng.uniform(-1, 1, (ic * N, H, W)).astype(np.float32)
    f = make_filter(rng, oc, fs * fs * ic // groups)
    assert capi.kernel_name(N, W, H, ic, groups, pad, stride, fs, oc) == "conv_igemm"
    got = run_dev(capi, torch, x, f, N, W, H, ic, groups, pad, stride, fs, oc, act, capi.FFGPU.K_AUTO)
    OH, OW = (H + 2 * pad - fs) // stride + 1, (W + 2 * pad - fs) // stride + 1
    xf = x.reshape(ic, N, H, W)
    for n in range(N):
        o = orc.groupconv(np.ascontiguousarray(xf[:, n]), f, groups, pad, stride, fs, act)
        check(got.reshape(oc, N, OH, OW)[:, n], o, "grouped igemm %s frame %d vs oracle" % (shape, n))


def test_conv_igemm_frame_major_input_and_residual(env, orc):
    """the strides the executor hands over for a first layer (frame-major batch input) and a fused shortcut"""
    capi, torch = env
    ic, oc, N, H, W = 16, 24, 3, 10, 14
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, (N, ic, H, W)).astype(np.float32)                   # frame-major
    f = make_filter(rng, oc, 9 * ic)
    ref = np.stack([orc.groupconv(x[n], f, 1, 1, 1, 3, 2) for n in range(N)], 1)   # (oc, N, H, W)
    cn = np.ascontiguousarray(x.transpose(1, 0, 2, 3)).reshape(ic * N, H, W)
    got = run_dev(capi, torch, cn, f, N, W, H, ic, 1, 1, 1, 3, oc, 2, capi.FFGPU.K_IGEMM)
    check(got.reshape(oc, N, H, W), ref, "igemm CNHW")


DWPW_SHAPES = [  # (C, OC, N, H, W, fs, actd, actp)
    (120, 120, 3, 20, 20, 5, 2, 0), (96, 96, 2, 10, 10, 5, 2, 0), (16, 24, 2, 13, 7, 3, 2, 2), (20, 100, 1, 9, 33, 5, 1, 0),
    (8, 8, 1, 4, 4, 3, 0, 0), (40, 17, 2, 40, 40, 5, 2, 1), (12, 20, 1, 6, 6, 3, 2, 0), (30, 128, 1, 17, 5, 5, 2, 0), (4, 4, 5, 4, 64, 3, 2, 0),
]


@pytest.mark.parametrize("shape", DWPW_SHAPES)
def test_dwpw_fused_pair(env, orc, shape):
    """depthwise KxK + pointwise 1x1 in one kernel (k_dwpw: the heads' pairs of yolo-fastest and ragged relatives: channel counts
    off the 8 / 16 grids, planes that do not divide into tiles, one-tile and many-tile planes) against the oracle's two
    groupconv calls per frame"""
    capi, torch = env
    C, OC, N, H, W, fs, actd, actp = shape
    rng = np.random.default_rng(hash(shape) & 0xffff)
    x = rng.uniform(-1, 1, (C * N, H, W)).astype(np.float32)
    fd, fp = make_filter(rng, C, fs * fs), make_filter(rng, OC, C)
    fp[:, :C] *= 4.0 / np.sqrt(C)
    t = [torch.from_numpy(a).cuda() for a in (x, fd, fp)]
    out = torch.full((OC * N, H, W), float("nan"), device="cuda")
    capi.dwpw_dev(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), out.data_ptr(), N, W, H, C, OC, fs, actd, actp)
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(OC, N, H, W)
    xf = x.reshape(C, N, H, W)
    for n in range(N):
        o1 = orc.groupconv(np.ascontiguousarray(xf[:, n]), fd, C, fs // 2, 1, fs, actd)
        o2 = orc.groupconv(o1, fp, 1, 0, 1, 1, actp)
        check(got[:, n], o2, "dwpw %s frame %d vs oracle" % (shape, n))


@pytest.mark.parametrize("shape", [(256, 512, 3, 20, 20, 2), (128, 130, 2, 10, 10, 0), (160, 255, 1, 12, 12, 1), (136, 128, 2, 20, 20, 2)])
def test_pw_bf16(env, orc, shape):
    """the OPT-IN bf16 pointwise variant (FFGPU_BF16_PW; never the default): inputs and weights rounded to bf16, fp32
    accumulation.  Its own tolerance, stated here: |d| <= 2^-7 * |scale'| * sum_k |w_k x_k| + 1e-5 per output (two roundings
    of relative size 2^-9 per product, worst case) against the fp32 oracle; and it must really differ from fp32 (i.e. run)."""
    capi, torch = env
    ic, oc, N, H, W, act = shape
    rng = np.random.default_rng(hash(shape) & 0xffff)
    x = rng.uniform(-1, 1, (ic * N, H, W)).astype(np.float32)
    f = make_filter(rng, oc, ic)
    assert capi.kernel_name(N, W, H, ic, 1, 0, 1, 1, oc, capi.FFGPU.K_PW_BF16) == "pw_bf16"
    dx, df = torch.from_numpy(x).cuda(), torch.from_numpy(f).cuda()
    dy = torch.full((oc * N, H, W), float("nan"), device="cuda")
    capi.groupconv_dev(dx.data_ptr(), df.data_ptr(), dy.data_ptr(), N, W, H, ic, 1, 0, 1, 1, oc, act, capi.FFGPU.BF16_PW, capi.FFGPU.K_PW_BF16, None)
    torch.cuda.synchronize()
    got = dy.cpu().numpy().reshape(oc, N, H * W)
    assert not np.isnan(got).any()
    k4 = (ic + 3) & ~3
    w, sc = f[:, :ic], np.abs(f[:, k4])
    xf = x.reshape(ic, N, H * W)
    worst = 0.0
    for n in range(N):
        ref = orc.groupconv(np.ascontiguousarray(x.reshape(ic, N, H, W)[:, n]), f, 1, 0, 1, 1, act).reshape(oc, H * W)
        bound = 2.0 ** -7 * sc[:, None] * (np.abs(w) @ np.abs(xf[:, n])) + 1e-5
        err = np.abs(got[:, n] - ref)
        assert (err <= bound).all(), "pw_bf16 %s frame %d: max excess %.3g" % (shape, n, (err - bound).max())
        worst = max(worst, float(err.max()))
    assert worst > 1e-5, "bf16 variant returned fp32-exact results: it did not run"
    # AUTO never picks it unless the flag is passed
    assert capi.kernel_name(N, W, H, ic, 1, 0, 1, 1, oc) in ("pw_gemm", "pw_mfma", "pw_x3", "pw_x3s")


def test_unsupported_variant_fails_loudly(env):
    capi, torch = env
    x = torch.zeros((4, 7, 7), device="cuda")          # W % 4 != 0: the stream kernel must refuse
    f = torch.zeros((4, 16), device="cuda")
    y = torch.zeros((4, 7, 7), device="cuda")
    with pytest.raises(RuntimeError, match="does not support"):
        capi.groupconv_dev(x.data_ptr(), f.data_ptr(), y.data_ptr(), 1, 7, 7, 4, 4, 1, 1, 3, 4, 2, 0, capi.FFGPU.K_DW_STREAM)


def test_full_size_properties(env):
    """BASELINE config[1] size (320x320x64, batch 64): linearity + shift properties instead of an oracle run."""
    capi, torch = env
    C, N, H, W = 64, 8, 320, 320          # 8 frames here; bench.py runs the 64-frame case
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.rand((C * N, H, W), device="cuda", generator=g) - 0.5
    f = torch.zeros((C, 16), device="cuda")
    f[:, :9] = torch.rand((C, 9), device="cuda", generator=g) - 0.5
    f[:, 12] = 1.0
    y1, y2, y3 = (torch.empty_like(x) for _ in range(3))

    def run(inp, out):
        capi.groupconv_dev(inp.data_ptr(), f.data_ptr(), out.data_ptr(), N, W, H, C, C, 1, 1, 3, C, 0, 0, capi.FFGPU.K_DW_STREAM)
    run(x, y1)
    x2 = 2.0 * x
    run(x2, y2)
    torch.cuda.synchronize()
    assert torch.allclose(y2, 2 * y1, rtol=1e-5, atol=1e-5)            # linear layer, bias 0
    # a plane made of one impulse reproduces the flipped 3x3 taps around it (checks column/row neighbours,
    # including across the lane-63/lane-0 seam at column 255/256)
    imp = torch.zeros_like(x)
    imp[:, 100, 255] = 1.0
    imp[:, 200, 256] = 1.0
    run(imp, y3)
    torch.cuda.synchronize()
    taps = f[:, :9].reshape(C, 3, 3).repeat_interleave(N, dim=0)
    for (r, c) in ((100, 255), (200, 256)):
        patch = y3[:, r - 1:r + 2, c - 1:c + 2]
        assert torch.allclose(patch, torch.flip(taps, dims=(1, 2)), atol=1e-6), (r, c)


IRB_SHAPES = [  # (ic, ec, oc, stride, N, H, W, residual)
    (8, 32, 8, 1, 2, 80, 80, True), (4, 24, 8, 2, 2, 160, 160, False), (8, 48, 16, 1, 3, 40, 40, False),
    (16, 96, 16, 1, 2, 40, 40, True), (16, 96, 24, 2, 2, 40, 40, False), (24, 136, 24, 1, 3, 20, 20, True),
    (24, 136, 48, 2, 2, 20, 20, False), (48, 224, 48, 1, 3, 10, 10, True), (8, 8, 4, 1, 1, 32, 48, False),
    (4, 8, 4, 1, 2, 16, 16, True), (12, 40, 20, 1, 1, 12, 20, True), (6, 30, 10, 2, 2, 16, 12, False),
    (8, 32, 8, 1, 2, 13, 11, True), (16, 50, 30, 1, 1, 9, 22, True), (3, 20, 5, 2, 3, 21, 17, True), (8, 64, 16, 2, 1, 30, 26, False),
]


@pytest.mark.parametrize("shape", IRB_SHAPES)
def test_irb_fused_block(env, shape, orc=None):
    """fused expand->dw3x3->project(+shortcut) == three generic groupconv launches + add on the same tensors, and == the
    ORACLE's three groupconv calls + shortcut per frame (conv-v0.c:7-31, ffcnn.c:418-423)"""
    capi, torch = env
    if orc is None:
        from oracle import orc as _orc
        _orc.build()
        orc = _orc
    ic, ec, oc, stride, N, H, W, use_res = shape
    rng = np.random.default_rng(hash(shape) & 0xffff)
    x = rng.uniform(-1, 1, (ic * N, H, W)).astype(np.float32)
    f1, fd, f2 = make_filter(rng, ec, ic), make_filter(rng, ec, 9), make_filter(rng, oc, ec)
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    res = rng.uniform(-1, 1, (oc * N, OH, OW)).astype(np.float32)
    e1 = run_dev(capi, torch, x, f1, N, W, H, ic, 1, 0, 1, 1, ec, 2, capi.FFGPU.K_GENERIC)
    e2 = run_dev(capi, torch, e1, fd, N, W, H, ec, ec, 1, stride, 3, ec, 2, capi.FFGPU.K_GENERIC)
    ref = run_dev(capi, torch, e2, f2, N, OW, OH, ec, 1, 0, 1, 1, oc, 0, capi.FFGPU.K_GENERIC)
    if use_res:
        ref = ref + res
    t = [torch.from_numpy(a).cuda() for a in (x, f1, fd, f2, res)]
    out = torch.full((oc * N, OH, OW), float("nan"), device="cuda")
    capi.irb_dev(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), t[4].data_ptr() if use_res else None,
                 out.data_ptr(), N, W, H, ic, ec, oc, stride)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    check(got, ref, "irb %s" % (shape,))
    xf, rf, gf = x.reshape(ic, N, H, W), res.reshape(oc, N, OH, OW), got.reshape(oc, N, OH, OW)
    for n in range(N):
        o1 = orc.groupconv(np.ascontiguousarray(xf[:, n]), f1, 1, 0, 1, 1, 2)
        o2 = orc.groupconv(o1, fd, ec, 1, stride, 3, 2)
        o3 = orc.groupconv(o2, f2, 1, 0, 1, 1, 0)
        if use_res:
            o3 = orc.shortcut(o3, np.ascontiguousarray(rf[:, n]), 0)
        check(gf[:, n], o3, "irb %s frame %d vs oracle" % (shape, n))


@pytest.mark.parametrize("band", [0, 1, 3, 5, 16])
@pytest.mark.parametrize("shape", [(8, 8, 4, 1, 2, 37, 160, False), (4, 8, 4, 1, 3, 21, 48, True), (8, 8, 8, 1, 1, 16, 256, True),
                                   (4, 8, 8, 1, 2, 9, 4, False), (4, 8, 4, 1, 1, 1, 8, True), (8, 8, 4, 1, 2, 2, 12, False)])
def test_irb_thin_bands(env, shape, band, monkeypatch):
    """the 8-channel streaming block (k_irb_thin): every channel-count instantiation, rows split into bands of any length
    (ragged last band, bands longer than the plane, single-row planes), 1 to 64 lanes per row"""
    if band:
        monkeypatch.setenv("FFGPU_THIN_BAND", str(band))
    test_irb_fused_block(env, shape)


@pytest.mark.parametrize("tile", [(0, 0), (5, 6), (4, 8), (2, 10), (8, 4)])
@pytest.mark.parametrize("shape", [(8, 32, 8, 1, 2, 80, 80, True), (8, 48, 16, 1, 3, 40, 40, False), (16, 96, 16, 1, 2, 40, 40, True),
                                   (8, 32, 8, 1, 2, 13, 11, True), (5, 20, 7, 1, 1, 23, 30, True)])
def test_irb_wave_two_strips(env, shape, tile, monkeypatch):
    """two output strips per wave (k_irbw2): forced on small test planes, several tile shapes incl. ragged edges"""
    monkeypatch.setenv("FFGPU_IRBW2_MIN_TILES", "1")
    monkeypatch.setenv("FFGPU_IRBW_G", "1")
    if tile[0]:
        monkeypatch.setenv("FFGPU_IRBW2_TWQ", str(tile[0]))
        monkeypatch.setenv("FFGPU_IRBW2_TH", str(tile[1]))
    test_irb_fused_block(env, shape)


@pytest.mark.parametrize("shape", [(8, 32, 8, 1, 7, 80, 80, True), (16, 96, 16, 1, 5, 40, 40, True), (24, 136, 24, 1, 11, 20, 20, True), (48, 224, 48, 1, 13, 10, 10, True),
                                   (4, 24, 8, 2, 3, 160, 160, False), (16, 96, 24, 2, 9, 40, 40, False)])
def test_irb_wave_tile_order(env, shape, monkeypatch):
    """the fused blocks walk their tiles XCD by XCD (irbw_xcd_block, round 5): workgroup counts that are no multiple of 8, one-wave-per-tile and group-split
    launches -- against the oracle (every tile computed exactly once), and bit for bit the same as in workgroup-id order (a tile's arithmetic does not depend on who runs it)"""
    capi, torch = env
    ic, ec, oc, stride, N, H, W, use_res = shape
    rng = np.random.default_rng(7)
    x = rng.uniform(-1, 1, (ic * N, H, W)).astype(np.float32)
    f1, fd, f2 = make_filter(rng, ec, ic), make_filter(rng, ec, 9), make_filter(rng, oc, ec)
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    res = rng.uniform(-1, 1, (oc * N, OH, OW)).astype(np.float32)
    t = [torch.from_numpy(a).cuda() for a in (x, f1, fd, f2, res)]
    outs = []
    for order in ("0", "3"):
        monkeypatch.setenv("FFGPU_IRBW_XCD", order)
        out = torch.full((oc * N, OH, OW), float("nan"), device="cuda")
        capi.irb_dev(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), t[4].data_ptr() if use_res else None, out.data_ptr(), N, W, H, ic, ec, oc, stride)
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy())
    assert not np.isnan(outs[1]).any() and np.array_equal(outs[0], outs[1])
    monkeypatch.setenv("FFGPU_IRBW_XCD", "3")
    test_irb_fused_block(env, shape)


@pytest.mark.parametrize("G", [1, 2, 3, 5, 8])
@pytest.mark.parametrize("shape", [(24, 136, 24, 1, 3, 20, 20, True), (48, 224, 48, 1, 2, 10, 10, True), (8, 48, 16, 1, 2, 40, 40, False),
                                   (4, 24, 8, 2, 2, 48, 32, True)])
def test_irb_wave_group_split(env, shape, G, monkeypatch):
    """wave-autonomous fused block with the channel groups of a tile split over G waves (fixed-order LDS reduction)"""
    monkeypatch.setenv("FFGPU_IRBW_G", str(G))
    test_irb_fused_block(env, shape)

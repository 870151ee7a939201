"""Round 4: fp32-equivalent products on the bf16 matrix cores ("X3": each operand = three exact bf16 parts, six partial products, fp32
accumulation; ffcnn_amd/csrc/ffgpu_irb_wave.inc, ffgpu_pw_x3.inc).  The claim under test is that the split form is NOT a reduced
precision: it must sit as close to the oracle (conv-v0.c restated, oracle/) as the fp32-MFMA form of the same kernel does, launch after
launch -- the second half matters because the one defect of the round was timing dependent (a packed FMA with op_sel broadcasts right
behind bf16 MFMAs dropped its addend on a quarter of a wave about once in 10^4 tiles; only back-to-back launches showed it)."""
import os

import numpy as np
import pytest

from test_gpu_kernels import make_filter

YOLO_KIND = 7                                                   # LAYER_TYPE_YOLO (include/ffcnn.h:45)


@pytest.fixture(scope="module")
def env():
    import torch
    from ffcnn_amd import capi
    from oracle import orc
    orc.build()
    capi.lib()
    return capi, torch, orc


# (ic, ec, oc, N, H, W): the three families that run the split expand GEMM by default + the two that can (FFGPU_IRBW_X3 bit 0)
X3_BLOCKS = [(16, 96, 16, 24, 40, 40), (24, 136, 24, 24, 20, 20), (48, 224, 48, 24, 10, 10), (8, 48, 8, 24, 40, 40), (8, 32, 8, 8, 80, 80)]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", X3_BLOCKS)
def test_x3_fused_block_is_an_fp32_reorder(env, shape, monkeypatch):
    capi, torch, orc = env
    ic, ec, oc, N, H, W = shape
    rng = np.random.default_rng(hash(shape) & 0xffff)
    x = rng.uniform(-1, 1, (ic * N, H, W)).astype(np.float32)
    f1, fd, f2 = make_filter(rng, ec, ic), make_filter(rng, ec, 9), make_filter(rng, oc, ec)
    res = rng.uniform(-1, 1, (oc * N, H, W)).astype(np.float32)
    t = [torch.from_numpy(a).cuda() for a in (x, f1, fd, f2, res)]
    outs = {}
    for mode in ("0", "15"):
        monkeypatch.setenv("FFGPU_IRBW_X3", mode)
        out = torch.full((oc * N, H, W), float("nan"), device="cuda")
        # 40 launches back to back, the last one's output is checked (a timing-dependent slip shows as a handful of wrong tiles)
        capi.irb_dev(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), t[4].data_ptr(), out.data_ptr(), N, W, H, ic, ec, oc, 1, warmup=0, iters=40)
        torch.cuda.synchronize()
        outs[mode] = out.cpu().numpy().reshape(oc, N, H, W)
    assert not np.isnan(outs["15"]).any()
    xf, rf = x.reshape(ic, N, H, W), res.reshape(oc, N, H, W)
    worst = {"0": 0.0, "15": 0.0}
    for n in range(0, N, 5):
        o1 = orc.groupconv(np.ascontiguousarray(xf[:, n]), f1, 1, 0, 1, 1, 2)
        o2 = orc.groupconv(o1, fd, ec, 1, 1, 3, 2)
        o3 = orc.shortcut(orc.groupconv(o2, f2, 1, 0, 1, 1, 0), np.ascontiguousarray(rf[:, n]), 0)
        for k in outs:
            worst[k] = max(worst[k], float(np.abs(outs[k][:, n] - o3).max()))
    scale = float(np.abs(outs["0"]).max())
    # both forms within a few fp32 ulps of the block's largest value of the oracle, the split form no further than twice the fp32 form
    assert worst["0"] <= 8e-7 * scale + 1e-6 and worst["15"] <= 8e-7 * scale + 1e-6, (worst, scale)
    assert worst["15"] <= 2.0 * worst["0"] + 1e-6, worst
    assert float(np.abs(outs["15"] - outs["0"]).max()) <= 8e-7 * scale + 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(256, 512, 64, 20, 20), (120, 255, 32, 20, 20), (96, 255, 32, 10, 10), (72, 300, 21, 20, 20)])
@pytest.mark.parametrize("mt", [4, 2, 1])
def test_pw_x3_back_to_back(env, shape, mt, monkeypatch):
    """k_pw_x3 with leaky activation, random scale' / bias', 30 launches back to back: the regression test of the dropped-addend defect"""
    capi, torch, orc = env
    monkeypatch.setenv("FFGPU_PWX3_MT", str(mt))
    ic, oc, N, H, W = shape
    rng = np.random.default_rng(hash(shape) & 0xffff)
    x = rng.uniform(-1, 1, (ic * N, H, W)).astype(np.float32)
    f = make_filter(rng, oc, ic)
    dx, df = torch.from_numpy(x).cuda(), torch.from_numpy(f).cuda()
    ref = torch.empty((oc * N, H, W), device="cuda")
    capi.groupconv_dev(dx.data_ptr(), df.data_ptr(), ref.data_ptr(), N, W, H, ic, 1, 0, 1, 1, oc, 2, 0, capi.FFGPU.K_PW_MFMA, None)
    for rep in range(3):
        y = torch.full((oc * N, H, W), float("nan"), device="cuda")
        capi.groupconv_time_dev(dx.data_ptr(), df.data_ptr(), y.data_ptr(), N, W, H, ic, 1, 0, 1, 1, oc, act=2, variant=capi.FFGPU.K_PW_X3, warmup=0, iters=30)
        torch.cuda.synchronize()
        d = (y - ref).abs()
        assert not torch.isnan(y).any() and float(d.max()) <= 1e-4, "rep %d: %d outputs off, max |d| %.3g" % (rep, int((d > 1e-4).sum()), float(d.max()))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(120, 255, 64, 20, 20, 1), (256, 512, 64, 20, 20, 1), (96, 96, 64, 10, 10, 1), (64, 128, 16, 26, 26, 3), (32, 64, 8, 52, 52, 3)])
def test_conv_x3_back_to_back(env, shape):
    """both forms of k_conv_x3 (pointwise / 3x3) with leaky activation and random scale' / bias', 30 launches back to back, three times: the watch for
    timing-dependent slips behind bf16 MFMAs (the dropped-addend hazard of section 5.10 showed only this way)"""
    capi, torch, orc = env
    ic, oc, N, H, W, fs = shape
    rng = np.random.default_rng(hash(shape) & 0xffff)
    x = rng.uniform(-1, 1, (ic * N, H, W)).astype(np.float32)
    f = make_filter(rng, oc, fs * fs * ic)
    f[:, :fs * fs * ic] *= 3.0 / np.sqrt(fs * fs * ic)
    pad = fs // 2
    dx, df = torch.from_numpy(x).cuda(), torch.from_numpy(f).cuda()
    ref = torch.empty((oc * N, H, W), device="cuda")
    capi.groupconv_dev(dx.data_ptr(), df.data_ptr(), ref.data_ptr(), N, W, H, ic, 1, pad, 1, fs, oc, 2, 0, capi.FFGPU.K_PW_MFMA if fs == 1 else capi.FFGPU.K_IGEMM, None)
    for rep in range(3):
        y = torch.full((oc * N, H, W), float("nan"), device="cuda")
        capi.groupconv_time_dev(dx.data_ptr(), df.data_ptr(), y.data_ptr(), N, W, H, ic, 1, pad, 1, fs, oc, act=2, variant=capi.FFGPU.K_CONV_X3, warmup=0, iters=30)
        torch.cuda.synchronize()
        d = (y - ref).abs()
        assert not torch.isnan(y).any() and float(d.max()) <= 1e-4, "rep %d: %d outputs off, max |d| %.3g" % (rep, int((d > 1e-4).sum()), float(d.max()))


# ---- dense 3x3 layers on the split form (ffgpu_conv_x3.inc) inside a PLANNED net: plan-time image, MT frozen in the plan, fused shortcut,
# ragged widths (52 -> 26 -> 13), maxpool / route / upsample around them -- a darknet-tiny-style backbone
def boxes_match_up_to_threshold_flips(got, want, thresholds, what):
    """the boxes of a darknet-style cfg against the oracle's, by class, score (1e-4) and corners (0.05), in the reference's score order (ffcnn.c:301).
    A box may be missing on either side only if its score lies within 1e-4 of a yolo layer's ignore_thres (ffcnn.c:259-289 keeps `conf >= thres` and stores conf
    as the score: a conf within rounding of the threshold flips between two fp32 orders of the same dot product).  Such a box has the lowest score of its frame,
    so it suppresses nothing above it in the NMS.  At most 2 % of a frame's boxes (one at least) may flip.
    Corners: 0.05 pixel + 1e-4 of the box's own scale.  The sizes are exp(t) x anchor (ffcnn.c:259-289), so a difference d in the yolo layer's input t is a
    RELATIVE d in the size; with random weights these cfgs produce boxes up to 3e7 pixels tall, where the 1e-5 of the yolo-fastest helpers cannot hold:
    measured on dark3.cfg (tools/diag_boxes.py) the yolo inputs sit within 5.9e-5 of the oracle's (the layer check admits 1e-3), every box has its partner at
    the same place with the score within 1.3e-6 and the corners within a relative 4.1e-5."""
    def scale(b):
        return max(abs(float(b["x2"]) - float(b["x1"])), abs(float(b["y2"]) - float(b["y1"])), max(abs(float(b[c])) for c in ("x1", "y1", "x2", "y2")))
    used = np.zeros(len(got), bool)
    pairs, lost = [], []
    for k, w in enumerate(want):
        hit = -1
        for j in range(len(got)):
            g = got[j]
            if (not used[j] and int(g["type"]) == int(w["type"]) and abs(float(g["score"]) - float(w["score"])) <= 1e-4 and
                    all(abs(float(g[c]) - float(w[c])) <= 0.05 + 1e-4 * scale(w) for c in ("x1", "y1", "x2", "y2"))):
                hit = j
                break
        if hit < 0:
            lost.append(("oracle", k, w))
        else:
            used[hit] = True
            pairs.append((k, hit))
    lost += [("hip", j, got[j]) for j in range(len(got)) if not used[j]]
    for side, k, b in lost:
        assert min(abs(float(b["score"]) - t) for t in thresholds) <= 1e-4, "%s: %s box %d %r has no counterpart and is not at a threshold %r" % (what, side, k, b, thresholds)
    assert len(lost) <= max(1, len(want) // 50), "%s: %d boxes flipped at the threshold (%d reference boxes)" % (what, len(lost), len(want))
    for (k0, j0), (k1, j1) in zip(pairs, pairs[1:]):              # same order, except between scores an ulp apart
        assert j1 > j0 or abs(float(want[k1]["score"]) - float(want[k0]["score"])) <= 2e-6, "%s: boxes %d / %d change places across a real score gap" % (what, k0, k1)


def _yolo_thresholds(n):
    t = sorted(set(round(float(L.ignore_thres), 6) for L in (n.layer(i) for i in range(n.layer_num)) if L.type == YOLO_KIND))
    assert t, "no yolo layer"
    return t


def _c(filters, size, stride, act, bn=1):
    return "[convolutional]\n%sfilters=%d\nsize=%d\nstride=%d\npad=1\nactivation=%s\n\n" % ("batch_normalize=1\n" if bn else "", filters, size, stride, act)


DENSE3_CFG = "[net]\nwidth=104\nheight=104\nchannels=3\n\n" + _c(16, 3, 1, "leaky") + "[maxpool]\nsize=2\nstride=2\n\n" + _c(32, 3, 1, "leaky") + \
    "[maxpool]\nsize=2\nstride=2\n\n" + _c(64, 3, 1, "leaky") + _c(64, 3, 1, "linear") + "[shortcut]\nfrom=-2\nactivation=leaky\n\n" + \
    "[maxpool]\nsize=2\nstride=2\n\n" + _c(128, 3, 1, "leaky") + _c(40, 1, 1, "leaky") + _c(72, 3, 1, "leaky") + _c(21, 1, 1, "linear", bn=0) + \
    "[yolo]\nmask = 0,1,2\nanchors = 6,8, 10,14, 20,18, 30,40, 50,44, 70,80\nclasses=2\nignore_thresh = .55\nscale_x_y = 1.05\n\n" + \
    "[route]\nlayers = -4\n\n" + _c(24, 1, 1, "leaky") + "[upsample]\nstride=2\n\n" + "[route]\nlayers = -1, 5\n\n" + _c(48, 3, 1, "leaky") + _c(21, 1, 1, "linear", bn=0) + \
    "[yolo]\nmask = 3,4,5\nanchors = 6,8, 10,14, 20,18, 30,40, 50,44, 70,80\nclasses=2\nignore_thresh = .55\nscale_x_y = 1.05\n\n"


# a yolov3-style backbone: 3x3 stride-2 downsampling layers, 1x1 / 3x3 residual blocks, two heads with a route + upsample
DARK_CFG = "[net]\nwidth=96\nheight=128\nchannels=3\n\n" + _c(16, 3, 1, "leaky") + _c(32, 3, 2, "leaky") + _c(16, 1, 1, "leaky") + _c(32, 3, 1, "leaky") + \
    "[shortcut]\nfrom=-3\nactivation=linear\n\n" + _c(64, 3, 2, "leaky") + _c(32, 1, 1, "leaky") + _c(64, 3, 1, "leaky") + "[shortcut]\nfrom=-3\nactivation=linear\n\n" + \
    _c(128, 3, 2, "leaky") + _c(64, 1, 1, "leaky") + _c(128, 3, 1, "leaky") + "[shortcut]\nfrom=-3\nactivation=linear\n\n" + _c(21, 1, 1, "linear", bn=0) + \
    "[yolo]\nmask = 0,1,2\nanchors = 6,8, 10,14, 20,18, 30,40, 50,44, 70,80\nclasses=2\nignore_thresh = .55\nscale_x_y = 1.05\n\n" + \
    "[route]\nlayers = -3\n\n" + _c(32, 1, 1, "leaky") + "[upsample]\nstride=2\n\n" + "[route]\nlayers = -1, 8\n\n" + _c(64, 3, 1, "leaky") + _c(21, 1, 1, "linear", bn=0) + \
    "[yolo]\nmask = 3,4,5\nanchors = 6,8, 10,14, 20,18, 30,40, 50,44, 70,80\nclasses=2\nignore_thresh = .55\nscale_x_y = 1.05\n\n"


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["tiny", "dark"])
@pytest.mark.parametrize("batch", [3, 16])
def test_conv_x3_layers_in_an_executor(env, tmp_path, batch, which, monkeypatch):
    from test_gpu_parity import _write_random_weights, close
    capi, torch, orc = env
    monkeypatch.setenv("FFGPU_IGX3_MIN_WGS", "1")                 # small planes too: every eligible layer on the split form
    if which == "tiny":
        # the shapes of this cfg the split form must take (16 -> 32 on 52x52, 32 -> 64 / 64 -> 64 on 26x26, 64 -> 128 / 40 -> 72 on 13x13, 88 -> 48 on 26x26)
        for (ic, oc, hw) in ((16, 32, 52), (32, 64, 26), (64, 64, 26), (64, 128, 13), (40, 72, 13), (88, 48, 26)):
            assert capi.kernel_name(batch, hw, hw, ic, 1, 1, 1, 3, oc) == "conv_x3", (ic, oc, hw)
    else:
        # stride 2: 16 -> 32 from 96x128, 32 -> 64 from 48x64, 64 -> 128 from 24x32; stride 1: 16 -> 32 on 48x64, 32 -> 64 on 24x32, 64 -> 128 on 12x16, 96 -> 64 on 24x32
        for (ic, oc, w, h, st) in ((16, 32, 96, 128, 2), (32, 64, 48, 64, 2), (64, 128, 24, 32, 2), (16, 32, 48, 64, 1), (32, 64, 24, 32, 1), (64, 128, 12, 16, 1), (96, 64, 24, 32, 1)):
            assert capi.kernel_name(batch, w, h, ic, 1, 1, st, 3, oc) == "conv_x3", (ic, oc, w, h, st)
    cfg = str(tmp_path / "dense3.cfg")
    open(cfg, "w").write(DENSE3_CFG if which == "tiny" else DARK_CFG)
    o = orc.Oracle(cfg=cfg, weights=None)
    wpath = str(tmp_path / "dense3.weights")
    _write_random_weights(wpath, o, 41)
    o.close()
    o = orc.Oracle(cfg=cfg, weights=wpath)
    rng = np.random.default_rng(42)
    frames = rng.uniform(0, 1, (batch, 3, 104, 104) if which == "tiny" else (batch, 3, 128, 96)).astype(np.float32)
    with capi.Net(cfg, wpath) as n:
        for flags in (capi.FFGPU.KEEP_ALL | capi.FFGPU.NO_FUSE, capi.FFGPU.KEEP_ALL, capi.FFGPU.CONCURRENT, capi.FFGPU.NO_GRAPH):
            with n.executor(batch, flags) as ex:
                ex.set_scale(1, 1)
                for rep in range(2):                               # (the second forward replays the graph)
                    ex.forward_host(frames)
                for f in range(0, batch, 5):
                    o.input[...] = frames[f]
                    o.n.s1, o.n.s2 = 1, 1
                    o.forward(0)
                    seen = 0
                    for i in range(o.nlayers):
                        ref = o.layer_out(i)
                        if ref is None or not (flags & capi.FFGPU.KEEP_ALL):
                            continue
                        try:
                            a = ex.read_layer(i, f)
                        except RuntimeError as e:
                            assert "not materialised" in str(e)
                            continue
                        seen += 1
                        close(a, ref, "dense3 cfg flags %d frame %d layer %d" % (flags, f, i))
                    assert seen >= 8 or not (flags & capi.FFGPU.KEEP_ALL)
                    boxes_match_up_to_threshold_flips(ex.read_boxes(f), o.boxes, _yolo_thresholds(n), "dense3 cfg %s flags %d frame %d" % (which, flags, f))
    o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 5, 11, 13, 27, 29])
def test_random_generic_nets_with_conv_x3(tmp_path, seed, monkeypatch):
    """the random generic nets of test_gpu_fuzz_nets.py with every eligible dense 3x3 layer forced onto the split form"""
    from oracle import orc
    from test_gpu_fuzz_nets import test_random_nets_fused_vs_oracle as run_net
    orc.build()
    monkeypatch.setenv("FFGPU_IGX3_MIN_WGS", "1")
    monkeypatch.setenv("FFGPU_IGX3_MIN_IC", "8")
    run_net(orc, tmp_path, seed, "generic")


# (test_concurrent_executors_reproduce_themselves -- 11 head layers of ~560 executor-forwards against their own first round -- became
#  tests/test_gpu_round5.py::test_concurrent_executors_against_the_oracle_then_themselves in round 5: round 0 against the ORACLE, every later round on EVERY
#  materialised layer and frame, the bench's own 4 x 64 configuration with fp32 and u8 frames, thousands of executor-forwards.)


@pytest.mark.gpu
@pytest.mark.parametrize("batch,force", [(2, True), (9, True), (4, False)])
def test_dark3_cfg_against_the_oracle(env, tmp_path, batch, force, monkeypatch):
    """tests/data/dark3.cfg (yolov3-shaped: 3x3 stride-2 downsampling layers, 1x1 + 3x3 residual blocks, two heads; what tools/other_nets.py times at 416x416) at
    128x96 so that the oracle finishes in seconds: every layer and the boxes, with every eligible layer forced onto the split-bf16 kernels and with the default picks"""
    from conftest import ROOT
    from test_gpu_parity import _write_random_weights, close
    capi, torch, orc = env
    if force:
        for k, v in (("FFGPU_IGX3_MIN_WGS", "1"), ("FFGPU_PWX3S_MIN_WGS", "1"), ("FFGPU_PWX3S_MIN_IC", "8"), ("FFGPU_PWX3S_MIN_OC", "8")):
            monkeypatch.setenv(k, v)
    txt = open(os.path.join(ROOT, "tests", "data", "dark3.cfg")).read().replace("width=416", "width=128").replace("height=416", "height=96")
    cfg = str(tmp_path / "dark3_small.cfg")
    open(cfg, "w").write(txt)
    o = orc.Oracle(cfg=cfg, weights=None)
    wpath = str(tmp_path / "dark3.weights")
    _write_random_weights(wpath, o, 23)
    o.close()
    o = orc.Oracle(cfg=cfg, weights=wpath)
    rng = np.random.default_rng(24)
    frames = rng.uniform(0, 1, (batch, 3, 96, 128)).astype(np.float32)
    refs = []
    for f in range(batch):
        o.input[...] = frames[f]
        o.n.s1, o.n.s2 = 1, 1
        o.forward(0)
        refs.append(({i: o.layer_out(i).copy() for i in range(o.nlayers) if o.layer_out(i) is not None}, o.boxes.copy()))
    with capi.Net(cfg, wpath) as n:
        assert n.layer_num == o.nlayers
        if force:
            names = set(capi.kernel_name(batch, L.w, L.h, L.c, 1, L.pad, L.stride, L.fs, L.fn) for L in (n.layer(i) for i in range(n.layer_num)) if L.type == 0)
            assert "conv_x3" in names and "pw_x3s" in names, names
        for flags in (capi.FFGPU.KEEP_ALL | capi.FFGPU.NO_FUSE, capi.FFGPU.KEEP_ALL, 0):
            with n.executor(batch, flags) as ex:
                ex.set_scale(1, 1)
                ex.forward_host(frames)
                for f in range(batch):
                    if flags & capi.FFGPU.KEEP_ALL:
                        for i, ref in refs[f][0].items():
                            try:
                                a = ex.read_layer(i, f)
                            except RuntimeError as e:
                                assert "not materialised" in str(e)
                                continue
                            close(a, ref, "dark3 flags %d frame %d layer %d" % (flags, f, i))
                    boxes_match_up_to_threshold_flips(ex.read_boxes(f), refs[f][1], _yolo_thresholds(n), "dark3 flags %d frame %d" % (flags, f))
    o.close()

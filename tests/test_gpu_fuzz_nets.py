"""Seeded random darknet cfgs in yolo-fastest's grammar (stem 3x3 s2, inverted-residual blocks = 1x1 expand / depthwise 3x3 s1|s2
/ 1x1 linear project [+ dropout + shortcut], one or two 1x1 + yolo heads) through the PLANNER and the fused kernels, against the
oracle (which restates ffcnn.c:476-520 layer by layer): every tensor the fused executor still materialises, for every frame,
at plane sizes and channel counts the shipped model never has (k_front / k_irb_thin / k_irbw / k_irbw2 / k_irb tile logic,
group splits, stride-2 halos, ragged last tiles, 8..200 expanded channels), default and FFGPU_CONCURRENT plans, batches 1-6."""
import os

import numpy as np
import pytest

from test_gpu_parity import _write_random_weights, boxes_match, close

pytestmark = pytest.mark.gpu


def boxes_match_near_ties(got, want, what):
    """boxes_match, except that boxes whose scores agree to 2e-6 may come in either order: the reference sorts by score
    (ffcnn.c:301), random nets produce thousands of boxes with scores within an ulp of each other, and a last-bit difference
    in a score swaps two neighbours without being an error.  Every box must have its partner (class, score within 1e-4,
    corners within 0.05) at most 8 places away."""
    try:
        boxes_match(got, want, what)
        return
    except AssertionError:
        pass
    assert len(got) == len(want), "%s: %d vs %d boxes" % (what, len(got), len(want))
    used = np.zeros(len(want), bool)
    for i, g in enumerate(got):
        lo, hi = max(0, i - 8), min(len(want), i + 9)
        ok = -1
        for j in range(lo, hi):
            w = want[j]
            if (not used[j] and int(g["type"]) == int(w["type"]) and abs(float(g["score"]) - float(w["score"])) <= 1e-4 and
                    all(abs(float(g[k]) - float(w[k])) <= 0.05 for k in ("x1", "y1", "x2", "y2"))):
                ok = j
                break
        assert ok >= 0, "%s: box %d %r has no partner near its place" % (what, i, g)
        assert ok == i or abs(float(want[ok]["score"]) - float(want[i]["score"])) <= 2e-6, "%s: box %d moved %d places across a real score gap" % (what, i, ok - i)
        used[ok] = True


def conv(filters, size, stride, act, groups=1, bn=1):
    s = "[convolutional]\n"
    if groups > 1:
        s += "groups=%d\n" % groups
    return s + "filters=%d\nsize=%d\nstride=%d\npad=%d\n%sactivation=%s\n\n" % (filters, size, stride, 1 if size > 1 else 0, "batch_normalize=1\n" if bn else "", act)


def random_cfg(rng):
    """-> (cfg text, (H, W)).  Layers are counted while the text grows so that routes can name absolute indices."""
    big = rng.random() < 0.3                                    # inputs large enough for the big-plane plans (k_front, k_irb_thin)
    W = int(rng.choice([256, 320]) if big else rng.choice([64, 96, 128, 160, 224]))
    H = int(rng.choice([192, 320]) if big else rng.choice([64, 96, 128, 160]))
    classes = int(rng.choice([1, 2, 5]))
    L = {"n": 0, "txt": "[net]\nwidth=%d\nheight=%d\nchannels=3\n\n" % (W, H)}

    def add(t, k=1):
        L["txt"] += t
        L["n"] += k
        return L["n"] - 1                                       # index of the last layer added

    c = 8 if big else int(rng.choice([4, 8, 8, 12, 16]))
    add(conv(c, 3, 2, "leaky"))
    w, h = W // 2, H // 2
    if big or rng.random() < 0.5:                               # yolo-fastest's opening: pointwise, depthwise, linear projection
        c2 = 4 if big else int(rng.choice([4, 8]))
        add(conv(c, 1, 1, "leaky") + conv(c, 3, 1, "leaky", groups=c) + conv(c2, 1, 1, "linear"), 3)
        c = c2
    taps = []                                                   # (layer index, channels, w, h) of block outputs, for the second head
    nblocks = int(rng.integers(3, 9))
    for b in range(nblocks):
        if rng.random() < 0.15:                                 # not an inverted-residual block: dense 3x3 + 1x1 with the shortcut in the epilogue
            last = add(conv(c, 3, 1, "leaky") + conv(c, 1, 1, "linear") + "[shortcut]\nfrom=-3\nactivation=%s\n\n" % rng.choice(["linear", "leaky", "relu"]), 3)
            taps.append((last, c, w, h))
            continue
        if rng.random() < 0.1:                                  # ... or the shortcut straight behind a dense / grouped 3x3 (the implicit-GEMM epilogue)
            g = 2 if (c % 2 == 0 and rng.random() < 0.4) else 1
            last = add(conv(c, 3, 1, str(rng.choice(["linear", "leaky"])), groups=g) + "[shortcut]\nfrom=-2\nactivation=%s\n\n" % rng.choice(["linear", "leaky"]), 2)
            taps.append((last, c, w, h))
            continue
        stride = 2 if (rng.random() < 0.35 and min(w, h) >= 8) else 1
        ec = int(rng.choice([8, 16, 24, 32, 40, 48, 72, 96, 136, 200]))
        oc = c if (stride == 1 and rng.random() < 0.6) else int(rng.choice([4, 8, 16, 24, 48]))
        last = add(conv(ec, 1, 1, "leaky") + conv(ec, 3, stride, "leaky", groups=ec) + conv(oc, 1, 1, "linear"), 3)
        if stride == 1 and oc == c:
            last = add("[dropout]\nprobability=.15\n\n[shortcut]\nfrom=-5\nactivation=linear\n\n", 2)
        c = oc
        w, h = (w + stride - 1) // stride, (h + stride - 1) // stride
        taps.append((last, c, w, h))
    if rng.random() < 0.4:                                      # SPP: three max pools of the same tensor + the tensor (ffcnn.c:381-394)
        base = L["n"] - 1
        add("[maxpool]\nsize=3\nstride=1\n\n[route]\nlayers = %d\n\n[maxpool]\nsize=5\nstride=1\n\n[route]\nlayers = %d\n\n[maxpool]\nsize=9\nstride=1\n\n"
            "[route]\nlayers = -1,-3,-5,%d\n\n" % (base, base, base), 6)
        c *= 4
        add(conv(int(rng.choice([16, 24, 48])), 1, 1, "leaky"))
        c = None

    def head(mask, dw5):
        if dw5:                                                 # the heads' 5x5 depthwise + pointwise pairs (layers 116-119 / 125-128)
            hc = int(rng.choice([16, 24, 40, 96]))
            add(conv(hc, 1, 1, "leaky") + "[convolutional]\ngroups=%d\nfilters=%d\nsize=5\nstride=1\npad=1\nbatch_normalize=1\nactivation=leaky\n\n" % (hc, hc) + conv(hc, 1, 1, "linear"), 3)
        add(conv(3 * (5 + classes), 1, 1, "linear", bn=0))
        add("[yolo]\nmask = %s\nanchors = 6,8, 10,14, 20,18, 30,40, 50,44, 70,80\nclasses=%d\nignore_thresh = .55\nscale_x_y = 1.05\n\n" % (mask, classes))

    feat = L["n"] - 1                                           # what the first head reads
    head("3,4,5", rng.random() < 0.5)
    r = rng.random()
    if r < 0.35:                                                # a second head on the same tensor through a route
        add("[route]\nlayers = %d\n\n" % feat)
        add(conv(int(rng.choice([16, 32])), 1, 1, "leaky"))
        head("0,1,2", rng.random() < 0.5)
    elif r < 0.7:                                               # ... or on the upsampled tensor joined with an earlier block output of twice the size
        wf, hf = w, h
        cand = [t for t in taps if t[2] == 2 * wf and t[3] == 2 * hf]
        if cand:
            t = cand[int(rng.integers(0, len(cand)))]
            add("[route]\nlayers = %d\n\n" % feat)
            add("[upsample]\nstride=2\n\n")
            add("[route]\nlayers = -1,%d\n\n" % t[0])
            head("0,1,2", rng.random() < 0.5)
    return L["txt"], (H, W)


def random_generic_cfg(rng):
    """A random layer GRAPH over everything the ffcnn grammar has (ffcnn.c:127-240): convolutions of any size / stride /
    padding / grouping / activation, max and average pools (size 2..5, stride 1..3), upsample, dropout, shortcuts that reach
    back over several layers, routes joining one to three earlier tensors, one to three yolo heads.  Shapes are tracked so that
    every join is legal; -> (text, (H, W))."""
    W, H = int(rng.choice([32, 64, 96, 128])), int(rng.choice([32, 64, 96]))
    if rng.random() < 0.3:                                      # a cfg's own size is taken as it is (ffcnn.c:131-134): any integers
        W, H = int(rng.integers(17, 131)), int(rng.integers(17, 101))
    CIN = int(rng.choice([3, 3, 3, 1, 4]))
    classes = int(rng.choice([1, 3]))
    txt = "[net]\nwidth=%d\nheight=%d\nchannels=%d\n\n" % (W, H, CIN)
    shapes = []                                                 # per layer: (c, w, h)
    cur = (CIN, W, H)

    def emit(t, shape):
        nonlocal txt, cur
        txt += t
        shapes.append(shape)
        cur = shape

    def conv_any():
        c, w, h = cur
        fs = int(rng.choice([1, 1, 3, 3, 3, 5, 2]))
        stride = int(rng.choice([1, 1, 1, 2])) if min(w, h) >= 8 else 1
        pad = int(rng.choice([1, 1, 0])) if fs > 1 else 0      # darknet: pad=1 -> size / 2
        padpx = fs // 2 if pad else 0
        ow, oh = (w + 2 * padpx - fs) // stride + 1, (h + 2 * padpx - fs) // stride + 1
        if ow < 1 or oh < 1:
            return False
        oc = int(rng.choice([4, 6, 8, 12, 16, 24, 32, 40, 64]))
        groups = 1
        r = rng.random()
        if r < 0.2 and fs > 1:
            oc, groups = c, c                                   # depthwise
        elif r < 0.35:
            for g in (2, 3, 4, 8):
                if c % g == 0 and rng.random() < 0.5:
                    groups = g
                    oc = g * int(rng.choice([2, 4, 8, 12]))
                    break
        act = str(rng.choice(["leaky", "leaky", "relu", "linear"]))
        bn = int(rng.random() < 0.8)
        t = "[convolutional]\n" + ("groups=%d\n" % groups if groups > 1 else "") + "filters=%d\nsize=%d\nstride=%d\npad=%d\n%sactivation=%s\n\n" % (
            oc, fs, stride, pad, "batch_normalize=1\n" if bn else "", act)
        emit(t, (oc, ow, oh))
        return True

    emit("[convolutional]\nfilters=%d\nsize=3\nstride=%d\npad=1\nbatch_normalize=1\nactivation=leaky\n\n" % (int(rng.choice([4, 8, 16])), int(rng.choice([1, 2]))),
         None)
    shapes[-1] = cur = (int(txt.split("filters=")[1].split("\n")[0]), (W + 2 - 3) // int(txt.split("stride=")[1].split("\n")[0]) + 1,
                        (H + 2 - 3) // int(txt.split("stride=")[1].split("\n")[0]) + 1)
    nheads = 0
    for step in range(int(rng.integers(6, 18))):
        c, w, h = cur
        r = rng.random()
        if r < 0.5:
            conv_any()
        elif r < 0.6 and min(w, h) >= 4:
            kind = str(rng.choice(["maxpool", "avgpool", "max", "avg"]))
            fs, st = int(rng.choice([2, 3, 3, 5])), int(rng.choice([1, 2, 2, 3]))
            if w // st >= 1 and h // st >= 1:
                emit("[%s]\nsize=%d\nstride=%d\n\n" % (kind, fs, st), (c, w // st, h // st))
        elif r < 0.65 and max(w, h) <= 24:
            st = int(rng.choice([2, 2, 3]))
            emit("[upsample]\nstride=%d\n\n" % st, (c, w * st, h * st))
        elif r < 0.7:
            emit("[dropout]\nprobability=.2\n\n", cur)
        elif r < 0.82:                                          # shortcut with any earlier layer of the same shape
            cand = [i for i, sh in enumerate(shapes[:-1]) if sh == cur]
            if cand:
                i = cand[int(rng.integers(0, len(cand)))]
                emit("[shortcut]\nfrom=%d\nactivation=%s\n\n" % (i - len(shapes), rng.choice(["linear", "leaky", "relu"])), cur)
        elif r < 0.93:                                          # route: the previous tensor joined with earlier ones of the same plane size
            # (ffcnn.c:178-183: a positive number is an absolute layer index, anything else counts back from here -- layer 0 has no
            #  absolute name; the channels add up, the plane size is the last source's)
            cand = [i for i, sh in enumerate(shapes[:-1]) if sh[1:] == cur[1:]]
            k = int(rng.integers(0, min(3, len(cand)) + 1))
            pick = [len(shapes) - 1] + [cand[int(j)] for j in rng.choice(len(cand), k, replace=False)] if cand and k else [[i for i in range(len(shapes)) if shapes[i][0] > 0][-1 - int(rng.integers(0, min(3, sum(1 for sh in shapes if sh[0] > 0))))]]
            ctot = sum(shapes[i][0] for i in pick)
            if ctot <= 200:
                emit("[route]\nlayers = %s\n\n" % ",".join(str(i if (i > 0 and rng.random() < 0.5) else i - len(shapes)) for i in pick), (ctot, shapes[pick[0]][1], shapes[pick[0]][2]))
        elif nheads < 2 and step > 3:
            feat = len(shapes) - 1
            emit("[convolutional]\nfilters=%d\nsize=1\nstride=1\npad=1\nactivation=linear\n\n" % (3 * (5 + classes)), (3 * (5 + classes), w, h))
            emit("[yolo]\nmask = %s\nanchors = 4,6, 8,12, 16,14, 24,30, 40,36, 60,70\nclasses=%d\nignore_thresh = .6\nscale_x_y = 1.05\n\n" % (
                rng.choice(["0,1,2", "3,4,5", "1,3,5"]), classes), (0, 0, 0))      # (a yolo layer has no output tensor: never a source)
            nheads += 1
            emit("[route]\nlayers = %d\n\n" % (feat if feat > 0 else feat - len(shapes)), shapes[feat])
    c, w, h = cur
    emit("[convolutional]\nfilters=%d\nsize=1\nstride=1\npad=1\nactivation=linear\n\n" % (3 * (5 + classes)), (3 * (5 + classes), w, h))
    emit("[yolo]\nmask = 0,1,2\nanchors = 4,6, 8,12, 16,14, 24,30, 40,36, 60,70\nclasses=%d\nignore_thresh = .6\nscale_x_y = 1.05\n\n" % classes, (0, 0, 0))
    return txt, (H, W, CIN)


SEEDS = range(int(os.environ.get("FFCNN_FUZZ_SEED0", "0")), int(os.environ.get("FFCNN_FUZZ_SEED0", "0")) + int(os.environ.get("FFCNN_FUZZ_NETS", "12")))


@pytest.mark.parametrize("grammar", ["mobile", "generic"])
@pytest.mark.parametrize("seed", SEEDS)
def test_random_nets_fused_vs_oracle(orc, tmp_path, seed, grammar):
    from ffcnn_amd import capi as F
    F.lib()
    rng = np.random.default_rng((9100 if grammar == "mobile" else 20000) + seed)
    txt, hw = random_cfg(rng) if grammar == "mobile" else random_generic_cfg(rng)
    H, W, CIN = hw if len(hw) == 3 else (hw[0], hw[1], 3)
    cfg = str(tmp_path / "rnd.cfg")
    open(cfg, "w").write(txt)
    o = orc.Oracle(cfg=cfg, weights=None)
    wpath = str(tmp_path / "rnd.weights")
    _write_random_weights(wpath, o, 100 + seed)
    o.close()
    o = orc.Oracle(cfg=cfg, weights=wpath)
    B = int(rng.integers(1, 7))
    if H * W <= 128 * 128 and rng.random() < 0.25:               # plans are functions of the batch: now and then a bigger, odd one
        B = int(rng.choice([9, 17, 33, 64]))
    frames = rng.uniform(0, 1, (B, CIN, H, W)).astype(np.float32)
    acts, cands, boxes = [], [], []
    for f in range(B):
        o.input[...] = frames[f]
        o.n.s1, o.n.s2 = 1, 1
        o.forward(0)
        acts.append({i: o.layer_out(i).copy() for i in range(o.nlayers) if o.layer_out(i) is not None})
        cands.append(o.candidates)
        boxes.append(o.boxes)
    # a candidate whose confidence sits within 2e-3 of the threshold may legitimately flip under the 1e-3 activation tolerance
    edge = any(min(abs(float(c["score"]) - 0.55), abs(float(c["score"]) - 0.6)) < 2e-3 for cs in cands for c in cs)
    with F.Net(cfg, wpath) as n:
        assert n.layer_num == o.nlayers and np.array_equal(n.weights_host(), o.weights())
        extra = int(rng.choice([F.FFGPU.SPLIT2, F.FFGPU.SPLIT2 | F.FFGPU.CONCURRENT, F.FFGPU.HOST_DETS, F.FFGPU.NO_GRAPH | F.FFGPU.CONCURRENT, F.FFGPU.NO_FUSE]))
        for flags in (F.FFGPU.KEEP_ALL, F.FFGPU.KEEP_ALL | F.FFGPU.CONCURRENT, 0, extra):
            with n.executor(B, flags) as ex:
                ex.set_scale(1, 1)
                ex.forward_host(frames)
                dets = ex.read_dets()
                seen = 0
                if flags & F.FFGPU.KEEP_ALL:
                    for i in sorted(acts[0]):
                        if n.layer(i).type == 4:
                            continue
                        try:
                            ex.read_layer(i, 0)
                        except RuntimeError as e:
                            assert "not materialised" in str(e)
                            continue
                        seen += 1
                        for f in range(B):
                            close(ex.read_layer(i, f), acts[f][i], "seed %d flags %d frame %d layer %d (%s)\n%s" % (seed, flags, f, i, ex.plan_text() if hasattr(ex, "plan_text") else "", ""))
                    assert seen >= 3 or grammar != "mobile"
                if not edge:
                    for f in range(B):
                        # more candidates than bbox_max (ffcnn.c:243,463: input bytes / 24): the reference keeps the first bbox_max in
                        # emission order, and so must the boxes here; the record's ncand is the untruncated count
                        if len(cands[f]) < o.n.cap:
                            assert dets[f]["ncand"] == len(cands[f]), "seed %d flags %d frame %d" % (seed, flags, f)
                        else:
                            assert dets[f]["ncand"] >= len(cands[f])
                        # random weights can drive exp(tw) to 1e10 pixels: boxes are compared where the 0.05-pixel tolerance
                        # means something (every candidate within +-2000 pixels); the full list, not the 128 of the record
                        # ... and where greedy NMS is not chaotic: with thousands of overlapping boxes one overlap ratio within rounding
                        # of the 0.5 threshold changes who survives (seen once in 4 500 nets: 3 403 vs 3 401 survivors)
                        if len(boxes[f]) > 300:
                            assert abs(len(ex.read_boxes(f)) - len(boxes[f])) <= max(2, len(boxes[f]) // 200), "seed %d flags %d frame %d: box count" % (seed, flags, f)
                        elif all(abs(float(c[k])) < 2000 for c in cands[f] for k in ("x1", "y1", "x2", "y2")):
                            try:
                                boxes_match_near_ties(ex.read_boxes(f), boxes[f], "seed %d flags %d frame %d boxes" % (seed, flags, f))
                            except AssertionError:
                                # two candidates of one class whose scores agree to the last bits: which of them is sorted first -- and
                                # suppresses the other -- is decided by rounding; only without such a pair is a difference an error
                                sc = np.sort(np.array([float(c["score"]) + 10.0 * int(c["type"]) for c in cands[f]], np.float64))
                                if len(sc) < 2 or np.min(np.diff(sc)) > 2e-6:
                                    raise
    o.close()


GROUPED_CFG = "[net]\nwidth=64\nheight=64\nchannels=3\n\n" + conv(16, 3, 2, "leaky") + conv(48, 1, 1, "leaky") + \
    conv(48, 3, 1, "linear", groups=2) + "[shortcut]\nfrom=-2\nactivation=leaky\n\n" + conv(64, 3, 2, "leaky", groups=4) + \
    conv(32, 3, 1, "leaky", groups=2) + conv(21, 1, 1, "linear", bn=0) + \
    "[yolo]\nmask = 0,1,2\nanchors = 6,8, 10,14, 20,18, 30,40, 50,44, 70,80\nclasses=2\nignore_thresh = .55\nscale_x_y = 1.05\n\n"


def test_grouped_dense_layers_in_an_executor(orc, tmp_path):
    """grouped 3x3 convolutions with 8+ channels per group inside a PLANNED net (the implicit-GEMM kernel with one plan-time
    weight image per group): found by the random nets -- the image buffer was sized for one group and the second group's
    image overwrote the next layer's.  Every layer, fused and unfused, three frames."""
    from ffcnn_amd import capi as F
    F.lib()
    cfg = str(tmp_path / "grouped.cfg")
    open(cfg, "w").write(GROUPED_CFG)
    o = orc.Oracle(cfg=cfg, weights=None)
    wpath = str(tmp_path / "grouped.weights")
    _write_random_weights(wpath, o, 11)
    o.close()
    o = orc.Oracle(cfg=cfg, weights=wpath)
    rng = np.random.default_rng(12)
    frames = rng.uniform(0, 1, (3, 3, 64, 64)).astype(np.float32)
    with F.Net(cfg, wpath) as n:
        for flags in (F.FFGPU.KEEP_ALL | F.FFGPU.NO_FUSE, F.FFGPU.KEEP_ALL):
            with n.executor(3, flags) as ex:
                ex.set_scale(1, 1)
                ex.forward_host(frames)
                for f in range(3):
                    o.input[...] = frames[f]
                    o.n.s1, o.n.s2 = 1, 1
                    o.forward(0)
                    for i in range(o.nlayers):
                        ref = o.layer_out(i)
                        if ref is None:
                            continue
                        try:
                            a = ex.read_layer(i, f)
                        except RuntimeError as e:
                            assert "not materialised" in str(e)
                            continue
                        close(a, ref, "grouped cfg flags %d frame %d layer %d" % (flags, f, i))
    o.close()

"""Pin the oracle to the reference ITSELF (oracle/_ref, compiled unmodified from
/root/reference by oracle/Makefile).  Skipped where the prebuilt _ref is absent."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def need_ref(orc):
    if not orc.have_ref("v0"):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    return orc


def test_every_layer_bit_exact(need_ref, test_image):
    orc = need_ref
    bgr, w, h = test_image
    o, r = orc.Oracle(), orc.Ref("v0")
    try:
        o.set_input_image(bgr, w, h)
        r.set_input_image(bgr, w, h)
        assert np.array_equal(o.input, r.input)
        assert np.array_equal(o.weights(), np.ctypeslib.as_array(r.n.weight_buf, (r.n.weight_size,)))
        o.forward(0)
        outs = r.forward(keep_activations=True)
        for i, a in outs.items():
            assert np.array_equal(o.layer_out(i), a), "layer %d" % i
        assert np.array_equal(o.boxes, r.boxes)
    finally:
        o.close()
        r.close()


def test_layer_table_fields(need_ref):
    orc = need_ref
    o, r = orc.Oracle(), orc.Ref("v0")
    try:
        assert o.nlayers == r.n.layer_num
        for i in range(o.nlayers):
            a, b, nxt = o.layer(i), r.layer(i), r.layer(i + 1)
            assert (a.kind, a.iw, a.ih, a.ic) == (b.type, b.w, b.h, b.c), i
            if a.kind != 7:
                assert (a.ow, a.oh, a.oc) == (nxt.w, nxt.h, nxt.c), i
            if a.kind == 0:
                assert (a.fs, a.fn, a.stride, a.pad, a.groups, a.batchnorm, a.act) == \
                       (b.fs, b.fn, b.stride, b.pad, b.groups, b.batchnorm, b.activation), i
            assert list(a.dep)[:a.ndep] == list(b.depend_list)[:b.depend_num]
    finally:
        o.close()
        r.close()


def test_random_groupconv_vs_v2_im2col(need_ref):
    """conv-v2.c (im2col+GEMM, the variant the north star names) is bit-identical too."""
    orc = need_ref
    if not orc.have_ref("v2"):
        pytest.skip("v2 not built")
    r = orc.Ref("v2")
    rng = np.random.default_rng(7)
    try:
        for (ic, ih, iw, g, fs, s, p, fn) in [(8, 9, 11, 1, 1, 1, 0, 12), (6, 8, 8, 6, 3, 1, 1, 6), (3, 10, 10, 1, 3, 2, 1, 8)]:
            K = fs * fs * ic // g
            k4 = (K + 3) & ~3
            x = rng.uniform(-1, 1, (ic, ih, iw)).astype(np.float32)
            f = np.zeros((fn, k4 + 4), np.float32)
            f[:, :K] = rng.uniform(-.5, .5, (fn, K))
            f[:, k4] = 1.25
            f[:, k4 + 1] = 0.03
            assert np.array_equal(orc.groupconv(x, f, g, p, s, fs, 2), r.groupconv(x, f, g, p, s, fs, 2))
    finally:
        r.close()


@pytest.mark.parametrize("bbox_max", [0, 700, 40])
def test_candidate_capacity_regimes_bit_exact(need_ref, test_image, bbox_max):
    """ignore_thresh = 0 (1 500 candidates per frame, hundreds of boxes after NMS) and net->bbox_max lowered until the
    reference's emission-order truncation bites (ffcnn.c:463): the oracle's candidate cap reproduces both bit for bit"""
    orc = need_ref
    bgr, w, h = test_image
    o, r = orc.Oracle(), orc.Ref("v0")
    try:
        for i in range(o.nlayers):
            if o.layer(i).kind == 7:
                o.layer(i).thresh = 0.0
                r.layer(i).ignore_thres = 0.0
        if bbox_max:
            o.n.cap = bbox_max
            r.n.bbox_max = bbox_max
        o.set_input_image(bgr, w, h)
        r.set_input_image(bgr, w, h)
        o.forward(0)
        r.forward()
        assert o.n.ncand == (bbox_max or 1500)
        assert o.boxes.tobytes() == r.boxes.tobytes() and len(o.boxes) > (100 if not bbox_max else 5)
    finally:
        o.close()
        r.close()

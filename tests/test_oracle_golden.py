"""The CPU oracle against the golden vectors produced by the unmodified reference
(oracle/gen_golden.py).  Runs everywhere (no GPU, no /root/reference)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLD


def _stats_match(a, g, key, exact, rtol=0.0, atol=0.0):
    assert tuple(g[key + "_shape"]) == a.shape
    flat = a.reshape(-1)
    samp = flat[g[key + "_idx"]]
    if exact:
        assert np.array_equal(samp, g[key + "_samples"]), key
        assert np.isclose(flat.astype(np.float64).sum(), g[key + "_sum"], rtol=1e-12, atol=1e-9), key
        assert np.isclose(np.abs(flat.astype(np.float64)).sum(), g[key + "_asum"], rtol=1e-12, atol=1e-9), key
    else:
        assert np.allclose(samp, g[key + "_samples"], rtol=rtol, atol=atol), key
        assert np.isclose(np.abs(flat.astype(np.float64)).sum(), g[key + "_asum"], rtol=1e-5), key


@pytest.fixture(scope="module")
def oracle_320(orc, test_image):
    bgr, w, h = test_image
    o = orc.Oracle()
    o.set_input_image(bgr, w, h)
    yield o
    o.close()


def test_input_matches_reference(oracle_320):
    g = np.load(os.path.join(GOLD, "input_320.npz"))
    inp = oracle_320.input
    assert np.array_equal(inp.reshape(-1)[g["idx"]], g["samples"])
    assert np.isclose(inp.astype(np.float64).sum(), g["sum"], rtol=1e-12)
    assert (oracle_320.n.s1, oracle_320.n.s2) == (int(g["s1"]), int(g["s2"])) == (640, 320)
    # top-left letterbox: rows >= nonzero_rows stay zero (ffcnn.c:242,278-288)
    assert int(g["nonzero_rows"]) == 212 and not inp[:, 212:, :].any()


def test_layers_bit_exact_vs_v0(oracle_320):
    g = np.load(os.path.join(GOLD, "layers_320.npz"))
    oracle_320.forward(0)
    layers = list(g["v0_layers"])
    assert len(layers) == 111        # 131 layers minus 18 dropout (pointer moves) minus 2 yolo
    for i in layers:
        _stats_match(oracle_320.layer_out(int(i)), g, "v0_L%d" % i, exact=True)


def test_layers_compat_v6(oracle_320):
    """compat mode reproduces conv-v6.c's 5x5 row omission; rounding order differs -> 1e-4."""
    g = np.load(os.path.join(GOLD, "layers_320.npz"))
    oracle_320.forward(1)
    for i in g["v6_layers"]:
        _stats_match(oracle_320.layer_out(int(i)), g, "v6_L%d" % i, exact=False, rtol=1e-4, atol=1e-4)
    # and the defect is real: v0 and v6 disagree downstream of the first 5x5 layer
    assert abs(float(g["v0_L129_asum"]) - float(g["v6_L129_asum"])) > 1.0


def test_heads_full_tensors(oracle_320):
    g = np.load(os.path.join(GOLD, "heads_320.npz"))
    oracle_320.forward(0)
    assert np.array_equal(oracle_320.layer_out(120), g["L120"])
    assert np.array_equal(oracle_320.layer_out(129), g["L129"])


@pytest.mark.parametrize("geom,wh", [("net_320x320", (0, 0)), ("cli_640x448", None)])
def test_boxes(orc, test_image, geom, wh):
    bgr, w, h = test_image
    gold = json.load(open(os.path.join(GOLD, "boxes.json")))
    o = orc.Oracle(w=0 if wh else w, h=0 if wh else h)
    try:
        o.set_input_image(bgr, w, h)
        for compat, v in ((0, "v0"), (1, "v6")):
            o.forward(compat)
            want = gold["%s_%s" % (geom, v)]
            got = o.boxes
            assert (o.n.in_w, o.n.in_h) == (want["w"], want["h"])
            assert len(got) == len(want["boxes"]) == 3
            for b, wb in zip(got, want["boxes"]):
                assert int(b["type"]) == wb["type"]
                tol = 0 if compat == 0 else 2e-4
                assert abs(float(b["score"]) - wb["score"]) <= tol
                for k in ("x1", "y1", "x2", "y2"):
                    assert abs(float(b[k]) - wb[k]) <= tol * 100
        if geom == "net_320x320":
            o.forward(0)
            assert len(o.candidates) == 20      # 8 + 12 candidates before NMS
    finally:
        o.close()


def test_cli_printed_boxes():
    """The reference CLI prints (int) coordinates: (188 96 273 365) (397 125 601 345) (68 264 201 350)."""
    gold = json.load(open(os.path.join(GOLD, "boxes.json")))["cli_640x448_v6"]["boxes"]
    got = [(b["type"], int(b["x1"]), int(b["y1"]), int(b["x2"]), int(b["y2"])) for b in gold]
    assert got == [(0, 188, 96, 273, 365), (18, 397, 125, 601, 345), (16, 68, 264, 201, 350)]


def test_groupconv_cases(orc):
    g = np.load(os.path.join(GOLD, "groupconv_cases.npz"))
    meta = json.loads(bytes(g["meta_json"]).decode())
    assert len(meta) >= 20
    for m in meta:
        x, f = g[m["name"] + "_x"], g[m["name"] + "_f"]
        out = orc.groupconv(x, f, m["groups"], m["pad"], m["stride"], m["fs"], m["act"], 0)
        assert np.array_equal(out, g[m["name"] + "_out_v0"]), m["name"]
        if "v6" in m["variants"]:
            out6 = orc.groupconv(x, f, m["groups"], m["pad"], m["stride"], m["fs"], m["act"], 1)
            assert np.allclose(out6, g[m["name"] + "_out_v6"], rtol=1e-5, atol=1e-5), m["name"]
    # the v6 5x5 defect shows up in exactly one output row
    d = np.abs(g["dw5s1_out_v0"] - g["dw5s1_out_v6"]).max(axis=(0, 2))
    assert d[8] > 1e-3 and d[[0, 1, 2, 3, 4, 5, 6, 7, 9]].max() < 1e-5


def test_net_dump_text(orc):
    o = orc.Oracle()
    try:
        assert o.dump() == open(os.path.join(GOLD, "net_dump.txt")).read()
        assert o.nlayers == 131 and o.n.nweights == 356576 and o.n.weights_consumed == 346062
    finally:
        o.close()


def test_missing_files(orc):
    assert not orc.lib().orc_load(b"/nonexistent.cfg", None, 0, 0)
    o = orc.Oracle(weights="/nonexistent.weights")      # tolerated: all-zero filters (ffcnn.c:213-220)
    try:
        assert not o.weights().any()
    finally:
        o.close()


def test_nms_semantics(orc):
    """class-aware, inter/min(area) > 0.5 suppresses, survivors rescaled by s1/s2 (ffcnn.c:298-335)."""
    c = np.zeros(5, orc.BOX_DTYPE)
    c[0] = (1, 0.9, 0, 0, 10, 10)
    c[1] = (1, 0.8, 1, 1, 5, 5)        # inside box 0 -> inter/min = 1 -> suppressed
    c[2] = (2, 0.7, 1, 1, 5, 5)        # other class -> kept
    c[3] = (1, 0.6, 8, 8, 20, 20)      # inter 4 / min(100,144) -> kept
    c[4] = (1, 0.95, 100, 100, 110, 110)
    out = orc.nms(c, 0.5, 1, 2, 1)
    assert [int(b["type"]) for b in out] == [1, 1, 2, 1]
    assert [round(float(b["score"]), 2) for b in out] == [0.95, 0.9, 0.7, 0.6]
    assert float(out[0]["x1"]) == 200.0 and float(out[3]["x2"]) == 40.0
    assert len(orc.nms(np.zeros(0, orc.BOX_DTYPE))) == 0

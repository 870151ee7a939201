#!/usr/bin/env python3
"""Generate tests/golden/* by RUNNING THE UNMODIFIED REFERENCE (oracle/_ref, built
by oracle/Makefile from /root/reference with -O2 -ffp-contract=off).

Run in the build container only (needs /root/reference for the _ref build):
    python oracle/gen_golden.py
Fixtures are data (inputs + the reference's outputs); no reference source text.

  boxes.json            final detections, CLI geometry (640x448) and 320x320,
                        v0 and v6
  layers_320.npz        per layer (v0 and v6): shape, float64 sum, abs-sum,
                        64 strided samples, at 320x320 on data/test.bmp
  heads_320.npz         full inputs of the two YOLO heads (outputs of L120, L129), v0
  groupconv_cases.npz   seeded single-layer groupconv vectors per (fs,stride,pad,
                        groups) class incl. odd sizes, outputs of v0 (and v6 where
                        its specialised path is defined)
  net_dump.txt          the reference's layer table
  input_320.npz         net_input result checksum + samples
  cli.json              the reference CLI (oracle/_ref/ffcnn_ref_cli_v6 = build.sh's default link) on
                        data/test.bmp: printed detection lines and the sha256 of the out.bmp it writes
"""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
NSAMP = 64


def stats(a):
    flat = a.reshape(-1)
    idx = np.linspace(0, flat.size - 1, NSAMP).astype(np.int64)
    return dict(shape=np.array(a.shape, np.int32), sum=np.float64(flat.astype(np.float64).sum()),
                asum=np.float64(np.abs(flat.astype(np.float64)).sum()), idx=idx, samples=flat[idx].copy())


def boxes_json(b):
    return [dict(type=int(x["type"]), score=float(x["score"]), x1=float(x["x1"]), y1=float(x["y1"]),
                 x2=float(x["x2"]), y2=float(x["y2"])) for x in b]


# (name, ic, ih, iw, groups, fs, stride, pad, fn, act, variants)
CASES = [
    ("dense3x3s2", 3, 16, 16, 1, 3, 2, 1, 8, 2, ("v0", "v6")),
    ("dense3x3s2_odd", 3, 7, 5, 1, 3, 2, 1, 5, 2, ("v0", "v6")),
    ("dense3x3s1_nopad", 2, 6, 7, 1, 3, 1, 0, 3, 1, ("v0",)),       # v6 im2row overruns its scratch when pad==0, fs>1
    ("pw_8_4", 8, 6, 6, 1, 1, 1, 0, 4, 0, ("v0", "v6")),
    ("pw_16_12", 16, 5, 7, 1, 1, 1, 0, 12, 2, ("v0", "v6")),
    ("pw_24_255", 24, 4, 4, 1, 1, 1, 0, 255, 0, ("v0", "v6")),
    ("pw_ic6_generic", 6, 5, 5, 1, 1, 1, 0, 7, 2, ("v0",)),     # v6's 1x1 path assumes ic%4==0
    ("pw_sigmoid", 4, 3, 3, 1, 1, 1, 0, 4, 3, ("v0", "v6")),
    ("dw3s1", 8, 12, 12, 8, 3, 1, 1, 8, 2, ("v0", "v6")),
    ("dw3s1_1x1", 3, 1, 1, 3, 3, 1, 1, 3, 2, ("v0",)),          # v6 reads OOB rows when oh==1
    ("dw3s1_2x2", 3, 2, 2, 3, 3, 1, 1, 3, 2, ("v0", "v6")),
    ("dw3s1_3x3", 3, 3, 3, 3, 3, 1, 1, 3, 0, ("v0", "v6")),
    ("dw3s1_7x9", 5, 7, 9, 5, 3, 1, 1, 5, 2, ("v0", "v6")),
    ("dw3s1_wide", 2, 5, 37, 2, 3, 1, 1, 2, 2, ("v0", "v6")),
    ("dw3s2", 4, 12, 12, 4, 3, 2, 1, 4, 2, ("v0", "v6")),
    ("dw3s2_odd", 4, 7, 7, 4, 3, 2, 1, 4, 2, ("v0", "v6")),
    ("dw5s1", 4, 10, 10, 4, 5, 1, 2, 4, 2, ("v0", "v6")),       # v6 differs on row oh-2 (conv-v6.c:422-441)
    ("dw5s1_6x8", 3, 6, 8, 3, 5, 1, 2, 3, 2, ("v0", "v6")),
    ("grouped2_3x3", 4, 6, 6, 2, 3, 1, 1, 6, 2, ("v0", "v6")),
    ("dense5x5s2", 2, 9, 9, 1, 5, 2, 2, 3, 2, ("v0", "v6")),
]


def make_case(rng, ic, ih, iw, groups, fs, fn):
    K = fs * fs * (ic // groups)
    k4 = (K + 3) & ~3
    x = rng.uniform(-1, 1, (ic, ih, iw)).astype(np.float32)
    f = np.zeros((fn, k4 + 4), np.float32)
    f[:, :K] = rng.uniform(-0.5, 0.5, (fn, K))
    f[:, k4] = rng.uniform(0.5, 1.5, fn)
    f[:, k4 + 1] = rng.uniform(-0.1, 0.1, fn)
    return x, f


def cli_golden():
    import hashlib
    import tempfile
    exe = os.path.join(HERE, "_ref", "ffcnn_ref_cli_v6")
    with tempfile.TemporaryDirectory() as d:
        out = subprocess.run([exe, "1", orc.BMP, orc.CFG, orc.WEIGHTS], capture_output=True, text=True, check=True, cwd=d).stdout
        sha = hashlib.sha256(open(os.path.join(d, "out.bmp"), "rb").read()).hexdigest()
    lines = out.splitlines()
    res = dict(detections=[l for l in lines if l.startswith("score:")], out_bmp_sha256=sha,
               head=[l.split(":")[0].strip() for l in lines[:3]],
               layer_table=[l for l in lines if l[:3].strip().isdigit() or l.startswith("layer")])
    json.dump(res, open(os.path.join(GOLD, "cli.json"), "w"), indent=1)
    return res


def main():
    orc.build()
    os.makedirs(GOLD, exist_ok=True)
    bgr, bw, bh = orc.load_bmp()

    # (i) detections ----------------------------------------------------------
    boxes = {}
    for geom, (gw, gh) in (("cli_640x448", (bw, bh)), ("net_320x320", (0, 0))):
        for v in ("v0", "v6"):
            r = orc.Ref(v, w=gw, h=gh)
            r.set_input_image(bgr, bw, bh)
            r.forward()
            boxes["%s_%s" % (geom, v)] = dict(s1=r.n.s1, s2=r.n.s2, w=r.layer(0).w, h=r.layer(0).h, boxes=boxes_json(r.boxes))
            r.close()
    json.dump(boxes, open(os.path.join(GOLD, "boxes.json"), "w"), indent=1)

    # (ii)+(iii) per-layer activations at 320x320 ------------------------------
    lay = {}
    heads = {}
    for v in ("v0", "v6"):
        r = orc.Ref(v)
        r.set_input_image(bgr, bw, bh)
        if v == "v0":
            inp = r.input.copy()
            np.savez_compressed(os.path.join(GOLD, "input_320.npz"), **{k: val for k, val in stats(inp).items()},
                                s1=r.n.s1, s2=r.n.s2, nonzero_rows=int(np.abs(inp).sum(axis=(0, 2)).nonzero()[0].max()) + 1)
        outs = r.forward(keep_activations=True)
        for i, a in outs.items():
            for k, val in stats(a).items():
                lay["%s_L%d_%s" % (v, i, k)] = val
        if v == "v0":
            heads["L120"] = outs[120]
            heads["L129"] = outs[129]
        lay["%s_layers" % v] = np.array(sorted(outs), np.int32)
        r.close()
    np.savez_compressed(os.path.join(GOLD, "layers_320.npz"), **lay)
    np.savez_compressed(os.path.join(GOLD, "heads_320.npz"), **heads)

    # (iv) single-layer groupconv vectors --------------------------------------
    rng = np.random.default_rng(20240612)
    gc = {}
    refs = {v: orc.Ref(v) for v in ("v0", "v6")}
    meta = []
    for (name, ic, ih, iw, groups, fs, stride, pad, fn, act, variants) in CASES:
        x, f = make_case(rng, ic, ih, iw, groups, fs, fn)
        gc[name + "_x"] = x
        gc[name + "_f"] = f
        for v in variants:
            gc["%s_out_%s" % (name, v)] = refs[v].groupconv(x, f, groups, pad, stride, fs, act)
        meta.append(dict(name=name, ic=ic, ih=ih, iw=iw, groups=groups, fs=fs, stride=stride, pad=pad, fn=fn, act=act,
                         variants=list(variants)))
    for r in refs.values():
        r.close()
    gc["meta_json"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    np.savez_compressed(os.path.join(GOLD, "groupconv_cases.npz"), **gc)

    # (v) net_dump text (the reference printf()s it: capture a child's stdout) ---
    code = ("import sys; sys.path.insert(0, %r); from oracle import orc; "
            "r = orc.Ref('v0'); r.L.net_dump(r.p); import ctypes; ctypes.CDLL(None).fflush(None)") % ROOT
    txt = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True).stdout
    open(os.path.join(GOLD, "net_dump.txt"), "w").write(txt)
    cli_golden()
    print("golden written to", GOLD)
    for fn_ in sorted(os.listdir(GOLD)):
        print("  %-24s %8d B" % (fn_, os.path.getsize(os.path.join(GOLD, fn_))))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""What would the net's activations / scores look like if the 1x1 layers' products ran on the bf16 matrix cores as SPLIT operands
(x = x1 + x2 [+ x3], each part a bf16; fp32 accumulation)?  A numpy emulation over the oracle's layer graph (CPU only):
every other layer is the oracle's own fp32 arithmetic.  Prints, per variant, the worst |d| / (1e-3 + 1e-3 |ref|) over all layers,
the head inputs' max |d| and the score / box deviations of the surviving boxes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import orc

def bf16_trunc(x):
    return (x.view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)

def bf16_rne(x):
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7fff + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)

def split(x, parts, first=bf16_rne):
    out, rem = [], x.astype(np.float32)
    for i in range(parts):
        p = (first if i == 0 else bf16_rne)(rem)
        out.append(p)
        rem = (rem - p).astype(np.float32)
    return out

def pw_emul(x, filt, act, wparts, xparts, maxord, scope):
    """x (ic, h, w), filt rows [w .. | scale', bias']: out = act(scale' * sum + bias'); the sum from split products, fp32-ish accumulate"""
    ic, h, w = x.shape
    k4 = (ic + 3) & ~3
    W = np.ascontiguousarray(filt[:, :ic], np.float32)
    X = x.reshape(ic, -1).astype(np.float32)
    if wparts == 0:                                  # exact reference: float64 sum rounded once
        acc = (W.astype(np.float64) @ X.astype(np.float64)).astype(np.float32)
    else:
        ws, xs = split(W, wparts), split(X, xparts)
        acc = np.zeros((W.shape[0], X.shape[1]), np.float32)
        # smallest terms first would be the most accurate order; the MFMA adds them inside one fp32 accumulation, emulate with fp32 adds
        for i in range(wparts):
            for j in range(xparts):
                if i + j <= maxord:
                    acc = (acc + (ws[i].astype(np.float64) @ xs[j].astype(np.float64)).astype(np.float32)).astype(np.float32)
    s, b = filt[:, k4].astype(np.float32)[:, None], filt[:, k4 + 1].astype(np.float32)[:, None]
    t = (acc * s + b).astype(np.float32)
    if act == 2: t = np.where(t > 0, t, np.float32(0.1) * t).astype(np.float32)
    elif act == 1: t = np.maximum(t, 0)
    elif act == 3: t = (1.0 / (1.0 + np.exp(-t.astype(np.float64)))).astype(np.float32)
    return t.reshape(-1, h, w)

def run(o, variant, frames):
    """variant: (wparts, xparts, maxord, which) -- which in {'expand', 'block', 'all'}"""
    wparts, xparts, maxord, which = variant
    n = o.nlayers
    res = {}
    worst = 0.0; worst_at = None
    outs = [None] * n
    lay = [o.layer(i) for i in range(n)]
    def is_pw(i): return lay[i].kind == 0 and lay[i].fs == 1 and lay[i].groups == 1
    def is_dw(i): return lay[i].kind == 0 and lay[i].groups == lay[i].ic and lay[i].groups > 1
    def emul(i):
        if not is_pw(i): return False
        if which == 'all': return True
        exp = i + 1 < n and is_dw(i + 1) and lay[i + 1].fs == 3           # expand of a fused block
        prj = i >= 1 and is_dw(i - 1) and lay[i - 1].fs == 3 and i >= 2 and is_pw(i - 2)
        return exp if which == 'expand' else (exp or prj)
    def out_of(i):
        if i < 0: return o.input.copy()
        if lay[i].kind == 4: return out_of(i - 1)
        return outs[i]
    cands = []
    for i in range(n):
        L = lay[i]
        x = out_of(i - 1)
        if L.kind == 0:
            filt = o.filter_rows(i)
            if emul(i): y = pw_emul(x, filt, L.act, wparts, xparts, maxord, which)
            else: y = orc.groupconv(x, filt, L.groups, L.pad, L.stride, L.fs, L.act)
        elif L.kind in (1, 2): y = orc.pool(x, L.fs, L.stride, 1 if L.kind == 2 else 0)
        elif L.kind == 3: y = orc.upsample(x, L.stride)
        elif L.kind == 4: y = None
        elif L.kind == 5: y = orc.shortcut(x, out_of(L.dep[0]), L.act)
        elif L.kind == 6: y = np.concatenate([out_of(L.dep[k]) for k in range(L.ndep)])
        else:
            anchors = [(L.anchors[a][0], L.anchors[a][1]) for a in range(3)]
            cands.append(orc.yolo(x, L.classes, anchors, L.thresh, L.scale_xy, o.n.in_w, o.n.in_h))
            y = None
        outs[i] = y
        if y is not None:
            ref = o.layer_out(i)
            r = float(np.max(np.abs(y - ref) / (1e-3 + 1e-3 * np.abs(ref))))
            if r > worst: worst, worst_at = r, i
    c = np.concatenate(cands) if cands else np.zeros(0, orc.BOX_DTYPE)
    boxes = orc.nms(c, 0.5, 1, o.n.s1, o.n.s2)
    return worst, worst_at, outs, c, boxes

def main():
    bgr, w, h = orc.load_bmp()
    rng = np.random.default_rng(7)
    frames = [("test.bmp", None), ("noise", rng.integers(0, 256, bgr.shape, dtype=np.uint8)), ("shifted", np.roll(bgr, 37, axis=1))]
    variants = [("exact float64 sum (what 'fp32 in another order' costs)", (0, 0, 0, 'all')),
                ("w 2 parts, x 2 parts, 3 terms, expand only", (2, 2, 1, 'expand')),
                ("w 2 parts, x 2 parts, 4 terms, expand only", (2, 2, 2, 'expand')),
                ("w 2 parts, x 2 parts, 4 terms, expand + project", (2, 2, 2, 'block')),
                ("w 2 parts, x 2 parts, 4 terms, every 1x1", (2, 2, 2, 'all')),
                ("w 3 parts, x 3 parts, 6 terms, expand only", (3, 3, 2, 'expand')),
                ("w 3 parts, x 3 parts, 6 terms, expand + project", (3, 3, 2, 'block')),
                ("w 3 parts, x 2 parts, 5 terms (i + j <= 2), expand + project", (3, 2, 2, 'block')),
                ("w 3 parts, x 3 parts, 6 terms, every 1x1", (3, 3, 2, 'all'))]
    for fname, img in frames:
        o = orc.Oracle()
        o.set_input_image(img if img is not None else bgr, w, h)
        o.forward(0)
        print("frame %s: reference %d candidates, %d boxes" % (fname, len(o.candidates), len(o.boxes)))
        for name, v in variants:
            worst, at, outs, c, boxes = run(o, v, None)
            ref_c, ref_b = o.candidates, o.boxes
            same = len(c) == len(ref_c) and len(boxes) == len(ref_b)
            ds = float(np.max(np.abs(c["score"] - ref_c["score"]))) if same and len(c) else float("nan")
            db = max([float(np.max(np.abs(boxes[k] - ref_b[k]))) for k in ("x1", "y1", "x2", "y2")]) if same and len(boxes) else float("nan")
            h1, h2 = [i for i in range(o.nlayers) if o.layer(i).kind == 7]
            d1 = float(np.max(np.abs(outs[h1 - 1] - o.layer_out(h1 - 1)))); d2 = float(np.max(np.abs(outs[h2 - 1] - o.layer_out(h2 - 1))))
            print("  %-62s worst tol ratio %.4f (layer %s)  head |d| %.2e %.2e  counts %s  score |d| %.2e  box |d| %.3e px" %
                  (name, worst, at, d1, d2, "same" if same else "DIFFER (%d/%d cand)" % (len(c), len(ref_c)), ds, db))
        o.close()

if __name__ == "__main__":
    main()

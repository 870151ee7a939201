"""lab: the failing case of test_dark3_cfg_against_the_oracle[2-True] (tests/data/dark3.cfg at 128x96, seed 23 / 24): every oracle box without a partner under
the test's tolerances, the nearest HIP box of the same class by score, and how far the yolo layers' inputs are from the oracle's"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
for k, v in (("FFGPU_IGX3_MIN_WGS", "1"), ("FFGPU_PWX3S_MIN_WGS", "1"), ("FFGPU_PWX3S_MIN_IC", "8"), ("FFGPU_PWX3S_MIN_OC", "8")):
    os.environ[k] = v
from ffcnn_amd import capi
from oracle import orc
from test_gpu_parity import _write_random_weights
tmp = tempfile.mkdtemp()
txt = open(os.path.join(ROOT, "tests", "data", "dark3.cfg")).read().replace("width=416", "width=128").replace("height=416", "height=96")
cfg = os.path.join(tmp, "d.cfg"); open(cfg, "w").write(txt)
wpath = os.path.join(tmp, "d.weights")
o = orc.Oracle(cfg=cfg, weights=None); _write_random_weights(wpath, o, 23); o.close()
o = orc.Oracle(cfg=cfg, weights=wpath)
batch = 2
frames = np.random.default_rng(24).uniform(0, 1, (batch, 3, 96, 128)).astype(np.float32)
C4 = ("x1", "y1", "x2", "y2")
with capi.Net(cfg, wpath) as n:
    yl = [i for i in range(n.layer_num) if n.layer(i).type == 7]
    for flags in (capi.FFGPU.KEEP_ALL | capi.FFGPU.NO_FUSE, 0):
        with n.executor(batch, flags) as ex:
            ex.set_scale(1, 1); ex.forward_host(frames)
            for f in range(batch):
                o.input[...] = frames[f]; o.n.s1, o.n.s2 = 1, 1; o.forward(0)
                want, got = o.boxes.copy(), ex.read_boxes(f)
                print("flags %d frame %d: %d oracle boxes, %d hip boxes; extent of oracle corners: max |c| = %.1f" % (flags, f, len(want), len(got), max(abs(float(w[c])) for w in want for c in C4)))
                if flags & capi.FFGPU.KEEP_ALL:
                    for i in yl:
                        a, r = ex.read_layer(i - 1, f), o.layer_out(i - 1)
                        print("   yolo input layer %d: max |d| %.3g, max |ref| %.3g" % (i - 1, np.abs(a - r).max(), np.abs(r).max()))
                used = np.zeros(len(got), bool)
                for k, w in enumerate(want):
                    hit = -1
                    for j, g in enumerate(got):
                        if not used[j] and int(g["type"]) == int(w["type"]) and abs(float(g["score"]) - float(w["score"])) <= 1e-4 and all(abs(float(g[c]) - float(w[c])) <= 0.05 + 1e-5 * abs(float(w[c])) for c in C4):
                            hit = j; break
                    if hit >= 0:
                        used[hit] = True; continue
                    cand = [j for j in range(len(got)) if not used[j] and int(got[j]["type"]) == int(w["type"])]
                    j = min(cand, key=lambda j: abs(float(got[j]["score"]) - float(w["score"])) + 1e-9 * sum(abs(float(got[j][c]) - float(w[c])) for c in C4)) if cand else -1
                    print("   oracle box %d: %r" % (k, w))
                    if j >= 0:
                        g = got[j]
                        rel = max(abs(float(g[c]) - float(w[c])) / max(1.0, abs(float(w[c]))) for c in C4)
                        print("      nearest hip box %d: %r   d(score) %.3g, max d(corner) %.4g, max relative %.3g" % (j, g, abs(float(g["score"]) - float(w["score"])), max(abs(float(g[c]) - float(w[c])) for c in C4), rel))
                    else:
                        print("      NO hip box of that class left")
                print("   hip boxes without partner: %d" % int((~used).sum()))
o.close()

#!/usr/bin/env python3
"""as pw_x3_check.py but the launches run BACK TO BACK (ffgpu_groupconv_time_dev, 30 launches, no sync between them):
explicit variant = the weight image is re-packed into scratch before every launch; auto = packed once"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
ACT = int(os.environ.get('ACT', '2'))
shapes = [(256, 512, 256, 20, 20), (120, 255, 64, 20, 20), (96, 255, 64, 10, 10)]
for ic, oc, N, H, W in shapes:
    g = torch.Generator(device="cuda").manual_seed(1235)
    x = torch.rand((ic * N, H, W), device="cuda", generator=g) * 2 - 1
    y = torch.empty((oc * N, H, W), device="cuda")
    k4 = (ic + 3) & ~3
    filt = torch.zeros((oc, k4 + 4), device="cuda")
    filt[:, :ic] = torch.rand((oc, ic), device="cuda", generator=g) - 0.5
    filt[:, k4] = torch.rand((oc,), device="cuda", generator=g) + 0.5 if ACT else 1.0
    if ACT:
        filt[:, k4 + 1] = torch.rand((oc,), device="cuda", generator=g) * 0.2 - 0.1
    os.environ["FFGPU_PW_X3"] = "0"
    capi.groupconv_dev(x.data_ptr(), filt.data_ptr(), y.data_ptr(), N, W, H, ic, 1, 0, 1, 1, oc, ACT, 0, capi.FFGPU.K_AUTO, None)
    torch.cuda.synchronize()
    y0 = y.clone()
    os.environ["FFGPU_PW_X3"] = "1"
    os.environ["FFGPU_PWX3_MIN_IC"] = "1"; os.environ["FFGPU_PWX3_MIN_OC"] = "1"; os.environ["FFGPU_PWX3_MIN_P"] = "1"
    for mt in (4, 2, 1):
        os.environ["FFGPU_PWX3_MT"] = str(mt)
        for mode, variant in (("explicit", capi.FFGPU.K_PW_X3), ("auto", capi.FFGPU.K_AUTO)):
            counts = []
            for rep in range(6):
                y.fill_(float("nan"))
                capi.groupconv_time_dev(x.data_ptr(), filt.data_ptr(), y.data_ptr(), N, W, H, ic, 1, 0, 1, 1, oc, act=ACT, variant=variant, warmup=0, iters=30)
                torch.cuda.synchronize()
                bad = ~((y - y0).abs().view(oc, -1) <= 1e-3)
                counts.append(int(bad.sum()))
                if bad.any() and not os.environ.get("QUIET"):
                    os.environ["QUIET"] = "1"
                    idx = bad.nonzero()[:48].tolist()
                    yv, y0v = y.view(oc, -1), y0.view(oc, -1)
                    print("   first wrong outputs (o, px, got, want):", [(o, px, round(float(yv[o, px]), 5), round(float(y0v[o, px]), 5)) for o, px in idx[:24]])
                    o, px = idx[0]
                    # pre-activation offset of that output and which row's bias would explain it
                    inv = lambda v: v if v > 0 or ACT != 2 else v * 10.0
                    off = inv(float(yv[o, px])) - inv(float(y0v[o, px]))
                    bi = filt[:, k4 + 1].cpu().numpy(); sc = filt[:, k4].cpu().numpy()
                    cand = np.argsort(np.abs((bi - bi[o]) - off))[:3]
                    print("   offset %.6f at row %d (sc %.5f bi %.5f); rows whose bias would explain it: %s; offset / sc = %.6f" %
                          (off, o, sc[o], bi[o], [(int(c), round(float(bi[c] - bi[o] - off), 6)) for c in cand], off / sc[o]))
                    cand2 = np.argsort(np.abs(bi * sc[o] - off))[:3]; cand3 = np.argsort(np.abs(bi - off / sc[o]))[:3]
                    print("   ... or a stale accumulator start c with sc * c = offset: c = %.6f; rows with bias == c: %s; rows with sc == c: %s" %
                          (off / sc[o], [(int(c), round(float(bi[c] - off / sc[o]), 6)) for c in cand3], [(int(c), round(float(sc[c] - off / sc[o]), 6)) for c in np.argsort(np.abs(sc - off / sc[o]))[:2]]))
                    print("   row %d around px %d: got %s want %s" % (o, px, [round(float(v), 4) for v in yv[o, (px // 64) * 64:(px // 64) * 64 + 64:4]], [round(float(v), 4) for v in y0v[o, (px // 64) * 64:(px // 64) * 64 + 64:4]]))
            print("%d -> %d x %d  MT=%d %-8s (%s): wrong outputs after 30 back-to-back launches %s" % (ic, oc, N, mt, mode, capi.kernel_name(N, W, H, ic, 1, 0, 1, 1, oc, variant), counts), flush=True)

#!/usr/bin/env python3
"""Sweep group split G and tile shape of the wave-autonomous fused block for every distinct block of yolo-fastest (batch 64)."""
import itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi

SHAPES = [(4, 24, 8, 2, 160, False), (8, 32, 8, 1, 80, True), (8, 32, 8, 2, 80, False),
          (8, 48, 8, 1, 40, True), (16, 96, 16, 1, 40, True), (16, 96, 24, 2, 40, False), (24, 136, 24, 1, 20, True),
          (48, 224, 48, 1, 10, True), (24, 136, 48, 2, 20, False)]
N = 64
only = [int(a) for a in sys.argv[1:]]
for si, (ic, ec, oc, s, HW, res) in enumerate(SHAPES):
    if only and si not in only:
        continue
    OH = (HW - 1) // s + 1
    x = torch.rand((ic * N, HW, HW), device="cuda") - 0.5
    f1 = torch.rand((ec, ((ic + 3) & ~3) + 4), device="cuda") - 0.5
    fd = torch.rand((ec, 16), device="cuda") - 0.5
    f2 = torch.rand((oc, ((ec + 3) & ~3) + 4), device="cuda") - 0.5
    r = torch.rand((oc * N, OH, OH), device="cuda")
    out = torch.empty((oc * N, OH, OH), device="cuda")

    def run():
        return capi.irb_dev(x.data_ptr(), f1.data_ptr(), fd.data_ptr(), f2.data_ptr(), r.data_ptr() if res else None, out.data_ptr(),
                            N, HW, HW, ic, ec, oc, s, warmup=2, iters=10)
    results = []
    tiles = [(0, 0)] + [(a, b) for a in (1, 2, 3, 4, 5) for b in (1, 2, 3, 4, 5, 8) if a * b <= 16 and a * b >= 6 and (a - 1) * 4 < OH]
    quick = os.environ.get("TUNE_QUICK")
    if quick:
        tiles = [(0, 0)]
    for (tq, th), G, big in itertools.product(tiles, (1, 2, 3, 4, 6, 7, 8), (0, 1)):
        if G > (ec + 15) // 16:
            continue
        os.environ.update(FFGPU_IRBW_G=str(G), FFGPU_IRBW_TWQ=str(tq), FFGPU_IRBW_TH=str(th), FFGPU_IRBW_BIG=str(big))
        try:
            results.append((run(), tq, th, G * 10 + big))
        except RuntimeError:
            continue
    for k in ("FFGPU_IRBW_G", "FFGPU_IRBW_TWQ", "FFGPU_IRBW_TH", "FFGPU_IRBW_BIG"):
        os.environ.pop(k, None)
    auto = run()
    os.environ["FFGPU_NO_IRBW"] = "1"
    old = run()
    os.environ.pop("FFGPU_NO_IRBW")
    results.sort()
    print("block %2d  %3dx%-3d %2d->%3d->%2d s%d: auto %.1f us (workgroup kernel %.1f); best: %s" %
          (si, HW, HW, ic, ec, oc, s, auto, old, "  ".join("%dx%d/G%d%s %.1f" % (4 * tq, th, G // 10, "b" if G % 10 else "", us) for us, tq, th, G in results[:10])))

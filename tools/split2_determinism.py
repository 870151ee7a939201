#!/usr/bin/env python3
"""Repeats one forward of a FFGPU_SPLIT2 executor (two half-batch chains as parallel graph branches) from the same u8 frames and counts the runs
whose detection records differ from the first one: a determinism watch for kernels that race only under concurrency
(python tools/split2_determinism.py [batch] [iters] [flags])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ffcnn_amd import capi as F
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
flags = int(sys.argv[3]) if len(sys.argv) > 3 else F.FFGPU.SPLIT2
rng = np.random.default_rng(5)
imgs = rng.integers(0, 256, (batch, 320, 960)).astype(np.uint8)
imgs[::3] = (np.arange(960)[None, :] * np.arange(320)[:, None] // 7) % 256
d = torch.from_numpy(imgs).cuda()
x = torch.rand((batch, 3, 320, 320), device="cuda")
with F.Net(F.CFG, F.WEIGHTS) as net, net.executor(batch, flags) as ex:
    for mode in ("u8", "f32"):
        first, bad, where = None, 0, []
        for it in range(iters):
            if mode == "u8":
                ex.forward_bgr_dev(d.data_ptr(), 320, 320)
            else:
                ex.forward_dev(x.data_ptr())
            torch.cuda.synchronize()
            b = ex.read_dets().tobytes()
            if first is None:
                first = b
            elif b != first:
                bad += 1
                where.append(it)
        print("%s batch %d flags %d: %d of %d runs differ from the first %s" % (mode, batch, flags, bad, iters, where[:10]), flush=True)

import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from ffcnn_amd import capi
from test_gpu_kernels import run_dev, make_filter
shape = (64, 256, 164, 20, 20, 0)
ic, oc, N, H, W, act = shape
rng = np.random.default_rng(1)
x = rng.uniform(-1, 1, (ic * N, H, W)).astype(np.float32)
f = make_filter(rng, oc, ic)
got = run_dev(capi, torch, x, f, N, W, H, ic, 1, 0, 1, 1, oc, act, capi.FFGPU.K_PW_X3T).reshape(oc, -1)
ref = run_dev(capi, torch, x, f, N, W, H, ic, 1, 0, 1, 1, oc, act, capi.FFGPU.K_GENERIC).reshape(oc, -1)
bad = np.abs(got - ref) > 1e-3
print("bad", bad.sum(), "of", bad.size)
rows = np.where(bad.any(1))[0]; cols = np.where(bad.any(0))[0]
print("rows", rows[:40], len(rows)); print("cols", cols[:40], cols[-10:], len(cols))

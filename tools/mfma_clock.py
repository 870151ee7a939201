#!/usr/bin/env python3
"""Sustained MFMA rate: 16 independent v_mfma_f32_16x16x4_f32 per trip, four waves per SIMD, for up to 0.9 s (no clock throttling: 32.2 cycles at 2.4 GHz throughout)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
L = capi.diag()
for it in (2000, 20000, 200000, 1000000):
    t = L.ffgpu_pipe_probe2(0, 0, 1024, it, None)
    print("iters %7d: %10.1f us -> %.2f ns per MFMA per SIMD = %.1f cycles at 2.4 GHz" % (it, t, t * 1e3 / (it * 16 * 4), t * 1e3 / (it * 16 * 4) * 2.4))

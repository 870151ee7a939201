#!/usr/bin/env python3
"""Does vector-ALU work hide in the shadow of an MFMA?  Hand-placed streams (ffgpu_pipe_probe2), 1 / 2 / 4 waves per SIMD."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
L = capi.diag()
it = 2000
for wps in (1, 2, 4):
    blocks = 256 * wps
    t = lambda m, n: L.ffgpu_pipe_probe2(m, n, blocks, it, None)
    print("waves/SIMD %d: 16 MFMA %.0f us | + ns plain v_fma each: ns=2 %.0f  4 %.0f  6 %.0f  8 %.0f | + ns/2 v_pk_fma each: 2 %.0f  4 %.0f  6 %.0f  8 %.0f | alone: 16x4 fma %.0f  16x8 fma %.0f  16x2 pk %.0f  16x4 pk %.0f" %
          (wps, t(0, 0), t(1, 2), t(1, 4), t(1, 6), t(1, 8), t(2, 2), t(2, 4), t(2, 6), t(2, 8), t(3, 4), t(3, 8), t(4, 4), t(4, 8)))

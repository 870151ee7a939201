#!/usr/bin/env python3
"""How MFMA and VALU work share a SIMD: one wave vs several waves per SIMD (ffgpu_pipe_probe)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
L = capi.diag()
it = 2000
print("waves/SIMD  mfma16  valu64  mfma16+valu64  valu128  mfma16+valu128   (us; 16 MFMA = 512 cycles, 64 FMA = 256 cycles of issue)")
for wps in (1, 2, 4):
    blocks = 256 * wps                     # 4 waves per block, one block per CU per wave-per-SIMD
    r = [L.ffgpu_pipe_probe(m, v, blocks, it, None) for (m, v) in ((16, 0), (0, 64), (16, 64), (0, 128), (16, 128))]
    print("%9d  %7.1f %7.1f %13.1f %8.1f %14.1f" % ((wps,) + tuple(r)))

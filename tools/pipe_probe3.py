#!/usr/bin/env python3
"""Do the BF16 matrix cores and the vector ALU overlap?  Hand-placed streams (ffgpu_pipe_probe3): inside a wave, and between waves of a SIMD."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffcnn_amd import capi
L = capi.diag()
it = 2000
def t(m, n, blocks, thr=256):
    return L.ffgpu_pipe_probe3(m, n, blocks, thr, it, None)
print("one kind of stream per wave (256 threads per workgroup); us per launch of %d trips x 16 slots" % it)
for wps in (1, 2, 4):
    b = 256 * wps
    print("waves/SIMD %d: 16 bf16 MFMA %.0f | + ns/2 v_pk_fma each: ns=2 %.0f  4 %.0f  8 %.0f  16 %.0f | pk alone: 2 %.0f  4 %.0f  8 %.0f  16 %.0f" %
          (wps, t(0, 0, b), t(2, 2, b), t(2, 4, b), t(2, 8, b), t(2, 16, b), t(4, 2, b), t(4, 4, b), t(4, 8, b), t(4, 16, b)))
print("two kinds of wave per SIMD (512 threads per workgroup: waves 0-3 16 MFMAs per trip, waves 4-7 16 x ns/2 v_pk_fma)")
for wps in (2, 4):
    b = 256 * wps // 2
    print("waves/SIMD %d: bf16 MFMA waves + pk waves: ns=4 %.0f  8 %.0f  16 %.0f | fp32 MFMA waves + pk waves: 4 %.0f  8 %.0f  16 %.0f   (same-kind references at this occupancy: bf16 MFMA only %.0f)" %
          (wps, t(5, 4, b, 512), t(5, 8, b, 512), t(5, 16, b, 512), t(6, 4, b, 512), t(6, 8, b, 512), t(6, 16, b, 512), t(0, 0, b, 512)))

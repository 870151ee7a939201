#!/usr/bin/env python3
"""The drop-in API on ONE frame at a time (what an unmodified ffcnn application does): net_input + net_forward in a loop,
milliseconds per call and frames/s; next to it net_forward alone (input already in layer_list[0].data)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ffcnn_amd import capi as F
F.lib()
bgr, w, h = F.load_bmp(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "test.bmp"))
with F.Net() as n:
    for _ in range(20):
        n.set_input_image(bgr, w, h); n.forward()
    N = 500
    t = time.perf_counter()
    for _ in range(N):
        F.net_input(n.p, bgr, w, h, (0.0, 0.0, 0.0), (1 / 255.0,) * 3); n.forward()
    a = (time.perf_counter() - t) / N
    t = time.perf_counter()
    for _ in range(N):
        n.forward()
    b = (time.perf_counter() - t) / N
    print("net_input + net_forward: %.3f ms per frame (%.0f frames/s); net_forward alone: %.3f ms (%.0f frames/s); %d boxes" % (a * 1e3, 1 / a, b * 1e3, 1 / b, len(n.boxes)))

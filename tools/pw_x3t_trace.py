#!/usr/bin/env python3
"""Per-tile timeline of k_pw_x3t on BASELINE config[2] (lab build: tools/build_lab_lib.sh xttrace -DXT_TRACE=1; FFCNN_HIP_LIB=tools/lab/lib/libffcnn_hip_xttrace.so):
wave 0 of every tile stamps start / first barrier / main loop done / stores issued (s_memrealtime, 10 ns) and its CU; prints the phase means and what
happens on a CU between one tile's last store and the next tile's start."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ffcnn_amd import capi
ic, oc, N, H, W = 256, 512, int(os.environ.get('XT_TRACE_N', '256')), 20, 20
s = torch.cuda.Stream()
x = torch.rand((ic * N, H, W), device="cuda") - 0.5
filt = torch.zeros((oc, ic + 4), device="cuda")
filt[:, :ic] = (torch.rand((oc, ic), device="cuda") - 0.5) / ic ** 0.5
filt[:, ic] = 1.0
y = torch.empty((oc * N, H, W), device="cuda")
run = lambda it: capi.groupconv_time_dev(x.data_ptr(), filt.data_ptr(), y.data_ptr(), N, W, H, ic, 1, 0, 1, 1, oc, act=2, variant=0, warmup=it, iters=it, stream=s.cuda_stream)
print("untraced: %.1f us" % run(40))
path = "/tmp/xt_trace.txt"
os.environ["FFGPU_PWXT_TRACE"] = path
run(1)
del os.environ["FFGPU_PWXT_TRACE"]
t = np.loadtxt(path, dtype=np.uint64).astype(np.int64)
t0 = t[:, 0].min()
T = (t[:, :4] - t0) * 0.01                       # us
hw, xcc = t[:, 4], t[:, 5] & 15
cu = ((hw >> 8) & 15) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)
print("%d tiles on %d CUs; launch span %.1f us" % (len(T), len(set(cu.tolist())), T[:, 3].max()))
d = np.stack([T[:, 1] - T[:, 0], T[:, 2] - T[:, 1], T[:, 3] - T[:, 2]], 1)
print("per tile (us): prologue %.2f  main loop %.2f  epilogue (to last store issued) %.2f   [p10 / p90 main loop %.2f / %.2f]" % (d[:, 0].mean(), d[:, 1].mean(), d[:, 2].mean(), np.percentile(d[:, 1], 10), np.percentile(d[:, 1], 90)))
# per CU: tiles in start order; how many are resident at once, and the gap from a tile's end (stores issued) to the next start on that CU beyond the one already resident
gaps, busy2, busy1, busy0 = [], 0.0, 0.0, 0.0
for c in set(cu.tolist()):
    m = T[cu == c]
    ev = sorted([(a, 1) for a in m[:, 0]] + [(b, -1) for b in m[:, 3]])
    lvl, last = 0, ev[0][0]
    for tt, dv in ev:
        if lvl >= 2: busy2 += tt - last
        elif lvl == 1: busy1 += tt - last
        else: busy0 += tt - last
        lvl += dv; last = tt
    ends = np.sort(m[:, 3]); starts = np.sort(m[:, 0])
    for e in ends:
        nxt = starts[starts > e]
        if len(nxt): gaps.append(nxt[0] - e)
ncu = len(set(cu.tolist()))
print("per CU (us, mean): two tiles resident %.1f, one %.1f, none (between first start and last end) %.1f" % (busy2 / ncu, busy1 / ncu, busy0 / ncu))
if gaps:
    print("end of a tile (last store issued) -> next tile start on that CU: mean %.2f us, median %.2f, p90 %.2f" % (np.mean(gaps), np.median(gaps), np.percentile(gaps, 90)))
starts = np.sort(T[:, 0])
print("tile starts: first round over after %.2f us; tiles started by 25 / 50 / 75 %% of the span: %d / %d / %d" % (starts[min(511, len(starts) - 1)], *(int((starts < T[:, 3].max() * f).sum()) for f in (0.25, 0.5, 0.75))))

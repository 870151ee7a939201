"""Does one of the four chains lag?  Per-chain completion of the last round of a 20-step run, several repetitions in one process; CHAINS_PRIO / NSTREAMS_BEFORE vary the set-up"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ffcnn_amd import capi
os.environ.setdefault("FFGPU_BRANCH", "0")
S, K = 4, 20
pre = [torch.cuda.Stream(priority=-1) for _ in range(int(os.environ.get("NSTREAMS_BEFORE", "0")))]
net = capi.Net()
exs = [net.executor(64, capi.FFGPU.HOST_DETS | capi.FFGPU.CONCURRENT) for _ in range(S)]
pr = os.environ.get("CHAINS_PRIO", "-1")
sts = [torch.cuda.Stream(priority=int(pr)) if pr != "none" else torch.cuda.Stream() for _ in range(S)]
xs = [torch.rand((64, 3, 320, 320), device="cuda") for _ in range(8)]
out = []
for rep in range(int(os.environ.get('REPS', '6'))):
    for i in range(8):
        exs[i % S].forward_dev(xs[i % 8].data_ptr(), sts[i % S].cuda_stream)
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t0.record(sts[0])
    evs, host = [], []
    al = float(os.environ.get("ALIGN_US", "0"))
    if al > 0:                                                   # chain j's first launch is enqueued ~j x 19 us after chain 0's: hold the earlier chains back so that all four start together
        for j in range(S - 1):
            with torch.cuda.stream(sts[j]):
                torch.cuda._sleep(int((S - 1 - j) * al * 1e-6 * 2.4e9))
    h0 = time.perf_counter()
    syn = int(os.environ.get("SYNC_ROUNDS", "0"))             # 1: a chain starts round r only when every chain has finished round r - 1 (events between the chains)
    for i in range(K):
        if syn and i >= S and (i // S) % syn == 0:
            for k in range(S):
                if k != i % S:
                    sts[i % S].wait_event(evs[(i // S - 1) * S + k])
        exs[i % S].forward_dev(xs[i % 8].data_ptr(), sts[i % S].cuda_stream)
        host.append((time.perf_counter() - h0) * 1e3)
        e = torch.cuda.Event(enable_timing=True); e.record(sts[i % S]); evs.append(e)
    torch.cuda.synchronize()
    t = [t0.elapsed_time(e) for e in evs]
    out.append("rep %d: first round %s | last round %s | %.4f ms/step | host enqueue done at (ms) %s" % (rep, " ".join("%.2f" % v for v in t[:4]), " ".join("%.2f" % v for v in t[-4:]), max(t) / K, " ".join("%.3f" % h for h in host[:8])))
import re
v = sorted(float(re.search(r"\| ([0-9.]+) ms/step", o).group(1)) for o in out[2:])
print("\n".join(sorted(out[2:], key=lambda o: float(re.search(r"\| ([0-9.]+) ms/step", o).group(1)))[-3:]))
print("reps %d: median %.4f  mean %.4f  worst %.4f ms/step; reps above 0.325: %d" % (len(v), v[len(v) // 2], sum(v) / len(v), v[-1], sum(1 for x in v if x > 0.325)))

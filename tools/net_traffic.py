#!/usr/bin/env python3
"""HBM traffic of ONE whole forward of yolo-fastest-1.1 at batch 64 (BASELINE config[3]) from PMC counters.
  run        (under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`, one counter per pass: tools/net_traffic.sh) -- a single chain with the bench's plan
             (FFGPU.CONCURRENT | HOST_DETS, FFGPU_BRANCH=0), launched kernel by kernel (NO_GRAPH: the counters are collected per dispatch either way), from u8 BGR
             frames, REPS forwards; beside it a bare 16-byte copy of a known byte count (k_membench<0>) that calibrates both counters as the guide prescribes;
             writes the plan's step model (ffgpu_exec_step_model) to gpurun_out/net_traffic_model.json
  summary    FETCH.db WRITE.db -> gpurun_out/net_traffic.json + a table: per launch of a forward (dispatch order = step order) measured read / written bytes
             against the model's, the forward's totals, the worst launches by bytes over model."""
import json, os, sqlite3, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REPS, B = 6, 64
CAL = 256 << 20

if sys.argv[1] == "run":
    import torch
    os.environ.setdefault("FFGPU_BRANCH", "0")
    from ffcnn_amd import capi
    F = capi.FFGPU
    net = capi.Net()
    ex = net.executor(B, F.NO_GRAPH | F.CONCURRENT | F.HOST_DETS)
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randint(0, 256, (B, 320, 960), dtype=torch.uint8, device="cuda", generator=g)
    a = torch.empty(CAL // 4, device="cuda"); b = torch.empty(CAL // 4, device="cuda")
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(3):
        capi.diag().ffgpu_membench(b.data_ptr(), a.data_ptr(), CAL, 0, 1024, 1, s.cuda_stream)
    torch.cuda.synchronize()
    for _ in range(REPS):
        ex.forward_bgr_dev(x.data_ptr(), 320, 320, stream=s.cuda_stream)
        torch.cuda.synchronize()
    model = [list(m) for m in ex.step_model()]
    by, fl = ex.work_model()
    u8_less = B * 3 * 320 * 320 * 3                                  # the first kernel reads 3 bytes per pixel, not 12 (bench.py prices its byte model the same way)
    by -= u8_less
    model[0][1] -= u8_less
    kc = ex.kernel_count() if callable(ex.kernel_count) else ex.kernel_count
    json.dump({"steps": model, "kernel_count": kc, "model_bytes": by, "model_flops": fl, "batch": B, "reps": REPS, "cal_bytes": CAL},
              open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "net_traffic_model.json"), "w"))
    print("forward: %d steps, %d kernel launches, model %.1f MB" % (len(model), kc, by / 1e6))
    sys.exit(0)


def dispatches(dbpath, counter):
    db = sqlite3.connect(dbpath)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    key = "dispatch_id" if "dispatch_id" in cols else "id"
    rows = db.execute("select %s, kernel_name, sum(value) from counters_collection where counter_name = ? group by %s, kernel_name order by %s" % (key, key, key), (counter,)).fetchall()
    return [(k.split("(")[0].replace("void ", ""), v * 1024.0) for _, k, v in rows]


fe, wr = dispatches(sys.argv[2], "FETCH_SIZE"), dispatches(sys.argv[3], "WRITE_SIZE")
M = json.load(open("gpurun_out/net_traffic_model.json"))
cal_r = [v for k, v in fe if "k_membench" in k][-1]
cal_w = [v for k, v in wr if "k_membench" in k][-1]
fr, fw = M["cal_bytes"] / cal_r, M["cal_bytes"] / cal_w
skip = ("k_membench", "pack", "k_params", "Cijk", "at::", "elementwise", "vectorized", "distribution", "fill")
nf = [(k, v) for k, v in fe if not any(t in k for t in skip)]
nw = [(k, v) for k, v in wr if not any(t in k for t in skip)]
K = M["kernel_count"]
assert len(nf) >= K * M["reps"] and len(nw) >= K * M["reps"], (len(nf), len(nw), K)
nf, nw = nf[-K * M["reps"]:], nw[-K * M["reps"]:]                 # the forwards are the last dispatches of the run (plan-time packing comes first)
rows = []
for i in range(K):
    names = {nf[r * K + i][0] for r in range(M["reps"])}
    assert len(names) == 1 and nw[i][0] in names, (i, names, nw[i][0])
    rd = sum(nf[r * K + i][1] for r in range(1, M["reps"])) / (M["reps"] - 1) * fr     # (the first forward warms the caches: left out)
    wt = sum(nw[r * K + i][1] for r in range(1, M["reps"])) / (M["reps"] - 1) * fw
    rows.append({"launch": i, "kernel": names.pop(), "read_bytes": round(rd), "written_bytes": round(wt)})
steps = M["steps"]
per_step = len(steps) == K
for i, r in enumerate(rows):
    if per_step:
        r["layer"], r["model_bytes"] = steps[i][0], round(steps[i][1])
        r["over_model"] = round((r["read_bytes"] + r["written_bytes"]) / steps[i][1], 3) if steps[i][1] > 0 else None
tot_r, tot_w = sum(r["read_bytes"] for r in rows), sum(r["written_bytes"] for r in rows)
res = {"workload": "yolo-fastest-1.1.cfg, 320x320, batch %d, u8 BGR frames, ONE chain (the bench's plan: FFGPU_CONCURRENT, launched kernel by kernel)" % M["batch"],
       "launches_per_forward": K, "calibration": {"copy_bytes_each_way": M["cal_bytes"], "raw_FETCH_copy": cal_r, "raw_WRITE_copy": cal_w, "read_factor": round(fr, 4), "write_factor": round(fw, 4)},
       "hbm_read_bytes_per_forward": round(tot_r), "hbm_write_bytes_per_forward": round(tot_w), "hbm_bytes_per_forward": round(tot_r + tot_w),
       "fused_model_bytes_per_forward": round(M["model_bytes"]), "traffic_over_fused_model": round((tot_r + tot_w) / M["model_bytes"], 4),
       "note": "a single chain's working set (75 MB arena) sits in the 256 MB Infinity Cache between producer and consumer: bytes the counters do not see never reached HBM; "
               "four chains in flight (the reported configuration) share that cache", "launches": rows}
json.dump(res, open("gpurun_out/net_traffic.json", "w"), indent=1)
print("per forward: read %.1f MB + written %.1f MB = %.1f MB against the fused model's %.1f MB: x %.3f   (calibration: read x %.3f, write x %.3f)" %
      (tot_r / 1e6, tot_w / 1e6, (tot_r + tot_w) / 1e6, M["model_bytes"] / 1e6, (tot_r + tot_w) / M["model_bytes"], fr, fw))
print("%3s %5s %-44s %10s %10s %10s %7s" % ("#", "layer", "kernel", "read MB", "write MB", "model MB", "x"))
for r in rows:
    print("%3d %5s %-44s %10.2f %10.2f %10s %7s" % (r["launch"], r.get("layer", ""), r["kernel"][:44], r["read_bytes"] / 1e6, r["written_bytes"] / 1e6,
                                                     "%.2f" % (r["model_bytes"] / 1e6) if "model_bytes" in r else "", r.get("over_model", "")))
if per_step:
    worst = sorted([r for r in rows if r.get("over_model")], key=lambda r: -(r["read_bytes"] + r["written_bytes"] - r["model_bytes"]))[:3]
    print("worst by bytes over model: " + "; ".join("launch %d (layer %d, %s): +%.1f MB, x %.2f" % (r["launch"], r["layer"], r["kernel"][:24], (r["read_bytes"] + r["written_bytes"] - r["model_bytes"]) / 1e6, r["over_model"]) for r in worst))

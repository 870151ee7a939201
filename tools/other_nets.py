#!/usr/bin/env python3
"""Throughput of the SAME executor on what SURVEY 8(f) rows 2 / 3 name beside yolo-fastest at 320x320: another darknet cfg
(tests/data/tiny3.cfg, yolov3-tiny-shaped, and tests/data/dark3.cfg, yolov3-shaped with 3x3 stride-2 layers and residual blocks: random weights, 416x416) and yolo-fastest at
the reference CLI's geometry (640x448, ffcnn.c:574).  Batch 16 per step, four chains; frames/s + the per-launch table of one chain.
Used by bench.py (config.other_nets) and stand-alone:  python tools/other_nets.py [--table]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_random_weights(capi, cfg, path, w, h, seed=7):
    """darknet .weights for an arbitrary cfg from the PRODUCT's own layer table (net_load without weights): 20-byte header, then per
    conv layer biases | [scales, means, variances] | taps"""
    rng = np.random.default_rng(seed)
    with capi.Net(cfg, None, w, h) as n, open(path, "wb") as fp:
        fp.write(np.array([0, 2, 5], "<i4").tobytes() + np.array([0], "<u8").tobytes())
        for i in range(n.layer_num):
            L = n.layer(i)
            if L.type != 0:                                     # LAYER_TYPE_CONV
                continue
            K = L.fs * L.fs * (L.c // L.groups)
            # (the linear layers feed the [yolo] heads: a bias of -6 keeps the objectness of random weights below the threshold -- with every
            #  anchor a candidate, 2 500 per frame, the greedy O(n^2) suppression of one workgroup per frame takes 8 ms and hides the conv stack)
            bias = rng.uniform(-0.2, 0.2, L.fn) if L.activation != 0 else np.full(L.fn, -6.0)
            fp.write(bias.astype("<f4").tobytes())
            if L.batchnorm:
                fp.write(rng.uniform(0.5, 1.5, L.fn).astype("<f4").tobytes())
                fp.write(rng.uniform(-0.3, 0.3, L.fn).astype("<f4").tobytes())
                fp.write(rng.uniform(0.2, 1.0, L.fn).astype("<f4").tobytes())
            # (... and small taps for them: with 1024 inputs of unit scale the pre-activations otherwise spread over +-5 and a bias of -6 no longer keeps them down)
            fp.write((rng.uniform(-1, 1, L.fn * K) * ((1.6 if L.activation != 0 else 0.1) / np.sqrt(K))).astype("<f4").tobytes())


def rate(torch, capi, cfg, weights, w, h, batch=16, chains=4, steps=80, table=False):
    with capi.Net(cfg, weights, w, h) as net:
        c, H, W = net.input_shape
        flags = capi.FFGPU.HOST_DETS | (capi.FFGPU.CONCURRENT if chains >= 3 else 0)
        exs = [net.executor(batch, flags) for _ in range(chains)]
        sts = [torch.cuda.Stream(priority=-1) for _ in range(chains)]
        xs = [torch.rand((batch, c, H, W), device="cuda") for _ in range(4)]
        out = {"net": os.path.basename(cfg), "input": "%dx%d" % (W, H), "batch": batch, "launches_per_step": exs[0].kernel_count}
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                exs[i % chains].forward_dev(xs[i % 4].data_ptr(), sts[i % chains].cuda_stream)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        out["value"] = round(batch * steps / dt, 1)
        out["unit"] = "frames/s"
        b, f = exs[0].work_model()
        out["TFLOPs"] = round(f * steps / dt / 1e12, 2)
        if table:
            rows = exs[0].profile_steps(xs[0].data_ptr())
            out["table"] = [(lay, round(us, 1)) for lay, us in rows]
        for e in exs:
            e.close()
        return out


def rows(torch, capi, table=False, tmpdir="/tmp"):
    cfg3 = os.path.join(ROOT, "tests", "data", "tiny3.cfg")
    w3 = os.path.join(tmpdir, "tiny3_bench_%d.weights" % os.getpid())
    write_random_weights(capi, cfg3, w3, 416, 416)
    try:
        r = [rate(torch, capi, cfg3, w3, 416, 416, table=table)]
    finally:
        os.unlink(w3)
    cfgd = os.path.join(ROOT, "tests", "data", "dark3.cfg")                  # yolov3-shaped (reduced depth): 3x3 stride-2 layers + residual blocks
    wd = os.path.join(tmpdir, "dark3_bench_%d.weights" % os.getpid())
    write_random_weights(capi, cfgd, wd, 416, 416)
    try:
        r.append(rate(torch, capi, cfgd, wd, 416, 416, steps=40, table=table))
    finally:
        os.unlink(wd)
    cfgg = os.path.join(ROOT, "tests", "data", "group3.cfg")                 # ResNeXt-shaped: grouped 3x3 layers (4 .. 64 channels per group), all groups of a layer in one grid
    wg = os.path.join(tmpdir, "group3_bench_%d.weights" % os.getpid())
    write_random_weights(capi, cfgg, wg, 416, 416)
    try:
        r.append(rate(torch, capi, cfgg, wg, 416, 416, steps=40, table=table))
    finally:
        os.unlink(wg)
    r.append(rate(torch, capi, capi.CFG, capi.WEIGHTS, 640, 448, table=table))
    return r


if __name__ == "__main__":
    import torch
    from ffcnn_amd import capi
    for r in rows(torch, capi, table="--table" in sys.argv):
        t = r.pop("table", None)
        print(r)
        if t:
            print("   per launch (layer, us): " + " ".join("%d:%.1f" % (a, b) for a, b in t))

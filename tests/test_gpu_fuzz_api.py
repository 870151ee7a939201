"""A seeded random WALK over the executor API: several executors of one net (different batch sizes and flags, each with its own
captured graph), and a random sequence of calls -- host and device-resident forwards on different torch streams and buffers,
source-scale changes, a record ring attached / restarted / detached, re-reads -- with the records checked against the oracle
after every forward.  What it hunts: state that one call leaves behind for the next (the device parameter block, the ring
counter, graph reuse across buffers, executors sharing a net's weights)."""
import os

import numpy as np
import pytest

from test_gpu_parity import boxes_match

pytestmark = pytest.mark.gpu

SCALES = [(1, 1), (2, 1), (3, 2)]


@pytest.fixture(scope="module")
def pool(orc, test_image):
    """8 frames x 3 source scales through the oracle: candidates (unscaled) and boxes (scaled by s1 / s2, ffcnn.c:519)"""
    bgr, w, h = test_image
    o = orc.Oracle()
    o.set_input_image(bgr, w, h)
    img = o.input.copy()
    rng = np.random.default_rng(31)
    fr = np.zeros((8, 3, 320, 320), np.float32)
    fr[0] = img
    fr[1] = np.roll(img, 53, axis=2)
    fr[2] = img[:, ::-1, :]
    fr[3] = np.clip(img + rng.normal(0, 0.1, img.shape), 0, 1)
    fr[4] = 0.0
    fr[5] = np.roll(img, -77, axis=1)
    fr[6] = img[:, :, ::-1]
    fr[7] = rng.uniform(0, 1, img.shape)
    want = {}
    for k in range(8):
        for (s1, s2) in SCALES:
            o.input[...] = fr[k]
            o.n.s1, o.n.s2 = s1, s2
            o.forward(0)
            want[(k, s1, s2)] = (len(o.candidates), o.boxes)
    o.close()
    return fr, want


@pytest.mark.parametrize("seed", range(int(os.environ.get("FFCNN_FUZZ_SEED0", "0")), int(os.environ.get("FFCNN_FUZZ_SEED0", "0")) + int(os.environ.get("FFCNN_FUZZ_WALKS", "4"))))
def test_random_api_walk(pool, seed):
    import torch
    from ffcnn_amd import capi as F
    F.lib()
    fr, want = pool
    rng = np.random.default_rng(600 + seed)
    streams = [None] + [torch.cuda.Stream() for _ in range(2)]
    with F.Net() as n:
        exs = []
        for _ in range(3):
            B = int(rng.choice([1, 2, 3, 5, 8]))
            flags = int(rng.choice([0, F.FFGPU.CONCURRENT, F.FFGPU.HOST_DETS, F.FFGPU.NO_GRAPH, F.FFGPU.NO_FUSE]))
            exs.append(dict(ex=n.executor(B, flags), B=B, scale=(1, 1), ring=None, k=0, flags=flags))
        try:
            for op in range(45):
                e = exs[int(rng.integers(0, len(exs)))]
                ex, B = e["ex"], e["B"]
                r = rng.random()
                if r < 0.15:
                    e["scale"] = SCALES[int(rng.integers(0, 3))]
                    ex.set_scale(*e["scale"])
                    continue
                if r < 0.25:                                    # attach (or restart) a ring of 2-4 slots, or detach it
                    if e["ring"] is not None and rng.random() < 0.4:
                        ex.set_ring(None, 0)
                        e["ring"] = None
                    else:
                        slots = int(rng.integers(2, 5))
                        nbytes = ex.dets_dev()[1]
                        e["ring"] = (torch.zeros((slots, nbytes), dtype=torch.uint8, device="cuda"), slots)
                        ex.set_ring(e["ring"][0].data_ptr(), slots)
                        e["k"] = 0
                    continue
                pick = [int(v) for v in rng.integers(0, 8, B)]
                frames = np.ascontiguousarray(fr[pick])
                st = streams[int(rng.integers(0, 3))]
                if rng.random() < 0.5:
                    ex.forward_host(frames)
                else:
                    buf = torch.from_numpy(frames).cuda()       # a fresh device buffer every time
                    if st is not None:
                        st.wait_stream(torch.cuda.current_stream())
                    ex.forward_dev(buf.data_ptr(), st.cuda_stream if st is not None else None)
                    (st or torch.cuda.current_stream()).synchronize()
                dets = ex.read_dets()
                s1, s2 = e["scale"]
                for f in range(B):
                    nc, bx = want[(pick[f], s1, s2)]
                    assert dets[f]["ncand"] == nc, "op %d frame %d (source %d): %d candidates, oracle %d" % (op, f, pick[f], dets[f]["ncand"], nc)
                    boxes_match(ex.boxes(f, dets), bx, "op %d executor batch %d flags %d frame %d scale %r" % (op, B, e["flags"], f, e["scale"]))
                if e["ring"] is not None:                       # the same records sit in slot k % slots of the ring
                    ring, slots = e["ring"]
                    torch.cuda.synchronize()
                    rec = np.frombuffer(ring[e["k"] % slots].cpu().numpy().tobytes(), F.DETS_DTYPE)
                    assert np.array_equal(rec["count"], dets["count"]) and np.array_equal(rec["ncand"], dets["ncand"]), "ring slot %d" % (e["k"] % slots)
                    for f in range(B):
                        assert np.array_equal(rec[f]["box"][:rec[f]["count"]], dets[f]["box"][:dets[f]["count"]])
                    e["k"] += 1
                assert ex.graph_captures <= 1
        finally:
            for e in exs:
                e["ex"].close()


@pytest.mark.isolated
@pytest.mark.parametrize("seed", range(int(os.environ.get("FFCNN_FUZZ_SEED0", "0")), int(os.environ.get("FFCNN_FUZZ_SEED0", "0")) + int(os.environ.get("FFCNN_FUZZ_WALKS", "4"))))
def test_random_node_walk(pool, seed):
    """the C node path (ffgpu_node_*: shards, per-rank executors, gather offsets, pipelined slots) on one device through the
    loopback transport: random rank counts, uneven totals, depths 1-4, scale changes, synchronous forwards mixed with
    submit / wait in any legal order -- every step's records in global frame order against the oracle"""
    from ffcnn_amd import capi as F
    F.lib()
    fr, want = pool
    rng = np.random.default_rng(900 + seed)
    ranks = int(rng.integers(1, 5))
    total = int(rng.integers(ranks, 11))
    depth = int(rng.integers(1, 5))
    flags = int(rng.choice([0, F.FFGPU.CONCURRENT]))
    with F.Net() as n, F.Node(n, ranks, total, exec_flags=flags, node_flags=F.Node.LOOPBACK | F.Node.DEPTH(depth)) as nd:
        scale = (1, 1)
        inflight = []                                           # (ticket, pick, scale at submit)

        def check(dets, pick, sc, what):
            for f in range(total):
                nc, bx = want[(pick[f], sc[0], sc[1])]
                assert dets[f]["ncand"] == nc, "%s frame %d: %d candidates, oracle %d" % (what, f, dets[f]["ncand"], nc)
                boxes_match(dets[f]["box"][:dets[f]["count"]], bx, "%s ranks %d total %d depth %d frame %d" % (what, ranks, total, depth, f))

        for op in range(14):
            r = rng.random()
            if r < 0.15 and not inflight:                       # (the scale belongs to the node: changed between steps only)
                scale = SCALES[int(rng.integers(0, 3))]
                nd.set_scale(*scale)
                continue
            pick = [int(v) for v in rng.integers(0, 8, total)]
            frames = np.ascontiguousarray(fr[pick])
            if r < 0.45 and not inflight:
                check(nd.forward_host(frames), pick, scale, "op %d forward_host" % op)
            elif len(inflight) < depth and (r < 0.8 or not inflight):
                inflight.append((nd.submit(frames), pick, scale))
            else:
                t, pk, sc = inflight.pop(0)
                check(nd.wait(t), pk, sc, "op %d wait(%d)" % (op, t))
        for t, pk, sc in inflight:
            check(nd.wait(t), pk, sc, "drain wait(%d)" % t)


@pytest.mark.isolated
@pytest.mark.parametrize("seed", range(int(os.environ.get("FFCNN_FUZZ_SEED0", "0")), int(os.environ.get("FFCNN_FUZZ_SEED0", "0")) + int(os.environ.get("FFCNN_FUZZ_WALKS", "4"))))
def test_random_lifecycle_walk(pool, test_image, seed):
    """object lifetimes: several NETs (native geometry and others) and their executors created, used through both API levels
    (net_input + net_forward of the drop-in API, batched executors) and destroyed in random order -- a net may be freed
    between two forwards of another one, and an executor whose net has been freed is an orphan: it refuses to run (loudly)
    and can still be destroyed, at any later point"""
    import ctypes as C
    from ffcnn_amd import capi as F
    F.lib()
    fr, want = pool
    bgr, w, h = test_image
    rng = np.random.default_rng(1300 + seed)
    nets, exs = [], []
    try:
        for op in range(30):
            r = rng.random()
            if r < 0.2 and len(nets) < 4:
                geo = [(0, 0), (0, 0), (640, 448), (96, 64)][int(rng.integers(0, 4))]
                nets.append(dict(n=F.Net(w=geo[0], h=geo[1]), geo=geo))
            elif r < 0.4 and nets:
                e = nets[int(rng.integers(0, len(nets)))]
                if e["geo"] == (0, 0):
                    B = int(rng.choice([1, 2, 4]))
                    exs.append(dict(ex=e["n"].executor(B, int(rng.choice([0, F.FFGPU.CONCURRENT]))), B=B, net=e, orphan=False))
            elif r < 0.5 and nets:
                e = nets.pop(int(rng.integers(0, len(nets))))
                e["n"].close()
                for x in exs:
                    if x["net"] is e:
                        x["orphan"] = True
            elif r < 0.6 and exs:
                exs.pop(int(rng.integers(0, len(exs))))["ex"].close()
            elif r < 0.8 and exs:
                e = exs[int(rng.integers(0, len(exs)))]
                pick = [int(v) for v in rng.integers(0, 8, e["B"])]
                if e["orphan"]:                                 # (straight through the C-ABI: the Python wrapper would look at the freed NET)
                    two = np.ascontiguousarray(fr[pick])
                    assert F.lib().ffgpu_exec_forward_host(e["ex"].h, two.ctypes.data_as(C.POINTER(C.c_float))) < 0 and "has been freed" in F.last_error()
                    continue
                e["ex"].set_scale(1, 1)
                e["ex"].forward_host(np.ascontiguousarray(fr[pick]))
                dets = e["ex"].read_dets()
                for f in range(e["B"]):
                    boxes_match(e["ex"].boxes(f, dets), want[(pick[f], 1, 1)][1], "op %d executor frame %d" % (op, f))
            elif nets:
                e = nets[int(rng.integers(0, len(nets)))]
                if e["geo"] == (0, 0):                          # the drop-in path on the native geometry: test.bmp -> the golden boxes
                    e["n"].set_input_image(bgr, w, h)
                    e["n"].forward()
                    assert len(e["n"].boxes) == 3, "op %d: net_forward found %d boxes" % (op, len(e["n"].boxes))
                else:
                    e["n"].input[...] = 0.25
                    e["n"].forward()
    finally:
        for e in exs:
            e["ex"].close()
        for e in nets:
            e["n"].close()


@pytest.mark.isolated
def test_no_device_or_host_memory_leak(pool):
    """200 x (net_load, two executors, forwards through both API levels, destroy everything): the device's free memory and the
    process's resident set come back to where they were (a production host reloads models and re-plans executors for years)"""
    import resource
    import torch
    from ffcnn_amd import capi as F
    F.lib()
    fr, want = pool

    def cycle(k):
        with F.Net() as n:
            with n.executor(1 + k % 4, F.FFGPU.CONCURRENT if k % 2 else 0) as ex, n.executor(2, F.FFGPU.HOST_DETS) as ex2:
                ex.forward_host(np.ascontiguousarray(fr[:1 + k % 4]))
                ex2.forward_host(np.ascontiguousarray(fr[:2]))
                ex.read_dets()
            n.input[...] = fr[0]
            n.forward()

    for k in range(10):                                         # allocator pools, code objects, RCCL-free paths settle first
        cycle(k)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    for k in range(200):
        cycle(k)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    assert free0 - free1 < 64 << 20, "device memory: %.1f MB gone after 200 load / plan / run / free cycles" % ((free0 - free1) / 2**20)
    assert rss1 - rss0 < 256 << 10, "host memory: peak RSS grew by %.1f MB over 200 cycles" % ((rss1 - rss0) / 1024)

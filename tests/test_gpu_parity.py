"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle and the
reference's golden vectors.  Tolerance for fp32 activations:
|d| <= 1e-3 + 1e-3*|ref|; boxes within 0.05 px, scores within 1e-4, same class/count."""
import json
import os

import numpy as np
import pytest

from conftest import GOLD

pytestmark = pytest.mark.gpu

ATOL, RTOL = 1e-3, 1e-3


def close(a, ref, what=""):
    err = np.abs(a - ref) - (ATOL + RTOL * np.abs(ref))
    assert err.max() <= 0, "%s: max excess %.3g (max |d| %.3g)" % (what, err.max(), np.abs(a - ref).max())


def boxes_match(got, want, what=""):
    assert len(got) == len(want), "%s: %d vs %d boxes" % (what, len(got), len(want))
    for g, w in zip(got, want):
        assert int(g["type"]) == int(w["type"]), what
        assert abs(float(g["score"]) - float(w["score"])) <= 1e-4, what
        for k in ("x1", "y1", "x2", "y2"):
            assert abs(float(g[k]) - float(w[k])) <= 0.05, what


@pytest.fixture(scope="module")
def F():
    import ffcnn_amd
    from ffcnn_amd import capi
    capi.lib()
    return capi


@pytest.fixture(scope="module")
def net(F):
    n = F.Net()
    yield n
    n.close()


@pytest.fixture(scope="module")
def frames(orc, test_image):
    """frame 0: letterboxed test.bmp (as net_input makes it); 1..3: seeded noise / edge content."""
    bgr, w, h = test_image
    o = orc.Oracle()
    o.set_input_image(bgr, w, h)
    rng = np.random.default_rng(1236)
    fr = np.zeros((4, 3, 320, 320), np.float32)
    fr[0] = o.input
    fr[1] = rng.uniform(0, 1, (3, 320, 320))
    fr[2] = np.roll(o.input, 37, axis=2)           # shifted image: different detections
    fr[3] = 0.0                                     # all-zero frame
    o.close()
    return fr


@pytest.fixture(scope="module")
def oracle_runs(orc, frames):
    """oracle activations, candidates and boxes per frame (s1 = s2 = 1 -> network pixels)."""
    res = []
    o = orc.Oracle()
    for f in frames:
        o.input[...] = f
        o.n.s1, o.n.s2 = 1, 1
        o.forward(0)
        acts = {i: o.layer_out(i).copy() for i in range(o.nlayers) if o.layer_out(i) is not None}
        res.append(dict(acts=acts, cand=o.candidates, boxes=o.boxes))
    o.close()
    return res


def test_groupconv_dropin_golden(F):
    """conv.h groupconv with host pointers on every golden single-layer case."""
    g = np.load(os.path.join(GOLD, "groupconv_cases.npz"))
    for m in json.loads(bytes(g["meta_json"]).decode()):
        out = F.groupconv(g[m["name"] + "_x"], g[m["name"] + "_f"], m["groups"], m["pad"], m["stride"], m["fs"], m["act"])
        close(out, g[m["name"] + "_out_v0"], m["name"])


def test_net_api_single_frame(F, net, test_image):
    """ffcnn.h path: net_load(cfg, weights, 0, 0) + net_input(test.bmp) + net_forward."""
    bgr, w, h = test_image
    gold = json.load(open(os.path.join(GOLD, "boxes.json")))["net_320x320_v0"]
    net.set_input_image(bgr, w, h)
    g = np.load(os.path.join(GOLD, "input_320.npz"))
    assert np.array_equal(net.input.reshape(-1)[g["idx"]], g["samples"])
    assert (net.n.s1, net.n.s2) == (640, 320)
    for _ in range(2):                              # second call replays the captured graph
        net.forward()
        boxes_match(net.boxes, gold["boxes"], "net_forward")
    assert net.layer_num == 131 and net.n.weight_size == 356576


def test_cli_geometry_640x448(F, test_image):
    """what the reference CLI does: net_load with the BMP size -> 640x448 (ffcnn.c:574,133-134)."""
    bgr, w, h = test_image
    gold = json.load(open(os.path.join(GOLD, "boxes.json")))["cli_640x448_v0"]
    with F.Net(w=w, h=h) as n:
        assert n.input_shape == (3, 448, 640)
        n.set_input_image(bgr, w, h)
        n.forward()
        boxes_match(n.boxes, gold["boxes"], "cli geometry")


def test_every_layer_keep_all(F, net, frames, oracle_runs):
    """unfused executor, batch 4: every layer of every frame against the oracle."""
    with net.executor(4, F.FFGPU.KEEP_ALL | F.FFGPU.NO_FUSE | F.FFGPU.NO_GRAPH) as ex:
        ex.forward_host(frames)
        for f in range(4):
            for i, ref in oracle_runs[f]["acts"].items():
                if net.layer(i).type == 4:          # dropout: alias
                    continue
                close(ex.read_layer(i, f), ref, "frame %d layer %d" % (f, i))


def test_fused_executor_materialised_layers(F, net, frames, oracle_runs):
    """fused executor (IRB blocks, shortcut-in-epilogue, concat in place), no arena reuse: every tensor
    that still exists must equal the oracle's activation of that layer; the fused-away ones must refuse."""
    with net.executor(4, F.FFGPU.KEEP_ALL) as ex:
        ex.forward_host(frames)
        seen, gone = 0, 0
        for i, ref in oracle_runs[0]["acts"].items():
            if net.layer(i).type == 4:
                continue
            try:
                a = ex.read_layer(i, 0)
            except RuntimeError as e:
                assert "not materialised" in str(e)
                gone += 1
                continue
            seen += 1
            for f in range(4):
                close(ex.read_layer(i, f), oracle_runs[f]["acts"][i], "fused: frame %d layer %d" % (f, i))
        assert seen >= 40 and gone >= 40, (seen, gone)      # 20 blocks x (2 expanded tensors + projection absorbed by the shortcut)
        assert ex.kernel_count <= 60


def test_golden_layer_samples(F, net, frames):
    """frame 0 against the reference's own per-layer samples (not via the oracle)."""
    g = np.load(os.path.join(GOLD, "layers_320.npz"))
    with net.executor(1, F.FFGPU.KEEP_ALL | F.FFGPU.NO_FUSE) as ex:
        ex.forward_host(frames[:1])
        for i in g["v0_layers"]:
            a = ex.read_layer(int(i), 0).reshape(-1)
            close(a[g["v0_L%d_idx" % i]], g["v0_L%d_samples" % i], "golden layer %d" % i)
        hd = np.load(os.path.join(GOLD, "heads_320.npz"))
        close(ex.read_layer(120, 0), hd["L120"], "head L120")
        close(ex.read_layer(129, 0), hd["L129"], "head L129")


@pytest.mark.parametrize("flags", [0, 4, 8, 12])
def test_fused_graph_executor_boxes(F, net, frames, oracle_runs, flags):
    """default (fused, graph) and the NO_GRAPH / NO_FUSE variants: candidates + boxes per frame."""
    with net.executor(4, flags) as ex:
        for rep in range(2):
            ex.forward_host(frames)
            dets = ex.read_dets()
            for f in range(4):
                want = oracle_runs[f]
                assert dets[f]["ncand"] == len(want["cand"]) and dets[f]["overflow"] == 0
                cand = ex.read_candidates(f)
                boxes_match(cand, want["cand"], "cand frame %d" % f)
                boxes_match(ex.boxes(f, dets), want["boxes"], "boxes frame %d" % f)
        assert ex.kernel_count <= 140


def test_host_mirror_of_records(F, net, frames, oracle_runs):
    """FFGPU_HOST_DETS: the NMS kernel's pinned host mirror holds the same records as the device buffer."""
    with net.executor(4, F.FFGPU.HOST_DETS) as ex:
        for rep in range(2):
            ex.forward_host(frames)
            dets = ex.read_dets()                       # synchronises, reads the mirror
            mirror = ex.dets_host()
            assert mirror.tobytes() == dets.tobytes()
            for f in range(4):
                boxes_match(ex.boxes(f, dets), oracle_runs[f]["boxes"], "boxes frame %d" % f)
    with net.executor(1, 0) as ex:
        with pytest.raises(RuntimeError, match="HOST_DETS"):
            ex.dets_host()


@pytest.mark.parametrize("flags", [32, 32 | 4, 32 | 16, 32 | 16 | 8])
def test_split_executor(F, net, frames, oracle_runs, flags):
    """FFGPU_SPLIT2: two half-batch chains as parallel graph branches give the same records (also mirrored / ring)"""
    import torch
    with net.executor(4, flags) as ex:
        rec_bytes = F.DETS_DTYPE.itemsize * 4
        ring = torch.zeros((2, rec_bytes), dtype=torch.uint8, device="cuda")
        ex.set_ring(ring.data_ptr(), 2)
        for rep, fr in enumerate((frames, frames[::-1].copy(), frames)):
            ex.forward_host(fr)
            dets = ex.read_dets()
            want = oracle_runs if rep != 1 else oracle_runs[::-1]
            for f in range(4):
                assert dets[f]["ncand"] == len(want[f]["cand"]) and dets[f]["overflow"] == 0
                boxes_match(ex.read_candidates(f), want[f]["cand"], "cand frame %d" % f)
                boxes_match(ex.boxes(f, dets), want[f]["boxes"], "boxes frame %d" % f)
            slot = np.frombuffer(ring[rep % 2].cpu().numpy().tobytes(), F.DETS_DTYPE, 4)
            assert slot.tobytes() == dets.tobytes()
            if flags & 16:
                assert ex.dets_host().tobytes() == dets.tobytes()


@pytest.mark.parametrize("flags", [0, 32, 32 | 16])
def test_repeated_forwards_are_bit_identical(F, net, frames, flags):
    """60 graph replays on the same frames: records identical every time (races between the parallel chains /
    branches, the counter hand-over between k_nms and the next forward's heads, ring slots)"""
    import torch
    with net.executor(4, flags) as ex:
        rec_bytes = F.DETS_DTYPE.itemsize * 4
        ring = torch.zeros((5, rec_bytes), dtype=torch.uint8, device="cuda")
        ex.set_ring(ring.data_ptr(), 5)
        x = torch.from_numpy(frames).cuda()
        first = None
        for k in range(60):
            ex.forward_dev(x.data_ptr())
            if k % 7 == 0 or k == 59:
                d = ex.read_dets().tobytes()
                first = first or d
                assert d == first, "replay %d differs" % k
                assert ring[k % 5].cpu().numpy().tobytes() == first, "ring slot of replay %d" % k
        assert sum(int(c) for c in np.frombuffer(first, F.DETS_DTYPE, 4)["count"]) > 0


def test_strided_ring_shared_by_executors(F, net, frames):
    """two executors taking turns share one ring: executor e writes slots e, e + 2, ... (ffgpu_exec_set_ring_strided)"""
    import torch
    rec = F.DETS_DTYPE.itemsize * 4
    ring = torch.zeros((4, rec), dtype=torch.uint8, device="cuda")
    rev = frames[::-1].copy()
    with net.executor(4, 0) as e0, net.executor(4, 0) as e1:
        e0.set_ring(ring.data_ptr(), 2, 2 * 4)
        e1.set_ring(ring.data_ptr() + rec, 2, 2 * 4)
        want = {}
        for step in range(6):
            ex, fr = (e0, frames) if step % 2 == 0 else (e1, rev)
            ex.forward_host(fr)
            want[step % 4] = ex.read_dets().tobytes()
            got = ring.cpu().numpy()
            for slot, b in want.items():
                assert got[slot].tobytes() == b, "step %d slot %d" % (step, slot)


def test_record_ring(F, net, frames, oracle_runs):
    """ffgpu_exec_set_ring: forward k also lands in slot k % slots of a caller-owned device ring"""
    import torch
    with net.executor(4, 0) as ex:
        rec_bytes = F.DETS_DTYPE.itemsize * 4
        ring = torch.zeros((3, rec_bytes), dtype=torch.uint8, device="cuda")
        ex.set_ring(ring.data_ptr(), 3)
        order = [frames, frames[::-1].copy(), frames, frames[::-1].copy(), frames]
        for k, fr in enumerate(order):
            ex.forward_host(fr)
            dets = ex.read_dets()
            slot = np.frombuffer(ring[k % 3].cpu().numpy().tobytes(), F.DETS_DTYPE, 4)
            assert slot.tobytes() == dets.tobytes(), "forward %d" % k
        ex.set_ring(None, 0)                          # detached: the ring stays as it is
        before = ring.cpu().numpy().copy()
        ex.forward_host(frames[::-1].copy())
        ex.read_dets()
        assert (ring.cpu().numpy() == before).all()


def test_branch_parallel_executor(F, net, frames, oracle_runs, monkeypatch):
    """the first detection head as a parallel graph branch (own stream, disjoint arena) and with FFGPU_BRANCH=0"""
    for flags, br in ((0, "1"), (4, "1"), (0, "0")):
        monkeypatch.setenv("FFGPU_BRANCH", br)
        with net.executor(4, flags) as ex:
            for rep in range(3):
                ex.forward_host(frames)
                dets = ex.read_dets()
                for f in range(4):
                    boxes_match(ex.read_candidates(f), oracle_runs[f]["cand"], "cand frame %d" % f)
                    boxes_match(ex.boxes(f, dets), oracle_runs[f]["boxes"], "boxes frame %d" % f)


def test_compat_v6_executor(F, net, frames, orc):
    """FFGPU_COMPAT_V6 reproduces conv-v6.c's 5x5 row omission (layers 116.. and 125..)."""
    o = orc.Oracle()
    o.input[...] = frames[0]
    o.forward(1)
    with net.executor(1, F.FFGPU.KEEP_ALL | F.FFGPU.COMPAT_V6) as ex:
        ex.forward_host(frames[:1])
        for i in (116, 118, 120, 125, 127, 129):
            close(ex.read_layer(i, 0), o.layer_out(i), "compat layer %d" % i)
    g = np.load(os.path.join(GOLD, "layers_320.npz"))
    with net.executor(1, F.FFGPU.KEEP_ALL | F.FFGPU.COMPAT_V6) as ex:
        ex.forward_host(frames[:1])
        a = ex.read_layer(129, 0).reshape(-1)
        close(a[g["v6_L129_idx"]], g["v6_L129_samples"], "golden v6 L129")
    o.close()


def test_scale_and_bgr_input(F, net, test_image, orc):
    """device-side batched net_input (u8 BGR -> letterbox) + box rescale == reference net_input path."""
    import torch
    bgr, w, h = test_image
    gold = json.load(open(os.path.join(GOLD, "boxes.json")))["net_320x320_v0"]
    d = torch.from_numpy(np.stack([bgr, bgr[::-1].copy()])).cuda()        # frame 1: upside-down image
    with net.executor(2) as ex:
        ex.forward_bgr_dev(d.data_ptr(), w, h)
        dets = ex.read_dets()
        boxes_match(ex.boxes(0, dets), gold["boxes"], "bgr frame 0")
        o = orc.Oracle()
        o.set_input_image(np.ascontiguousarray(bgr[::-1]), w, h)
        o.forward(0)
        boxes_match(ex.boxes(1, dets), o.boxes, "bgr frame 1")
        o.close()


def test_forward_dev_torch_stream(F, net, frames, oracle_runs):
    """device-resident input on a non-default torch stream (what bench.py does)."""
    import torch
    s = torch.cuda.Stream()
    x = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()
    with net.executor(4) as ex:
        with torch.cuda.stream(s):
            for _ in range(3):
                ex.forward_dev(x.data_ptr(), s.cuda_stream)
        s.synchronize()
        dets = ex.read_dets()
        for f in range(4):
            boxes_match(ex.boxes(f, dets), oracle_runs[f]["boxes"], "frame %d" % f)


def test_executor_life_cycle_frees_device_memory(F, net, frames):
    """create / forward / destroy: device memory in use must not grow (a graph with a forked branch kept its arena's worth
    of memory per cycle until ffgpu_exec_destroy synchronised the device first)"""
    import torch

    def used():
        torch.cuda.synchronize()
        free, total = torch.cuda.mem_get_info()
        return total - free

    marks = {}
    for it in range(12):
        for flags in (0, F.FFGPU.HOST_DETS, F.FFGPU.SPLIT2):
            with net.executor(4, flags) as ex:
                ex.forward_host(frames)
                ex.read_dets()
        if it in (3, 11):
            marks[it] = used()
    assert marks[11] - marks[3] < 4 << 20, "device memory grew by %.1f MB over 8 cycles" % ((marks[11] - marks[3]) / 1e6)


def test_packed_records(F, net, frames, oracle_runs):
    """ffgpu_pack_records (what the multi-GPU gather moves) == its numpy mirror, and unpacking gives the records back;
    a step with more boxes than the budget keeps the first ones and flags the rest"""
    import torch
    from ffcnn_amd import dist as ffdist
    from packref import pack_records
    with net.executor(4, 0) as ex:
        rec_bytes = F.DETS_DTYPE.itemsize * 4
        ring = torch.zeros((3, rec_bytes), dtype=torch.uint8, device="cuda")
        ex.set_ring(ring.data_ptr(), 3)
        for fr in (frames, frames[::-1].copy(), frames):
            ex.forward_host(fr)
        torch.cuda.synchronize()
        full = np.frombuffer(ring.cpu().numpy().tobytes(), F.DETS_DTYPE).reshape(3, 4)
        assert full["count"].sum() > 0
        for cap in (64, 5, 1):
            pb = F.packed_records_bytes(4, cap)
            assert pb == ffdist.packed_bytes(4, cap)
            out = torch.full((3, pb), 0xAB, dtype=torch.uint8, device="cuda")
            F.pack_records_dev(ring.data_ptr(), 3, 4, 4, cap, out.data_ptr())
            torch.cuda.synchronize()
            got = out.cpu().numpy()
            for s in range(3):
                want = pack_records(full[s], cap)
                hdr = got[s][:16].view(np.int32)
                nb = 16 + 16 * 4 + 24 * int(hdr[0])                 # header, frame table and the boxes that exist
                assert got[s][:nb].tobytes() == want[:nb].tobytes(), (cap, s)
                back = ffdist.unpack_records(got[s], F.DETS_DTYPE)
                total = int(full[s]["count"].sum())
                assert int(hdr[1]) == (total > cap)
                if total <= cap:
                    for n in range(4):
                        assert back[n]["count"] == full[s][n]["count"] and back[n]["ncand"] == full[s][n]["ncand"]
                        assert back[n]["box"][: back[n]["count"]].tobytes() == full[s][n]["box"][: full[s][n]["count"]].tobytes()
                else:
                    assert int(back["count"].sum()) == cap and (back["overflow"] & 2).any()


def test_front_kernel_layers(F, net, frames, oracle_runs, monkeypatch):
    """first layer + first thin block as one kernel (k_front; forced on a small batch): its output (layer 3) and everything
    behind it against the oracle, frame by frame; the tensor it no longer writes (layer 0) must refuse to be read"""
    if any(os.environ.get(k, "0") not in ("", "0") for k in ("FFGPU_NO_FRONT", "FFGPU_NO_THIN", "FFGPU_NO_FUSE")):
        pytest.skip("front kernel switched off by the environment")
    monkeypatch.setenv("FFGPU_FRONT_MIN_PX", "1")
    for band in ("", "1", "3", "16"):
        if band:
            monkeypatch.setenv("FFGPU_FRONT_BAND", band)
        with net.executor(4, F.FFGPU.KEEP_ALL) as ex:
            ex.forward_host(frames)
            with pytest.raises(RuntimeError, match="not materialised"):
                ex.read_layer(0, 0)
            for i in (3, 8, 11, 21, 37, 57, 80, 108, 120, 129):
                for f in range(4):
                    close(ex.read_layer(i, f), oracle_runs[f]["acts"][i], "front band %s: frame %d layer %d" % (band, f, i))
            assert ex.kernel_count <= 59


@pytest.mark.parametrize("flags", [0, 64, 64 | 16])
def test_batch64_plans(F, net, frames, oracle_runs, flags):
    """batch 64 (the bench's batch: tile splits, band lengths and kernel choices that only big batches take), planned for one
    chain in flight (0) and for several (FFGPU_CONCURRENT): every frame's records against the oracle's"""
    big = np.ascontiguousarray(np.tile(frames, (16, 1, 1, 1)))
    with net.executor(64, flags) as ex:
        for rep in range(2):
            ex.forward_host(big)
            dets = ex.read_dets()
            for f in range(64):
                want = oracle_runs[f % 4]
                assert dets[f]["ncand"] == len(want["cand"]) and dets[f]["overflow"] == 0
                boxes_match(ex.read_candidates(f), want["cand"], "cand frame %d" % f)
                boxes_match(ex.boxes(f, dets), want["boxes"], "boxes frame %d" % f)
            if flags & 16:
                assert ex.dets_host().tobytes() == dets.tobytes()


def test_batch_sizes_and_arena(F, net, frames, oracle_runs):
    sizes = {}
    for b in (1, 2, 3):
        with net.executor(b) as ex:
            ex.forward_host(frames[:b])
            dets = ex.read_dets()
            for f in range(b):
                boxes_match(ex.boxes(f, dets), oracle_runs[f]["boxes"], "batch %d frame %d" % (b, f))
            sizes[b] = ex.arena_bytes
    with net.executor(1, F.FFGPU.KEEP_ALL | F.FFGPU.NO_FUSE) as ex:
        keep = ex.arena_bytes
    assert sizes[1] < keep / 3          # liveness reuse: far smaller than one-buffer-per-layer
    assert sizes[2] <= 2 * sizes[1] + 4096 * 131


def test_missing_weights_and_bad_cfg(F):
    assert F.net_load("/nonexistent.cfg", None) is None
    with F.Net(weights="/nonexistent.weights") as n:      # tolerated (ffcnn.c:213-220): zero filters
        assert not n.weights_host().any()
        n.forward()
        assert n.n.bbox_num == 0


def test_reference_main_on_hip_groupconv():
    """Level-1 drop-in: the reference's unmodified main/ffcnn.c (oracle/_ref/ffcnn_ref_dropin, built from
    /root/reference by oracle/Makefile) with ONLY groupconv taken from libffcnn_hip.so prints the CLI's boxes."""
    import re
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "oracle", "_ref", "ffcnn_ref_dropin")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ffcnn_ref_dropin not built (needs /root/reference at build time)")
    data = os.path.join(ROOT, "data")
    out = subprocess.run([exe, "2", os.path.join(data, "test.bmp"), os.path.join(data, "yolo-fastest-1.1.cfg"),
                          os.path.join(data, "yolo-fastest-1.1.weights")], capture_output=True, text=True, timeout=300, cwd="/tmp")
    assert out.returncode == 0, out.stderr[-500:]
    boxes = re.findall(r"score: ([0-9.]+), category:\s*(\d+), rect: \(\s*(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\)", out.stdout)
    got = [(int(c), int(a), int(b), int(cc), int(d)) for _, c, a, b, cc, d in boxes]
    assert got == [(0, 188, 96, 273, 365), (18, 397, 125, 601, 345), (16, 68, 264, 201, 350)], out.stdout[-800:]


def _write_random_weights(path, orc_net, seed):
    """darknet .weights for an arbitrary cfg: 20-byte header, then per conv layer biases | [scales, means, vars] | taps"""
    rng = np.random.default_rng(seed)
    with open(path, "wb") as fp:
        fp.write(np.array([0, 2, 5], "<i4").tobytes() + np.array([0], "<u8").tobytes())
        for i in range(orc_net.nlayers):
            L = orc_net.layer(i)
            if L.kind != 0:
                continue
            K = L.fs * L.fs * (L.ic // L.groups)
            fp.write(rng.uniform(-0.2, 0.2, L.fn).astype("<f4").tobytes())
            if L.batchnorm:
                fp.write(rng.uniform(0.5, 1.5, L.fn).astype("<f4").tobytes())
                fp.write(rng.uniform(-0.3, 0.3, L.fn).astype("<f4").tobytes())
                fp.write(rng.uniform(0.2, 1.0, L.fn).astype("<f4").tobytes())
            fp.write((rng.uniform(-1, 1, L.fn * K) * (1.6 / np.sqrt(K))).astype("<f4").tobytes())


def test_dwpw_plan_opt_in(F, net, frames, oracle_runs, monkeypatch):
    """FFGPU_DWPW=1: the heads' depthwise 5x5 + pointwise pairs as one launch each (k_dwpw; off by default: measured slower):
    4 launches fewer, same activations behind the pairs, same records"""
    with net.executor(4, F.FFGPU.KEEP_ALL) as ex:
        base = ex.kernel_count
    monkeypatch.setenv("FFGPU_DWPW", "1")
    with net.executor(4, F.FFGPU.KEEP_ALL) as ex:
        assert ex.kernel_count == base - 4
        ex.forward_host(frames)
        with pytest.raises(RuntimeError, match="not materialised"):
            ex.read_layer(125, 0)
        dets = ex.read_dets()
        for f in range(4):
            for i in (117, 119, 120, 126, 128, 129):
                close(ex.read_layer(i, f), oracle_runs[f]["acts"][i], "dwpw plan: frame %d layer %d" % (f, i))
            boxes_match(ex.boxes(f, dets), oracle_runs[f]["boxes"], "dwpw plan boxes frame %d" % f)


@pytest.mark.parametrize("batch", [1, 5])
@pytest.mark.parametrize("flags", [1, 0])
def test_tiny3_cfg_implicit_gemm(F, orc, tmp_path, flags, batch):
    """tests/data/tiny3.cfg: a yolov3-tiny-shaped net (dense 3x3 stack, 1x1 bottlenecks, a dense 5x5, two heads through route +
    upsample): every conv behind the first runs on the implicit-GEMM MFMA kernel -- every layer and the boxes against the oracle"""
    from conftest import ROOT
    cfg = os.path.join(ROOT, "tests", "data", "tiny3.cfg")
    o = orc.Oracle(cfg=cfg, weights=None)
    wpath = str(tmp_path / "tiny3.weights")
    _write_random_weights(wpath, o, 7)
    o.close()
    o = orc.Oracle(cfg=cfg, weights=wpath)
    rng = np.random.default_rng(11)
    frames = rng.uniform(0, 1, (batch, 3, 64, 96)).astype(np.float32)
    keep = F.FFGPU.KEEP_ALL | F.FFGPU.NO_FUSE if flags else 0
    with F.Net(cfg, wpath) as n:
        assert n.layer_num == o.nlayers == 22
        for i in (2, 4, 6, 8, 10, 12, 19, 20):
            L = n.layer(i)
            assert F.kernel_name(batch, L.w, L.h, L.c, 1, L.pad, L.stride, L.fs, L.fn) == "conv_igemm", i
        with n.executor(batch, keep) as ex:
            ex.forward_host(frames)
            dets = ex.read_dets()
            for f in range(batch):
                o.input[...] = frames[f]
                o.n.s1, o.n.s2 = 1, 1
                o.forward(0)
                if keep:
                    for i in range(o.nlayers):
                        ref = o.layer_out(i)
                        if ref is None:
                            continue
                        close(ex.read_layer(i, f), ref, "tiny3 frame %d layer %d" % (f, i))
                assert dets[f]["ncand"] == len(o.candidates)
                boxes_match(ex.boxes(f, dets), o.boxes, "tiny3 boxes frame %d" % f)
    o.close()


@pytest.mark.parametrize("flags", [1, 0])      # 1: KEEP_ALL | NO_FUSE with per-layer checks, 0: default fused graph executor
def test_other_cfg_generic_path(F, orc, tmp_path, flags):
    """a cfg that is NOT yolo-fastest (tests/data/mini.cfg: grouped 3x3, dense 5x5 s2, unpadded 3x3, avgpool, relu,
    shortcut with leaky, multi-source route, upsample, two heads with 2 classes): every layer and the boxes against
    the oracle -- the generic kernels and the planner on a graph they were not tuned for."""
    from conftest import ROOT
    cfg = os.path.join(ROOT, "tests", "data", "mini.cfg")
    o = orc.Oracle(cfg=cfg, weights=None)
    wpath = str(tmp_path / "mini.weights")
    _write_random_weights(wpath, o, 42)
    o.close()
    o = orc.Oracle(cfg=cfg, weights=wpath)
    rng = np.random.default_rng(5)
    frames = rng.uniform(0, 1, (3, 3, 48, 64)).astype(np.float32)
    keep = F.FFGPU.KEEP_ALL | F.FFGPU.NO_FUSE if flags else 0
    with F.Net(cfg, wpath) as n:
        assert n.layer_num == o.nlayers == 18
        assert np.array_equal(n.weights_host(), o.weights())
        with n.executor(3, keep) as ex:
            ex.set_scale(3, 2)
            ex.forward_host(frames)
            dets = ex.read_dets()
            for f in range(3):
                o.input[...] = frames[f]
                o.n.s1, o.n.s2 = 3, 2
                o.forward(0)
                if keep:
                    for i in range(o.nlayers):
                        ref = o.layer_out(i)
                        if ref is None or n.layer(i).type == 4:
                            continue
                        close(ex.read_layer(i, f), ref, "mini frame %d layer %d" % (f, i))
                assert dets[f]["ncand"] == len(o.candidates)
                boxes_match(ex.boxes(f, dets), o.boxes, "mini boxes frame %d" % f)
    o.close()


def test_demo_cli_matches_reference_cli(tmp_path):
    """ffcnn_hip_demo (the C harness on libffcnn_hip.so) against the reference CLI's own output on test.bmp:
    same layer table, same printed detections, byte-identical out.bmp (green outlines included)."""
    import hashlib
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "ffcnn_amd", "bin", "ffcnn_hip_demo")
    gold = json.load(open(os.path.join(GOLD, "cli.json")))
    data = os.path.join(ROOT, "data")
    out = subprocess.run([exe, "3", os.path.join(data, "test.bmp"), os.path.join(data, "yolo-fastest-1.1.cfg"),
                          os.path.join(data, "yolo-fastest-1.1.weights")], capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-400:]
    lines = out.stdout.splitlines()
    assert [l for l in lines if l.startswith("score:")] == gold["detections"]
    assert [l for l in lines if l[:3].strip().isdigit() or l.startswith("layer")] == gold["layer_table"]
    assert hashlib.sha256(open(tmp_path / "out.bmp", "rb").read()).hexdigest() == gold["out_bmp_sha256"]

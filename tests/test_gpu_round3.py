"""Round 3: u8 BGR frames straight into the first kernel (ffcnn.c:259-289 fused into k_front), NET.bbox_max
re-read on every forward (ffcnn.c:461-463) by a captured graph, transfers staged through library-owned pinned memory."""
import os

import numpy as np
import pytest

from test_gpu_parity import boxes_match
from test_gpu_round2 import F, close, net  # noqa: F401  (fixtures / helpers)

pytestmark = pytest.mark.gpu
LAYERS = (3, 11, 57, 120, 129)


@pytest.fixture(scope="module")
def u8set(orc, test_image):
    """8 u8 BGR images of the net's own geometry (320 x 320: net_input copies pixel for pixel) + the oracle's run of each, for two
    (mean, norm) settings"""
    bgr, w, h = test_image                                          # 640 x 448, row stride ALIGN(3 w, 4)
    src = np.frombuffer(bgr, np.uint8).reshape(h, (3 * w + 3) & ~3)[:, :3 * w].reshape(h, w, 3)
    rng = np.random.default_rng(31)
    imgs = np.zeros((8, 320, 960), np.uint8)
    imgs[0] = src[64:384, 150:470].reshape(320, 960)               # a crop with the dog and the bicycle
    imgs[1] = src[::1, ::2][:320, :320].reshape(320, 960)          # squeezed horizontally
    imgs[2] = rng.integers(0, 256, (320, 960))
    imgs[3] = 0
    imgs[4] = 255
    imgs[5] = imgs[0][::-1]
    imgs[6] = np.clip(imgs[0].astype(np.int32) + rng.integers(-40, 40, (320, 960)), 0, 255)
    imgs[7] = (np.arange(960)[None, :] * np.arange(320)[:, None] // 7) % 256
    settings = [((0.0, 0.0, 0.0), (1 / 255.0,) * 3), ((104.0, 117.0, 123.0), (0.017, 0.0175, 0.0171))]
    runs = {}
    o = orc.Oracle()
    for si, (mean, norm) in enumerate(settings):
        for k in range(8):
            o.set_input_image(np.ascontiguousarray(imgs[k]), 320, 320, mean, norm)
            o.forward(0)
            runs[(si, k)] = dict(acts={i: o.layer_out(i).copy() for i in LAYERS}, cand=o.candidates, boxes=o.boxes, s=(o.n.s1, o.n.s2))
    o.close()
    return imgs, settings, runs


@pytest.mark.parametrize("batch,flags", [(16, 0), (64, 64), (11, 64), (32, 32), (37, 0)])
def test_u8_frames_into_the_first_kernel(F, net, u8set, batch, flags, monkeypatch):
    """ffgpu_exec_forward_bgr_dev with images of the net's geometry on plans that start with k_front (>= 11 frames; 32 = FFGPU_SPLIT2):
    activations and records of every frame against the oracle's net_input + net_forward; no fp32 batch exists afterwards; and the
    two-kernel path (k_input_bgr4 + fp32 k_front, FFGPU_NO_U8_FRONT) gives byte-identical records"""
    import torch
    imgs, settings, runs = u8set
    order = [(5 * f + f // 8) % 8 for f in range(batch)]
    d = torch.from_numpy(np.ascontiguousarray(imgs[order])).cuda()
    keep = F.FFGPU.KEEP_ALL if not flags & F.FFGPU.SPLIT2 else 0
    with net.executor(batch, keep | flags) as ex:
        for si, (mean, norm) in enumerate(settings):
            ex.forward_bgr_dev(d.data_ptr(), 320, 320, mean, norm)
            torch.cuda.synchronize()
            dets = ex.read_dets()
            for f in range(batch):
                want = runs[(si, order[f])]
                if keep:
                    for i in LAYERS:
                        close(ex.read_layer(i, f), want["acts"][i], "batch %d setting %d frame %d layer %d" % (batch, si, f, i))
                assert dets[f]["ncand"] == len(want["cand"]), "setting %d frame %d" % (si, f)
                boxes_match(ex.boxes(f, dets), want["boxes"], "setting %d frame %d" % (si, f))
            if keep:
                with pytest.raises(RuntimeError, match="no fp32 input tensor exists"):
                    ex.read_layer(-1, 0)
            fused = dets.tobytes()
            monkeypatch.setenv("FFGPU_NO_U8_FRONT", "1")
            ex.forward_bgr_dev(d.data_ptr(), 320, 320, mean, norm)
            torch.cuda.synchronize()
            assert ex.read_dets().tobytes() == fused, "two-kernel path differs (setting %d)" % si
            if keep:
                assert ex.read_layer(-1, 0).shape == (3, 320, 320)
            monkeypatch.delenv("FFGPU_NO_U8_FRONT")
        assert ex.graph_captures == 2                               # the fp32 graph and the u8 one, each captured once
        # fp32 frames afterwards: the parameter block switches back
        x = torch.zeros((batch, 3, 320, 320), device="cuda")
        ex.forward_dev(x.data_ptr())
        torch.cuda.synchronize()
        assert ex.graph_captures == 2


def test_u8_resized_images_keep_the_two_kernel_path(F, net, orc, test_image):
    """a source image of another size (letterbox resize, ffcnn.c:267-289) is not the fused case: k_input_bgr4 in front, same results"""
    import torch
    bgr, w, h = test_image
    o = orc.Oracle()
    o.set_input_image(bgr, w, h)
    o.forward(0)
    want = o.boxes
    o.close()
    img = np.frombuffer(bgr, np.uint8).reshape(1, h, -1)
    d = torch.from_numpy(np.ascontiguousarray(np.repeat(img, 16, 0))).cuda()
    with net.executor(16, F.FFGPU.KEEP_ALL) as ex:
        ex.forward_bgr_dev(d.data_ptr(), w, h)
        torch.cuda.synchronize()
        dets = ex.read_dets()
        for f in (0, 7, 15):
            boxes_match(ex.boxes(f, dets), want, "frame %d" % f)
        assert ex.read_layer(-1, 0).shape == (3, 320, 320) and ex.graph_captures == 1


def test_bbox_max_is_read_on_every_forward(F, orc, test_image):
    """NET.bbox_max changed BETWEEN two net_forward calls (the reference re-reads it: ffcnn.c:461-463): the captured graph honours
    the new value (it travels in the parameter block), as FFGPU_NO_GRAPH always did"""
    bgr, w, h = test_image
    wants = {}
    for cap in (0, 2):
        o = orc.Oracle()
        if cap:
            o.n.cap = cap
        o.set_input_image(bgr, w, h)
        o.forward(0)
        wants[cap] = o.boxes
        o.close()
    assert len(wants[0]) == 3 and len(wants[2]) < 3
    with F.Net() as n:
        full = n.n.bbox_max
        for cap in (0, 2, 0, 2):
            n.n.bbox_max = cap if cap else full
            n.set_input_image(bgr, w, h)
            n.forward()
            boxes_match(n.boxes, wants[cap], "bbox_max %d" % cap)


def test_transfers_from_any_caller_memory(F, net, orc, test_image):
    """forward_host / read_layer / read_dets with caller buffers of every provenance -- a sliced numpy view copied on the fly, an
    mmap-sized array, the NET's own page-locked input tensor -- give the same records: every transfer is staged through
    library-owned pinned memory, never through the caller's pages"""
    bgr, w, h = test_image
    net.set_input_image(bgr, w, h)
    img = np.array(net.input)
    with net.executor(3) as ex:
        ex.set_scale(net.n.s1, net.n.s2)
        ex.forward_host(np.stack([img, img * 0, img]))
        want = ex.read_dets().tobytes()
        big = np.zeros((64, 3, 320, 320), np.float32)               # 78 MB: an mmap'd block
        big[10], big[12] = img, img
        for _ in range(3):
            ex.forward_host(big[10:13])
            assert ex.read_dets().tobytes() == want
    with net.executor(1) as ex1:                                     # the NET's own tensor: recognised, one DMA
        ex1.set_scale(net.n.s1, net.n.s2)
        F._check(F.lib().ffgpu_exec_forward_host(ex1.h, net.n.layer_list[0].data), "forward_host")
        assert ex1.read_dets()[0]["count"] == 3


def test_node_run_is_the_submit_wait_loop(F, net, orc, test_image):
    """ffgpu_node_run(steps): the pipelined loop in C; its last step's records equal a synchronous forward of the same slot inputs;
    refused while a step is outstanding"""
    import torch
    bgr, w, h = test_image
    net.set_input_image(bgr, w, h)
    img = torch.from_numpy(np.array(net.input)).cuda()
    with F.Node(net, 2, 6, node_flags=F.Node.LOOPBACK | F.Node.DEPTH(3)) as nd:
        nd.set_scale(net.n.s1, net.n.s2)
        for slot in range(3):
            for r in range(2):
                lo, hi, dev = nd.shard(r)
                from bench import DevBuf
                v = torch.as_tensor(DevBuf(nd.input_slot_dev(r, slot), (hi - lo) * 3 * 320 * 320 * 4, "<f4"), device="cuda").view(hi - lo, 3, 320, 320)
                v.zero_()
                v[(slot + r) % (hi - lo)] = img
        torch.cuda.synchronize()
        want = [nd.forward().copy() for _ in range(3)]              # steps 0, 1, 2 -> slots 0, 1, 2
        for steps in (1, 2, 3, 7):
            got = nd.run(steps)                                      # continues with the slot after the last one used
        # 3 + 1 + 2 + 3 + 7 = 16 steps so far: the last one ran on slot 15 % 3 = 0
        assert got.tobytes() == want[0].tobytes()
        assert [int(c) for c in got["count"]] == [int(c) for c in want[0]["count"]] and int(got["count"].sum()) == 6
        t = nd.submit()
        with pytest.raises(RuntimeError, match="still outstanding"):
            nd.run(1)
        nd.wait(t)

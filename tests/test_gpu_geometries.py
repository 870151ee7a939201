"""yolo-fastest-1.1 at input geometries other than its native 320x320 -- net_load(cfg, weights, w, h) takes any multiple of 32
(ffcnn.c:133-134) -- down to 32x32 (the last planes are 1x1) and up to 704x352, fused executor, two frames: every tensor the
executor still materialises and the boxes against the oracle at the same geometry."""
import numpy as np
import pytest

from test_gpu_parity import boxes_match, close

pytestmark = pytest.mark.gpu

SIZES = [(32, 32), (64, 32), (32, 96), (96, 160), (160, 64), (224, 224), (352, 288), (704, 352), (416, 416), (128, 512)]


@pytest.mark.parametrize("wh", SIZES)
def test_geometry_fused_vs_oracle(orc, wh):
    from ffcnn_amd import capi as F
    F.lib()
    w, h = wh
    rng = np.random.default_rng(w * 1000 + h)
    o = orc.Oracle(w=w, h=h)
    assert (o.n.in_w, o.n.in_h) == (w, h)
    frames = rng.uniform(0, 1, (2, 3, h, w)).astype(np.float32)
    with F.Net(w=w, h=h) as n:
        assert n.input_shape == (3, h, w)
        for flags in (F.FFGPU.KEEP_ALL, F.FFGPU.KEEP_ALL | F.FFGPU.CONCURRENT, 0):
            with n.executor(2, flags) as ex:
                ex.set_scale(1, 1)
                ex.forward_host(frames)
                dets = ex.read_dets()
                for f in range(2):
                    o.input[...] = frames[f]
                    o.n.s1, o.n.s2 = 1, 1
                    o.forward(0)
                    if flags & F.FFGPU.KEEP_ALL:
                        seen = 0
                        for i in range(o.nlayers):
                            ref = o.layer_out(i)
                            if ref is None or n.layer(i).type == 4:
                                continue
                            try:
                                a = ex.read_layer(i, f)
                            except RuntimeError as e:
                                assert "not materialised" in str(e)
                                continue
                            seen += 1
                            close(a, ref, "%dx%d flags %d frame %d layer %d" % (w, h, flags, f, i))
                        assert seen >= 30
                    assert dets[f]["ncand"] == len(o.candidates)
                    boxes_match(ex.read_boxes(f), o.boxes, "%dx%d flags %d frame %d boxes" % (w, h, flags, f))
    o.close()

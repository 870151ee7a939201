"""The batched device net_input (ffgpu_exec_forward_bgr_dev: k_input_bgr / k_input_bgr4 in front of the net) on seeded random
source-image sizes, from 1x1 up to larger than the net in both directions, random pixels, means and norms, batches 1-3:
the letterboxed fp32 input tensor against the oracle's restatement of ffcnn.c:259-289 (nearest sampling by integer ratio
s1/s2, row stride ALIGN(3w,4), zero border)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(6))
def test_random_source_images(orc, seed):
    import torch
    from ffcnn_amd import capi as F
    F.lib()
    rng = np.random.default_rng(4400 + seed)
    o = orc.Oracle()
    with F.Net() as n:
        for case in range(12):
            B = int(rng.integers(1, 4))
            w = int(rng.choice([1, 2, 3, 5, 31, 160, 319, 320, 321, 333, 640, 641, 900, int(rng.integers(1, 1000))]))
            h = int(rng.choice([1, 2, 7, 240, 319, 320, 321, 448, 700, int(rng.integers(1, 800))]))
            stride = (3 * w + 3) & ~3
            imgs = rng.integers(0, 256, (B, h, stride), dtype=np.uint8)
            mean = tuple(float(v) for v in rng.uniform(0, 128, 3)) if rng.random() < 0.5 else (0.0, 0.0, 0.0)
            norm = tuple(float(v) for v in rng.uniform(0.002, 0.02, 3)) if rng.random() < 0.5 else (1 / 255.0,) * 3
            # the host net_input of the drop-in API (ffcnn_host.c) on the same image: bit-identical to the reference's arithmetic,
            # including the scale factors net_forward will use (NET.s1 / s2)
            o.set_input_image(np.ascontiguousarray(imgs[0]), w, h, mean, norm)
            n.set_input_image(np.ascontiguousarray(imgs[0]), w, h, mean, norm)
            assert np.array_equal(np.array(n.input), np.array(o.input)), "host net_input %dx%d" % (w, h)
            assert (n.n.s1, n.n.s2) == (o.n.s1, o.n.s2), "scale %dx%d: %r vs %r" % (w, h, (n.n.s1, n.n.s2), (o.n.s1, o.n.s2))
            d = torch.from_numpy(imgs).cuda()
            with n.executor(B, F.FFGPU.KEEP_ALL) as ex:
                ex.forward_bgr_dev(d.data_ptr(), w, h, mean, norm)
                torch.cuda.synchronize()
                for f in range(B):
                    o.set_input_image(np.ascontiguousarray(imgs[f]), w, h, mean, norm)
                    got = ex.read_layer(-1, f)
                    ref = np.array(o.input)
                    err = np.abs(got - ref)
                    assert err.max() <= 1e-6 + 1e-6 * np.abs(ref).max(), "%dx%d frame %d of %d: max |d| %.3g at %s" % (
                        w, h, f, B, err.max(), np.unravel_index(err.argmax(), err.shape))
    o.close()

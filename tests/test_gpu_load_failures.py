"""net_load's behaviour on bad files, against the reference itself (oracle/_ref): a cfg that does not exist -> NULL; a weights
file that does not exist -> a net with all-zero filter rows (ffcnn.c:205-207: calloc'd weight_buf, no fopen check); a weights
file cut short at an arbitrary byte -> whatever arrived is folded, the rest stays zero (ffcnn.c:211-235 ignores short reads).
The folded weight buffers must be equal float for float."""
import os

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CFG = os.path.join(ROOT, "data", "yolo-fastest-1.1.cfg")
WEIGHTS = os.path.join(ROOT, "data", "yolo-fastest-1.1.weights")


def test_missing_cfg_is_null(orc):
    from ffcnn_amd import capi as F
    L = F.lib()
    assert not L.net_load(b"/nonexistent/x.cfg", WEIGHTS.encode(), 0, 0)
    if orc.have_ref("v0"):
        with pytest.raises(RuntimeError):
            orc.Ref("v0", "/nonexistent/x.cfg", WEIGHTS)


@pytest.mark.parametrize("cut", [None, 0, 19, 20, 21, 4096, 4098, 123457, 400001, -3, -4])
def test_missing_or_truncated_weights_like_the_reference(orc, tmp_path, cut):
    from ffcnn_amd import capi as F
    F.lib()
    if not orc.have_ref("v0"):
        pytest.skip("oracle/_ref not built")
    if cut is None:
        wpath = str(tmp_path / "does_not_exist.weights")
    else:
        raw = open(WEIGHTS, "rb").read()
        wpath = str(tmp_path / "cut.weights")
        open(wpath, "wb").write(raw[:cut] if cut >= 0 else raw[:len(raw) + cut])
    r = orc.Ref("v0", CFG, wpath)
    wr = np.ctypeslib.as_array(r.n.weight_buf, (r.n.weight_size,)).copy()
    with F.Net(CFG, wpath) as n:
        wn = n.weights_host().copy()                              # (a view of NET.weight_buf: gone with the net)
    r.close()
    assert wn.shape == wr.shape
    same = (wn == wr) | (np.isnan(wn) & np.isnan(wr))
    assert same.all(), "cut %r: %d of %d floats differ, first at %d: %r vs reference %r" % (
        cut, int((~same).sum()), wn.size, int(np.argmax(~same)), wn[np.argmax(~same)], wr[np.argmax(~same)])
    if cut is None:
        assert not wn.any()

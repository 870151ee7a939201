"""The cfg / weights loader (ffcnn_host.c, restating ffcnn.c:35-247) on the SAME network written in the styles the reference's
strstr-based parser accepts -- short section names ([conv], [max], [avg]; ffcnn.c:51), spaces around '=', CRLF line ends,
keys in another order, trailing blanks -- against the reference itself (oracle/_ref, its own ffcnn.h structs): every LAYER
field the loader fills and every folded weight must be equal, style by style."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

FIELDS = ("type", "w", "h", "c", "pad", "stride", "fn", "fs", "groups", "batchnorm", "activation", "depend_num", "class_num")


def styles(txt):
    secs = re.split(r"(?m)^(?=\[)", txt)
    def reorder(sec):
        lines = sec.splitlines()
        head, body = lines[0], [ln for ln in lines[1:] if ln.strip() and not ln.lstrip().startswith("#")]
        return "\n".join([head] + body[::-1]) + "\n\n"
    yield "original", txt
    yield "short section names", txt.replace("[convolutional]", "[conv]").replace("[maxpool]", "[max]").replace("[avgpool]", "[avg]")
    yield "spaces around =", re.sub(r"(?m)^([a-z_]+)\s*=\s*", r"\1 = ", txt)
    yield "no spaces around =", re.sub(r"(?m)^([a-z_]+)\s*=\s*", r"\1=", txt)
    yield "CRLF", txt.replace("\n", "\r\n")
    yield "keys reversed", secs[0] + "".join(reorder(s) for s in secs[1:])
    yield "trailing blanks", re.sub(r"(?m)^([a-z_]+\s*=\s*\S.*?)$", r"\1  ", txt)


@pytest.mark.parametrize("cfgname", ["yolo-fastest-1.1.cfg", "mini.cfg", "tiny3.cfg"])
def test_cfg_styles_load_like_the_reference(orc, tmp_path, cfgname):
    from ffcnn_amd import capi as F
    from test_gpu_parity import _write_random_weights
    if not orc.have_ref("v0"):
        pytest.skip("oracle/_ref not built")
    F.lib()
    src = os.path.join(ROOT, "data", cfgname) if cfgname.startswith("yolo") else os.path.join(ROOT, "tests", "data", cfgname)
    txt = open(src).read()
    if cfgname.startswith("yolo"):
        wpath = os.path.join(ROOT, "data", "yolo-fastest-1.1.weights")
    else:
        o = orc.Oracle(cfg=src, weights=None)
        wpath = str(tmp_path / "w.weights")
        _write_random_weights(wpath, o, 7)
        o.close()
    for name, t in styles(txt):
        cfg = str(tmp_path / "style.cfg")
        open(cfg, "w", newline="").write(t)
        r = orc.Ref("v0", cfg, wpath)
        with F.Net(cfg, wpath) as n:
            assert n.layer_num == r.n.layer_num, name
            for i in range(n.layer_num):
                a, b = n.layer(i), r.layer(i)
                for fld in FIELDS:
                    assert getattr(a, fld) == getattr(b, fld), "%s: layer %d field %s: %r vs reference %r" % (name, i, fld, getattr(a, fld), getattr(b, fld))
                assert list(a.depend_list)[:a.depend_num] == list(b.depend_list)[:b.depend_num], "%s: layer %d depend_list" % (name, i)
                if a.type == 7:
                    assert [tuple(x) for x in a.anchor_list] == [tuple(x) for x in b.anchor_list], "%s: layer %d anchors" % (name, i)
                    assert a.ignore_thres == b.ignore_thres and a.scale_x_y == b.scale_x_y, "%s: layer %d yolo floats" % (name, i)
            assert n.n.weight_size == r.n.weight_size, name
            wr = np.ctypeslib.as_array(r.n.weight_buf, (r.n.weight_size,))
            assert np.array_equal(n.weights_host(), wr), name
        r.close()

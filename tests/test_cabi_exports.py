"""CPU-side checks of the C-ABI boundary: the library builds, loads, exports every
symbol include/*.h declares, and refuses to run without a GPU (no CPU fallback)."""
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def capi():
    from ffcnn_amd import capi as m
    m.build_library()
    return m


def test_headers_and_exports_agree(capi):
    L = capi.lib()
    declared = set()
    for h in ("ffcnn.h", "conv.h", "ffcnn_hip.h"):
        txt = open(os.path.join(ROOT, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        declared |= set(re.findall(r"\b((?:net|ffgpu)_[a-z0-9_]+|groupconv)\s*\(", txt))
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    for sym in declared:
        assert hasattr(L, sym), sym


def test_lab_equipment_is_not_in_the_product(capi):
    """ffcnn_hip_diag.h (membench, pipe probes) lives in its own library; the product .so exports none of it and no
    launch-dropping debug switch is compiled into it"""
    import ctypes as C
    txt = open(os.path.join(ROOT, "include", "ffcnn_hip_diag.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    declared = set(re.findall(r"\b(ffgpu_[a-z0-9_]+)\s*\(", txt))
    assert declared == set(capi.DIAG_EXPORTS)
    D, L = C.CDLL(capi.diag_path()), capi.lib()
    for sym in declared:
        assert hasattr(D, sym) and not hasattr(L, sym), sym
    blob = open(capi.library_path(), "rb").read()
    assert b"FFGPU_DBG_SKIP" not in blob and b"FFGPU_DBG_KEEP" not in blob


def test_struct_abi(capi):
    import ctypes as C
    assert (C.sizeof(capi.LAYER), C.sizeof(capi.BBOX), C.sizeof(capi.NET)) == (120, 24, 104)
    assert capi.LAYER.data.offset == 8 and capi.LAYER.filter.offset == 16 and capi.LAYER.w.offset == 24
    assert capi.LAYER.depend_list.offset == 64 and capi.LAYER.class_num.offset == 84 and capi.LAYER.scale_x_y.offset == 116
    assert capi.NET.bbox_list.offset == 16 and capi.NET.weight_buf.offset == 48 and capi.NET.timeused.offset == 68


def test_no_cpu_fallback(capi):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert capi.lib().ffgpu_device_count() == 0
    assert capi.net_load() is None
    assert "no HIP device" in capi.last_error()
    assert capi.net_load("/nonexistent.cfg", None) is None
    assert "cannot read cfg" in capi.last_error()


def test_product_does_not_touch_oracle():
    """The product package and its C sources never reference oracle/ (parity would be void)."""
    for base, _, files in os.walk(os.path.join(ROOT, "ffcnn_amd")):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hpp", ".hip", ".inc", "Makefile")):
                txt = open(os.path.join(base, f), errors="replace").read()
                assert "oracle" not in txt.lower(), os.path.join(base, f)


def test_shard_range_c_equals_python(capi):
    """ffgpu_shard_range (what the C node path cuts a batch with) == ffcnn_amd.dist.shard_range (the torchrun path);
    shards are contiguous, cover the batch and differ by at most one frame"""
    from ffcnn_amd import dist as ffdist
    for total in (1, 2, 7, 32, 64, 255, 256, 257):
        for world in (1, 2, 3, 4, 8):
            prev = 0
            sizes = []
            for r in range(world):
                lo, hi = capi.shard_range(total, r, world)
                assert (lo, hi) == ffdist.shard_range(total, r, world)
                assert lo == prev and hi >= lo
                prev = hi
                sizes.append(hi - lo)
            assert prev == total and max(sizes) - min(sizes) <= 1


def test_node_needs_a_net_with_device_state(capi):
    import ctypes as C
    assert not capi.lib().ffgpu_node_create(None, 1, None, 4, 0, 0)
    assert "NULL net" in capi.last_error()

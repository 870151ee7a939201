"""ctypes bindings for the CPU checker (oracle/liborc.so) and for the unmodified
reference built into oracle/_ref/ (oracle/Makefile).  TEST INFRASTRUCTURE ONLY:
imported by tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline
leg; the product package ffcnn_amd never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
DATA = os.path.join(ROOT, "data")
CFG = os.path.join(DATA, "yolo-fastest-1.1.cfg")
WEIGHTS = os.path.join(DATA, "yolo-fastest-1.1.weights")
BMP = os.path.join(DATA, "test.bmp")

f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int)


def _fp(a):
    return a.ctypes.data_as(f32p)


def build(force=False):
    """Compile liborc.so (and oracle/_ref when /root/reference is mounted)."""
    so = os.path.join(HERE, "liborc.so")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(HERE, "ffcnn_oracle.c")):
        subprocess.check_call(["make", "-C", HERE, "liborc.so"], stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/ffcnn.c"):
        subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)
    return so


def load_bmp(path=BMP):
    """24-bit BMP -> (bgr bytes top-down with stride ALIGN(3w,4), w, h)."""
    raw = open(path, "rb").read()
    w, h = int.from_bytes(raw[18:22], "little"), int.from_bytes(raw[22:26], "little")
    pitch = (w * 3 + 3) & ~3
    rows = np.frombuffer(raw, np.uint8, pitch * h, 54).reshape(h, pitch)[::-1]
    return np.ascontiguousarray(rows), w, h


# --------------------------------------------------------------------------
class OrcLayer(C.Structure):
    _fields_ = [("kind", C.c_int), ("iw", C.c_int), ("ih", C.c_int), ("ic", C.c_int),
                ("ow", C.c_int), ("oh", C.c_int), ("oc", C.c_int),
                ("fs", C.c_int), ("fn", C.c_int), ("stride", C.c_int), ("pad", C.c_int),
                ("groups", C.c_int), ("batchnorm", C.c_int), ("act", C.c_int),
                ("ndep", C.c_int), ("dep", C.c_int * 4),
                ("classes", C.c_int), ("anchors", (C.c_int * 2) * 3),
                ("thresh", C.c_float), ("scale_xy", C.c_float),
                ("filt", f32p), ("out", f32p)]


class Box(C.Structure):
    _fields_ = [("type", C.c_int), ("score", C.c_float), ("x1", C.c_float), ("y1", C.c_float),
                ("x2", C.c_float), ("y2", C.c_float)]

    def tup(self):
        return (self.type, self.score, self.x1, self.y1, self.x2, self.y2)


BOX_DTYPE = np.dtype([("type", "<i4"), ("score", "<f4"), ("x1", "<f4"), ("y1", "<f4"), ("x2", "<f4"), ("y2", "<f4")])


class OrcNet(C.Structure):
    _fields_ = [("nlayers", C.c_int), ("layers", C.POINTER(OrcLayer)),
                ("in_w", C.c_int), ("in_h", C.c_int), ("in_c", C.c_int), ("input", f32p),
                ("nweights", C.c_int), ("weights", f32p),
                ("boxes", C.POINTER(Box)), ("nboxes", C.c_int),
                ("cand", C.POINTER(Box)), ("ncand", C.c_int), ("cap", C.c_int),
                ("s1", C.c_int), ("s2", C.c_int), ("weights_consumed", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(HERE, "liborc.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.orc_load.restype = C.POINTER(OrcNet)
        L.orc_load.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int]
        L.orc_free.argtypes = [C.POINTER(OrcNet)]
        L.orc_input.argtypes = [C.POINTER(OrcNet), C.c_void_p, C.c_int, C.c_int, f32p, f32p]
        L.orc_forward.argtypes = [C.POINTER(OrcNet), C.c_int]
        L.orc_layer_out.restype = f32p
        L.orc_layer_out.argtypes = [C.POINTER(OrcNet), C.c_int]
        L.orc_dump.argtypes = [C.POINTER(OrcNet), C.c_char_p, C.c_int]
        L.orc_groupconv.argtypes = [f32p, f32p, f32p] + [C.c_int] * 13
        L.orc_pool.argtypes = [f32p, f32p] + [C.c_int] * 6
        L.orc_upsample.argtypes = [f32p, f32p] + [C.c_int] * 4
        L.orc_shortcut.argtypes = [f32p, f32p, f32p, C.c_int, C.c_int]
        L.orc_yolo.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_float,
                               C.c_int, C.c_int, C.POINTER(Box), i32p, C.c_int]
        L.orc_nms.restype = C.c_int
        L.orc_nms.argtypes = [C.POINTER(Box), C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        _lib = L
    return _lib


def conv_out_dim(i, fs, pad, stride):
    return (i - fs + 2 * pad) // stride + 1


def groupconv(x, filt, groups, pad, stride, fs, act, compat_v6=0):
    """x: (ic, ih, iw) float32; filt: (fn, K4+4) float32 -> (oc, oh, ow)."""
    ic, ih, iw = x.shape
    fn = filt.shape[0]
    oh, ow = conv_out_dim(ih, fs, pad, stride), conv_out_dim(iw, fs, pad, stride)
    out = np.empty((fn, oh, ow), np.float32)
    x = np.ascontiguousarray(x, np.float32)
    filt = np.ascontiguousarray(filt, np.float32)
    lib().orc_groupconv(_fp(x), _fp(filt), _fp(out), iw, ih, ic, groups, pad, stride, fs, fn, ow, oh, fn, act, compat_v6)
    return out


def pool(x, fs, stride, is_max=1):
    c, h, w = x.shape
    out = np.empty((c, h // stride, w // stride), np.float32)
    x = np.ascontiguousarray(x, np.float32)
    lib().orc_pool(_fp(x), _fp(out), w, h, c, fs, stride, is_max)
    return out


def upsample(x, stride):
    c, h, w = x.shape
    out = np.empty((c, h * stride, w * stride), np.float32)
    x = np.ascontiguousarray(x, np.float32)
    lib().orc_upsample(_fp(x), _fp(out), w, h, c, stride)
    return out


def shortcut(a, b, act=0):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    out = np.empty_like(a)
    lib().orc_shortcut(_fp(a), _fp(b), _fp(out), a.size, act)
    return out


def yolo(x, classes, anchors, thresh, scale_xy, netw, neth, cap=4096):
    """x: (3*(5+classes), h, w) -> structured array of candidates (emission order)."""
    _, h, w = x.shape
    x = np.ascontiguousarray(x, np.float32)
    anc = ((C.c_int * 2) * 3)(*[(C.c_int * 2)(*a) for a in anchors])
    cand = (Box * cap)()
    n = C.c_int(0)
    lib().orc_yolo(_fp(x), w, h, classes, anc, thresh, scale_xy, netw, neth, cand, C.byref(n), cap)
    return np.frombuffer(cand, BOX_DTYPE, n.value).copy()


def nms(cands, thresh=0.5, use_min=1, s1=1, s2=1):
    n = len(cands)
    buf = (Box * max(n, 1))()
    if n:
        C.memmove(buf, np.ascontiguousarray(cands).ctypes.data, n * C.sizeof(Box))
    k = lib().orc_nms(buf, n, thresh, use_min, s1, s2)
    return np.frombuffer(buf, BOX_DTYPE, k).copy()


class Oracle:
    """Whole-net oracle: keeps every layer's activations."""

    def __init__(self, cfg=CFG, weights=WEIGHTS, w=0, h=0):
        self.L = lib()
        self.p = self.L.orc_load(cfg.encode(), weights.encode() if weights else None, w, h)
        if not self.p:
            raise RuntimeError("orc_load failed for %s" % cfg)
        self.n = self.p.contents

    def close(self):
        if self.p:
            self.L.orc_free(self.p)
            self.p = None

    @property
    def nlayers(self):
        return self.n.nlayers

    def layer(self, i):
        return self.n.layers[i]

    @property
    def input(self):
        n = self.n
        return np.ctypeslib.as_array(n.input, (n.in_c, n.in_h, n.in_w))

    def set_input_image(self, bgr, w, h, mean=(0, 0, 0), norm=(1 / 255.0,) * 3):
        m = (C.c_float * 3)(*mean)
        s = (C.c_float * 3)(*norm)
        self.input[...] = 0          # the reference's calloc'd letterbox border
        self.L.orc_input(self.p, bgr.ctypes.data, w, h, m, s)

    def forward(self, compat_v6=0):
        self.L.orc_forward(self.p, compat_v6)

    def layer_out(self, i):
        lay = self.n.layers[i]
        ptr = self.L.orc_layer_out(self.p, i)
        if not ptr or lay.kind == 7:
            return None
        return np.ctypeslib.as_array(ptr, (lay.oc, lay.oh, lay.ow))

    @property
    def boxes(self):
        return np.frombuffer((Box * max(self.n.nboxes, 1)).from_address(C.addressof(self.n.boxes.contents)), BOX_DTYPE, self.n.nboxes).copy()

    @property
    def candidates(self):
        return np.frombuffer((Box * max(self.n.ncand, 1)).from_address(C.addressof(self.n.cand.contents)), BOX_DTYPE, self.n.ncand).copy()

    def weights(self):
        return np.ctypeslib.as_array(self.n.weights, (self.n.nweights,))

    def filter_rows(self, i):
        lay = self.n.layers[i]
        k4 = (lay.fs * lay.fs * (lay.ic // lay.groups) + 3) & ~3
        return np.ctypeslib.as_array(lay.filt, (lay.fn, k4 + 4))

    def dump(self):
        buf = C.create_string_buffer(1 << 16)
        n = self.L.orc_dump(self.p, buf, len(buf))
        return buf.raw[:n].decode()


# --------------------------------------------------------------------------
# the reference itself (oracle/_ref), through its own ffcnn.h ABI (ffcnn.h:16-52)
class RefLayer(C.Structure):
    _fields_ = [("type", C.c_int), ("refcnt", C.c_int), ("data", f32p), ("filter", f32p),
                ("w", C.c_int), ("h", C.c_int), ("c", C.c_int), ("pad", C.c_int), ("stride", C.c_int),
                ("fn", C.c_int), ("fs", C.c_int), ("groups", C.c_int),
                ("batchnorm", C.c_int), ("activation", C.c_int),
                ("depend_list", C.c_int * 4), ("depend_num", C.c_int),
                ("class_num", C.c_int), ("anchor_list", (C.c_int * 2) * 3),
                ("ignore_thres", C.c_float), ("scale_x_y", C.c_float)]


class RefNet(C.Structure):
    _fields_ = [("layer_list", C.POINTER(RefLayer)), ("layer_num", C.c_int),
                ("bbox_list", C.POINTER(Box)), ("bbox_num", C.c_int), ("bbox_max", C.c_int),
                ("s1", C.c_int), ("s2", C.c_int), ("weight_size", C.c_int),
                ("weight_buf", f32p), ("cnntempbuf", f32p), ("cnnbufsize", C.c_int),
                ("timeused", C.c_int * 8)]


assert C.sizeof(RefLayer) == 120 and C.sizeof(RefNet) == 104 and C.sizeof(Box) == 24


def ref_path(name):
    return os.path.join(HERE, "_ref", "libffcnn_ref_%s.so" % name)


def have_ref(name="v0"):
    return os.path.exists(ref_path(name))


def bind_ffcnn_abi(L):
    """Attach the ffcnn.h / conv.h prototypes to a CDLL (reference or product)."""
    L.net_load.restype = C.POINTER(RefNet)
    L.net_load.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int]
    L.net_free.argtypes = [C.POINTER(RefNet)]
    L.net_input.argtypes = [C.POINTER(RefNet), C.c_void_p, C.c_int, C.c_int, f32p, f32p]
    L.net_forward.argtypes = [C.POINTER(RefNet)]
    L.net_dump.argtypes = [C.POINTER(RefNet)]
    L.groupconv.argtypes = [f32p, f32p, f32p] + [C.c_int] * 12 + [C.POINTER(f32p), i32p]
    return L


class Ref:
    """The reference net through its own C API; variant in {v0, v2, v6, v6_fast_*}."""

    def __init__(self, variant="v0", cfg=CFG, weights=WEIGHTS, w=0, h=0):
        self.L = bind_ffcnn_abi(C.CDLL(ref_path(variant)))
        self.p = self.L.net_load(cfg.encode(), weights.encode(), w, h)
        if not self.p:
            raise RuntimeError("reference net_load failed")
        self.n = self.p.contents

    def close(self):
        if self.p:
            self.L.net_free(self.p)
            self.p = None

    def layer(self, i):
        return self.n.layer_list[i]

    def set_input_image(self, bgr, w, h, mean=(0, 0, 0), norm=(1 / 255.0,) * 3):
        m = (C.c_float * 3)(*mean)
        s = (C.c_float * 3)(*norm)
        self.L.net_input(self.p, bgr.ctypes.data, w, h, m, s)

    @property
    def input(self):
        l0 = self.n.layer_list[0]
        return np.ctypeslib.as_array(l0.data, (l0.c, l0.h, l0.w))

    def forward(self, keep_activations=False):
        """keep_activations: pin every tensor (refcnt trick, ffcnn.c:511-516) and
        return {layer index -> output ndarray}; the pinned buffers are released
        afterwards so the NET stays reusable."""
        n = self.n
        if not keep_activations:
            self.L.net_forward(self.p)
            return None
        for i in range(1, n.layer_num + 1):
            n.layer_list[i].refcnt += 1
        self.L.net_forward(self.p)
        outs = {}
        libc = C.CDLL(None)
        libc.free.argtypes = [C.c_void_p]
        for i in range(n.layer_num):
            o = n.layer_list[i + 1]
            if o.data and n.layer_list[i].type != 7:
                outs[i] = np.ctypeslib.as_array(o.data, (o.c, o.h, o.w)).copy()
        # boxes alias the input tensor (ffcnn.c:243-244): copy them before cleanup
        self._boxes = self.boxes
        seen = set()
        for i in range(1, n.layer_num + 1):
            lay = n.layer_list[i]
            lay.refcnt -= 1
            addr = C.cast(lay.data, C.c_void_p).value
            if addr and addr not in seen:
                seen.add(addr)
                libc.free(addr)
            lay.data = None
        return outs

    @property
    def boxes(self):
        k = self.n.bbox_num
        if k == 0:
            return np.zeros(0, BOX_DTYPE)
        return np.frombuffer((Box * k).from_address(C.addressof(self.n.bbox_list.contents)), BOX_DTYPE, k).copy()

    def filter_rows(self, i):
        lay = self.n.layer_list[i]
        k4 = (lay.fs * lay.fs * (lay.c // lay.groups) + 3) & ~3
        return np.ctypeslib.as_array(lay.filter, (lay.fn, k4 + 4))

    def groupconv(self, x, filt, groups, pad, stride, fs, act):
        ic, ih, iw = x.shape
        fn = filt.shape[0]
        oh, ow = conv_out_dim(ih, fs, pad, stride), conv_out_dim(iw, fs, pad, stride)
        out = np.empty((fn, oh, ow), np.float32)
        x = np.ascontiguousarray(x, np.float32)
        filt = np.ascontiguousarray(filt, np.float32)
        buf = f32p()
        size = C.c_int(0)
        self.L.groupconv(_fp(x), _fp(filt), _fp(out), iw, ih, ic, groups, pad, stride, fs, fn, ow, oh, fn, act,
                         C.byref(buf), C.byref(size))
        if buf:
            libc = C.CDLL(None)
            libc.free.argtypes = [C.c_void_p]
            libc.free(C.cast(buf, C.c_void_p))
        return out

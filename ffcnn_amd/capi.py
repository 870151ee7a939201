"""ctypes mirror of include/ffcnn.h, include/conv.h and include/ffcnn_hip.h."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
DATA = os.path.join(ROOT, "data")
CFG = os.path.join(DATA, "yolo-fastest-1.1.cfg")
WEIGHTS = os.path.join(DATA, "yolo-fastest-1.1.weights")

f32p = C.POINTER(C.c_float)


class FFGPU:
    MAX_DET = 128
    KEEP_ALL, COMPAT_V6, NO_GRAPH, NO_FUSE, HOST_DETS, SPLIT2, CONCURRENT, BF16_PW = 1, 2, 4, 8, 16, 32, 64, 128
    K_AUTO, K_GENERIC, K_DW_STREAM, K_DW_LDS, K_PW_MFMA, K_PW_GEMM, _K6, K_DENSE_SMALL, K_IGEMM, K_PW_BF16, K_GROUP_THIN, K_PW_X3, K_CONV_X3, K_PW_X3T = range(14)


class LAYER(C.Structure):            # include/ffcnn.h (120 bytes)
    _fields_ = [("type", C.c_int), ("refcnt", C.c_int), ("data", f32p), ("filter", f32p),
                ("w", C.c_int), ("h", C.c_int), ("c", C.c_int), ("pad", C.c_int), ("stride", C.c_int),
                ("fn", C.c_int), ("fs", C.c_int), ("groups", C.c_int),
                ("batchnorm", C.c_int), ("activation", C.c_int),
                ("depend_list", C.c_int * 4), ("depend_num", C.c_int),
                ("class_num", C.c_int), ("anchor_list", (C.c_int * 2) * 3),
                ("ignore_thres", C.c_float), ("scale_x_y", C.c_float)]


class BBOX(C.Structure):             # 24 bytes
    _fields_ = [("type", C.c_int), ("score", C.c_float), ("x1", C.c_float), ("y1", C.c_float),
                ("x2", C.c_float), ("y2", C.c_float)]


class NET(C.Structure):              # 104 bytes
    _fields_ = [("layer_list", C.POINTER(LAYER)), ("layer_num", C.c_int),
                ("bbox_list", C.POINTER(BBOX)), ("bbox_num", C.c_int), ("bbox_max", C.c_int),
                ("s1", C.c_int), ("s2", C.c_int), ("weight_size", C.c_int),
                ("weight_buf", f32p), ("cnntempbuf", f32p), ("cnnbufsize", C.c_int),
                ("timeused", C.c_int * 8)]


class FrameDets(C.Structure):        # ffgpu_frame_dets
    _fields_ = [("count", C.c_int), ("ncand", C.c_int), ("overflow", C.c_int), ("nfull", C.c_int),
                ("box", BBOX * FFGPU.MAX_DET)]


assert C.sizeof(LAYER) == 120 and C.sizeof(NET) == 104 and C.sizeof(BBOX) == 24
assert C.sizeof(FrameDets) == 16 + 24 * FFGPU.MAX_DET

BOX_DTYPE = np.dtype([("type", "<i4"), ("score", "<f4"), ("x1", "<f4"), ("y1", "<f4"), ("x2", "<f4"), ("y2", "<f4")])
DETS_DTYPE = np.dtype([("count", "<i4"), ("ncand", "<i4"), ("overflow", "<i4"), ("nfull", "<i4"),
                       ("box", BOX_DTYPE, (FFGPU.MAX_DET,))])

# every symbol include/*.h declares; tests check the built library exports all of them
EXPORTS = ["net_load", "net_free", "net_input", "net_forward", "net_dump", "net_profile", "groupconv",
           "ffgpu_device_count", "ffgpu_set_device", "ffgpu_last_error", "ffgpu_build_info",
           "ffgpu_net_weights_dev", "ffgpu_net_weights_commit",
           "ffgpu_exec_create", "ffgpu_exec_destroy", "ffgpu_exec_batch", "ffgpu_exec_arena_bytes",
           "ffgpu_exec_kernel_count", "ffgpu_exec_work_model", "ffgpu_exec_set_scale", "ffgpu_exec_forward_dev", "ffgpu_exec_forward_host",
           "ffgpu_exec_forward_bgr_dev", "ffgpu_exec_dets_dev", "ffgpu_exec_dets_host", "ffgpu_exec_set_ring", "ffgpu_exec_set_ring_strided", "ffgpu_exec_read_dets", "ffgpu_exec_read_layer", "ffgpu_exec_hash_layers",
           "ffgpu_exec_read_boxes", "ffgpu_exec_cand_capacity", "ffgpu_exec_graph_captures",
           "ffgpu_exec_profile", "ffgpu_exec_profile_steps", "ffgpu_exec_step_model", "ffgpu_groupconv_dev", "ffgpu_groupconv_kernel_name", "ffgpu_groupconv_time_dev", "ffgpu_irb_dev", "ffgpu_dwpw_dev", "ffgpu_packed_records_bytes", "ffgpu_pack_records", "ffgpu_unpack_records",
           "ffgpu_shard_range", "ffgpu_node_create", "ffgpu_node_destroy", "ffgpu_node_ndev", "ffgpu_node_shard", "ffgpu_node_set_scale",
           "ffgpu_node_input_dev", "ffgpu_node_input_slot_dev", "ffgpu_node_depth", "ffgpu_node_rccl_ranks", "ffgpu_node_forward", "ffgpu_node_forward_host",
           "ffgpu_node_submit", "ffgpu_node_wait", "ffgpu_node_run"]
# include/ffcnn_hip_diag.h (libffcnn_hip_diag.so: lab equipment, its own library)
DIAG_EXPORTS = ["ffgpu_membench", "ffgpu_pipe_probe", "ffgpu_pipe_probe2", "ffgpu_pipe_probe3", "ffgpu_mfma_floor", "ffgpu_diag_x3_term", "ffgpu_diag_xl_op", "ffgpu_clock_probe"]


def library_path():
    return os.environ.get("FFCNN_HIP_LIB") or os.path.join(HERE, "lib", "libffcnn_hip.so")     # (override: tuning builds)


def build_library(force=False):
    """Compile libffcnn_hip.so in-tree (gcc + hipcc --offload-arch=gfx950)."""
    cmd = ["make", "-C", os.path.join(HERE, "csrc")]
    if force:
        subprocess.check_call(cmd + ["clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return library_path()


_lib = None


def lib():
    """The loaded C-ABI library.  Raises if it was not built: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the HIP extension is the only compute path)" % path)
    L = C.CDLL(path)
    vp, i, sz = C.c_void_p, C.c_int, C.c_size_t
    L.net_load.restype = C.POINTER(NET)
    L.net_load.argtypes = [C.c_char_p, C.c_char_p, i, i]
    L.net_free.argtypes = [C.POINTER(NET)]
    L.net_input.argtypes = [C.POINTER(NET), vp, i, i, f32p, f32p]
    L.net_forward.argtypes = [C.POINTER(NET)]
    L.net_dump.argtypes = [C.POINTER(NET)]
    L.net_profile.argtypes = [C.POINTER(NET)]
    L.groupconv.argtypes = [f32p, f32p, f32p] + [i] * 12 + [C.POINTER(f32p), C.POINTER(i)]
    L.ffgpu_last_error.restype = C.c_char_p
    L.ffgpu_build_info.restype = C.c_char_p
    # a lab build (make DIAG=1: FFGPU_DBG_SKIP / FFGPU_DBG_KEEP drop launches, results are wrong by design) is only loaded when the caller says so
    if b" DIAG " in L.ffgpu_build_info() and os.environ.get("FFCNN_HIP_ALLOW_DIAG") != "1":
        raise RuntimeError("%s is a DIAG (lab) build of libffcnn_hip; set FFCNN_HIP_ALLOW_DIAG=1 to load it on purpose" % path)
    L.ffgpu_set_device.argtypes = [i]
    L.ffgpu_net_weights_dev.argtypes = [C.POINTER(NET), C.POINTER(vp), C.POINTER(sz)]
    L.ffgpu_net_weights_commit.argtypes = [C.POINTER(NET), vp]
    L.ffgpu_exec_create.restype = vp
    L.ffgpu_exec_create.argtypes = [C.POINTER(NET), i, i]
    L.ffgpu_exec_destroy.argtypes = [vp]
    L.ffgpu_exec_batch.argtypes = [vp]
    L.ffgpu_exec_arena_bytes.restype = sz
    L.ffgpu_exec_arena_bytes.argtypes = [vp]
    L.ffgpu_exec_kernel_count.argtypes = [vp]
    L.ffgpu_exec_work_model.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.ffgpu_exec_set_scale.argtypes = [vp, i, i]
    L.ffgpu_exec_forward_dev.argtypes = [vp, vp, vp]
    L.ffgpu_exec_forward_host.argtypes = [vp, f32p]
    L.ffgpu_exec_forward_bgr_dev.argtypes = [vp, vp, i, i, f32p, f32p, vp]
    L.ffgpu_exec_dets_dev.argtypes = [vp, C.POINTER(vp), C.POINTER(sz)]
    L.ffgpu_exec_dets_host.restype = vp; L.ffgpu_exec_dets_host.argtypes = [vp]
    L.ffgpu_exec_set_ring.argtypes = [vp, vp, C.c_int]
    L.ffgpu_exec_set_ring_strided.argtypes = [vp, vp, C.c_int, C.c_int]
    L.ffgpu_exec_read_dets.argtypes = [vp, vp, i]
    L.ffgpu_exec_read_layer.argtypes = [vp, i, i, f32p, sz]
    L.ffgpu_exec_hash_layers.argtypes = [vp, vp, i]
    L.ffgpu_exec_profile.argtypes = [vp, vp, f32p]
    L.ffgpu_exec_profile_steps.argtypes = [vp, vp, C.POINTER(i), f32p, i]
    L.ffgpu_exec_step_model.argtypes = [vp, C.POINTER(i), C.POINTER(C.c_double), i]
    L.ffgpu_groupconv_dev.argtypes = [vp, vp, vp] + [i] * 15 + [vp]
    L.ffgpu_groupconv_kernel_name.restype = C.c_char_p
    L.ffgpu_groupconv_kernel_name.argtypes = [i] * 10
    L.ffgpu_groupconv_time_dev.restype = C.c_float
    L.ffgpu_groupconv_time_dev.argtypes = [vp, vp, vp] + [i] * 17 + [vp]
    L.ffgpu_irb_dev.restype = C.c_float
    L.ffgpu_irb_dev.argtypes = [vp] * 6 + [i] * 13 + [vp]
    L.ffgpu_exec_read_boxes.argtypes = [vp, i, vp, i]
    L.ffgpu_exec_cand_capacity.argtypes = [vp]
    L.ffgpu_exec_graph_captures.argtypes = [vp]
    L.ffgpu_dwpw_dev.restype = C.c_float
    L.ffgpu_dwpw_dev.argtypes = [vp] * 4 + [i] * 10 + [vp]
    L.ffgpu_packed_records_bytes.restype = C.c_size_t
    L.ffgpu_packed_records_bytes.argtypes = [i, i]
    L.ffgpu_pack_records.restype = i
    L.ffgpu_pack_records.argtypes = [vp, i, C.c_long, i, i, vp, vp]
    L.ffgpu_shard_range.argtypes = [i, i, i, C.POINTER(i), C.POINTER(i)]
    L.ffgpu_shard_range.restype = None
    L.ffgpu_node_create.restype = vp
    L.ffgpu_node_create.argtypes = [C.POINTER(NET), i, C.POINTER(i), i, i, i]
    L.ffgpu_node_destroy.argtypes = [vp]
    L.ffgpu_node_ndev.argtypes = [vp]
    L.ffgpu_node_shard.argtypes = [vp, i, C.POINTER(i), C.POINTER(i), C.POINTER(i)]
    L.ffgpu_node_set_scale.argtypes = [vp, i, i]
    L.ffgpu_node_input_dev.restype = vp
    L.ffgpu_node_input_dev.argtypes = [vp, i]
    L.ffgpu_node_input_slot_dev.restype = vp
    L.ffgpu_node_input_slot_dev.argtypes = [vp, i, i]
    L.ffgpu_node_depth.argtypes = [vp]
    if hasattr(L, "ffgpu_node_rccl_ranks"):                     # (absent from older tuning builds loaded through FFCNN_HIP_LIB)
        L.ffgpu_node_rccl_ranks.argtypes = [vp]
    L.ffgpu_node_submit.restype = C.c_long
    L.ffgpu_node_submit.argtypes = [vp, f32p]
    L.ffgpu_node_wait.argtypes = [vp, C.c_long, vp]
    L.ffgpu_node_run.argtypes = [vp, C.c_long, vp]
    L.ffgpu_unpack_records.argtypes = [vp, i, i, vp]
    L.ffgpu_node_forward.argtypes = [vp, vp]
    L.ffgpu_node_forward_host.argtypes = [vp, f32p, vp]
    _lib = L
    return L


_diag = None


def diag_path():
    return os.path.join(HERE, "lib", "libffcnn_hip_diag.so")


def diag():
    """libffcnn_hip_diag.so (include/ffcnn_hip_diag.h): HBM stream calibration and pipe probes; not the product."""
    global _diag
    if _diag is None:
        D = C.CDLL(diag_path())
        vp, i, sz = C.c_void_p, C.c_int, C.c_size_t
        D.ffgpu_membench.restype = C.c_float
        D.ffgpu_membench.argtypes = [vp, vp, sz, i, i, i, vp]
        D.ffgpu_pipe_probe.restype = C.c_float
        D.ffgpu_pipe_probe.argtypes = [i, i, i, i, vp]
        D.ffgpu_pipe_probe2.restype = C.c_float
        D.ffgpu_pipe_probe2.argtypes = [i, i, i, i, vp]
        D.ffgpu_pipe_probe3.restype = C.c_float
        D.ffgpu_pipe_probe3.argtypes = [i, i, i, i, i, vp]
        D.ffgpu_clock_probe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        D.ffgpu_mfma_floor.restype = C.c_float
        D.ffgpu_mfma_floor.argtypes = [vp, i, i, i, vp]
        _diag = D
    return _diag


def load_bmp(path):
    """24-bit BMP -> (bgr rows top-down with stride ALIGN(3w,4), w, h); what the demo feeds net_input."""
    raw = open(path, "rb").read()
    w, h = int.from_bytes(raw[18:22], "little"), int.from_bytes(raw[22:26], "little")
    pitch = (w * 3 + 3) & ~3
    rows = np.frombuffer(raw, np.uint8, pitch * h, 54).reshape(h, pitch)[::-1]
    return np.ascontiguousarray(rows), w, h


def last_error():
    return lib().ffgpu_last_error().decode(errors="replace")


def _check(rc, what):
    if rc < 0:
        raise RuntimeError("%s failed: %s" % (what, last_error()))
    return rc


# ---- ffcnn.h mirror (same names / argument meaning as the reference) -------
def net_load(cfg=CFG, weights=WEIGHTS, inputw=0, inputh=0):
    """NET* or None (cfg unreadable, allocation failure, or no HIP device)."""
    p = lib().net_load(cfg.encode() if cfg else None, weights.encode() if weights else None, inputw, inputh)
    return p if p else None


def net_free(net):
    lib().net_free(net)


def net_input(net, bgr, w, h, mean=(0.0, 0.0, 0.0), norm=(1 / 255.0, 1 / 255.0, 1 / 255.0)):
    m, s = (C.c_float * 3)(*mean), (C.c_float * 3)(*norm)
    lib().net_input(net, bgr.ctypes.data, w, h, m, s)


def net_forward(net):
    lib().net_forward(net)


def net_dump(net):
    lib().net_dump(net)


def groupconv(x, filt, groups, pad, stride, fs, act):
    """conv.h drop-in with host arrays: x (ic, ih, iw), filt (fn, K4+4) -> (fn, oh, ow)."""
    ic, ih, iw = x.shape
    fn = filt.shape[0]
    oh, ow = (ih + 2 * pad - fs) // stride + 1, (iw + 2 * pad - fs) // stride + 1
    x = np.ascontiguousarray(x, np.float32)
    filt = np.ascontiguousarray(filt, np.float32)
    out = np.full((fn, oh, ow), np.nan, np.float32)
    buf, size = f32p(), C.c_int(0)
    lib().groupconv(x.ctypes.data_as(f32p), filt.ctypes.data_as(f32p), out.ctypes.data_as(f32p),
                    iw, ih, ic, groups, pad, stride, fs, fn, ow, oh, fn, act, C.byref(buf), C.byref(size))
    return out


def boxes_of(net):
    n = net.contents
    k = n.bbox_num
    if k <= 0:
        return np.zeros(0, BOX_DTYPE)
    return np.frombuffer((BBOX * k).from_address(C.addressof(n.bbox_list.contents)), BOX_DTYPE, k).copy()


# ---- convenience wrappers ---------------------------------------------------
class Net:
    """Owns a NET* (net_load/net_free) and exposes the layer table."""

    def __init__(self, cfg=CFG, weights=WEIGHTS, w=0, h=0):
        self.p = net_load(cfg, weights, w, h)
        if self.p is None:
            raise RuntimeError("net_load failed: %s" % last_error())
        self.n = self.p.contents

    def close(self):
        if self.p is not None:
            net_free(self.p)
            self.p = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def layer_num(self):
        return self.n.layer_num

    def layer(self, i):
        return self.n.layer_list[i]

    @property
    def input_shape(self):
        l0 = self.n.layer_list[0]
        return (l0.c, l0.h, l0.w)

    @property
    def input(self):
        return np.ctypeslib.as_array(self.n.layer_list[0].data, self.input_shape)

    def set_input_image(self, bgr, w, h, mean=(0.0, 0.0, 0.0), norm=(1 / 255.0,) * 3):
        self.input[...] = 0
        net_input(self.p, bgr, w, h, mean, norm)

    def forward(self):
        net_forward(self.p)

    @property
    def boxes(self):
        return boxes_of(self.p)

    def out_shape(self, i):
        o = self.n.layer_list[i + 1]
        return (o.c, o.h, o.w)

    def weights_host(self):
        return np.ctypeslib.as_array(self.n.weight_buf, (self.n.weight_size,))

    def weights_dev(self):
        ptr, nbytes = C.c_void_p(), C.c_size_t()
        _check(lib().ffgpu_net_weights_dev(self.p, C.byref(ptr), C.byref(nbytes)), "ffgpu_net_weights_dev")
        return ptr.value, nbytes.value

    def weights_commit(self, stream=None):
        _check(lib().ffgpu_net_weights_commit(self.p, stream), "ffgpu_net_weights_commit")

    def executor(self, batch, flags=0):
        return Executor(self, batch, flags)


class Executor:
    """A planned batched executor (ffgpu_exec_*)."""

    def __init__(self, net, batch, flags=0):
        self.net = net
        self.h = lib().ffgpu_exec_create(net.p, batch, flags)
        if not self.h:
            raise RuntimeError("ffgpu_exec_create failed: %s" % last_error())
        self.batch = batch
        self.flags = flags

    def close(self):
        if self.h:
            lib().ffgpu_exec_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def arena_bytes(self):
        return lib().ffgpu_exec_arena_bytes(self.h)

    @property
    def kernel_count(self):
        return lib().ffgpu_exec_kernel_count(self.h)

    def work_model(self):
        """(HBM bytes, flops) one forward of this plan must move / compute"""
        b, f = C.c_double(), C.c_double()
        _check(lib().ffgpu_exec_work_model(self.h, C.byref(b), C.byref(f)), "ffgpu_exec_work_model")
        return b.value, f.value

    def set_scale(self, s1, s2):
        _check(lib().ffgpu_exec_set_scale(self.h, s1, s2), "ffgpu_exec_set_scale")

    def forward_dev(self, dev_ptr, stream=None):
        _check(lib().ffgpu_exec_forward_dev(self.h, dev_ptr, stream), "ffgpu_exec_forward_dev")

    def forward_host(self, frames):
        frames = np.ascontiguousarray(frames, np.float32)
        assert frames.shape == (self.batch,) + self.net.input_shape, frames.shape
        _check(lib().ffgpu_exec_forward_host(self.h, frames.ctypes.data_as(f32p)), "ffgpu_exec_forward_host")

    def forward_bgr_dev(self, dev_ptr, w, h, mean=(0.0, 0.0, 0.0), norm=(1 / 255.0,) * 3, stream=None):
        m, s = (C.c_float * 3)(*mean), (C.c_float * 3)(*norm)
        _check(lib().ffgpu_exec_forward_bgr_dev(self.h, dev_ptr, w, h, m, s, stream), "ffgpu_exec_forward_bgr_dev")

    def dets_dev(self):
        ptr, nbytes = C.c_void_p(), C.c_size_t()
        _check(lib().ffgpu_exec_dets_dev(self.h, C.byref(ptr), C.byref(nbytes)), "ffgpu_exec_dets_dev")
        return ptr.value, nbytes.value

    def dets_host(self):
        """numpy view of the pinned host mirror (FFGPU.HOST_DETS executors); valid after the stream is synchronised"""
        ptr = lib().ffgpu_exec_dets_host(self.h)
        if not ptr:
            raise RuntimeError("ffgpu_exec_dets_host: " + last_error())
        buf = (C.c_char * (DETS_DTYPE.itemsize * self.batch)).from_address(ptr)
        return np.frombuffer(buf, DETS_DTYPE, self.batch)

    def set_ring(self, dev_ptr, slots, slot_records=None):
        """forward k also writes its records into slot k % slots of the device buffer at dev_ptr (None detaches);
        slot_records: distance between slots in records (default: the batch)"""
        if slot_records is None:
            _check(lib().ffgpu_exec_set_ring(self.h, dev_ptr, slots), "ffgpu_exec_set_ring")
        else:
            _check(lib().ffgpu_exec_set_ring_strided(self.h, dev_ptr, slots, slot_records), "ffgpu_exec_set_ring_strided")

    def read_dets(self):
        out = np.zeros(self.batch, DETS_DTYPE)
        _check(lib().ffgpu_exec_read_dets(self.h, out.ctypes.data, self.batch), "ffgpu_exec_read_dets")
        return out

    def boxes(self, frame=0, dets=None):
        d = self.read_dets() if dets is None else dets
        return d[frame]["box"][: d[frame]["count"]].copy()

    def read_layer(self, layer, frame=0):
        shape = self.net.out_shape(layer) if layer >= 0 else self.net.input_shape
        out = np.empty(shape, np.float32)
        _check(lib().ffgpu_exec_read_layer(self.h, layer, frame, out.ctypes.data_as(f32p), out.size), "ffgpu_exec_read_layer")
        return out

    def hash_layers(self):
        """one 64-bit hash per layer over the layer's whole batch tensor (0: not materialised); FFGPU.KEEP_ALL executors"""
        out = np.zeros(self.net.layer_num, np.uint64)
        _check(lib().ffgpu_exec_hash_layers(self.h, out.ctypes.data, out.size), "ffgpu_exec_hash_layers")
        return out

    @property
    def cand_capacity(self):
        return lib().ffgpu_exec_cand_capacity(self.h)

    @property
    def graph_captures(self):
        return lib().ffgpu_exec_graph_captures(self.h)

    def read_boxes(self, frame=0):
        """every box of `frame` that survived NMS (the record keeps the first FFGPU.MAX_DET)"""
        out = np.zeros(max(1, self.cand_capacity), BOX_DTYPE)
        n = _check(lib().ffgpu_exec_read_boxes(self.h, frame, out.ctypes.data, out.size), "ffgpu_exec_read_boxes")
        return out[:n].copy()

    def read_candidates(self, frame=0):
        out = np.zeros(max(1, self.cand_capacity), BOX_DTYPE)
        n = _check(lib().ffgpu_exec_read_layer(self.h, -2, frame, out.ctypes.data_as(f32p), out.size * 6), "read candidates")
        return out[:n].copy()

    def profile_steps(self, dev_ptr):
        cap = 512
        lay, us = (C.c_int * cap)(), (C.c_float * cap)()
        n = _check(lib().ffgpu_exec_profile_steps(self.h, dev_ptr, lay, us, cap), "ffgpu_exec_profile_steps")
        return [(lay[k], us[k]) for k in range(n)]

    def step_model(self):
        """[(layer, model HBM bytes)] per step of the plan (ffgpu_exec_step_model)"""
        cap = 512
        lay, by = (C.c_int * cap)(), (C.c_double * cap)()
        n = _check(lib().ffgpu_exec_step_model(self.h, lay, by, cap), "ffgpu_exec_step_model")
        return [(lay[k], by[k]) for k in range(n)]

    def profile(self, dev_ptr):
        us = (C.c_float * 8)()
        _check(lib().ffgpu_exec_profile(self.h, dev_ptr, us), "ffgpu_exec_profile")
        return list(us)


def shard_range(total, rank, world):
    lo, hi = C.c_int(), C.c_int()
    lib().ffgpu_shard_range(total, rank, world, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


class Node:
    """ffgpu_node_*: one process, ndev GPUs (RCCL broadcast of the weights, gather of the records)"""
    LOOPBACK = 1

    @staticmethod
    def DEPTH(n):
        return (n & 0xf) << 8

    def submit(self, frames=None):
        if frames is not None:
            frames = np.ascontiguousarray(frames, np.float32)
            assert frames.shape == (self.total,) + self.net.input_shape, frames.shape
        t = lib().ffgpu_node_submit(self.h, frames.ctypes.data_as(f32p) if frames is not None else None)
        if t < 0:
            raise RuntimeError("ffgpu_node_submit failed: %s" % last_error())
        return t                                                # (numpy memory is staged inside submit: `frames` is free again)

    def wait(self, ticket):
        out = np.zeros(self.total, DETS_DTYPE)
        _check(lib().ffgpu_node_wait(self.h, ticket, out.ctypes.data), "ffgpu_node_wait")
        return out

    def wait_into(self, ticket, out):
        """as wait(), into a caller-owned DETS_DTYPE array of `total` records"""
        assert out.dtype == DETS_DTYPE and len(out) == self.total and out.flags["C_CONTIGUOUS"]
        _check(lib().ffgpu_node_wait(self.h, ticket, out.ctypes.data), "ffgpu_node_wait")
        return out

    def rccl_ranks(self):
        return lib().ffgpu_node_rccl_ranks(self.h)

    def run(self, steps, out=None):
        """ffgpu_node_run: `steps` pipelined steps from the slots' input buffers (the loop runs in C); records of the last step"""
        if out is None:
            out = np.zeros(self.total, DETS_DTYPE)
        assert out.dtype == DETS_DTYPE and len(out) == self.total and out.flags["C_CONTIGUOUS"]
        _check(lib().ffgpu_node_run(self.h, steps, out.ctypes.data), "ffgpu_node_run")
        return out

    def input_slot_dev(self, rank, slot):
        return lib().ffgpu_node_input_slot_dev(self.h, rank, slot)

    def __init__(self, net, ndev, global_batch, devices=None, exec_flags=0, node_flags=0):
        self.net, self.ndev, self.total = net, ndev, global_batch
        dv = (C.c_int * ndev)(*devices) if devices is not None else None
        self.h = lib().ffgpu_node_create(net.p, ndev, dv, global_batch, exec_flags, node_flags)
        if not self.h:
            raise RuntimeError("ffgpu_node_create failed: %s" % last_error())

    def close(self):
        if self.h:
            lib().ffgpu_node_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def shard(self, rank):
        lo, hi, dev = C.c_int(), C.c_int(), C.c_int()
        _check(lib().ffgpu_node_shard(self.h, rank, C.byref(lo), C.byref(hi), C.byref(dev)), "ffgpu_node_shard")
        return lo.value, hi.value, dev.value

    def set_scale(self, s1, s2):
        _check(lib().ffgpu_node_set_scale(self.h, s1, s2), "ffgpu_node_set_scale")

    def input_dev(self, rank):
        return lib().ffgpu_node_input_dev(self.h, rank)

    def forward(self):
        out = np.zeros(self.total, DETS_DTYPE)
        _check(lib().ffgpu_node_forward(self.h, out.ctypes.data), "ffgpu_node_forward")
        return out

    def forward_host(self, frames):
        frames = np.ascontiguousarray(frames, np.float32)
        assert frames.shape == (self.total,) + self.net.input_shape, frames.shape
        out = np.zeros(self.total, DETS_DTYPE)
        _check(lib().ffgpu_node_forward_host(self.h, frames.ctypes.data_as(f32p), out.ctypes.data), "ffgpu_node_forward_host")
        return out


def groupconv_dev(d_in, d_filt, d_out, batch, iw, ih, ic, groups, pad, stride, fs, fn, act=0, flags=0, variant=0, stream=None):
    ow, oh = (iw + 2 * pad - fs) // stride + 1, (ih + 2 * pad - fs) // stride + 1
    _check(lib().ffgpu_groupconv_dev(d_in, d_filt, d_out, batch, iw, ih, ic, groups, pad, stride, fs, fn, ow, oh, fn,
                                     act, flags, variant, stream), "ffgpu_groupconv_dev")


def groupconv_time_dev(d_in, d_filt, d_out, batch, iw, ih, ic, groups, pad, stride, fs, fn, act=0, flags=0, variant=0,
                       warmup=5, iters=20, stream=None):
    ow, oh = (iw + 2 * pad - fs) // stride + 1, (ih + 2 * pad - fs) // stride + 1
    us = lib().ffgpu_groupconv_time_dev(d_in, d_filt, d_out, batch, iw, ih, ic, groups, pad, stride, fs, fn, ow, oh, fn,
                                        act, flags, variant, warmup, iters, stream)
    if us < 0:
        raise RuntimeError("ffgpu_groupconv_time_dev failed: %s" % last_error())
    return us


def packed_records_bytes(batch, cap):
    return int(lib().ffgpu_packed_records_bytes(batch, cap))


def pack_records_dev(d_records, nslots, slot_stride_records, batch, cap, d_out, stream=None):
    """pack nslots steps of `batch` fixed-size records (device) into compact blocks (device): what the multi-GPU gather moves"""
    _check(lib().ffgpu_pack_records(d_records, nslots, slot_stride_records, batch, cap, d_out, stream), "ffgpu_pack_records")


def kernel_name(batch, iw, ih, ic, groups, pad, stride, fs, fn, variant=0):
    return lib().ffgpu_groupconv_kernel_name(batch, iw, ih, ic, groups, pad, stride, fs, fn, variant).decode()


def dwpw_dev(d_in, d_wd, d_wp, d_out, batch, iw, ih, c, oc, fs, actd=2, actp=0, warmup=0, iters=0, stream=None):
    """fused depthwise KxK (s1, same padding) -> pointwise 1x1; returns us per launch when iters > 0"""
    us = lib().ffgpu_dwpw_dev(d_in, d_wd, d_wp, d_out, batch, iw, ih, c, oc, fs, actd, actp, warmup, iters, stream)
    if us < 0:
        raise RuntimeError("ffgpu_dwpw_dev failed: %s" % last_error())
    return us


def irb_dev(d_in, d_w1, d_wd, d_w2, d_res, d_out, batch, iw, ih, ic, ec, oc, stride, act1=2, actd=2, act2=0, res_act=0,
            warmup=0, iters=0, stream=None):
    """fused expand -> dw3x3 -> project [+ residual]; returns us per launch when iters > 0"""
    us = lib().ffgpu_irb_dev(d_in, d_w1, d_wd, d_w2, d_res, d_out, batch, iw, ih, ic, ec, oc, stride, act1, actd, act2, res_act,
                             warmup, iters, stream)
    if us < 0:
        raise RuntimeError("ffgpu_irb_dev failed: %s" % last_error())
    return us

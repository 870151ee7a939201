// ffgpu_exec.hip -- device executor of the ffcnn forward path + the C-ABI of ffcnn_hip.h.
//
// The reference walks its layer list once per frame, malloc()ing every output
// tensor and free()ing inputs as a per-forward refcount drops to zero
// (ffcnn.c:476-520).  Here the same liveness information is used ONCE, at plan
// time: every tensor of a batch gets a fixed offset in a single HBM arena
// (first-fit over lifetime intervals), the layer list becomes a static list of
// kernel launches, and that list is captured into a HIP graph so a forward is a
// single graph launch.  Plan-time rewrites (all disabled by FFGPU_NO_FUSE and
// by FFGPU_KEEP_ALL):
//   * dropout is a pointer alias (as in ffcnn.c:412-416),
//   * a route with one source is an alias; a multi-source route becomes
//     "concat in place": its sources are allocated inside the route's buffer
//     (a CNHW channel concat is a concatenation of whole blocks),
//   * conv -> [dropout] -> shortcut: the residual add moves into the conv epilogue.
#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "ffgpu_dev.hpp"
#include "conv.h"

// --------------------------------------------------------------------------
static thread_local char g_err[512] = "";

extern "C" void ffgpu_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    if (getenv("FFGPU_VERBOSE")) fprintf(stderr, "ffgpu: %s\n", g_err);
}

extern "C" const char *ffgpu_last_error(void) { return g_err; }

extern "C" const char *ffgpu_build_info(void)
{
#ifdef FFGPU_DIAG
    return "libffcnn_hip gfx950 (CDNA4) HIP DIAG (lab build: launch-dropping switches compiled in) " __DATE__ " " __TIME__;
#else
    return "libffcnn_hip gfx950 (CDNA4) HIP " __DATE__ " " __TIME__;
#endif
}

extern "C" int ffgpu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

extern "C" int ffgpu_set_device(int ordinal)
{
    FFGPU_CHECK(hipSetDevice(ordinal));
    return 0;
}

// -------------------------------------------------------------------------- host <-> device transfers
// Rule (round 3): the GPU never reads or writes a page of the CALLER's heap.  A copy from / to pageable
// memory makes the runtime map those pages for the device (hipHostRegister explicitly, a large hipMemcpy implicitly), and mapping
// and unmapping malloc pages that glibc hands out again, trims and re-grows took the process down about once in ten to thirty
// load / plan / run / free sequences ("Memory access fault by GPU ... on address <a heap address>", GPUTEST_r02's abort).  Every
// transfer between caller memory and the device therefore goes through page-locked memory that the HIP runtime itself allocated
// (hipHostMalloc) and this library owns: the input tensor of a NET, an executor's / node's staging buffers, or the process-wide
// bounce buffer below.  Memory handed out by ffgpu_host_alloc is recognised and copied from directly (one DMA).
namespace {
struct PinnedRange { const char *p; size_t bytes; };
std::mutex g_pin_mu;
std::vector<PinnedRange> g_pinned;

bool pinned_ours(const void *ptr, size_t bytes)
{
    const char *c = static_cast<const char *>(ptr);
    std::lock_guard<std::mutex> lk(g_pin_mu);
    for (const PinnedRange &r : g_pinned) if (c >= r.p && c + bytes <= r.p + r.bytes) return true;
    return false;
}

constexpr size_t BOUNCE_BYTES = (size_t)16 << 20;
// one staging buffer (and lock) per DEVICE, allocated on first use and kept for the life of the process: readers / writers of
// different GPUs (the node path, one thread per device in a caller) do not queue behind each other
struct Bounce { std::mutex mu; char *p = nullptr; };
Bounce g_bounces[32];

Bounce &bounce_cur()
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    return g_bounces[(unsigned)dev % 32u];
}

int bounce_ready(Bounce &b)          // (call with b.mu held)
{
    if (b.p) return 0;
    FFGPU_CHECK(hipHostMalloc((void **)&b.p, BOUNCE_BYTES, hipHostMallocPortable));
    return 0;
}
}

// synchronous copies between caller memory and device memory, in chunks through the bounce buffer
static int copy_h2d(void *d_dst, const void *h_src, size_t bytes)
{
    if (bytes == 0) return 0;
    if (pinned_ours(h_src, bytes)) { FFGPU_CHECK(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice)); return 0; }
    Bounce &bn = bounce_cur();
    std::lock_guard<std::mutex> lk(bn.mu);
    if (bounce_ready(bn)) return -1;
    char *const g_bounce = bn.p;
    for (size_t off = 0; off < bytes; off += BOUNCE_BYTES) {
        const size_t n = std::min(BOUNCE_BYTES, bytes - off);
        memcpy(g_bounce, static_cast<const char *>(h_src) + off, n);
        FFGPU_CHECK(hipMemcpy(static_cast<char *>(d_dst) + off, g_bounce, n, hipMemcpyHostToDevice));
    }
    return 0;
}

static int copy_d2h(void *h_dst, const void *d_src, size_t bytes)
{
    if (bytes == 0) return 0;
    if (pinned_ours(h_dst, bytes)) { FFGPU_CHECK(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost)); return 0; }
    Bounce &bn = bounce_cur();
    std::lock_guard<std::mutex> lk(bn.mu);
    if (bounce_ready(bn)) return -1;
    char *const g_bounce = bn.p;
    for (size_t off = 0; off < bytes; off += BOUNCE_BYTES) {
        const size_t n = std::min(BOUNCE_BYTES, bytes - off);
        FFGPU_CHECK(hipMemcpy(g_bounce, static_cast<const char *>(d_src) + off, n, hipMemcpyDeviceToHost));
        memcpy(static_cast<char *>(h_dst) + off, g_bounce, n);
    }
    return 0;
}

// --------------------------------------------------------------------------
#define FFGPU_INTERNAL_CHILD 0x40000000   /* executor flag used only inside this file: a half of a split executor */

enum StepKind { S_CONV, S_POOL, S_UPSAMPLE, S_ADD, S_COPY, S_YOLO, S_NMS, S_CLEAR, S_TOCNHW, S_IRB, S_FRONT, S_DWPW };

struct Step {
    StepKind kind;
    int      layer;          // reference layer index this step belongs to (-1: none)
    int      ltype;          // LAYER_TYPE_* for profiling
    ConvDesc conv;           // S_CONV (in == nullptr: patched to the batch input at launch); S_DWPW: the depthwise layer
    ConvDesc conv2;          // S_DWPW: the pointwise layer behind it
    bool     in_is_input;    // S_CONV / S_POOL / S_UPSAMPLE / S_TOCNHW read the batch input
    const float *a, *b;      // generic sources
    float   *out;
    long     n;              // element count (ADD/COPY) or planes
    int      w, h, c, fs, stride, flag;
    YoloHead head;
    IrbDesc  irb;            // S_IRB: fused expand -> depthwise -> project [+ shortcut]
    float   *out2[2];        // S_POOL: further stride-1 max pools of the same tensor merged into this launch (fs2[k] != 0)
    int      fs2[2];
    int      lane;           // 0: main stream, 1: side stream (a detection head that runs beside the rest of the net)
};

struct Tensor {
    long size = 0;           // floats
    int  first = 1 << 30, last = -1;
    long off = -1;           // arena offset (roots only)
    int  parent = -1;        // concat-in-place parent tensor
    long parent_off = 0;
    bool used = false;
};

struct ffgpu_netdev {
    NET   *net = nullptr;
    std::mutex mu;                     // guards execs (executors may be created / destroyed from several threads)
    std::vector<ffgpu_exec *> execs;   // every live executor of this net (they hold packings derived from d_weights)
    double us_acc[LAYER_TYPE_TOTOAL] = {};   // FFCNN_PROFILE=1: device time per layer kind, accumulated over net_forward calls
    float *d_weights = nullptr;
    size_t weight_bytes = 0;
    int    device = 0;
    ffgpu_exec *exec1 = nullptr;
};

struct ffgpu_exec {
    NET *net = nullptr;
    ffgpu_netdev *dev = nullptr;
    int  N = 1, flags = 0, device = 0;
    int  in_c = 0, in_h = 0, in_w = 0;
    std::vector<Step>   steps;
    std::vector<Tensor> tensors;       // index = layer id
    std::vector<int>    canon;         // layer id -> tensor id holding its output (-1: the batch input, -2: none, -3: fused away)
    std::vector<char>   readable;      // layer's own output value exists in memory after a forward
    float *arena = nullptr;
    size_t arena_floats = 0;
    float *d_input = nullptr;          // own staging buffer for host / bgr entry points
    float *d_pack = nullptr;           // packed constants of the fused blocks (derived from the weights)
    BBOX  *d_cand = nullptr;           // cand_cap candidate slots per frame: one per anchor of every head cell
    int   *d_cand_key = nullptr, *d_ncand = nullptr;
    int    cand_cap = 1, bbox_max = 1;
    BBOX  *d_full = nullptr;           // every box that survives NMS, score order, cand_cap slots per frame
    void  *d_nms_scratch = nullptr;    // k_nms work arrays when they do not fit LDS
    ExecParams *d_prm = nullptr;       // device parameter block: input pointer + box scale of the forward being enqueued
    bool   indirect = false;           // every launch that reads the batch input does so through d_prm->frames
    ExecParams prm_sent = {}; hipStream_t prm_stream = nullptr; bool prm_valid = false;      // what d_prm holds (or will, on prm_stream)
    const float *last_frames = nullptr;   // input of the last forward (read_layer(-1))
    ffgpu_frame_dets *d_dets = nullptr;
    ffgpu_frame_dets *h_dets = nullptr, *h_dets_dev = nullptr;   // FFGPU_HOST_DETS: pinned mirror and its device address
    // u8 BGR frames converted by the first kernel itself (ffgpu_exec_forward_bgr_dev without resize on a plan that starts with k_front)
    const unsigned char *bgr = nullptr; long bgr_frame = 0; int bgr_pitch = 0; float bgr_mean[3] = {}, bgr_norm[3] = {};
    bool u8_mode = false;              // the forward being enqueued / captured reads `bgr`
    hipGraphExec_t graph_u8 = nullptr; // its graph (the u8 form of the first kernel is another kernel): captured on first use
    float *h_stage = nullptr;          // ffgpu_exec_forward_host from caller memory: page-locked staging of one batch (on first use)
    ffgpu_frame_dets *ring = nullptr; int ring_slots = 0; int *d_ringctr = nullptr;   // ffgpu_exec_set_ring
    int ring_stride = 0;               // records per ring slot (the parent's batch for the halves of a split executor)
    static constexpr int MAXPART = 8;
    ffgpu_exec *child[MAXPART] = {};   // FFGPU_SPLIT2: part-batch executors that run as parallel graph branches
    int nchild = 0;
    hipStream_t part_stream[MAXPART] = {}; hipEvent_t part_ev[MAXPART] = {};   // branch c >= 1 runs on part_stream[c]
    bool is_child = false;             // records / mirror / ring belong to the parent
    int    s1 = 1, s2 = 1;
    hipStream_t own_stream = nullptr, last_stream = nullptr;
    hipStream_t side_stream = nullptr;              // second graph branch
    hipEvent_t  ev_fork = nullptr, ev_join = nullptr;
    int side_lo = -1, side_hi = -1;                 // layers [side_lo, side_hi] form the side branch
    // ONE instantiated graph per executor: the input pointer and the box scale reach the kernels through d_prm.  Only a
    // net whose first layer has no kernel that reads through the slot (indirect == false: e.g. a pool or a pointwise conv
    // straight on the input) falls back to graphs keyed by the input pointer, least recently used one evicted.
    hipGraphExec_t graph1 = nullptr;
    struct Keyed { const float *in; hipGraphExec_t g; unsigned long stamp; };
    std::vector<Keyed> graphs;
    unsigned long graph_stamp = 0;
    int captures = 0;                  // graphs captured + instantiated so far (ffgpu_exec_graph_captures)
    int kernel_count = 0;
};

static float *tensor_ptr(const ffgpu_exec *ex, int t)
{
    long off = 0;
    while (ex->tensors[t].parent >= 0) { off += ex->tensors[t].parent_off; t = ex->tensors[t].parent; }
    return ex->arena + ex->tensors[t].off + off;
}

static int root_of(const std::vector<Tensor> &ts, int t) { while (ts[t].parent >= 0) t = ts[t].parent; return t; }

// -------------------------------------------------------------------------- planning
static int plan(ffgpu_exec *ex)
{
    NET *net = ex->net;
    const int L = net->layer_num, N = ex->N;
    const bool keep_all = ex->flags & FFGPU_KEEP_ALL;
    const bool fuse = !(ex->flags & FFGPU_NO_FUSE);
    const LAYER *ll = net->layer_list;
    std::vector<Tensor> &T = ex->tensors;
    std::vector<int> &canon = ex->canon;
    T.assign(L, Tensor());
    canon.assign(L, -2);
    auto out_floats = [&](int i) { return (long)ll[i + 1].w * ll[i + 1].h * ll[i + 1].c * N; };
    auto src_tensor = [&](int i) { return i < 0 ? -1 : canon[i]; };          // tensor holding the output of layer i

    // pass 1: aliases and shortcut fusion targets
    std::vector<int> fused_into(L, -1);         // conv p -> shortcut s whose add it absorbs
    std::vector<int> nuses(L, 0);               // consumers per LAYER output (before aliasing)
    for (int i = 0; i < L; i++) {
        if (i > 0 && ll[i].type != LAYER_TYPE_ROUTE) nuses[i - 1]++;
        for (int k = 0; k < ll[i].depend_num; k++) {
            const int d = ll[i].depend_list[k];
            if (d < 0 || d >= i) { ffgpu_set_error("layer %d depends on layer %d (not earlier)", i, d); return -1; }
            nuses[d]++;
        }
    }
    for (int i = 0; i < L; i++) {
        switch (ll[i].type) {
        case LAYER_TYPE_DROPOUT: canon[i] = src_tensor(i - 1); break;
        case LAYER_TYPE_YOLO: canon[i] = -2; break;
        case LAYER_TYPE_ROUTE:
            canon[i] = (ll[i].depend_num == 1 && src_tensor(ll[i].depend_list[0]) >= 0) ? src_tensor(ll[i].depend_list[0]) : i;
            break;
        default: canon[i] = i;
        }
        if (fuse && ll[i].type == LAYER_TYPE_SHORTCUT && i > 0) {
            // walk back over dropouts to the producer of our input
            int p = i - 1;
            while (p > 0 && ll[p].type == LAYER_TYPE_DROPOUT && nuses[p] == 1) p--;
            bool chain_private = ll[p].type == LAYER_TYPE_CONV && nuses[p] == 1 && src_tensor(ll[i].depend_list[0]) >= 0;
            for (int q = p + 1; q < i && chain_private; q++) chain_private = ll[q].type == LAYER_TYPE_DROPOUT && nuses[q] == 1;
            if (chain_private) { fused_into[p] = i; for (int q = p; q < i; q++) canon[q] = i; }
        }
    }
    // pass 1b: 1x1 expand -> depthwise 3x3 -> 1x1 project triples become ONE fused kernel (ffgpu_irb.inc);
    // the two expanded tensors are never materialised
    std::vector<int> irb_tail(L, -1);           // layer p+2 -> p
    static const bool no_irb = getenv("FFGPU_NO_IRB") && atoi(getenv("FFGPU_NO_IRB"));
    if (fuse && !no_irb) {
        for (int p0 = 0; p0 + 2 < L; p0++) {
            const LAYER &a = ll[p0], &b = ll[p0 + 1], &c = ll[p0 + 2];
            if (a.type != LAYER_TYPE_CONV || b.type != LAYER_TYPE_CONV || c.type != LAYER_TYPE_CONV) continue;
            const bool pw_a = a.fs == 1 && a.stride == 1 && a.pad == 0 && a.groups == 1;
            const bool dw_b = b.fs == 3 && b.pad == 1 && (b.stride == 1 || b.stride == 2) && b.groups == b.c && b.fn == b.c;
            const bool pw_c = c.fs == 1 && c.stride == 1 && c.pad == 0 && c.groups == 1;
            if (!pw_a || !dw_b || !pw_c || nuses[p0] != 1 || nuses[p0 + 1] != 1 || src_tensor(p0 - 1) < 0) continue;
            if (canon[p0] != p0 || canon[p0 + 1] != p0 + 1) continue;
            IrbDesc d{};
            d.N = N; d.H = a.h; d.W = a.w; d.OH = ll[p0 + 3].h; d.OW = ll[p0 + 3].w;
            d.ic = a.c; d.ec = a.fn; d.oc = c.fn; d.stride = b.stride; d.flags = ex->flags & FFGPU_CONCURRENT;
            d.act1 = a.activation; d.actd = b.activation; d.act2 = c.activation;
            d.res_act = fused_into[p0 + 2] >= 0 ? ll[fused_into[p0 + 2]].activation : 0;
            static const int min_ec = getenv("FFGPU_IRB_MIN_EC") ? atoi(getenv("FFGPU_IRB_MIN_EC")) : 24;
            if ((d.ec < min_ec && !ffgpu_irb_is_thin(d)) || !ffgpu_irb_supported(d)) continue;   // thin blocks take the streaming fused kernel
            irb_tail[p0 + 2] = p0;
            canon[p0] = canon[p0 + 1] = -3;
            p0 += 2;
        }
    }
    // pass 1c (opt-in, FFGPU_DWPW=1): depthwise K x K (stride 1) -> 1x1 pairs that no fused block claimed (the heads' 5x5 + 1x1
    // of yolo-fastest) become one launch (ffgpu_dwpw.inc); the depthwise tensor is never materialised.  MEASURED (r02,
    // tools/dwpw_bench.py): correct but SLOWER than the two launches it replaces -- 61 vs 8 + 15 us on 20x20x120 at batch 64
    // (one workgroup per CU, a barrier chain of 15 short depthwise / MFMA phases with one wave per SIMD; 147.5 k frames/s
    // end to end against 182.7 k) -- so the planner leaves the pairs alone unless asked.
    auto layer_desc = [&](int i) {
        ConvDesc d{};
        const LAYER &a = ll[i], &b = ll[i + 1];
        d.N = N; d.iw = a.w; d.ih = a.h; d.ic = a.c; d.ow = b.w; d.oh = b.h; d.oc = b.c;
        d.fs = a.fs; d.stride = a.stride; d.pad = a.pad; d.groups = a.groups; d.act = a.activation;
        d.flags = ex->flags & FFGPU_COMPAT_V6;
        d.in_cs = (long)N * a.w * a.h; d.in_ns = (long)a.w * a.h;
        d.out_cs = (long)N * b.w * b.h; d.out_ns = (long)b.w * b.h;
        return d;
    };
    std::vector<int> dwpw_tail(L, -1);          // layer p+1 -> p
    if (fuse && getenv("FFGPU_DWPW") && atoi(getenv("FFGPU_DWPW"))) {
        for (int p0 = 0; p0 + 1 < L; p0++) {
            if (ll[p0].type != LAYER_TYPE_CONV || ll[p0 + 1].type != LAYER_TYPE_CONV) continue;
            if (canon[p0] != p0 || canon[p0 + 1] != p0 + 1 || nuses[p0] != 1 || fused_into[p0] >= 0 || fused_into[p0 + 1] >= 0 || src_tensor(p0 - 1) < 0) continue;
            if (!ffgpu_dwpw_ok(layer_desc(p0), layer_desc(p0 + 1))) continue;
            dwpw_tail[p0 + 1] = p0;
            canon[p0] = -3;
            p0++;
        }
    }
    for (int i = 0; i < L; i++) if (canon[i] == i) { T[i].used = true; T[i].size = out_floats(i); }
    ex->readable.assign(L, 0);
    for (int i = 0; i < L; i++) {
        if (canon[i] == i) ex->readable[i] = 1;
        else if (ll[i].type == LAYER_TYPE_DROPOUT && i > 0 && canon[i] >= 0 && canon[i] == canon[i - 1]) ex->readable[i] = ex->readable[i - 1];
        else if (ll[i].type == LAYER_TYPE_ROUTE && ll[i].depend_num == 1 && canon[i] >= 0) ex->readable[i] = ex->readable[ll[i].depend_list[0]];
    }

    // pass 2: concat-in-place for multi-source routes
    std::vector<char> route_inplace(L, 0);
    if (fuse) {
        for (int i = 0; i < L; i++) {
            if (ll[i].type != LAYER_TYPE_ROUTE || canon[i] != i || ll[i].depend_num < 2) continue;
            bool ok = true;
            std::vector<int> kids;
            for (int k = 0; k < ll[i].depend_num && ok; k++) {
                const int t = src_tensor(ll[i].depend_list[k]);
                ok = t >= 0 && T[t].parent < 0 && std::find(kids.begin(), kids.end(), t) == kids.end();
                kids.push_back(t);
            }
            if (!ok) continue;
            long off = 0;
            for (int t : kids) { T[t].parent = i; T[t].parent_off = off; off += T[t].size; }
            route_inplace[i] = 1;
        }
    }

    // pass 3: lifetimes (in layer-index time), folded onto concat roots
    auto touch = [&](int t, int when) { if (t >= 0) { T[t].first = std::min(T[t].first, when); T[t].last = std::max(T[t].last, when); } };
    for (int i = 0; i < L; i++) {
        if (ll[i].type == LAYER_TYPE_DROPOUT) continue;
        if (canon[i] >= 0 && !(ll[i].type == LAYER_TYPE_ROUTE && canon[i] != i)) touch(canon[i], i);     // written at i
        if (ll[i].type != LAYER_TYPE_ROUTE) touch(src_tensor(i - 1), i);                                  // chain input read at i
        for (int k = 0; k < ll[i].depend_num; k++) touch(src_tensor(ll[i].depend_list[k]), i);
        if (irb_tail[i] >= 0) touch(src_tensor(irb_tail[i] - 1), i);
        if (dwpw_tail[i] >= 0) touch(src_tensor(dwpw_tail[i] - 1), i);
        if (irb_tail[i] >= 0 && fused_into[i] >= 0) touch(src_tensor(ll[fused_into[i]].depend_list[0]), i);
    }
    for (int t = 0; t < L; t++) {
        if (!T[t].used || T[t].parent < 0) continue;
        const int r = root_of(T, t);
        T[r].first = std::min(T[r].first, T[t].first);
        T[r].last = std::max(T[r].last, T[t].last);
    }

    // pass 3b: the first detection head runs BESIDE the rest of the net.  In a yolo cfg the layer after a [yolo]
    // is a [route] back to a tensor produced at layer r: layers r+1 .. yolo only feed that head, everything after the
    // yolo only depends on layers <= r, so the two chains become parallel branches of the HIP graph (the head's
    // 10x10 kernels are latency-bound and hide under the other branch).  Their tensors must then not share arena
    // space: every tensor born after the fork stays live to the end.
    ex->side_lo = ex->side_hi = -1;
    // MEASURED (r01): 1.5 % slower than the single chain while the head kernels were slow, 2 % FASTER once they were
    // latency-sized (0.766 vs 0.782 ms per 64-frame batch) -> on by default; FFGPU_BRANCH=0 turns it off.
    // (not inside the halves of a split executor: a fork nested in a forked stream crashes graph capture on ROCm 7.0,
    //  and the second half-batch chain already fills the gaps the head branch was meant to fill)
    //  nor in an FFGPU_CONCURRENT plan: the other chains fill those gaps -- 0.489 vs 0.455 ms with the branch on)
    //  Round 3: OPT-IN (FFGPU_BRANCH=1).  A graph with a forked branch makes the runtime (ROCm 7.0) run it on internal streams of
    //  its own, and that part of the runtime is the fragile one: it kept an arena's worth of memory per destroyed graph until a
    //  device-wide sync was put in front of hipGraphExecDestroy (ffgpu_exec_destroy), and the one host crash that survived the
    //  round-3 transfer rework (SIGSEGV inside hipGraphLaunch of a freshly created executor, once in 30
    //  runs of 630 create / launch / destroy cycles each) was in its walk over those streams.  2 % of ONE chain's latency is not
    //  worth a process; throughput configurations (several chains, FFGPU_CONCURRENT) never used the branch.
    const bool branch = (getenv("FFGPU_BRANCH") ? atoi(getenv("FFGPU_BRANCH")) != 0 : false) && !ex->is_child;
    if (fuse && branch) {
        for (int y = 0; y + 1 < L; y++) {
            if (ll[y].type != LAYER_TYPE_YOLO || ll[y + 1].type != LAYER_TYPE_ROUTE || ll[y + 1].depend_num != 1) continue;
            const int r = ll[y + 1].depend_list[0];
            bool ok = r >= 0 && r < y;
            // nothing after the yolo may read a tensor produced inside (r, y]
            for (int i = y + 1; i < L && ok; i++)
                for (int k = 0; k < ll[i].depend_num; k++) if (ll[i].depend_list[k] > r && ll[i].depend_list[k] <= y) ok = false;
            // nothing inside (r, y] may be a fused block straddling the fork (its first layer must be > r)
            for (int i = r + 1; i <= y && ok; i++) if ((irb_tail[i] >= 0 && irb_tail[i] <= r) || (dwpw_tail[i] >= 0 && dwpw_tail[i] <= r)) ok = false;
            if (fused_into[r] > r) ok = false;       // the fork tensor's producer was absorbed by a later shortcut
            if (!ok) continue;
            ex->side_lo = r + 1; ex->side_hi = y;
            for (int t = 0; t < L; t++) if (T[t].used && T[t].first > r) T[t].last = L + 1;
            for (int i = r + 1; i <= y; i++) {       // ... and whatever the side branch READS (its kernels may run late)
                if (ll[i].type != LAYER_TYPE_ROUTE && src_tensor(i - 1) >= 0) T[src_tensor(i - 1)].last = L + 1;
                for (int k = 0; k < ll[i].depend_num; k++) if (src_tensor(ll[i].depend_list[k]) >= 0) T[src_tensor(ll[i].depend_list[k])].last = L + 1;
                if (irb_tail[i] >= 0 && src_tensor(irb_tail[i] - 1) >= 0) T[src_tensor(irb_tail[i] - 1)].last = L + 1;
                if (dwpw_tail[i] >= 0 && src_tensor(dwpw_tail[i] - 1) >= 0) T[src_tensor(dwpw_tail[i] - 1)].last = L + 1;
            }
            for (int t = 0; t < L; t++) {
                if (!T[t].used || T[t].parent < 0) continue;
                const int rt = root_of(T, t);
                T[rt].last = std::max(T[rt].last, T[t].last);
            }
            break;
        }
    }

    // pass 4: first-fit arena allocation of the roots, 256-byte granules
    struct Live { long off, size; int last; };
    std::vector<Live> live;
    long high = 0;
    std::vector<int> order;
    for (int t = 0; t < L; t++) if (T[t].used && T[t].parent < 0) order.push_back(t);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return T[a].first != T[b].first ? T[a].first < T[b].first : a < b; });
    for (int t : order) {
        const long need = (T[t].size + 63) & ~63L;
        if (keep_all) { T[t].off = high; high += need; continue; }
        live.erase(std::remove_if(live.begin(), live.end(), [&](const Live &l) { return l.last < T[t].first; }), live.end());
        std::sort(live.begin(), live.end(), [](const Live &a, const Live &b) { return a.off < b.off; });
        long at = 0;
        for (const Live &l : live) { if (at + need <= l.off) break; at = std::max(at, l.off + l.size); }
        T[t].off = at;
        live.push_back({ at, need, T[t].last });
        high = std::max(high, at + need);
    }
    ex->arena_floats = (size_t)std::max(high, 64L);
    if (hipMalloc(&ex->arena, ex->arena_floats * sizeof(float)) != hipSuccess) {
        ffgpu_set_error("arena hipMalloc(%zu bytes) failed", ex->arena_floats * sizeof(float));
        return -1;
    }

    // pass 5: steps
    std::vector<Step> &S = ex->steps;
    S.clear();
    { Step c{}; c.kind = S_CLEAR; c.layer = -1; c.ltype = LAYER_TYPE_YOLO; S.push_back(c); }
    int key_base = 0, nheads = 0;
    float *cnhw_input = nullptr;     // set when the first layer cannot read the frame-major input directly
    bool bad_chain = false;
    auto in_ptr = [&](int i, bool *is_input) -> const float * {
        const int t = src_tensor(i - 1);
        *is_input = t == -1;
        if (t < -1) bad_chain = true;            // e.g. a conv chained directly behind a yolo head
        return t >= 0 ? tensor_ptr(ex, t) : nullptr;
    };
    for (int i = 0; i < L; i++) {
        const LAYER &a = ll[i], &b = ll[i + 1];
        Step st{};
        st.layer = i; st.ltype = a.type;
        bool from_input = false;
        switch (a.type) {
        case LAYER_TYPE_CONV: {
            if (canon[i] == -3) break;                              // expanded tensors of a fused block
            if (irb_tail[i] >= 0) {
                const int p0 = irb_tail[i];
                const LAYER &la = ll[p0], &lb = ll[p0 + 1], &lc = ll[p0 + 2];
                IrbDesc &d = st.irb;
                st.kind = S_IRB;
                d.in = tensor_ptr(ex, src_tensor(p0 - 1));
                d.out = tensor_ptr(ex, canon[i]);
                d.w1 = ex->dev->d_weights + (la.filter - net->weight_buf);
                d.wd = ex->dev->d_weights + (lb.filter - net->weight_buf);
                d.w2 = ex->dev->d_weights + (lc.filter - net->weight_buf);
                d.N = N; d.H = la.h; d.W = la.w; d.OH = b.h; d.OW = b.w; d.flags = ex->flags & FFGPU_CONCURRENT;
                d.ic = la.c; d.ec = la.fn; d.oc = lc.fn; d.stride = lb.stride;
                d.act1 = la.activation; d.actd = lb.activation; d.act2 = lc.activation;
                if (fused_into[i] >= 0) {
                    const LAYER &sc = ll[fused_into[i]];
                    d.residual = tensor_ptr(ex, src_tensor(sc.depend_list[0]));
                    d.res_act = sc.activation;
                }
                S.push_back(st);
                break;
            }
            if (dwpw_tail[i] >= 0) {                                // depthwise (p0) + this pointwise layer: one launch
                const int p0 = dwpw_tail[i];
                st.kind = S_DWPW;
                st.conv = layer_desc(p0); st.conv2 = layer_desc(i);
                st.conv.in = tensor_ptr(ex, src_tensor(p0 - 1));
                st.conv.filt = ex->dev->d_weights + (ll[p0].filter - net->weight_buf);
                st.conv2.filt = ex->dev->d_weights + (a.filter - net->weight_buf);
                st.conv2.out = tensor_ptr(ex, canon[i]);
                S.push_back(st);
                break;
            }
            ConvDesc &d = st.conv;
            st.kind = S_CONV;
            d.in = in_ptr(i, &from_input);
            st.in_is_input = from_input;
            d.filt = ex->dev->d_weights + (a.filter - net->weight_buf);
            d.out = tensor_ptr(ex, canon[i]);
            d.N = N; d.iw = a.w; d.ih = a.h; d.ic = a.c; d.ow = b.w; d.oh = b.h; d.oc = b.c;
            d.fs = a.fs; d.stride = a.stride; d.pad = a.pad; d.groups = a.groups; d.act = a.activation;
            d.flags = ex->flags & (FFGPU_COMPAT_V6 | FFGPU_BF16_PW);
            if (from_input) { d.in_cs = (long)a.w * a.h; d.in_ns = (long)a.c * a.w * a.h; }     // frame-major batch input
            else            { d.in_cs = (long)N * a.w * a.h; d.in_ns = (long)a.w * a.h; }
            d.out_cs = (long)N * b.w * b.h; d.out_ns = (long)b.w * b.h;
            if (fused_into[i] >= 0) {
                const LAYER &sc = ll[fused_into[i]];
                d.residual = tensor_ptr(ex, src_tensor(sc.depend_list[0]));
                d.res_act = sc.activation; d.res_cs = d.out_cs; d.res_ns = d.out_ns;
            }
            S.push_back(st);
            break; }
        case LAYER_TYPE_AVGPOOL: case LAYER_TYPE_MAXPOOL: case LAYER_TYPE_UPSAMPLE: case LAYER_TYPE_SHORTCUT: {
            const float *src = in_ptr(i, &from_input);
            if (from_input && N > 1) {          // needs CNHW order: convert the input once
                if (!cnhw_input) {
                    if (hipMalloc(&cnhw_input, (size_t)a.w * a.h * a.c * N * sizeof(float)) != hipSuccess) { ffgpu_set_error("hipMalloc failed"); return -1; }
                    Step cv{}; cv.kind = S_TOCNHW; cv.layer = -1; cv.ltype = a.type; cv.out = cnhw_input; cv.c = a.c; cv.w = a.w; cv.h = a.h; cv.in_is_input = true;
                    S.push_back(cv);
                }
                src = cnhw_input; from_input = false;
            }
            st.in_is_input = from_input;
            st.a = src;
            if (a.type == LAYER_TYPE_SHORTCUT) {
                if (canon[i - 1] == i) break;                       // absorbed by the producing conv
                st.kind = S_ADD;
                st.b = tensor_ptr(ex, src_tensor(a.depend_list[0]));
                st.out = tensor_ptr(ex, canon[i]);
                st.n = out_floats(i); st.flag = a.activation;
            } else {
                st.kind = a.type == LAYER_TYPE_UPSAMPLE ? S_UPSAMPLE : S_POOL;
                st.out = tensor_ptr(ex, canon[i]);
                st.c = a.c; st.w = a.w; st.h = a.h; st.fs = a.fs; st.stride = a.stride;
                st.flag = a.type == LAYER_TYPE_MAXPOOL;
            }
            S.push_back(st);
            break; }
        case LAYER_TYPE_ROUTE: {
            if (canon[i] != i || route_inplace[i]) break;           // alias / concat in place
            long off = 0;
            for (int k = 0; k < a.depend_num; k++) {
                const int t = src_tensor(a.depend_list[k]);
                if (t < 0) { ffgpu_set_error("route %d: unsupported source", i); return -1; }
                const long cnt = (long)ll[a.depend_list[k] + 1].w * ll[a.depend_list[k] + 1].h * ll[a.depend_list[k] + 1].c * N;
                Step cp{}; cp.kind = S_COPY; cp.layer = i; cp.ltype = a.type;
                cp.a = tensor_ptr(ex, t); cp.out = tensor_ptr(ex, i) + off; cp.n = cnt;
                S.push_back(cp);
                off += cnt;
            }
            break; }
        case LAYER_TYPE_YOLO: {
            st.kind = S_YOLO;
            YoloHead &hd = st.head;
            hd.in = in_ptr(i, &from_input);
            if (from_input) { ffgpu_set_error("yolo head directly on the input is not supported"); return -1; }
            hd.w = a.w; hd.h = a.h; hd.classes = a.class_num;
            memcpy(hd.anchors, a.anchor_list, sizeof hd.anchors);
            hd.thresh = a.ignore_thres; hd.scale_xy = a.scale_x_y; hd.key_base = key_base;
            key_base += a.w * a.h * 3;
            nheads++;
            S.push_back(st);
            break; }
        default: break;                                             // dropout
        }
    }
    // first layer + the thin block behind it -> one streaming kernel (ffgpu_front.inc): the 8-channel tensor between them
    // is never written (its arena space stays reserved; read_layer refuses it like the inside of a fused block)
    if (fuse) {
        for (size_t k = 0; k + 1 < S.size(); k++) {
            Step &c0 = S[k];
            const Step &b0 = S[k + 1];
            if (c0.kind != S_CONV || b0.kind != S_IRB || c0.layer < 0 || nuses[c0.layer] != 1) continue;
            if (!ffgpu_front_ok(c0.conv, b0.irb)) continue;
            c0.kind = S_FRONT; c0.irb = b0.irb;
            ex->readable[c0.layer] = 0;
            S.erase(S.begin() + k + 1);
            break;
        }
    }
    {   // constants of the fused blocks, packed once into their LDS image
        size_t tot = 0;
        for (Step &st : S) {
            if (st.kind == S_IRB) { ffgpu_irb_plan(st.irb); tot += ffgpu_irb_pack_floats(st.irb); }
            if (st.kind == S_CONV) {                              // kernel choice, split-K and k_conv_x3's MT frozen with the plan
                if (st.in_is_input) st.conv.flags |= FFGPU_F_BATCH_INPUT;
                ffgpu_conv_plan(st.conv);
                tot += ffgpu_pw_pack_floats(st.conv);
            }
            if (st.kind == S_DWPW) tot += ffgpu_dwpw_pack_floats(st.conv, st.conv2);
        }
        if (tot) {
            if (hipMalloc(&ex->d_pack, tot * sizeof(float)) != hipSuccess) { ffgpu_set_error("hipMalloc(pack) failed"); return -1; }
            size_t off = 0;
            for (Step &st : S) {
                if (st.kind == S_IRB) { st.irb.pk = ex->d_pack + off; off += ffgpu_irb_pack_floats(st.irb); }
                if (st.kind == S_DWPW) { st.conv2.wpack = ex->d_pack + off; off += ffgpu_dwpw_pack_floats(st.conv, st.conv2); }
                if (st.kind == S_CONV && ffgpu_pw_pack_floats(st.conv)) { st.conv.wpack = ex->d_pack + off; off += ffgpu_pw_pack_floats(st.conv); }
            }
        }
    }
    if (bad_chain) { ffgpu_set_error("a layer consumes the (non-existent) output of a yolo head"); return -1; }
    // stride-1 max pools of one tensor that follow each other (the routes between them are aliases: an SPP block)
    // become one launch
    if (fuse) {
        for (size_t i = 0; i + 1 < S.size(); i++) {
            Step &a = S[i];
            if (a.kind != S_POOL || !a.flag || a.stride != 1 || a.in_is_input || (long)a.w * a.h > 8192) continue;
            int k = 0;
            while (k < 2 && i + 1 < S.size()) {
                const Step &b = S[i + 1];
                if (b.kind != S_POOL || !b.flag || b.stride != 1 || b.in_is_input || b.a != a.a || b.c != a.c || b.w != a.w || b.h != a.h) break;
                a.out2[k] = b.out; a.fs2[k] = b.fs; k++;
                S.erase(S.begin() + i + 1);
            }
        }
    }
    // with a detection head in the net no clear launch is needed: k_nms zeroes the candidate counters it has consumed
    // (they start zeroed) and the first head counts the forward for the record ring
    if (nheads > 0 && fuse) {
        S.erase(std::remove_if(S.begin(), S.end(), [](const Step &t) { return t.kind == S_CLEAR; }), S.end());
        for (Step &t : S) if (t.kind == S_YOLO) { t.flag = 1; break; }
    }
    { Step nm{}; nm.kind = S_NMS; nm.layer = -1; nm.ltype = LAYER_TYPE_YOLO; S.push_back(nm); }
    for (Step &st : S) st.lane = (ex->side_lo >= 0 && st.layer >= ex->side_lo && st.layer <= ex->side_hi) ? 1 : 0;
    // launches that read the batch input take its address from the parameter block when their kernel can (the first conv
    // of every darknet cfg: dense KxK from 3 channels): the captured graph is then independent of the input buffer
    ex->indirect = true;
    for (const Step &st : S)
        if (st.in_is_input && !(st.kind == S_FRONT || (st.kind == S_CONV && ffgpu_conv_supports_ind(st.conv)))) ex->indirect = false;
    if (getenv("FFGPU_NO_INDIRECT") && atoi(getenv("FFGPU_NO_INDIRECT"))) ex->indirect = false;      // tests: the keyed fallback
    if (ex->indirect)
        for (Step &st : S) if (st.in_is_input) st.conv.in_ind = reinterpret_cast<const float *const *>(ex->d_prm);   // &d_prm->frames
    ex->kernel_count = (int)S.size();
    (void)nheads;
    return 0;
}

static int repack(ffgpu_exec *ex, hipStream_t s)
{
    for (const Step &st : ex->steps) {
        if (st.kind == S_IRB && ffgpu_irb_pack(st.irb, const_cast<float *>(st.irb.pk), s)) return -1;
        if (st.kind == S_DWPW && ffgpu_dwpw_pack(st.conv, st.conv2, const_cast<float *>(st.conv2.wpack), s)) return -1;
        if (st.kind == S_CONV && st.conv.wpack && ffgpu_pw_pack(st.conv, const_cast<float *>(st.conv.wpack), s)) return -1;
    }
    return 0;
}

// -------------------------------------------------------------------------- running
static int issue_step(ffgpu_exec *ex, const Step &st, const float *d_frames, hipStream_t s)
{
#ifdef FFGPU_DIAG
    // tuning only (tools/ablate_layers.py): FFGPU_DBG_SKIP="lo:hi" drops the launches of layers lo..hi -- wrong results,
    // but the change in frames/s is what that stretch of the net costs with several batches in flight
    // (FFGPU_DBG_KEEP="lo:hi" is the complement: only that stretch runs -- tools/saturate_layers.py)
    if (getenv("FFGPU_DBG_SKIP") || getenv("FFGPU_DBG_KEEP")) {
        static bool warned = false;
        if (!warned) { warned = true; fprintf(stderr, "libffcnn_hip: FFGPU_DBG_SKIP / FFGPU_DBG_KEEP set -- launches are being dropped, RESULTS ARE WRONG (tuning only)\n"); }
    }
    if (const char *sk = getenv("FFGPU_DBG_SKIP")) {
        int lo = -1, hi = -1;
        if (sscanf(sk, "%d:%d", &lo, &hi) == 2 && st.layer >= lo && st.layer <= hi && st.kind != S_NMS) return 0;
    }
    if (const char *sk = getenv("FFGPU_DBG_KEEP")) {
        int lo = -1, hi = -1;
        if (sscanf(sk, "%d:%d", &lo, &hi) == 2 && (st.layer < lo || st.layer > hi) && st.kind != S_NMS) return 0;
    }
#endif
    switch (st.kind) {
    case S_CLEAR:
        return ffgpu_launch_clear(ex->d_ncand, ex->N, ex->d_ringctr, s);
    case S_CONV: {
        ConvDesc d = st.conv;
        if (st.in_is_input && !d.in_ind) d.in = d_frames;
        return ffgpu_launch_conv(d, FFGPU_K_AUTO, s); }
    case S_POOL:
        if (st.fs2[0]) {
            float *const outs[3] = { st.out, st.out2[0], st.out2[1] };
            const int fss[3] = { st.fs, st.fs2[0], st.fs2[1] };
            return ffgpu_launch_spp(st.a, outs, fss, st.fs2[1] ? 3 : 2, (long)ex->N * st.c, st.w, st.h, s);
        }
        return ffgpu_launch_pool(st.in_is_input ? d_frames : st.a, st.out, ex->N, st.c, st.w, st.h, st.fs, st.stride, st.flag, s);
    case S_UPSAMPLE:
        return ffgpu_launch_upsample(st.in_is_input ? d_frames : st.a, st.out, (long)ex->N * st.c, st.w, st.h, st.stride, s);
    case S_ADD:
        return ffgpu_launch_add_act(st.in_is_input ? d_frames : st.a, st.b, st.out, st.n, st.flag, s);
    case S_COPY:
        return ffgpu_launch_copy(st.a, st.out, st.n, s);
    case S_TOCNHW: {
        // frame-major -> CNHW: a 1x1 "identity" regrouping expressed as strided plane copies
        const long plane = (long)st.w * st.h;
        for (int c = 0; c < st.c; c++)
            FFGPU_CHECK(hipMemcpy2DAsync(st.out + (long)c * ex->N * plane, plane * sizeof(float),
                                         d_frames + (long)c * plane, (size_t)st.c * plane * sizeof(float),
                                         plane * sizeof(float), ex->N, hipMemcpyDeviceToDevice, s));
        return 0; }
    case S_IRB:
        return ffgpu_launch_irb(st.irb, s);
    case S_DWPW:
        return ffgpu_launch_dwpw(st.conv, st.conv2, st.conv2.wpack, s);
    case S_FRONT: {
        ConvDesc d = st.conv;
        if (st.in_is_input && !d.in_ind) d.in = d_frames;
        return ffgpu_launch_front(d, st.irb, s, ex->u8_mode); }
    case S_YOLO:
        return ffgpu_launch_yolo(st.head, ex->N, ex->in_w, ex->in_h, ex->d_cand, ex->d_cand_key, ex->d_ncand, ex->cand_cap,
                                 st.flag ? ex->d_ringctr : nullptr, s);          // forwards are counted whether or not a ring is attached
    case S_NMS:
        return ffgpu_launch_nms(ex->d_cand, ex->d_cand_key, ex->d_ncand, ex->cand_cap, ex->d_full, ex->d_nms_scratch,
                                ex->d_dets, ex->h_dets_dev, ex->d_ringctr, ex->N, 0.5f, 1, ex->d_prm, s);
    }
    return -1;
}

static int issue_all(ffgpu_exec *ex, const float *d_frames, hipStream_t s)
{
    bool forked = false, joined = true;
    for (const Step &st : ex->steps) {
        if (st.lane == 1) {
            // (the branch's stream exists only in plans that have one: every stream takes a share of the device's four hardware queues)
            if (!ex->side_stream) FFGPU_CHECK(hipStreamCreateWithFlags(&ex->side_stream, hipStreamNonBlocking));
            if (!forked) {                                   // fork: the side branch starts where the main one is now
                FFGPU_CHECK(hipEventRecord(ex->ev_fork, s));
                FFGPU_CHECK(hipStreamWaitEvent(ex->side_stream, ex->ev_fork, 0));
                forked = true; joined = false;
            }
            if (issue_step(ex, st, d_frames, ex->side_stream) != 0) return -1;
            continue;
        }
        if (st.kind == S_NMS && !joined) {                   // join before the candidates are consumed
            FFGPU_CHECK(hipEventRecord(ex->ev_join, ex->side_stream));
            FFGPU_CHECK(hipStreamWaitEvent(s, ex->ev_join, 0));
            joined = true;
        }
        if (issue_step(ex, st, d_frames, s) != 0) return -1;
    }
    if (!joined) {
        FFGPU_CHECK(hipEventRecord(ex->ev_join, ex->side_stream));
        FFGPU_CHECK(hipStreamWaitEvent(s, ex->ev_join, 0));
    }
    return 0;
}

// FFGPU_SPLIT2: the two halves of the batch are two independent chains of the same 40-odd launches; issued on two
// streams they become two parallel branches of one graph.  Most of the launches are bound by a wave's serial chain and
// a few microseconds of launch / first-load latency, not by a pipe: the second chain fills those gaps.
static int issue_all(ffgpu_exec *ex, const float *d_frames, hipStream_t s);
static int issue_split(ffgpu_exec *ex, const float *d_frames, hipStream_t s)
{
    const size_t part = (size_t)ex->child[0]->N * ex->in_c * ex->in_h * ex->in_w;
    FFGPU_CHECK(hipEventRecord(ex->ev_fork, s));
    for (int c = 1; c < ex->nchild; c++) FFGPU_CHECK(hipStreamWaitEvent(ex->part_stream[c], ex->ev_fork, 0));
    for (int c = 0; c < ex->nchild; c++)
        if (issue_all(ex->child[c], d_frames + c * part, c ? ex->part_stream[c] : s) != 0) return -1;
    for (int c = 1; c < ex->nchild; c++) {
        FFGPU_CHECK(hipEventRecord(ex->part_ev[c], ex->part_stream[c]));
        FFGPU_CHECK(hipStreamWaitEvent(s, ex->part_ev[c], 0));
    }
    return 0;
}

// the parameter block of this forward, written in stream order in front of the launches that read it; skipped when the
// block already holds (or, on the same stream, will hold by then) the same values
static int push_params(ffgpu_exec *ex, const float *d_frames, hipStream_t s)
{
    if (ex->child[0]) {
        const size_t part = (size_t)ex->child[0]->N * ex->in_c * ex->in_h * ex->in_w;
        for (int c = 0; c < ex->nchild; c++) {
            ffgpu_exec *ch = ex->child[c];
            ch->s1 = ex->s1; ch->s2 = ex->s2; ch->bbox_max = ex->bbox_max; ch->last_stream = s;
            ch->u8_mode = ex->u8_mode; ch->bgr = ex->bgr ? ex->bgr + (long)c * ch->N * ex->bgr_frame : nullptr; ch->bgr_frame = ex->bgr_frame; ch->bgr_pitch = ex->bgr_pitch;
            memcpy(ch->bgr_mean, ex->bgr_mean, sizeof ch->bgr_mean); memcpy(ch->bgr_norm, ex->bgr_norm, sizeof ch->bgr_norm);
            if (push_params(ch, ex->u8_mode ? nullptr : d_frames + c * part, s)) return -1;
        }
        return 0;
    }
    ex->last_frames = d_frames;
    ExecParams v;
    memset(&v, 0, sizeof v);
    v.frames = d_frames; v.s1 = ex->s1; v.s2 = ex->s2;
    if (ex->u8_mode) {
        v.bgr = ex->bgr; v.bgr_frame = ex->bgr_frame; v.bgr_pitch = ex->bgr_pitch;
        for (int k = 0; k < 3; k++) { v.mean[k] = ex->bgr_mean[k]; v.norm[k] = ex->bgr_norm[k]; }
    }
    v.bbox_max = ex->bbox_max;
    v.ring = ex->ring; v.ring_slots = ex->ring_slots; v.ring_stride = ex->ring_stride ? ex->ring_stride : ex->N;
    if (ex->prm_valid && ex->prm_stream == s && memcmp(&v, &ex->prm_sent, sizeof v) == 0) return 0;
    if (ffgpu_launch_set_params(ex->d_prm, v, s)) return -1;
    ex->prm_sent = v; ex->prm_stream = s; ex->prm_valid = true;
    return 0;
}

static bool graph_pointer_free(const ffgpu_exec *ex)
{
    if (!ex->child[0]) return ex->indirect;
    for (int c = 0; c < ex->nchild; c++) if (!ex->child[c]->indirect) return false;
    return true;
}

static int capture(ffgpu_exec *ex, const float *d_frames, hipGraphExec_t *out)
{
    hipGraph_t graph = nullptr;
    hipGraphExec_t gexec = nullptr;
    // capture on the executor's own stream (a user stream may be the legacy
    // null stream, which cannot be captured); replay goes to the caller's stream
    hipStream_t cs = ex->own_stream;
    FFGPU_CHECK(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
    const int rc = ex->child[0] ? issue_split(ex, d_frames, cs) : issue_all(ex, d_frames, cs);
    hipError_t e = hipStreamEndCapture(cs, &graph);
    if (rc != 0) { if (graph) (void)hipGraphDestroy(graph); return -1; }
    if (e != hipSuccess) { ffgpu_set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return -1; }
    e = hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) { ffgpu_set_error("hipGraphInstantiate: %s", hipGetErrorString(e)); return -1; }
    ex->captures++;
    *out = gexec;
    return 0;
}

static void drop_graphs(ffgpu_exec *ex)                       // caller has synchronised the streams the graphs ran on
{
    if (ex->graph1) { (void)hipGraphExecDestroy(ex->graph1); ex->graph1 = nullptr; }
    if (ex->graph_u8) { (void)hipGraphExecDestroy(ex->graph_u8); ex->graph_u8 = nullptr; }
    for (auto &g : ex->graphs) (void)hipGraphExecDestroy(g.g);
    ex->graphs.clear();
}

// The launch list is captured into its graph when the executor is created (the input pointer travels through the parameter
// block, so no buffer has to be known): kernel choices, tile splits and packed-constant layouts -- which the FFGPU_* tuning
// switches influence -- are thereby frozen together with the plan; a switch changed between ffgpu_exec_create and the first
// forward cannot make a launch disagree with the constants packed for it.  (Eager FFGPU_NO_GRAPH executors re-read the
// switches per forward: they are a debugging mode.)
static int capture_at_create(ffgpu_exec *ex)
{
    if ((ex->flags & FFGPU_NO_GRAPH) || !graph_pointer_free(ex)) return 0;
    return capture(ex, nullptr, &ex->graph1);
}

static int forward_on(ffgpu_exec *ex, const float *d_frames, hipStream_t s)
{
    ex->last_stream = s;
    if (push_params(ex, d_frames, s)) return -1;
    if (ex->flags & FFGPU_NO_GRAPH) return ex->child[0] ? issue_split(ex, d_frames, s) : issue_all(ex, d_frames, s);
    if (graph_pointer_free(ex)) {                            // the usual case: one graph, whatever the input buffer / scale
        hipGraphExec_t &g1 = ex->u8_mode ? ex->graph_u8 : ex->graph1;      // (+ one more when u8 frames go straight into the first kernel)
        if (!g1 && capture(ex, d_frames, &g1)) return -1;
        FFGPU_CHECK(hipGraphLaunch(g1, s));
        return 0;
    }
    // fallback (the first layer's kernel cannot read through the parameter block): graphs keyed by the input pointer
    ffgpu_exec::Keyed *hit = nullptr;
    for (auto &g : ex->graphs) if (g.in == d_frames) hit = &g;
    if (!hit) {
        if (ex->graphs.size() >= 8) {                        // evict the least recently used one -- after its last launch ended
            size_t lru = 0;
            for (size_t i = 1; i < ex->graphs.size(); i++) if (ex->graphs[i].stamp < ex->graphs[lru].stamp) lru = i;
            FFGPU_CHECK(hipDeviceSynchronize());             // (it may have run on another stream than this call's)
            (void)hipGraphExecDestroy(ex->graphs[lru].g);
            ex->graphs.erase(ex->graphs.begin() + lru);
        }
        hipGraphExec_t g = nullptr;
        if (capture(ex, d_frames, &g)) return -1;
        ex->graphs.push_back({ d_frames, g, 0 });
        hit = &ex->graphs.back();
    }
    hit->stamp = ++ex->graph_stamp;
    FFGPU_CHECK(hipGraphLaunch(hit->g, s));
    return 0;
}

// -------------------------------------------------------------------------- C-ABI: executor
// an executor whose NET was freed first (net_free orphans it) keeps its device buffers until ffgpu_exec_destroy, but its
// layer table and weights are gone: every entry point that would touch them fails instead
static bool alive(const ffgpu_exec *ex, const char *what)
{
    if (!ex) { ffgpu_set_error("%s: NULL executor", what); return false; }
    if (!ex->dev) { ffgpu_set_error("%s: the NET of this executor has been freed (destroy executors before net_free)", what); return false; }
    return true;
}

static ffgpu_netdev *netdev_of(NET *net)
{
    if (!net) { ffgpu_set_error("NULL net"); return nullptr; }
    ffcnn_ext *ext = ffcnn_ext_of(net);
    if (!ext || !ext->dev) { ffgpu_set_error("net was not created by this library's net_load"); return nullptr; }
    return (ffgpu_netdev *)ext->dev;
}

static ffgpu_exec *exec_create_on(ffgpu_netdev *dev, NET *net, int batch, int flags);
extern "C" ffgpu_exec *ffgpu_exec_create(NET *net, int batch, int flags)
{
    ffgpu_netdev *dev = netdev_of(net);
    if (!dev) return nullptr;
    return exec_create_on(dev, net, batch, flags);
}

// (an executor lives on ONE device: that of the ffgpu_netdev -- the device copy of the weights -- it is planned on; a
//  multi-GPU node holds one netdev + executor per device, ffgpu_node.inc)
static ffgpu_exec *exec_create_on(ffgpu_netdev *dev, NET *net, int batch, int flags)
{
    if (batch < 1) { ffgpu_set_error("batch must be >= 1"); return nullptr; }
    if (hipSetDevice(dev->device) != hipSuccess) { ffgpu_set_error("hipSetDevice failed"); return nullptr; }
    const char *env = getenv("FFCNN_COMPAT_V6");
    if (env && atoi(env)) flags |= FFGPU_COMPAT_V6;
    env = getenv("FFGPU_NO_GRAPH");
    if (env && atoi(env)) flags |= FFGPU_NO_GRAPH;
    env = getenv("FFGPU_BF16_PW");
    if (env && atoi(env)) flags |= FFGPU_BF16_PW;
    env = getenv("FFGPU_NO_FUSE");
    if (env && atoi(env)) flags |= FFGPU_NO_FUSE;
    const bool split = (flags & FFGPU_SPLIT2) && batch >= 2 && batch % 2 == 0 && !(flags & FFGPU_KEEP_ALL);
    const bool as_child = (flags & FFGPU_INTERNAL_CHILD) != 0;
    flags &= ~(FFGPU_SPLIT2 | FFGPU_INTERNAL_CHILD);
    ffgpu_exec *ex = new ffgpu_exec();
    ex->net = net; ex->dev = dev; ex->device = dev->device; ex->N = batch; ex->flags = flags; ex->is_child = as_child;
    ex->in_c = net->layer_list[0].c; ex->in_h = net->layer_list[0].h; ex->in_w = net->layer_list[0].w;
    // candidate slots per frame: one per anchor of every head cell, so the decode can never overflow; the reference's own
    // cap (bbox_max candidates in emission order, ffcnn.c:243,463) is applied by k_nms
    long slots = 0;
    for (int i = 0; i < net->layer_num; i++)
        if (net->layer_list[i].type == LAYER_TYPE_YOLO) slots += 3L * net->layer_list[i].w * net->layer_list[i].h;
    if (slots > (1L << 24) || slots * batch > (1L << 28)) { ffgpu_set_error("executor: %ld candidate slots per frame x %d frames is too many", slots, batch); delete ex; return nullptr; }
    ex->cand_cap = (int)std::max(slots, 1L);
    ex->bbox_max = std::max(net->bbox_max, 1);
    int cap_p2 = 1;
    while (cap_p2 < ex->cand_cap) cap_p2 <<= 1;
    const size_t ncs = (size_t)ex->cand_cap * (size_t)batch;
    bool ok = hipStreamCreateWithFlags(&ex->own_stream, hipStreamNonBlocking) == hipSuccess
           && hipEventCreateWithFlags(&ex->ev_fork, hipEventDisableTiming) == hipSuccess
           && hipEventCreateWithFlags(&ex->ev_join, hipEventDisableTiming) == hipSuccess
           && hipMalloc(&ex->d_cand, sizeof(BBOX) * ncs) == hipSuccess
           && hipMalloc(&ex->d_cand_key, sizeof(int) * ncs) == hipSuccess
           && hipMalloc(&ex->d_full, sizeof(BBOX) * ncs) == hipSuccess
           && hipMalloc(&ex->d_prm, sizeof(ExecParams)) == hipSuccess && hipMemset(ex->d_prm, 0, sizeof(ExecParams)) == hipSuccess
           && (ffgpu_nms_in_lds(cap_p2) || hipMalloc(&ex->d_nms_scratch, (size_t)13 * cap_p2 * batch) == hipSuccess)
           && hipMalloc(&ex->d_ncand, sizeof(int) * (size_t)batch) == hipSuccess && hipMemset(ex->d_ncand, 0, sizeof(int) * (size_t)batch) == hipSuccess
           && hipMalloc(&ex->d_dets, sizeof(ffgpu_frame_dets) * (size_t)batch) == hipSuccess
           && hipMalloc(&ex->d_ringctr, sizeof(int)) == hipSuccess && hipMemset(ex->d_ringctr, 0, sizeof(int)) == hipSuccess
           && hipMemset(ex->d_dets, 0, sizeof(ffgpu_frame_dets) * (size_t)batch) == hipSuccess
           // hipMemset of device memory is asynchronous on the NULL stream, and every stream this executor works on is
           // non-blocking (not ordered with it): wait here, or a late memset may clear counters a forward has begun to use
           && hipStreamSynchronize(nullptr) == hipSuccess;
    if (ok && (flags & FFGPU_HOST_DETS)) {
        ok = hipHostMalloc(&ex->h_dets, sizeof(ffgpu_frame_dets) * (size_t)batch, hipHostMallocMapped) == hipSuccess
          && hipHostGetDevicePointer((void **)&ex->h_dets_dev, ex->h_dets, 0) == hipSuccess;
        if (ok) memset(ex->h_dets, 0, sizeof(ffgpu_frame_dets) * (size_t)batch);
    }
    if (!ok) { ffgpu_set_error("executor buffers: %s", hipGetErrorString(hipGetLastError())); ffgpu_exec_destroy(ex); return nullptr; }
    ex->last_stream = ex->own_stream;
    if (split) {
        // the parent owns the records (and their mirror / ring); each part plans its own arena and writes its slice
        int K = 2;
        const char *ek = getenv("FFGPU_SPLIT_PARTS");            // tuning: 2 (default), 4 or 8 parallel chains
        if (ek && (atoi(ek) == 4 || atoi(ek) == 8) && batch % atoi(ek) == 0) K = atoi(ek);
        for (int c = 0; c < K; c++) {
            ffgpu_exec *ch = exec_create_on(dev, net, batch / K, (flags & ~FFGPU_HOST_DETS) | FFGPU_INTERNAL_CHILD);
            if (!ch) { ffgpu_exec_destroy(ex); return nullptr; }
            (void)hipFree(ch->d_dets); (void)hipFree(ch->d_full);
            ch->d_dets = ex->d_dets + (size_t)c * (batch / K);
            ch->d_full = ex->d_full + (size_t)c * (batch / K) * ex->cand_cap;
            ch->h_dets_dev = ex->h_dets_dev ? ex->h_dets_dev + (size_t)c * (batch / K) : nullptr;
            ex->child[c] = ch; ex->nchild = c + 1;
            ex->kernel_count += ch->kernel_count;
            ex->arena_floats += ch->arena_floats;
            if (c && (hipStreamCreateWithFlags(&ex->part_stream[c], hipStreamNonBlocking) != hipSuccess ||
                      hipEventCreateWithFlags(&ex->part_ev[c], hipEventDisableTiming) != hipSuccess)) {
                ffgpu_set_error("split executor: stream / event creation failed"); ffgpu_exec_destroy(ex); return nullptr; }
        }
        { std::lock_guard<std::mutex> lk(dev->mu); dev->execs.push_back(ex); }
        if (capture_at_create(ex)) { ffgpu_exec_destroy(ex); return nullptr; }
        return ex;
    }
    if (plan(ex) != 0 || repack(ex, ex->own_stream) != 0 || hipStreamSynchronize(ex->own_stream) != hipSuccess) { ffgpu_exec_destroy(ex); return nullptr; }
    { std::lock_guard<std::mutex> lk(dev->mu); dev->execs.push_back(ex); }
    if (!as_child && capture_at_create(ex)) { ffgpu_exec_destroy(ex); return nullptr; }
    return ex;
}

extern "C" void ffgpu_exec_destroy(ffgpu_exec *ex)
{
    if (!ex) return;
    int cur_dev = ex->device;
    (void)hipGetDevice(&cur_dev);
    if (cur_dev != ex->device) (void)hipSetDevice(ex->device);      // (a node's executors live on other devices than the caller's)
    if (ex->last_stream) (void)hipStreamSynchronize(ex->last_stream);
    // a graph with a forked branch (the head branch) keeps ~its arena's worth of device memory if its exec is destroyed after
    // a sync of the launch stream alone (ROCm 7.0; tools/leak_check.py: +27 MB per create / forward / destroy cycle) --
    // a device-wide sync first lets the runtime retire the branch's internal streams
    if (!ex->is_child) (void)hipDeviceSynchronize();
    for (int c = 0; c < ffgpu_exec::MAXPART; c++) {
        if (ex->child[c]) { ffgpu_exec_destroy(ex->child[c]); ex->child[c] = nullptr; }
        if (ex->part_stream[c]) (void)hipStreamDestroy(ex->part_stream[c]);
        if (ex->part_ev[c]) (void)hipEventDestroy(ex->part_ev[c]);
    }
    if (ex->is_child) { ex->d_dets = nullptr; ex->d_full = nullptr; }      // slices of the parent's buffers
    if (ex->dev) {
        std::lock_guard<std::mutex> lk(ex->dev->mu);
        ex->dev->execs.erase(std::remove(ex->dev->execs.begin(), ex->dev->execs.end(), ex), ex->dev->execs.end());
    }
    drop_graphs(ex);
    for (const Step &st : ex->steps) if (st.kind == S_TOCNHW) (void)hipFree(st.out);
    (void)hipFree(ex->arena); (void)hipFree(ex->d_input); (void)hipFree(ex->d_pack); (void)hipFree(ex->d_cand);
    (void)hipFree(ex->d_cand_key); (void)hipFree(ex->d_ncand); (void)hipFree(ex->d_dets);
    (void)hipFree(ex->d_full); (void)hipFree(ex->d_prm); (void)hipFree(ex->d_nms_scratch);
    if (ex->h_dets) (void)hipHostFree(ex->h_dets);
    if (ex->h_stage) (void)hipHostFree(ex->h_stage);
    (void)hipFree(ex->d_ringctr);
    if (ex->own_stream) (void)hipStreamDestroy(ex->own_stream);
    if (ex->side_stream) (void)hipStreamDestroy(ex->side_stream);
    if (ex->ev_fork) (void)hipEventDestroy(ex->ev_fork);
    if (ex->ev_join) (void)hipEventDestroy(ex->ev_join);
    if (cur_dev != ex->device) (void)hipSetDevice(cur_dev);
    delete ex;
}

extern "C" int ffgpu_exec_batch(const ffgpu_exec *ex) { return ex ? ex->N : 0; }
extern "C" size_t ffgpu_exec_arena_bytes(const ffgpu_exec *ex) { return ex ? ex->arena_floats * sizeof(float) : 0; }
extern "C" int ffgpu_exec_kernel_count(const ffgpu_exec *ex) { return ex ? ex->kernel_count : 0; }

// What one forward of this plan MUST move and compute: per launch the tensors it reads and writes once each (input,
// output, residual, filter rows; nothing for the tensors a fused launch keeps on chip), summed over the launch list; and
// 2 x multiply-adds of every conv layer of the net (fused or not).  bench.py prices the measured time per batch against
// these two numbers (HBM peak, fp32 matrix peak).
// what one step of the plan must move between HBM and the chip, as a model: its input tensor(s) + output tensor + filter rows, each once
static double step_model_bytes(const ffgpu_exec *ex, const Step &st)
{
    const double N = ex->N;
    auto rows = [](const ConvDesc &d) { return (double)d.oc * (conv_k4(d) + 4); };
    switch (st.kind) {
    case S_CONV: { const ConvDesc &d = st.conv;
        return 4.0 * (N * d.ic * d.ih * d.iw + N * d.oc * d.oh * d.ow * (d.residual ? 2 : 1) + rows(d)); }
    case S_IRB: { const IrbDesc &d = st.irb;
        return 4.0 * (N * d.ic * d.H * d.W + N * d.oc * d.OH * d.OW * (d.residual ? 2 : 1) + (double)d.ec * (d.ic + 4 + 16) + (double)d.oc * (d.ec + 4)); }
    case S_DWPW: { const ConvDesc &a = st.conv, &b = st.conv2;
        return 4.0 * (N * a.ic * a.ih * a.iw + N * b.oc * b.oh * b.ow + rows(a) + rows(b)); }
    case S_FRONT: { const ConvDesc &c = st.conv; const IrbDesc &d = st.irb;
        return 4.0 * (N * c.ic * c.ih * c.iw + N * d.oc * d.OH * d.OW + rows(c)); }
    case S_POOL: { const int np = 1 + (st.fs2[0] != 0) + (st.fs2[1] != 0);
        return 4.0 * N * st.c * ((double)st.w * st.h + (double)np * (st.w / st.stride) * (st.h / st.stride)); }
    case S_UPSAMPLE: return 4.0 * N * st.c * (double)st.w * st.h * (1 + st.stride * st.stride);
    case S_ADD: return 4.0 * 3 * st.n;
    case S_COPY: return 4.0 * 2 * st.n;
    case S_TOCNHW: return 4.0 * 2 * N * st.c * (double)st.w * st.h;
    case S_YOLO: return 4.0 * N * 3 * (5 + st.head.classes) * (double)st.head.w * st.head.h;
    default: return 0.0;
    }
}

/* the same model step by step (tools/net_traffic.py sets the counters of a rocprofv3 --pmc pass beside it): layer_of[i] / hbm_bytes[i] of step i */
extern "C" int ffgpu_exec_step_model(const ffgpu_exec *ex, int *layer_of, double *hbm_bytes, int cap)
{
    if (!ex || !ex->net || !layer_of || !hbm_bytes) { ffgpu_set_error("step_model: bad argument"); return -1; }
    if (ex->child[0]) { ffgpu_set_error("step_model: not available on a split executor"); return -1; }
    const int n = (int)std::min<size_t>(ex->steps.size(), (size_t)(cap > 0 ? cap : 0));
    for (int i = 0; i < n; i++) { layer_of[i] = ex->steps[i].layer; hbm_bytes[i] = step_model_bytes(ex, ex->steps[i]); }
    return n;
}

extern "C" int ffgpu_exec_work_model(const ffgpu_exec *ex, double *hbm_bytes, double *flops)
{
    if (!ex || !ex->net) { ffgpu_set_error("work_model: bad executor"); return -1; }
    double by = 0, fl = 0;
    if (ex->child[0]) {
        for (int c = 0; c < ex->nchild; c++) {
            double b1 = 0, f1 = 0;
            if (ffgpu_exec_work_model(ex->child[c], &b1, &f1)) return -1;
            by += b1; fl += f1;
        }
    } else {
        const double N = ex->N;
        for (const Step &st : ex->steps) by += step_model_bytes(ex, st);
        const LAYER *ll = ex->net->layer_list;
        for (int i = 0; i < ex->net->layer_num; i++)
            if (ll[i].type == LAYER_TYPE_CONV)
                fl += 2.0 * N * ll[i].fs * ll[i].fs * (ll[i].c / ll[i].groups) * (double)ll[i + 1].c * ll[i + 1].w * ll[i + 1].h;
    }
    if (hbm_bytes) *hbm_bytes = by;
    if (flops) *flops = fl;
    return 0;
}

extern "C" int ffgpu_exec_set_scale(ffgpu_exec *ex, int s1, int s2)
{
    if (!ex || s2 == 0) { ffgpu_set_error("set_scale: bad arguments"); return -1; }
    ex->s1 = s1; ex->s2 = s2;
    return 0;
}

extern "C" int ffgpu_exec_forward_dev(ffgpu_exec *ex, const float *d_frames, void *stream)
{
    if (!alive(ex, "forward_dev")) return -1;
    if (!d_frames) { ffgpu_set_error("forward_dev: NULL argument"); return -1; }
    return forward_on(ex, d_frames, stream ? (hipStream_t)stream : ex->own_stream);
}

static int ensure_input(ffgpu_exec *ex)
{
    if (ex->d_input) return 0;
    FFGPU_CHECK(hipMalloc(&ex->d_input, sizeof(float) * (size_t)ex->N * ex->in_c * ex->in_h * ex->in_w));
    return 0;
}

extern "C" int ffgpu_exec_forward_host(ffgpu_exec *ex, const float *h_frames)
{
    if (!alive(ex, "forward_host")) return -1;
    if (!h_frames) { ffgpu_set_error("forward_host: NULL argument"); return -1; }
    if (ensure_input(ex)) return -1;
    const size_t bytes = sizeof(float) * (size_t)ex->N * ex->in_c * ex->in_h * ex->in_w;
    if (pinned_ours(h_frames, bytes)) {                          // (a NET's input tensor: one DMA, in stream order with the forward)
        FFGPU_CHECK(hipMemcpyAsync(ex->d_input, h_frames, bytes, hipMemcpyHostToDevice, ex->own_stream));
    } else {
        // caller memory: through the executor's own page-locked staging buffer (allocated on the first such call)
        if (!ex->h_stage) FFGPU_CHECK(hipHostMalloc((void **)&ex->h_stage, bytes, hipHostMallocDefault));
        memcpy(ex->h_stage, h_frames, bytes);                    // (the previous forward_host of this executor ended with a stream sync)
        FFGPU_CHECK(hipMemcpyAsync(ex->d_input, ex->h_stage, bytes, hipMemcpyHostToDevice, ex->own_stream));
    }
    if (forward_on(ex, ex->d_input, ex->own_stream)) return -1;
    FFGPU_CHECK(hipStreamSynchronize(ex->own_stream));
    return 0;
}

// the plan's first launch is k_front reading through the parameter block (every leaf of a split executor alike)
static bool front_reads_u8(const ffgpu_exec *ex)
{
    if (ex->child[0]) {
        for (int c = 0; c < ex->nchild; c++) if (!front_reads_u8(ex->child[c])) return false;
        return true;
    }
    for (const Step &st : ex->steps)
        if (st.in_is_input) return st.kind == S_FRONT && st.conv.in_ind != nullptr && ex->indirect;
    return false;
}

extern "C" int ffgpu_exec_forward_bgr_dev(ffgpu_exec *ex, const unsigned char *d_bgr, int w, int h,
                                          const float mean[3], const float norm[3], void *stream)
{
    if (!alive(ex, "forward_bgr_dev")) return -1;
    if (!d_bgr || w <= 0 || h <= 0 || ex->in_c != 3) { ffgpu_set_error("forward_bgr_dev: bad arguments"); return -1; }
    hipStream_t s = stream ? (hipStream_t)stream : ex->own_stream;
    const int W = ex->in_w, H = ex->in_h;
    int sw, sh, s1, s2;                                          // ffcnn.c:267-273
    if ((long)w * H > (long)h * W) { sw = W; sh = (int)((long)sw * h / w); s1 = w; s2 = sw; }
    else                           { sh = H; sw = (int)((long)sh * w / h); s1 = h; s2 = sh; }
    ex->s1 = s1; ex->s2 = s2;
    const bool no_u8_front = getenv("FFGPU_NO_U8_FRONT") != nullptr;             // (tuning / tests: always the two-kernel path)
    if (w == W && h == H && (reinterpret_cast<uintptr_t>(d_bgr) & 3) == 0 && !no_u8_front && front_reads_u8(ex)) {
        // no resize (net_input copies pixel for pixel): the first kernel converts the bytes itself -- no fp32 batch in between
        const int pitch = (w * 3 + 3) & ~3;
        ex->bgr = d_bgr; ex->bgr_pitch = pitch; ex->bgr_frame = (long)pitch * h;
        for (int k = 0; k < 3; k++) { ex->bgr_mean[k] = mean[k]; ex->bgr_norm[k] = norm[k]; }
        ex->u8_mode = true;
        const int rc = forward_on(ex, nullptr, s);
        ex->u8_mode = false;
        return rc;
    }
    if (ensure_input(ex)) return -1;
    if (ffgpu_launch_input_bgr(d_bgr, ex->d_input, ex->N, w, h, W, H, sw, sh, s1, s2, mean, norm, s)) return -1;
    return forward_on(ex, ex->d_input, s);
}

extern "C" int ffgpu_exec_dets_dev(ffgpu_exec *ex, void **dev_ptr, size_t *bytes)
{
    if (!ex) { ffgpu_set_error("NULL executor"); return -1; }
    if (dev_ptr) *dev_ptr = ex->d_dets;
    if (bytes) *bytes = sizeof(ffgpu_frame_dets) * (size_t)ex->N;
    return 0;
}

extern "C" int ffgpu_exec_set_ring_strided(ffgpu_exec *ex, void *dev_ring, int slots, int slot_records)
{
    if (!ex || (dev_ring && (slots < 1 || slot_records < ex->N))) { ffgpu_set_error("set_ring: bad arguments"); return -1; }
    FFGPU_CHECK(hipStreamSynchronize(ex->last_stream));          // (the ring travels in the parameter block: the graph stays)
    ex->ring = (ffgpu_frame_dets *)dev_ring; ex->ring_slots = dev_ring ? slots : 0; ex->ring_stride = slot_records;
    FFGPU_CHECK(hipMemset(ex->d_ringctr, 0, sizeof(int)));
    for (int c = 0; c < ex->nchild; c++) {                      // each part writes its slice of every slot
        ffgpu_exec *ch = ex->child[c];
        ch->ring = dev_ring ? ex->ring + (size_t)c * ch->N : nullptr; ch->ring_slots = ex->ring_slots; ch->ring_stride = slot_records;
        FFGPU_CHECK(hipMemset(ch->d_ringctr, 0, sizeof(int)));
    }
    FFGPU_CHECK(hipStreamSynchronize(nullptr));                  // (the memsets above run on the NULL stream; forwards do not)
    return 0;
}

extern "C" int ffgpu_exec_set_ring(ffgpu_exec *ex, void *dev_ring, int slots)
{
    if (!ex) { ffgpu_set_error("set_ring: NULL executor"); return -1; }
    return ffgpu_exec_set_ring_strided(ex, dev_ring, slots, ex->N);
}

extern "C" const ffgpu_frame_dets *ffgpu_exec_dets_host(ffgpu_exec *ex)
{
    if (!ex) { ffgpu_set_error("NULL executor"); return nullptr; }
    if (!ex->h_dets) ffgpu_set_error("executor was not created with FFGPU_HOST_DETS");
    return ex->h_dets;
}

extern "C" int ffgpu_exec_read_dets(ffgpu_exec *ex, ffgpu_frame_dets *host_out, int max_frames)
{
    if (!ex || !host_out) { ffgpu_set_error("read_dets: NULL argument"); return -1; }
    const int n = std::max(0, std::min(max_frames, ex->N));
    FFGPU_CHECK(hipStreamSynchronize(ex->last_stream));
    if (ex->h_dets) memcpy(host_out, ex->h_dets, sizeof(ffgpu_frame_dets) * (size_t)n);
    else { if (copy_d2h(host_out, ex->d_dets, sizeof(ffgpu_frame_dets) * (size_t)n)) return -1; }
    return n;
}

extern "C" int ffgpu_exec_cand_capacity(const ffgpu_exec *ex) { return ex ? ex->cand_cap : 0; }
extern "C" int ffgpu_exec_graph_captures(const ffgpu_exec *ex)
{
    if (!ex) return 0;
    int n = ex->captures;
    for (int c = 0; c < ex->nchild; c++) n += ex->child[c]->captures;
    return n;
}

extern "C" int ffgpu_exec_read_boxes(ffgpu_exec *ex, int frame, BBOX *host_out, int cap)
{
    if (!ex || frame < 0 || frame >= ex->N || cap < 0 || (cap > 0 && !host_out)) { ffgpu_set_error("read_boxes: bad arguments"); return -1; }
    FFGPU_CHECK(hipStreamSynchronize(ex->last_stream));
    int nfull = 0;
    if (copy_d2h(&nfull, &ex->d_dets[frame].nfull, sizeof(int))) return -1;
    const int n = std::min(nfull, cap);
    if (n > 0 && copy_d2h(host_out, ex->d_full + (size_t)frame * ex->cand_cap, sizeof(BBOX) * (size_t)n)) return -1;
    return nfull;
}

// FFGPU_KEEP_ALL executors: one 64-bit hash per materialised layer over the layer's WHOLE batch tensor (every bit of every frame), computed on the
// device behind the last forward; 0 for layers this executor does not materialise.  What the concurrency soak compares round after round
// (tests/test_gpu_round5.py): 8 bytes per layer cross the bus instead of the activations.
extern "C" int ffgpu_exec_hash_layers(ffgpu_exec *ex, unsigned long long *host_out, int cap)
{
    if (!alive(ex, "hash_layers")) return -1;
    if (ex->child[0]) { ffgpu_set_error("hash_layers: not for FFGPU_SPLIT2 executors"); return -1; }
    if (!(ex->flags & FFGPU_KEEP_ALL)) { ffgpu_set_error("hash_layers needs an FFGPU_KEEP_ALL executor"); return -1; }
    const int L = ex->net->layer_num;
    if (!host_out || cap < L) { ffgpu_set_error("hash_layers: room for %d values needed", L); return -1; }
    unsigned long long *d = nullptr;
    FFGPU_CHECK(hipMalloc(&d, sizeof(unsigned long long) * L));
    hipStream_t s = ex->last_stream;
    int rc = 0, nh = 0;
    if (hipMemsetAsync(d, 0, sizeof(unsigned long long) * L, s) != hipSuccess) rc = -1;
    for (int i = 0; i < L && !rc; i++) {
        if (ex->canon[i] < 0 || !ex->readable[i]) continue;
        const LAYER &o = ex->net->layer_list[i + 1];
        if (ffgpu_launch_hash64(tensor_ptr(ex, ex->canon[i]), (long)o.c * ex->N * o.w * o.h, d + i, s)) rc = -1;
        nh++;
    }
    if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = -1;
    if (!rc && copy_d2h(host_out, d, sizeof(unsigned long long) * L)) rc = -1;
    (void)hipFree(d);
    if (rc) { ffgpu_set_error("hash_layers: a device call failed"); return -1; }
    return nh;
}

extern "C" int ffgpu_exec_read_layer(ffgpu_exec *ex, int layer, int frame, float *host_out, size_t cap_floats)
{
    if (!alive(ex, "read_layer")) return -1;
    if (!host_out || frame < 0 || frame >= ex->N) { ffgpu_set_error("read_layer: bad arguments"); return -1; }
    if (ex->child[0]) return ffgpu_exec_read_layer(ex->child[frame / ex->child[0]->N], layer, frame % ex->child[0]->N, host_out, cap_floats);
    FFGPU_CHECK(hipStreamSynchronize(ex->last_stream));
    if (layer == -2) {                                            // candidates in reference emission order
        int cnt = 0;
        if (copy_d2h(&cnt, &ex->d_dets[frame].ncand, sizeof(int))) return -1;    // (k_nms has reset d_ncand)
        cnt = std::min(cnt, ex->cand_cap);
        if ((size_t)cnt * 6 > cap_floats) { ffgpu_set_error("read_layer: buffer too small"); return -1; }
        std::vector<BBOX> b(cnt);
        std::vector<int> k(cnt), idx(cnt);
        if (cnt) {
            if (copy_d2h(b.data(), ex->d_cand + (size_t)frame * ex->cand_cap, sizeof(BBOX) * cnt)) return -1;
            if (copy_d2h(k.data(), ex->d_cand_key + (size_t)frame * ex->cand_cap, sizeof(int) * cnt)) return -1;
        }
        for (int i = 0; i < cnt; i++) idx[i] = i;
        std::sort(idx.begin(), idx.end(), [&](int a, int c) { return k[a] < k[c]; });
        for (int i = 0; i < cnt; i++) memcpy(host_out + 6 * i, &b[idx[i]], sizeof(BBOX));
        return cnt;
    }
    if (layer == -1) {                                            // the network input as the first layer saw it (frame-major)
        const size_t fl = (size_t)ex->in_c * ex->in_h * ex->in_w;
        if (!ex->last_frames && ex->bgr) { ffgpu_set_error("read_layer: the last forward's frames were u8 images converted by the first kernel -- no fp32 input tensor exists"); return -1; }
        if (!ex->last_frames) { ffgpu_set_error("read_layer: no forward has run yet"); return -1; }
        if (fl > cap_floats) { ffgpu_set_error("read_layer: buffer too small"); return -1; }
        if (copy_d2h(host_out, ex->last_frames + (size_t)frame * fl, fl * sizeof(float))) return -1;
        return (int)fl;
    }
    if (!(ex->flags & FFGPU_KEEP_ALL)) { ffgpu_set_error("read_layer needs an FFGPU_KEEP_ALL executor"); return -1; }
    if (layer < 0 || layer >= ex->net->layer_num || ex->canon[layer] < 0 || !ex->readable[layer]) {
        ffgpu_set_error("read_layer: layer %d is not materialised by this executor (fused away or no tensor)", layer);
        return -1;
    }
    const LAYER &o = ex->net->layer_list[layer + 1];
    const size_t plane = (size_t)o.w * o.h;
    if (plane * o.c > cap_floats) { ffgpu_set_error("read_layer: buffer too small"); return -1; }
    const float *src = tensor_ptr(ex, ex->canon[layer]) + (size_t)frame * plane;
    {   // the frame's planes (one per channel, N planes apart in CNHW) gathered into the bounce buffer, then into caller memory
        const size_t row = plane * sizeof(float), per = std::max<size_t>(1, BOUNCE_BYTES / row);
        if (row > BOUNCE_BYTES) {                                  // a plane larger than the staging buffer (> 2048 x 2048 floats): contiguous, chunked plane by plane
            for (size_t c = 0; c < (size_t)o.c; c++)
                if (copy_d2h(host_out + c * plane, src + c * plane * ex->N, row)) return -1;
            return (int)(plane * o.c);
        }
        Bounce &bn = bounce_cur();
        std::lock_guard<std::mutex> lk(bn.mu);
        if (bounce_ready(bn)) return -1;
        char *const g_bounce = bn.p;
        for (size_t c0 = 0; c0 < (size_t)o.c; c0 += per) {
            const size_t nc = std::min(per, (size_t)o.c - c0);
            FFGPU_CHECK(hipMemcpy2D(g_bounce, row, src + c0 * plane * ex->N, row * ex->N, row, nc, hipMemcpyDeviceToHost));
            memcpy(host_out + c0 * plane, g_bounce, nc * row);
        }
    }
    return (int)(plane * o.c);
}

extern "C" int ffgpu_exec_profile(ffgpu_exec *ex, const float *d_frames, float us_by_kind[LAYER_TYPE_TOTOAL])
{
    if (!alive(ex, "profile")) return -1;
    if (!d_frames || !us_by_kind) { ffgpu_set_error("profile: NULL argument"); return -1; }
    if (ex->child[0]) { ffgpu_set_error("profile: not available on a split executor"); return -1; }
    hipStream_t s = ex->own_stream;
    if (push_params(ex, d_frames, s)) return -1;
    std::vector<hipEvent_t> ev(ex->steps.size() + 1);
    for (auto &e : ev) FFGPU_CHECK(hipEventCreate(&e));
    if (issue_all(ex, d_frames, s)) return -1;                    // warm
    FFGPU_CHECK(hipEventRecord(ev[0], s));
    for (size_t i = 0; i < ex->steps.size(); i++) {
        if (issue_step(ex, ex->steps[i], d_frames, s)) return -1;
        FFGPU_CHECK(hipEventRecord(ev[i + 1], s));
    }
    FFGPU_CHECK(hipStreamSynchronize(s));
    for (int k = 0; k < LAYER_TYPE_TOTOAL; k++) us_by_kind[k] = 0.f;
    for (size_t i = 0; i < ex->steps.size(); i++) {
        float ms = 0.f;
        FFGPU_CHECK(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
        const int k = ex->steps[i].ltype;
        if (k >= 0 && k < LAYER_TYPE_TOTOAL) us_by_kind[k] += ms * 1000.f;
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
    ex->last_stream = s;
    return 0;
}

extern "C" int ffgpu_exec_profile_steps(ffgpu_exec *ex, const float *d_frames, int *layer_of, float *us, int cap)
{
    if (!alive(ex, "profile_steps")) return -1;
    if (!d_frames || !layer_of || !us) { ffgpu_set_error("profile_steps: NULL argument"); return -1; }
    if (ex->child[0]) { ffgpu_set_error("profile_steps: not available on a split executor"); return -1; }
    hipStream_t s = ex->own_stream;
    if (push_params(ex, d_frames, s)) return -1;
    const int n = (int)std::min<size_t>(ex->steps.size(), (size_t)cap);
    std::vector<hipEvent_t> ev(ex->steps.size() + 1);
    for (auto &e : ev) FFGPU_CHECK(hipEventCreate(&e));
    std::vector<float> acc(ex->steps.size(), 0.f);
    const int reps = 5;
    if (issue_all(ex, d_frames, s)) return -1;
    for (int r = 0; r < reps; r++) {
        FFGPU_CHECK(hipEventRecord(ev[0], s));
        for (size_t i = 0; i < ex->steps.size(); i++) {
            if (issue_step(ex, ex->steps[i], d_frames, s)) return -1;
            FFGPU_CHECK(hipEventRecord(ev[i + 1], s));
        }
        FFGPU_CHECK(hipStreamSynchronize(s));
        for (size_t i = 0; i < ex->steps.size(); i++) {
            float ms = 0.f;
            FFGPU_CHECK(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
            acc[i] += ms * 1000.f / reps;
        }
    }
    for (int i = 0; i < n; i++) { layer_of[i] = ex->steps[i].layer; us[i] = acc[i]; }
    for (auto &e : ev) (void)hipEventDestroy(e);
    ex->last_stream = s;
    return n;
}

// -------------------------------------------------------------------------- C-ABI: net device state
// device copy of the net's folded filter rows on `device` (-1: the calling thread's current device); upload == false leaves
// it zeroed -- the other GPUs of a node receive the weights over RCCL (ffgpu_node.inc)
static ffgpu_netdev *netdev_create_on(NET *net, int device, bool upload)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        ffgpu_set_error("no HIP device visible: libffcnn_hip has no CPU fallback");
        return nullptr;
    }
    ffgpu_netdev *dev = new ffgpu_netdev();
    dev->net = net;
    if (device >= 0) { dev->device = device; if (hipSetDevice(device) != hipSuccess) { ffgpu_set_error("hipSetDevice(%d) failed", device); delete dev; return nullptr; } }
    else if (hipGetDevice(&dev->device) != hipSuccess) dev->device = 0;
    dev->weight_bytes = sizeof(float) * (size_t)std::max(net->weight_size, 1);
    if (hipMalloc(&dev->d_weights, dev->weight_bytes) != hipSuccess ||
        (upload ? (copy_h2d(dev->d_weights, net->weight_buf, sizeof(float) * (size_t)net->weight_size) ? hipErrorUnknown : hipSuccess)
                : hipMemset(dev->d_weights, 0, dev->weight_bytes)) != hipSuccess ||
        hipStreamSynchronize(nullptr) != hipSuccess) {                // (the memset is asynchronous on the NULL stream: a broadcast on
                                                                      //  another stream must not be overtaken by it)
        ffgpu_set_error("weight upload failed: %s", hipGetErrorString(hipGetLastError()));
        (void)hipFree(dev->d_weights);
        delete dev;
        return nullptr;
    }
    return dev;
}

extern "C" void *ffgpu_netdev_create(NET *net) { return netdev_create_on(net, -1, true); }

extern "C" float *ffgpu_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    memset(p, 0, bytes);
    { std::lock_guard<std::mutex> lk(g_pin_mu); g_pinned.push_back({ (const char *)p, bytes }); }
    return (float *)p;
}

extern "C" void ffgpu_host_free(float *p)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        for (size_t i = 0; i < g_pinned.size(); i++) if (g_pinned[i].p == (const char *)p) { g_pinned.erase(g_pinned.begin() + i); break; }
    }
    (void)hipHostFree(p);
}

extern "C" void ffgpu_netdev_destroy(void *p)
{
    ffgpu_netdev *dev = (ffgpu_netdev *)p;
    if (!dev) return;
    if (dev->exec1) { ffgpu_exec_destroy(dev->exec1); dev->exec1 = nullptr; }
    // executors the caller still holds outlive the net as orphans: their steps point into d_weights and the layer table,
    // so they wait for their streams here and from now on refuse to run (alive()); ffgpu_exec_destroy still frees them
    std::vector<ffgpu_exec *> left;
    { std::lock_guard<std::mutex> lk(dev->mu); left.swap(dev->execs); }
    for (ffgpu_exec *ex : left) {
        if (ex->last_stream) (void)hipStreamSynchronize(ex->last_stream);
        ex->dev = nullptr; ex->net = nullptr;
    }
    (void)hipFree(dev->d_weights);
    delete dev;
}

// net_forward (ffcnn.c:476-520) for the one frame in layer_list[0].data: H2D, the batch-1 executor, boxes back into
// net->bbox_list -- ALL of them, up to net->bbox_max like the reference (the fixed-size record holds the first
// FFGPU_MAX_DET; the rest comes from the executor's full list).  profile != 0 (FFCNN_PROFILE=1, the counterpart of
// ENABLE_NET_PROFILE, ffcnn.c:33,494-510): the forward runs launch by launch between HIP events and the device time of
// every layer kind is added to net->timeused[] (whole milliseconds of the accumulated time, as the reference keeps them).
extern "C" int ffgpu_netdev_forward1(NET *net, void *p, int profile)
{
    ffgpu_netdev *dev = (ffgpu_netdev *)p;
    if (!dev->exec1) {
        // one frame at a time is latency: the records come back through the pinned host mirror (no device-to-host copy), and the
        // input tensor the application fills is page-locked memory of the runtime since net_load (ffgpu_host_alloc): one DMA up
        dev->exec1 = ffgpu_exec_create(net, 1, FFGPU_HOST_DETS);
        if (!dev->exec1) return -1;
    }
    ffgpu_exec *ex = dev->exec1;
    ex->s1 = net->s1 ? net->s1 : 1;
    ex->s2 = net->s2 ? net->s2 : 1;
    ex->bbox_max = std::max(net->bbox_max, 1);
    if (profile) {
        if (ensure_input(ex)) return -1;
        if (copy_h2d(ex->d_input, net->layer_list[0].data, sizeof(float) * (size_t)ex->in_c * ex->in_h * ex->in_w)) return -1;
        float us[LAYER_TYPE_TOTOAL];
        if (ffgpu_exec_profile(ex, ex->d_input, us)) return -1;
        for (int k = 0; k < LAYER_TYPE_TOTOAL; k++) { dev->us_acc[k] += us[k]; net->timeused[k] = (int)(dev->us_acc[k] / 1000.0 + 0.5); }
    } else if (ffgpu_exec_forward_host(ex, net->layer_list[0].data)) return -1;
    static thread_local ffgpu_frame_dets rec;
    if (ffgpu_exec_read_dets(ex, &rec, 1) != 1) return -1;
    const ffcnn_ext *ext = ffcnn_ext_of(net);
    const int nb = std::max(0, std::min(rec.nfull, std::min(net->bbox_max, ext ? ext->box_cap : FFGPU_MAX_DET)));
    if (nb <= FFGPU_MAX_DET) memcpy(net->bbox_list, rec.box, sizeof(BBOX) * (size_t)nb);
    else if (ffgpu_exec_read_boxes(ex, 0, net->bbox_list, nb) < 0) return -1;
    net->bbox_num = nb;
    return 0;
}

extern "C" int ffgpu_netdev_profile_us(void *p, double us_by_kind[LAYER_TYPE_TOTOAL])
{
    ffgpu_netdev *dev = (ffgpu_netdev *)p;
    if (!dev) return -1;
    for (int k = 0; k < LAYER_TYPE_TOTOAL; k++) us_by_kind[k] = dev->us_acc[k];
    return 0;
}

extern "C" int ffgpu_net_weights_dev(NET *net, void **dev_ptr, size_t *bytes)
{
    ffgpu_netdev *dev = netdev_of(net);
    if (!dev) return -1;
    if (dev_ptr) *dev_ptr = dev->d_weights;
    if (bytes) *bytes = sizeof(float) * (size_t)net->weight_size;
    return 0;
}

extern "C" int ffgpu_net_weights_commit(NET *net, void *stream)
{
    ffgpu_netdev *dev = netdev_of(net);
    if (!dev) return -1;
    std::lock_guard<std::mutex> lk(dev->mu);
    for (ffgpu_exec *ex : dev->execs)           // fused blocks keep their constants in a packed LDS image
        if (repack(ex, stream ? (hipStream_t)stream : ex->own_stream)) return -1;
    if (!stream) for (ffgpu_exec *ex : dev->execs) FFGPU_CHECK(hipStreamSynchronize(ex->own_stream));
    return 0;
}

#include "ffgpu_node.inc"

// -------------------------------------------------------------------------- C-ABI: single conv on device tensors
static void fill_desc(ConvDesc &d, const float *in, const float *filt, float *out, int batch,
                      int iw, int ih, int ic, int groups, int pad, int stride, int fs, int ow, int oh, int oc, int act, int flags)
{
    memset(&d, 0, sizeof d);
    d.in = in; d.filt = filt; d.out = out; d.N = batch;
    d.iw = iw; d.ih = ih; d.ic = ic; d.ow = ow; d.oh = oh; d.oc = oc;
    d.fs = fs; d.stride = stride; d.pad = pad; d.groups = groups; d.act = act; d.flags = flags & (FFGPU_COMPAT_V6 | FFGPU_BF16_PW);
    d.in_cs = (long)batch * iw * ih; d.in_ns = (long)iw * ih;
    d.out_cs = (long)batch * ow * oh; d.out_ns = (long)ow * oh;
}

extern "C" int ffgpu_groupconv_dev(const float *d_in, const float *d_filt, float *d_out, int batch,
                                   int iw, int ih, int ic, int groups, int pad, int stride,
                                   int fs, int fn, int ow, int oh, int oc, int act,
                                   int flags, int variant, void *stream)
{
    if (!d_in || !d_filt || !d_out || batch < 1 || fn != oc) { ffgpu_set_error("groupconv_dev: bad arguments"); return -1; }
    ConvDesc d;
    fill_desc(d, d_in, d_filt, d_out, batch, iw, ih, ic, groups, pad, stride, fs, ow, oh, oc, act, flags);
    return ffgpu_launch_conv(d, variant, (hipStream_t)stream);
}

extern "C" const char *ffgpu_groupconv_kernel_name(int batch, int iw, int ih, int ic, int groups, int pad,
                                                   int stride, int fs, int fn, int variant)
{
    ConvDesc d;
    const int ow = (iw + 2 * pad - fs) / stride + 1, oh = (ih + 2 * pad - fs) / stride + 1;
    fill_desc(d, nullptr, nullptr, nullptr, batch, iw, ih, ic, groups, pad, stride, fs, ow, oh, fn, 0, 0);
    return ffgpu_conv_kernel_name(d, variant);
}

extern "C" float ffgpu_groupconv_time_dev(const float *d_in, const float *d_filt, float *d_out, int batch,
                                          int iw, int ih, int ic, int groups, int pad, int stride,
                                          int fs, int fn, int ow, int oh, int oc, int act,
                                          int flags, int variant, int warmup, int iters, void *stream)
{
    if (iters < 1 || fn != oc) { ffgpu_set_error("groupconv_time_dev: bad arguments"); return -1.f; }
    hipStream_t s = (hipStream_t)stream;
    ConvDesc d;
    fill_desc(d, d_in, d_filt, d_out, batch, iw, ih, ic, groups, pad, stride, fs, ow, oh, oc, act, flags);
    // like the executor, hand pointwise kernels their plan-time weight image (packed once, outside the timed loop)
    float *pk = nullptr;
    if (variant == FFGPU_K_AUTO && ffgpu_pw_pack_floats(d) > 0) {
        if (hipMalloc(&pk, ffgpu_pw_pack_floats(d) * sizeof(float)) != hipSuccess || ffgpu_pw_pack(d, pk, s)) { ffgpu_set_error("pack failed"); return -1.f; }
        d.wpack = pk;
    }
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { ffgpu_set_error("hipEventCreate failed"); return -1.f; }
    for (int i = 0; i < warmup; i++) if (ffgpu_launch_conv(d, variant, s)) return -1.f;
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < iters; i++) if (ffgpu_launch_conv(d, variant, s)) return -1.f;
    (void)hipEventRecord(e1, s);
    if (hipEventSynchronize(e1) != hipSuccess) { ffgpu_set_error("hipEventSynchronize: %s", hipGetErrorString(hipGetLastError())); return -1.f; }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return ms * 1000.f / iters;
}

extern "C" float ffgpu_irb_dev(const float *d_in, const float *d_w1, const float *d_wd, const float *d_w2,
                               const float *d_res, float *d_out, int batch, int iw, int ih, int ic, int ec, int oc,
                               int stride, int act1, int actd, int act2, int res_act, int warmup, int iters, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    IrbDesc d{};
    d.in = d_in; d.out = d_out; d.residual = d_res; d.w1 = d_w1; d.wd = d_wd; d.w2 = d_w2;
    d.N = batch; d.H = ih; d.W = iw; d.OH = (ih + 2 - 3) / stride + 1; d.OW = (iw + 2 - 3) / stride + 1;
    d.ic = ic; d.ec = ec; d.oc = oc; d.stride = stride;
    d.act1 = act1; d.actd = actd; d.act2 = act2; d.res_act = res_act;
    if (!d_in || !d_out || !d_w1 || !d_wd || !d_w2 || !ffgpu_irb_supported(d)) { ffgpu_set_error("irb_dev: unsupported block shape"); return -1.f; }
    float *pk = nullptr;
    if (hipMalloc(&pk, ffgpu_irb_pack_floats(d) * sizeof(float)) != hipSuccess) { ffgpu_set_error("irb_dev: hipMalloc failed"); return -1.f; }
    d.pk = pk;
    float us = 0.f;
    int rc = ffgpu_irb_pack(d, pk, s);
    if (!rc && iters <= 0) rc = ffgpu_launch_irb(d, s);
    if (!rc && iters > 0) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int i = 0; i < warmup && !rc; i++) rc = ffgpu_launch_irb(d, s);
        (void)hipEventRecord(e0, s);
        for (int i = 0; i < iters && !rc; i++) rc = ffgpu_launch_irb(d, s);
        (void)hipEventRecord(e1, s);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        us = ms * 1000.f / iters;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    (void)hipStreamSynchronize(s);
    (void)hipFree(pk);
    return rc ? -1.f : us;
}

// depthwise K x K (stride 1, same padding) + pointwise 1x1 in one launch: two consecutive groupconv calls of the reference;
// d_wd / d_wp are the two layers' filter rows (conv.h layout).  iters > 0: mean microseconds per launch instead of 0.
extern "C" float ffgpu_dwpw_dev(const float *d_in, const float *d_wd, const float *d_wp, float *d_out, int batch, int iw, int ih,
                                int c, int oc, int fs, int actd, int actp, int warmup, int iters, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    ConvDesc dw, pw;
    fill_desc(dw, d_in, d_wd, nullptr, batch, iw, ih, c, c, fs / 2, 1, fs, iw, ih, c, actd, 0);
    fill_desc(pw, nullptr, d_wp, d_out, batch, iw, ih, c, 1, 0, 1, 1, iw, ih, oc, actp, 0);
    if (!d_in || !d_wd || !d_wp || !d_out || !ffgpu_dwpw_ok(dw, pw)) { ffgpu_set_error("dwpw_dev: unsupported layer pair"); return -1.f; }
    float *pk = nullptr;
    if (hipMalloc(&pk, ffgpu_dwpw_pack_floats(dw, pw) * sizeof(float)) != hipSuccess) { ffgpu_set_error("dwpw_dev: hipMalloc failed"); return -1.f; }
    float us = 0.f;
    int rc = ffgpu_dwpw_pack(dw, pw, pk, s);
    if (!rc && iters <= 0) rc = ffgpu_launch_dwpw(dw, pw, pk, s);
    if (!rc && iters > 0) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int i = 0; i < warmup && !rc; i++) rc = ffgpu_launch_dwpw(dw, pw, pk, s);
        (void)hipEventRecord(e0, s);
        for (int i = 0; i < iters && !rc; i++) rc = ffgpu_launch_dwpw(dw, pw, pk, s);
        (void)hipEventRecord(e1, s);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        us = ms * 1000.f / iters;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    (void)hipStreamSynchronize(s);
    (void)hipFree(pk);
    return rc ? -1.f : us;
}

// -------------------------------------------------------------------------- conv.h drop-in (host pointers)
// Literal replacement for a conv-vN.c object: stages through device scratch that
// grows on demand and is kept for the life of the process (one per thread).
struct HostConvScratch { float *in = nullptr, *filt = nullptr, *out = nullptr; size_t in_n = 0, filt_n = 0, out_n = 0; int dev = -1; };

static int grow(float **p, size_t *have, size_t need)
{
    if (*have >= need) return 0;
    (void)hipFree(*p);
    *p = nullptr; *have = 0;
    FFGPU_CHECK(hipMalloc(p, need * sizeof(float)));
    *have = need;
    return 0;
}

extern "C" void groupconv(float *in, float *filt, float *out,
                          int iw, int ih, int ic, int ig, int ipad, int istride,
                          int fs, int fn, int ow, int oh, int oc, int act,
                          float **scratch, int *scratch_floats)
{
    static thread_local HostConvScratch sc;
    (void)scratch; (void)scratch_floats;          // the reference's callee-grown host scratch is not needed
    if (!in || !filt || !out || ig < 1 || ic % ig) { fprintf(stderr, "ffcnn groupconv: bad arguments\n"); return; }
    const size_t n_in = (size_t)iw * ih * ic, n_out = (size_t)ow * oh * oc;
    const size_t n_f = (size_t)fn * ((((size_t)fs * fs * (ic / ig) + 3) & ~(size_t)3) + 4);
    const char *env = getenv("FFCNN_COMPAT_V6");
    const int flags = (env && atoi(env)) ? FFGPU_COMPAT_V6 : 0;
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (sc.dev != cur) {                          // the thread moved to another GPU (ffgpu_set_device): the buffers stay behind
        (void)hipFree(sc.in); (void)hipFree(sc.filt); (void)hipFree(sc.out);
        sc = HostConvScratch();
        sc.dev = cur;
    }
    int rc = grow(&sc.in, &sc.in_n, n_in) || grow(&sc.filt, &sc.filt_n, n_f) || grow(&sc.out, &sc.out_n, n_out);
    if (!rc) rc = copy_h2d(sc.in, in, n_in * sizeof(float)) || copy_h2d(sc.filt, filt, n_f * sizeof(float));   // (staged: DESIGN.md section 10)
    if (!rc) rc = ffgpu_groupconv_dev(sc.in, sc.filt, sc.out, 1, iw, ih, ic, ig, ipad, istride, fs, fn, ow, oh, oc, act, flags, FFGPU_K_AUTO, nullptr);
    if (!rc) rc = hipStreamSynchronize(nullptr) != hipSuccess || copy_d2h(out, sc.out, n_out * sizeof(float));
    if (rc) fprintf(stderr, "ffcnn groupconv: device path failed: %s\n", g_err[0] ? g_err : hipGetErrorString(hipGetLastError()));
}

// ffgpu_dev.hpp -- shared declarations of the HIP side of libffcnn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "ffgpu_internal.h"

#define FFGPU_CHECK(expr)                                                                   \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            ffgpu_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return -1;                                                                      \
        }                                                                                   \
    } while (0)

// One grouped convolution on device tensors.  Strides are in floats:
// element (c, n, y, x) of the input lives at in + c*in_cs + n*in_ns + y*iw + x.
// CNHW tensors have cs = N*h*w, ns = h*w; the frame-major batch input has
// cs = h*w, ns = C*h*w.
struct ConvDesc {
    const float *in;
    const float *const *in_ind;   // optional: device slot holding the input pointer (the executor's parameter block);
                                  // when set the kernel reads *in_ind instead of `in` -- a captured graph then serves
                                  // every input buffer.  Only the kernels ffgpu_conv_supports_ind() names honour it.
    const float *filt;     // fn rows of K4+4 floats (conv.h layout)
    float       *out;
    const float *residual; // optional: out = act2(conv_act(...) + residual), same layout as out (fused shortcut)
    const float *wpack;    // optional: plan-time LDS image of the weights (ffgpu_pw_pack), pointwise layers only
    int   N;
    int   iw, ih, ic;
    int   ow, oh, oc;
    int   fs, stride, pad, groups;
    int   act;            // activation of the conv itself
    int   res_act;        // activation applied after adding the residual
    int   flags;          // FFGPU_COMPAT_V6
    long  in_cs, in_ns, out_cs, out_ns, res_cs, res_ns;
    int   nsplit;         // implicit-GEMM split-K factor frozen at plan time (ffgpu_conv_plan); 0 = decided at launch (single-layer calls)
    int   kernel;         // FFGPU_K_* frozen at plan time (ffgpu_conv_plan): ffgpu_launch_conv(AUTO), the pack size and the pack kernel all follow it,
                          // so a later change of the FFGPU_* tuning environment cannot pair one kernel with another kernel's weight image; 0 = pick at launch
    int   x3_mt;          // k_conv_x3's MT (16-row blocks per wave) / k_pw_x3t's RB (32-row blocks per wave): the layout of the packed image, frozen with the plan; 0 = decided at launch
};
// internal bit of ConvDesc::flags (never part of the public flag set): the step reads the executor's BATCH INPUT (frame-major, possibly through the
// parameter block) -- kernels that cannot read through ConvDesc::in_ind are not picked for it, so the one-graph-for-every-input property survives
#define FFGPU_F_BATCH_INPUT (1 << 30)

static inline int conv_k4(const ConvDesc &d) { return (d.fs * d.fs * (d.ic / d.groups) + 3) & ~3; }

// Fused 1x1 expand -> depthwise 3x3 -> 1x1 project [+ residual] on CNHW tensors (ffgpu_irb.inc)
struct IrbDesc {
    const float *in; float *out; const float *residual;
    const float *w1, *wd, *w2;
    int N, H, W, OH, OW, ic, ec, oc, stride;
    int act1, actd, act2, res_act;
    const float *pk;          // packed constants (ffgpu_irb_pack_floats floats, filled by ffgpu_irb_pack)
    int flags;                // FFGPU_CONCURRENT: tile splits chosen for several chains in flight
    int half;                 // k_irbw's "half last group" form frozen with the plan (ffgpu_irb_plan): 0 = decided at every call (single-block calls), 1 = off, 2 = on
};
bool   ffgpu_irb_supported(const IrbDesc &d);
bool   ffgpu_irb_is_thin(const IrbDesc &d);      // 8 expanded channels: streaming VALU kernel instead of the MFMA/LDS one
size_t ffgpu_irb_pack_floats(const IrbDesc &d);
void   ffgpu_irb_plan(IrbDesc &d);                     // freezes what the packed image's layout depends on (FFGPU_IRBW_HALF is read once, here)
int    ffgpu_irb_pack(const IrbDesc &d, float *pk, hipStream_t s);
int    ffgpu_launch_irb(const IrbDesc &d, hipStream_t s);
bool   ffgpu_front_ok(const ConvDesc &c, const IrbDesc &d);      // first layer (3x3 s2, 3 -> 8) + thin block as one streaming kernel
int    ffgpu_launch_front(const ConvDesc &c, const IrbDesc &d, hipStream_t s, bool u8 = false);

// depthwise K x K (stride 1, same padding) + pointwise 1 x 1 as one launch (ffgpu_dwpw.inc); dw.out == pw.in is never written
bool   ffgpu_dwpw_ok(const ConvDesc &dw, const ConvDesc &pw);
size_t ffgpu_dwpw_pack_floats(const ConvDesc &dw, const ConvDesc &pw);
int    ffgpu_dwpw_pack(const ConvDesc &dw, const ConvDesc &pw, float *pk, hipStream_t s);
int    ffgpu_launch_dwpw(const ConvDesc &dw, const ConvDesc &pw, const float *wpack, hipStream_t s);

size_t ffgpu_pw_pack_floats(const ConvDesc &d);
void   ffgpu_conv_plan(ConvDesc &d);                   // freezes kernel / nsplit / x3_mt for this layer NOW (the tuning environment is read once, here)
int    ffgpu_pw_pack(const ConvDesc &d, float *pk, hipStream_t s);

// Per-executor parameter block in device memory: what changes from one forward to the next without changing the
// launch list.  A one-thread kernel (ffgpu_launch_set_params) rewrites it in stream order in front of the graph launch,
// so ONE instantiated graph serves every input buffer and every box scale (the first round keyed a graph cache on them).
struct ExecParams {
    const float *frames;      // this forward's batch input (frame-major N x C x H x W)
    int s1, s2;               // box rescale ratio (ffcnn.c:267-273), applied by k_nms
    ffgpu_frame_dets *ring;   // record ring of the multi-GPU gather (ffgpu_exec_set_ring), or NULL
    int ring_slots, ring_stride;
    int bbox_max;             // NET.bbox_max of this forward: the reference re-reads it on every net_forward (ffcnn.c:461-463)
    // u8 BGR frames of the net's own geometry, converted by the first kernel itself (k_front<.., true>; NULL: fp32 frames)
    const unsigned char *bgr;
    long  bgr_frame;          // bytes from one frame to the next
    int   bgr_pitch;          // bytes per image row (ALIGN(3 w, 4), ffcnn.c:262)
    float mean[3], norm[3];   // net_input's per-channel mean / norm (plane order R, G, B)
};
int  ffgpu_launch_set_params(ExecParams *d_prm, const ExecParams &v, hipStream_t s);
bool ffgpu_conv_supports_ind(const ConvDesc &d);    // the kernel ffgpu_launch_conv would pick reads ConvDesc::in_ind

// kernels.hip
int         ffgpu_launch_conv(const ConvDesc &d, int variant, hipStream_t s);
const char *ffgpu_conv_kernel_name(const ConvDesc &d, int variant);
int ffgpu_launch_pool(const float *in, float *out, int N, int c, int w, int h, int fs, int stride, int is_max, hipStream_t s);
int ffgpu_launch_spp(const float *in, float *const out[3], const int fs[3], int n, long planes, int w, int h, hipStream_t s);
int ffgpu_launch_upsample(const float *in, float *out, long planes, int w, int h, int stride, hipStream_t s);
int ffgpu_launch_add_act(const float *a, const float *b, float *out, long n, int act, hipStream_t s);
int ffgpu_launch_copy(const float *src, float *dst, long n, hipStream_t s);
int ffgpu_launch_hash64(const float *x, long n, unsigned long long *out, hipStream_t s);
int ffgpu_launch_input_bgr(const unsigned char *bgr, float *out, int N, int w, int h, int W, int H,
                           int sw, int sh, int s1, int s2, const float mean[3], const float norm[3], hipStream_t s);

struct YoloHead {
    const float *in;       // CNHW, 3*(5+classes) channels
    int   w, h, classes;
    int   anchors[3][2];
    float thresh, scale_xy;
    int   key_base;        // emission-order key of this head's first candidate
};
// cand / cand_key: `cap` slots per frame (cap = 3 * cells summed over the heads: every anchor of every cell has a slot, so
// the decode never drops a candidate -- the reference's buffer holds bbox_max = 51 200 of them, ffcnn.c:243,463)
int ffgpu_launch_yolo(const YoloHead &hd, int N, int netw, int neth, BBOX *cand, int *cand_key, int *ncand, int cap, int *ring_ctr, hipStream_t s);
// full (may be NULL): cap boxes per frame, ALL survivors in score order (the fixed-size record keeps the first FFGPU_MAX_DET)
// bbox_max: the reference stops appending candidates at net->bbox_max in emission order (ffcnn.c:463); same here
// scratch (cap_pow2 > FFGPU_NMS_LDS_CAP only): 12 bytes x cap_pow2 per frame of global memory instead of LDS
int ffgpu_launch_nms(const BBOX *cand, const int *cand_key, int *ncand, int cap, BBOX *full, void *scratch,
                     ffgpu_frame_dets *dets, ffgpu_frame_dets *dets_host, const int *ring_ctr, int N,
                     float thresh, int use_min, const ExecParams *prm, hipStream_t s);
#define FFGPU_NMS_LDS_CAP 8192
bool ffgpu_nms_in_lds(int cap_pow2);      // the work arrays of cap_pow2 slots fit the CURRENT device's LDS (else: global scratch, 13 bytes per slot and frame)
int ffgpu_launch_clear(int *ncand, int N, int *ring_ctr, hipStream_t s);

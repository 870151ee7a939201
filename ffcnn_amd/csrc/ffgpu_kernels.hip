// ffgpu_kernels.hip -- layer kernels of the ffcnn forward path for gfx950 (CDNA4).
//
// Every kernel works on CNHW device tensors (see ffcnn_hip.h) so that the batch
// folds into the plane index: a depthwise plane, a pointwise pixel run and an
// elementwise span are all contiguous regardless of the batch size.
//
// Reference arithmetic restated here (file:line into /root/reference):
//   conv + folded BN + activation      conv-v0.c:7-31, utils.h:15-23
//   max/avg pool                       ffcnn.c:337-394
//   upsample / shortcut / route        ffcnn.c:396-434
//   YOLO decode                        ffcnn.c:438-474
//   NMS                                ffcnn.c:298-335
//   net_input                          ffcnn.c:259-289
#include <algorithm>
#include <type_traits>
#include <vector>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include "ffgpu_dev.hpp"

#define WAVE 64

// A pointer READ FROM MEMORY (the executor's parameter block, ConvDesc::in_ind) is a generic ("flat") pointer to the compiler:
// its loads become flat_load_*, which count on lgkmcnt as well as vmcnt, and SIInsertWaitcnts then turns every wait for an LDS
// read into lgkmcnt(0) while one is in flight -- a kernel that prefetches input rows from HBM under arithmetic fed by LDS
// tables (k_front, k_conv_igemm) waited a full memory latency at its first table read.  The input tensor IS global memory, so
// say so: loads through a gfp are global_load_* (vmcnt only).
#define FFG __attribute__((address_space(1)))
typedef const FFG float *gfp;
__device__ __forceinline__ gfp to_global(const float *p) { return (gfp)p; }
template <class T> __device__ __forceinline__ T gld(gfp p) { return *(const FFG T *)p; }

__device__ __forceinline__ float act_apply(float x, int act)
{
    switch (act) {
    case 1: return x > 0.f ? x : 0.f;
    case 2: return x > 0.f ? x : 0.1f * x;
    case 3: return 1.0f / (1.0f + (float)exp((double)-x));
    default: return x;
    }
}

// ---------------------------------------------------------------------------
// Generic grouped convolution: one thread per output element, taps accumulated
// in the reference's order (channel, tap row, tap column; conv-v0.c:14-24).
// Covers every (fs, stride, pad, groups) the cfg grammar can express; the
// specialised kernels below take over the shapes that matter for throughput.
__global__ void k_conv_generic(ConvDesc d)
{
    const long total = (long)d.oc * d.N * d.oh * d.ow;
    const int gic = d.ic / d.groups, goc = d.oc / d.groups;
    const int k4 = (d.fs * d.fs * gic + 3) & ~3, rl = k4 + 4;
    const bool v6dw5 = (d.flags & FFGPU_COMPAT_V6) && d.pad == 2 && d.fs == 5 && d.stride == 1 && gic == 1;
    const gfp in0 = to_global(d.in_ind ? *d.in_ind : d.in);          // executor parameter block (one graph for every input buffer)
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % d.ow);
        long t = idx / d.ow;
        const int y = (int)(t % d.oh); t /= d.oh;
        const int n = (int)(t % d.N);
        const int o = (int)(t / d.N);
        const int g = o / goc;
        const float *w = d.filt + (long)o * rl;
        const gfp src = in0 + (long)g * gic * d.in_cs + (long)n * d.in_ns;
        float acc = 0.f;
        for (int ci = 0; ci < gic; ci++) {
            const gfp pl = src + (long)ci * d.in_cs;
            for (int ky = 0; ky < d.fs; ky++) {
                if (v6dw5 && d.oh > 2 && y == d.oh - 2 && ky == 0) continue;   // conv-v6.c:422-441
                const int sy = y * d.stride - d.pad + ky;
                if ((unsigned)sy >= (unsigned)d.ih) continue;
                for (int kx = 0; kx < d.fs; kx++) {
                    const int sx = x * d.stride - d.pad + kx;
                    if ((unsigned)sx >= (unsigned)d.iw) continue;
                    acc = fmaf(pl[(long)sy * d.iw + sx], w[(ci * d.fs + ky) * d.fs + kx], acc);
                }
            }
        }
        float v = act_apply(fmaf(acc, w[k4], w[k4 + 1]), d.act);
        const long ooff = (long)o * d.out_cs + (long)n * d.out_ns + (long)y * d.ow + x;
        if (d.residual) v = act_apply(v + d.residual[(long)o * d.res_cs + (long)n * d.res_ns + (long)y * d.ow + x], d.res_act);
        d.out[ooff] = v;
    }
}

// ---------------------------------------------------------------------------
// pooling: window [x-(fs-1)/2, +fs) clipped to the plane (ffcnn.c:337-372)
__global__ void k_pool(const float *in, float *out, long planes, int w, int h, int fs, int stride, int is_max)
{
    const int ow = w / stride, oh = h / stride;
    const long total = planes * oh * ow;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(idx % ow);
        const long t = idx / ow;
        const int oy = (int)(t % oh);
        const float *p = in + (t / oh) * (long)w * h;
        int x0 = ox * stride - (fs - 1) / 2, y0 = oy * stride - (fs - 1) / 2;
        int x1 = min(x0 + fs, w), y1 = min(y0 + fs, h);
        x0 = max(x0, 0); y0 = max(y0, 0);
        float v = is_max ? p[y0 * w + x0] : 0.f;
        for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) {
                const float q = p[y * w + x];
                if (is_max) v = v < q ? q : v; else v += q;
            }
        out[idx] = is_max ? v : v / (float)(fs * fs);
    }
}

// the 2x2 / stride 2 max pool of the darknet-tiny backbones on planes whose width is a multiple of 8: a thread owns FOUR adjacent outputs = two rows of eight
// inputs (four 16-byte loads, one 16-byte store, 32-bit offsets); same comparisons in the same order as k_pool (row, then column)
typedef float v4f_pl __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_pool2x2(const float *in, float *out, unsigned nquads, unsigned owq, unsigned m_owq, int w)
{
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    if (t >= nquads) return;
    const unsigned row = owq == 1 ? t : __umulhi(t, m_owq), xq = t - row * owq;      // row = (plane, output row): input rows 2 row, 2 row + 1 (h even)
    const gfp src = to_global(in) + ((size_t)row * 2u * (unsigned)w + 8u * xq);
    const v4f_pl a0 = gld<v4f_pl>(src), a1 = gld<v4f_pl>(src + 4), b0 = gld<v4f_pl>(src + w), b1 = gld<v4f_pl>(src + w + 4);
    auto mx = [](float p00, float p01, float p10, float p11) { float v = p00; v = v < p01 ? p01 : v; v = v < p10 ? p10 : v; v = v < p11 ? p11 : v; return v; };
    const v4f_pl r = { mx(a0.x, a0.y, b0.x, b0.y), mx(a0.z, a0.w, b0.z, b0.w), mx(a1.x, a1.y, b1.x, b1.y), mx(a1.z, a1.w, b1.z, b1.w) };
    *reinterpret_cast<v4f_pl *>(out + (size_t)t * 4) = r;
}

// several stride-1 max pools of ONE tensor (the SPP block of a yolo cfg: 3x3, 5x5, 9x9 of the same 10x10 planes) in one
// launch: a plane is staged in LDS once and every window size is read from there (same clipping as k_pool)
struct SppP { const float *in; float *out[3]; int fs[3]; int n; long planes; int w, h; int cascade; };

// cascade != 0 (odd, ascending sizes): max over a (2a+1) window of the max over a (2b+1) window is the max over the
// (2(a+b)+1) window -- also with windows clipped to the plane -- so size k is computed from the result of size k-1
// with a small separable (row pass, column pass) filter: 3x3, 5x5, 9x9 cost 3+3, 3+3, 5+5 LDS reads per pixel.
__global__ void __launch_bounds__(256) k_spp(SppP p)
{
    extern __shared__ float spp_s[];
    const int hw = p.w * p.h;
    constexpr int NP = 2;                                       // planes per workgroup trip
    float *cur = spp_s, *tmp = spp_s + NP * hw;
    for (long pl0 = (long)blockIdx.x * NP; pl0 < p.planes; pl0 += (long)gridDim.x * NP) {
        const int np = (int)min((long)NP, p.planes - pl0);
        __syncthreads();
        for (int i = threadIdx.x; i < np * hw; i += blockDim.x) cur[i] = p.in[pl0 * hw + i];
        __syncthreads();
        if (!p.cascade) {
            for (int i = threadIdx.x; i < np * hw; i += blockDim.x) {
                const int q = i / hw, r = i - q * hw, oy = r / p.w, ox = r - oy * p.w;
                for (int k = 0; k < p.n; k++) {
                    const int fs = p.fs[k];
                    int x0 = ox - (fs - 1) / 2, y0 = oy - (fs - 1) / 2;
                    const int x1 = min(x0 + fs, p.w), y1 = min(y0 + fs, p.h);
                    x0 = max(x0, 0); y0 = max(y0, 0);
                    float v = cur[q * hw + y0 * p.w + x0];
                    for (int y = y0; y < y1; y++)
                        for (int x = x0; x < x1; x++) { const float t = cur[q * hw + y * p.w + x]; v = v < t ? t : v; }
                    p.out[k][pl0 * hw + i] = v;
                }
            }
            continue;
        }
        int prev = 1;
        for (int k = 0; k < p.n; k++) {
            const int a = (p.fs[k] - prev) / 2;                 // half width of the incremental window
            for (int i = threadIdx.x; i < np * hw; i += blockDim.x) {
                const int q = i / hw, r = i - q * hw, oy = r / p.w, ox = r - oy * p.w;
                const float *row = cur + q * hw + oy * p.w;
                float v = row[ox];
                for (int x = max(ox - a, 0); x <= min(ox + a, p.w - 1); x++) v = v < row[x] ? row[x] : v;
                tmp[i] = v;
            }
            __syncthreads();
            for (int i = threadIdx.x; i < np * hw; i += blockDim.x) {
                const int q = i / hw, r = i - q * hw, oy = r / p.w, ox = r - oy * p.w;
                const float *col = tmp + q * hw + ox;
                float v = col[oy * p.w];
                for (int y = max(oy - a, 0); y <= min(oy + a, p.h - 1); y++) v = v < col[y * p.w] ? col[y * p.w] : v;
                p.out[k][pl0 * hw + i] = v;
                cur[i] = v;                                     // each thread rewrites the pixels it owns in both passes
            }
            __syncthreads();
            prev = p.fs[k];
        }
    }
}

__global__ void k_upsample(const float *in, float *out, long planes, int w, int h, int stride)
{
    const int ow = w * stride, oh = h * stride;
    const long total = planes * oh * ow;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % ow);
        const long t = idx / ow;
        const int y = (int)(t % oh);
        out[idx] = in[((t / oh) * h + y / stride) * w + x / stride];
    }
}

// the same with one thread per INPUT element and 32-bit index arithmetic (the 64-bit divisions above cost more than
// the memory traffic on the small planes a yolo cfg upsamples); m_w / m_h = ceil(2^32 / w), ceil(2^32 / h)
__global__ void __launch_bounds__(256) k_upsample32(const float *in, float *out, unsigned n_in, int w, int h, int stride, unsigned m_w, unsigned m_h)
{
    const int ow = w * stride;
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n_in; idx += gridDim.x * blockDim.x) {
        const unsigned t = m_w ? __umulhi(idx, m_w) : idx;             // idx / w
        const unsigned x = idx - t * w;
        const unsigned pl = m_h ? __umulhi(t, m_h) : t;                // t / h
        const unsigned y = t - pl * h;
        const float v = in[idx];
        float *o = out + ((size_t)pl * h * stride + y * stride) * ow + x * stride;
        for (int dy = 0; dy < stride; dy++)
            for (int dx = 0; dx < stride; dx++) o[dy * ow + dx] = v;
    }
}

// out = act(a + b), flat (ffcnn.c:418-423); 16-byte lanes with a scalar tail
__global__ void k_add_act(const float *a, const float *b, float *out, long n, int act)
{
    const long n4 = n >> 2;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x, gsz = (long)gridDim.x * blockDim.x;
    for (long i = gid; i < n4; i += gsz) {
        const float4 u = reinterpret_cast<const float4 *>(a)[i], v = reinterpret_cast<const float4 *>(b)[i];
        float4 r;
        r.x = act_apply(u.x + v.x, act); r.y = act_apply(u.y + v.y, act);
        r.z = act_apply(u.z + v.z, act); r.w = act_apply(u.w + v.w, act);
        reinterpret_cast<float4 *>(out)[i] = r;
    }
    for (long i = (n4 << 2) + gid; i < n; i += gsz) out[i] = act_apply(a[i] + b[i], act);
}

__global__ void k_copy(const float *src, float *dst, long n)
{
    const long n4 = n >> 2;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x, gsz = (long)gridDim.x * blockDim.x;
    for (long i = gid; i < n4; i += gsz) reinterpret_cast<float4 *>(dst)[i] = reinterpret_cast<const float4 *>(src)[i];
    for (long i = (n4 << 2) + gid; i < n; i += gsz) dst[i] = src[i];
}

// batched net_input (ffcnn.c:259-289): u8 BGR -> planar RGB fp32, nearest
// resize into the top-left sw x sh corner, zeros elsewhere.  Output is the
// frame-major batch input (N x 3 x H x W).
struct InputP { float mean[3], norm[3]; };
__global__ void k_input_bgr(const unsigned char *bgr, float *out, int N, int w, int h, int W, int H,
                            int sw, int sh, int s1, int s2, InputP p)
{
    const long total = (long)N * H * W;
    const long pitch = (long)((w * 3 + 3) & ~3);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % W);
        const long t = idx / W;
        const int y = (int)(t % H);
        const int n = (int)(t / H);
        float r = 0.f, g = 0.f, b = 0.f;
        if (x < sw && y < sh) {
            const unsigned char *px = bgr + (long)n * pitch * h + (long)((long)y * s1 / s2) * pitch + (long)((long)x * s1 / s2) * 3;
            r = ((float)px[2] - p.mean[0]) * p.norm[0];
            g = ((float)px[1] - p.mean[1]) * p.norm[1];
            b = ((float)px[0] - p.mean[2]) * p.norm[2];
        }
        float *o = out + (long)n * 3 * H * W + (long)y * W + x;
        o[0] = r; o[(long)H * W] = g; o[2L * H * W] = b;
    }
}

// the same for W % 4 == 0: a thread owns 4 consecutive output pixels of one row (one 32-bit division per thread, blockIdx.y =
// frame), reads their source bytes -- 12 contiguous bytes as three dwords when the image is not resized -- and writes
// one 16-byte store per colour plane.  Same arithmetic per pixel as above ((byte - mean) * norm, two roundings).
__global__ void __launch_bounds__(256) k_input_bgr4(const unsigned char *bgr, float *out, int w, int h, int W, int H,
                                                    int sw, int sh, int s1, int s2, InputP p)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    const unsigned wq = (unsigned)W >> 2, t = blockIdx.x * 256 + threadIdx.x;
    if (t >= wq * (unsigned)H) return;
    const int y = (int)(t / wq), x0 = (int)(t - (unsigned)y * wq) * 4, n = blockIdx.y;
    const long pitch = (long)((w * 3 + 3) & ~3);
    f4 r = { 0.f, 0.f, 0.f, 0.f }, g = r, b = r;
    if (y < sh && x0 < sw) {
        const unsigned char *row = bgr + (long)n * pitch * h + (long)((long)y * s1 / s2) * pitch;
        unsigned char px[4][3];
        if (s1 == s2 && x0 + 3 < sw) {                              // not resized: 12 contiguous bytes, dword aligned (pitch % 4 == 0, 3 x0 % 4 == 0)
            const unsigned *q = reinterpret_cast<const unsigned *>(row + 3 * x0);
            const unsigned d0 = q[0], d1 = q[1], d2 = q[2];
            const unsigned char by[12] = { (unsigned char)d0, (unsigned char)(d0 >> 8), (unsigned char)(d0 >> 16), (unsigned char)(d0 >> 24),
                                           (unsigned char)d1, (unsigned char)(d1 >> 8), (unsigned char)(d1 >> 16), (unsigned char)(d1 >> 24),
                                           (unsigned char)d2, (unsigned char)(d2 >> 8), (unsigned char)(d2 >> 16), (unsigned char)(d2 >> 24) };
#pragma unroll
            for (int i = 0; i < 4; i++) { px[i][0] = by[3 * i]; px[i][1] = by[3 * i + 1]; px[i][2] = by[3 * i + 2]; }
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int x = min(x0 + i, sw - 1);
                const unsigned char *s = row + (long)((long)x * s1 / s2) * 3;
                px[i][0] = s[0]; px[i][1] = s[1]; px[i][2] = s[2];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const bool in = x0 + i < sw;
            r[i] = in ? ((float)px[i][2] - p.mean[0]) * p.norm[0] : 0.f;
            g[i] = in ? ((float)px[i][1] - p.mean[1]) * p.norm[1] : 0.f;
            b[i] = in ? ((float)px[i][0] - p.mean[2]) * p.norm[2] : 0.f;
        }
    }
    float *o = out + (long)n * 3 * H * W + (long)y * W + x0;
    *reinterpret_cast<f4 *>(o) = r;
    *reinterpret_cast<f4 *>(o + (long)H * W) = g;
    *reinterpret_cast<f4 *>(o + 2L * H * W) = b;
}

// ---------------------------------------------------------------------------
// YOLO head decode (ffcnn.c:438-474).  One thread per (frame, cell, anchor);
// lanes run along x so the 85 strided channel reads are coalesced.  exp() is
// evaluated in double and narrowed exactly where the reference does, and FMA
// contraction is off so the confidence/box arithmetic rounds like the C code.
// ring_ctr (one head of a forward only, may be NULL): this forward's number for the record ring -- counted here, consumed by k_nms
__global__ void k_yolo(YoloHead hd, int N, int netw, int neth, BBOX *cand, int *cand_key, int *ncand, int cap, int *ring_ctr)
{
#pragma clang fp contract(off)
    if (ring_ctr && blockIdx.x == 0 && threadIdx.x == 0) *ring_ctr += 1;
    const int cells = hd.w * hd.h;
    const long total = (long)N * 3 * cells;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const long gidc = min(gid, total - 1);
    const int cell = (int)(gidc % cells);
    const int k = (int)((gidc / cells) % 3);
    const int n = (int)(gidc / (3L * cells));
    const int j = cell % hd.w, i = cell / hd.w;
    const long cs = (long)N * cells;                                  // CNHW channel stride
    const float *p = hd.in + (long)k * (5 + hd.classes) * cs + (long)n * cells + cell;
    const float bs = p[4 * cs];
    // conf = 1 / (1 + e^-bs (1 + e^-cs)) <= 1 / (1 + e^-bs): cells whose objectness alone cannot reach the
    // threshold (almost all of them) skip the 80-class scan.  0.1 % slack keeps the shortcut rounding-proof.
    const bool pass = gid < total && !(1.0f / (1.0f + __expf(-bs)) < hd.thresh * 0.999f);
    // class scan of a passing cell: the WAVE reads its scores (lane l -> classes l, l + 64, ...) and reduces to the
    // first maximum (the reference's `if (best < v)` scan, ffcnn.c:452-456) -- one lane walking 80 strided loads
    // alone set the run time of the whole kernel
    float cs_best = 0.f;
    int best = 0;
    for (unsigned long long todo = __ballot(pass); todo; todo &= todo - 1) {
        const int src = __ffsll((long long)todo) - 1;
        const long g = gid - lane + src;                              // that lane's (frame, anchor, cell)
        const int c2 = (int)(g % cells), k2 = (int)((g / cells) % 3), n2 = (int)(g / (3L * cells));
        const float *q = hd.in + (long)k2 * (5 + hd.classes) * cs + (long)n2 * cells + c2;
        float v = -3.0e38f;
        int vi = 1 << 30;
        for (int l = lane; l < hd.classes; l += 64) {
            const float t = q[(5 + l) * cs];
            if (v < t) { v = t; vi = l; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(v, off);
            const int oi = __shfl_xor(vi, off);
            if (v < ov || (v == ov && oi < vi)) { v = ov; vi = oi; }
        }
        if (lane == src) { cs_best = v; best = vi; }
    }
    if (!pass) return;
    const float conf = 1.0f / ((1.0f + (float)exp((double)-bs) * (1.0f + (float)exp((double)-cs_best))));
    if (!(conf >= hd.thresh)) return;
    const float tx = p[0], ty = p[cs], tw = p[2 * cs], th = p[3 * cs];
    const float sx = 1.0f / (1.0f + (float)exp((double)-tx));
    const float sy = 1.0f / (1.0f + (float)exp((double)-ty));
    const float cx = (j + sx) * netw / hd.w;
    const float cy = (i + sy) * neth / hd.h;
    const float bw = (float)exp((double)tw) * hd.anchors[k][0] * hd.scale_xy;
    const float bh = (float)exp((double)th) * hd.anchors[k][1] * hd.scale_xy;
    const int slot = atomicAdd(&ncand[n], 1);
    if (slot >= cap) return;                 // cannot happen: cap = 3 * cells over all heads (one slot per anchor)
    BBOX b;
    b.type = best; b.score = conf;
    b.x1 = cx - bw * 0.5f; b.y1 = cy - bh * 0.5f;
    b.x2 = cx + bw * 0.5f; b.y2 = cy + bh * 0.5f;
    cand[(long)n * cap + slot] = b;
    cand_key[(long)n * cap + slot] = hd.key_base + cell * 3 + k;   // reference emission order
}

// NMS (ffcnn.c:298-335): one workgroup per frame.  Candidates are ordered by
// (score desc, emission key asc) with a bitonic sort -- the key makes the
// order total where qsort's is unspecified -- then suppressed greedily per class
// with inter/min(area) (or IoU) > thresh, compacted and rescaled by s1/s2.
// Capacity: every anchor of every cell has a candidate slot (cap per frame), so nothing is dropped before the sort; the
// reference's own limit -- it stops appending at net->bbox_max in EMISSION order (ffcnn.c:463) -- is reproduced by a
// first sort on the emission key when a frame has more candidates than that (never with the default bbox_max = 51 200).
// Work arrays (score, key, index, alive: 13 bytes per slot of the next power of two) live in LDS up to
// FFGPU_NMS_LDS_CAP slots and in a global scratch buffer beyond (GLB).
// dets_host (may be NULL): pinned host mirror of the records, written by the same threads (FFGPU_HOST_DETS) so a
// single-GPU consumer needs no device-to-host copy after the forward
// ring (may be NULL): caller-owned device ring of ring_slots x N records; forward number *ring_ctr (counted by k_clear at
// the start of the forward) goes to slot (*ring_ctr - 1) % ring_slots -- the multi-GPU job gathers whole groups of slots
// full (may be NULL): ALL survivors of the frame in score order (cap slots per frame); the fixed-size record holds the first
// FFGPU_MAX_DET of them and their total number in `nfull`
template <bool GLB>
__global__ void __launch_bounds__(256) k_nms(const BBOX *cand, const int *cand_key, int *ncand, int cap, int cap_p2,
                                             BBOX *full, unsigned char *scratch,
                                             ffgpu_frame_dets *dets, ffgpu_frame_dets *dets_host,
                                             const int *ring_ctr, float thresh, int use_min, const ExecParams *prm)
{
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) unsigned char nms_lds[];
    unsigned char *wb = GLB ? scratch + (size_t)blockIdx.x * cap_p2 * 13 : nms_lds;
    float *s_score = reinterpret_cast<float *>(wb);
    int   *s_key = reinterpret_cast<int *>(wb + (size_t)4 * cap_p2);
    int   *s_idx = reinterpret_cast<int *>(wb + (size_t)8 * cap_p2);
    unsigned char *s_alive = wb + (size_t)12 * cap_p2;
    const int n = blockIdx.x, tid = threadIdx.x;
    const int s1 = prm->s1, s2 = prm->s2, bbox_max = prm->bbox_max;   // (in the parameter block: a graph replays with this forward's values)
    ffgpu_frame_dets *const ring = prm->ring;                   // (the ring travels in the parameter block too: attaching or
    const int ring_slots = prm->ring_slots, ring_stride = prm->ring_stride;   //  restarting it does not invalidate the graph)
    const int total = ncand[n];
    int m = min(total, cap);
    const BBOX *c = cand + (long)n * cap;
    ffgpu_frame_dets *out = dets + n;
    int pow2 = 1;
    while (pow2 < m) pow2 <<= 1;
    for (int i = tid; i < pow2; i += blockDim.x) {
        s_score[i] = i < m ? c[i].score : -1.f;
        s_key[i] = i < m ? cand_key[(long)n * cap + i] : 0x7fffffff;
        s_idx[i] = i;
    }
    __syncthreads();
    // BYKEY: ascending emission key only (the pass that reproduces the reference's truncation); else (score desc, key asc)
    auto sort = [&](bool bykey) {
        for (int k = 2; k <= pow2; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < pow2; i += blockDim.x) {
                    const int l = i ^ j;
                    if (l > i) {
                        const bool up = (i & k) == 0;
                        // "a precedes b"
                        const bool a_first = bykey ? s_key[i] < s_key[l]
                                                   : (s_score[i] > s_score[l] || (s_score[i] == s_score[l] && s_key[i] < s_key[l]));
                        if (a_first != up) {
                            const float ts = s_score[i]; s_score[i] = s_score[l]; s_score[l] = ts;
                            const int tk = s_key[i]; s_key[i] = s_key[l]; s_key[l] = tk;
                            const int ti = s_idx[i]; s_idx[i] = s_idx[l]; s_idx[l] = ti;
                        }
                    }
                }
                __syncthreads();
            }
    };
    if (m > bbox_max) {                                   // uniform; keep the first bbox_max in emission order (ffcnn.c:463)
        sort(true);
        for (int i = bbox_max + tid; i < pow2; i += blockDim.x) { s_score[i] = -1.f; s_key[i] = 0x7fffffff; }
        m = bbox_max;
        __syncthreads();
    }
    sort(false);
    for (int i = tid; i < m; i += blockDim.x) s_alive[i] = 1;
    __syncthreads();
    for (int a = 0; a < m; a++) {
        if (!s_alive[a]) continue;                       // uniform: read after a barrier
        const BBOX ba = c[s_idx[a]];
        // (width and height of a box are kept out of ONE register pair: hipcc otherwise forms v_pk_mul_f32 v[a:b], v[a:b], v[a:b] op_sel:[0,1] ..., the operand
        //  pattern of the packed-math slip under concurrent bf16 MFMAs -- see pw_fma4_apart in ffgpu_pw_mfma.inc)
        auto area = [](float x1, float y1, float x2, float y2) { float w = x2 - x1; asm volatile("" : "+v"(w)); return w * (y2 - y1); };
        const float area_a = area(ba.x1, ba.y1, ba.x2, ba.y2);
        for (int j = a + 1 + tid; j < m; j += blockDim.x) {
            if (!s_alive[j]) continue;
            const BBOX bj = c[s_idx[j]];
            if (bj.type != ba.type) continue;
            const float xa = ba.x1 > bj.x1 ? ba.x1 : bj.x1, ya = ba.y1 > bj.y1 ? ba.y1 : bj.y1;
            const float xb = ba.x2 < bj.x2 ? ba.x2 : bj.x2, yb = ba.y2 < bj.y2 ? ba.y2 : bj.y2;
            const float inter = (xa < xb && ya < yb) ? area(xa, ya, xb, yb) : 0.f;
            const float area_j = area(bj.x1, bj.y1, bj.x2, bj.y2);
            const float uni = area_a + area_j - inter;
            const float metric = use_min ? inter / (area_a < area_j ? area_a : area_j) : inter / uni;
            if (metric > thresh) s_alive[j] = 0;
        }
        __syncthreads();
    }
    // survivors in score order: one thread walks the list (the key array has served its purpose and becomes the list),
    // every thread then writes slots
    __shared__ int s_nkeep;
    if (tid == 0) {
        int keep = 0;
        for (int i = 0; i < m; i++) if (s_alive[i]) s_key[keep++] = s_idx[i];
        s_nkeep = keep;
    }
    __syncthreads();
    const int nfull = s_nkeep, nrec = min(nfull, FFGPU_MAX_DET);
    auto scaled = [&](int i) {
        const BBOX b = c[s_key[i]];
        BBOX r;
        r.type = b.type; r.score = b.score;
        r.x1 = b.x1 * s1 / s2; r.y1 = b.y1 * s1 / s2;
        r.x2 = b.x2 * s1 / s2; r.y2 = b.y2 * s1 / s2;
        return r;
    };
    if (full) for (int i = tid; i < nfull; i += blockDim.x) full[(long)n * cap + i] = scaled(i);
    ffgpu_frame_dets *outs[3] = { out, dets_host ? dets_host + n : nullptr, nullptr };
    if (ring) outs[2] = ring + (size_t)((unsigned)(*ring_ctr - 1) % (unsigned)ring_slots) * ring_stride + n;
    // slots at or beyond both the previous and the new count are zero already in the record and its host mirror (both
    // start zeroed and are only ever written here): the mirror, which sits across PCIe, is touched only where it
    // changes.  A ring slot last held some older forward's record, so it is written in full.
    const int nwrite = max(nrec, min(max(out->count, 0), FFGPU_MAX_DET));
    __syncthreads();                                               // every thread has read the old count
    for (int i = tid; i < FFGPU_MAX_DET; i += blockDim.x) {
        BBOX r = { 0, 0.f, 0.f, 0.f, 0.f, 0.f };                   // reference zeroes the tail (ffcnn.c:333)
        if (i < nrec) r = scaled(i);
        if (i < nwrite) { outs[0]->box[i] = r; if (outs[1]) outs[1]->box[i] = r; }
        if (outs[2]) outs[2]->box[i] = r;
    }
    if (tid == 0) {
        for (int k = 0; k < 3; k++) if (outs[k]) {
            outs[k]->count = nrec;
            outs[k]->ncand = total;
            outs[k]->overflow = (total > bbox_max ? 1 : 0) | (nfull > FFGPU_MAX_DET ? 4 : 0);
            outs[k]->nfull = nfull;
        }
        ncand[n] = 0;                                              // the next forward's heads start counting from zero (no clear launch)
    }
}

// the executor's parameter block (ExecParams): rewritten in stream order in front of a graph launch
__global__ void k_set_params(ExecParams *prm, ExecParams v)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) *prm = v;
}

// start of a forward: no candidates yet; one more forward for the record ring
__global__ void k_clear(int *ncand, int N, int *ring_ctr)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) ncand[i] = 0;
    if (ring_ctr && blockIdx.x == 0 && threadIdx.x == 0) *ring_ctr += 1;
}

// ---------------------------------------------------------------------------
// launchers
static inline int grid_for(long total, int block, int cap = 256 * 16)
{
    long g = (total + block - 1) / block;
    if (g < 1) g = 1;
    return (int)(g > cap ? cap : g);
}

#define LAUNCH_OK(what)                                                                      \
    do {                                                                                     \
        hipError_t e_ = hipGetLastError();                                                   \
        if (e_ != hipSuccess) { ffgpu_set_error("%s launch failed: %s", what, hipGetErrorString(e_)); return -1; } \
    } while (0)

// Dynamic LDS above 64 KB has to be allowed per kernel AND per device: a node (ffgpu_node.inc) plans the same kernels on every
// GPU of the box from one process, so "raised once per process" is not enough.
static int lds_allow(const void *fn, size_t lds, const char *what)
{
    if (lds <= 64 * 1024) return 0;
    struct Seen { const void *fn; int dev; size_t lds; };
    static std::mutex mu;
    static std::vector<Seen> seen;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(mu);
    Seen *e = nullptr;
    for (auto &k : seen) if (k.fn == fn && k.dev == dev) e = &k;
    if (e && e->lds >= lds) return 0;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        ffgpu_set_error("%s: cannot raise dynamic LDS to %zu bytes", what, lds);
        return -1;
    }
    if (e) e->lds = lds; else seen.push_back({ fn, dev, lds });
    return 0;
}

#include "ffgpu_conv_kernels.inc"

int ffgpu_launch_pool(const float *in, float *out, int N, int c, int w, int h, int fs, int stride, int is_max, hipStream_t s)
{
    if (stride < 1 || fs < 1) { ffgpu_set_error("pool: bad size/stride"); return -1; }
    const long planes = (long)N * c, total = planes * (h / stride) * (w / stride);
    if (is_max && fs == 2 && stride == 2 && w % 8 == 0 && h % 2 == 0 && total < (1L << 32) && ((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 15) == 0 && !env_int("FFGPU_NO_POOL2X2", 0)) {
        const unsigned owq = (unsigned)(w / 8), nquads = (unsigned)(total / 4);
        const unsigned m = owq == 1 ? 0u : (unsigned)(((1ULL << 32) + owq - 1) / owq);
        if ((unsigned long long)nquads * owq < (1ULL << 32))         // (umulhi division exact)
        {
        hipLaunchKernelGGL(k_pool2x2, dim3((nquads + 255) / 256), dim3(256), 0, s, in, out, nquads, owq, m, w);
        LAUNCH_OK("pool2x2");
        return 0;
        }
    }
    hipLaunchKernelGGL(k_pool, dim3(grid_for(total, 256)), dim3(256), 0, s, in, out, planes, w, h, fs, stride, is_max);
    LAUNCH_OK("pool");
    return 0;
}

int ffgpu_launch_spp(const float *in, float *const out[3], const int fs[3], int n, long planes, int w, int h, hipStream_t s)
{
    if (n < 1 || n > 3 || (long)w * h > 8192) { ffgpu_set_error("spp: bad pool count / plane size"); return -1; }
    SppP p; p.in = in; p.n = n; p.planes = planes; p.w = w; p.h = h;
    for (int k = 0; k < 3; k++) { p.out[k] = k < n ? out[k] : nullptr; p.fs[k] = k < n ? fs[k] : 1; }
    p.cascade = 1;
    for (int k = 0, prev = 1; k < n; prev = fs[k], k++) if (!(fs[k] & 1) || fs[k] <= prev) p.cascade = 0;
    hipLaunchKernelGGL(k_spp, dim3((unsigned)std::min((planes + 1) / 2, 4096L)), dim3(256), (size_t)4 * w * h * sizeof(float), s, p);
    LAUNCH_OK("spp");
    return 0;
}

int ffgpu_launch_upsample(const float *in, float *out, long planes, int w, int h, int stride, hipStream_t s)
{
    const long total = planes * h * stride * w * stride, n_in = planes * h * w;
    if (n_in * std::max(w, h) < (1L << 32) && total < (1L << 32)) {      // umulhi division exact, offsets fit 32 bits
        auto magic = [](long dv) { return dv == 1 ? 0u : (unsigned)(((1ULL << 32) + (unsigned long long)dv - 1) / (unsigned long long)dv); };
        hipLaunchKernelGGL(k_upsample32, dim3(grid_for(n_in, 256)), dim3(256), 0, s, in, out, (unsigned)n_in, w, h, stride, magic(w), magic(h));
        LAUNCH_OK("upsample");
        return 0;
    }
    hipLaunchKernelGGL(k_upsample, dim3(grid_for(total, 256)), dim3(256), 0, s, in, out, planes, w, h, stride);
    LAUNCH_OK("upsample");
    return 0;
}

int ffgpu_launch_add_act(const float *a, const float *b, float *out, long n, int act, hipStream_t s)
{
    hipLaunchKernelGGL(k_add_act, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, s, a, b, out, n, act);
    LAUNCH_OK("add_act");
    return 0;
}

// position-dependent 64-bit hash of a tensor's bits: sum over i of mix(bits[i], i) mod 2^64 -- integer addition, so the result does not depend on the
// order the threads arrive in (ffgpu_exec_hash_layers: every bit of every frame of a layer in one number, compared on the host)
__global__ void k_hash64(const unsigned *x, long n, unsigned long long *out)
{
    unsigned long long h = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        unsigned long long v = ((unsigned long long)x[i] << 32 | (unsigned long long)(unsigned)i) + 0x9e3779b97f4a7c15ull * (unsigned long long)(i >> 32);
        v ^= v >> 30; v *= 0xbf58476d1ce4e5b9ull; v ^= v >> 27; v *= 0x94d049bb133111ebull; v ^= v >> 31;      // splitmix64 finaliser
        h += v;
    }
    for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, h);
}
int ffgpu_launch_hash64(const float *x, long n, unsigned long long *out, hipStream_t s)
{
    hipLaunchKernelGGL(k_hash64, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, s, reinterpret_cast<const unsigned *>(x), n, out);
    LAUNCH_OK("hash64");
    return 0;
}

int ffgpu_launch_copy(const float *src, float *dst, long n, hipStream_t s)
{
    hipLaunchKernelGGL(k_copy, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, s, src, dst, n);
    LAUNCH_OK("copy");
    return 0;
}

int ffgpu_launch_input_bgr(const unsigned char *bgr, float *out, int N, int w, int h, int W, int H,
                           int sw, int sh, int s1, int s2, const float mean[3], const float norm[3], hipStream_t s)
{
    InputP p;
    for (int i = 0; i < 3; i++) { p.mean[i] = mean[i]; p.norm[i] = norm[i]; }
    if (W % 4 == 0 && N <= 65535 && (long)W * H < (1L << 31)) {
        hipLaunchKernelGGL(k_input_bgr4, dim3((unsigned)(((long)(W / 4) * H + 255) / 256), (unsigned)N), dim3(256), 0, s, bgr, out, w, h, W, H, sw, sh, s1, s2, p);
        LAUNCH_OK("input_bgr4");
        return 0;
    }
    hipLaunchKernelGGL(k_input_bgr, dim3(grid_for((long)N * H * W, 256)), dim3(256), 0, s, bgr, out, N, w, h, W, H, sw, sh, s1, s2, p);
    LAUNCH_OK("input_bgr");
    return 0;
}

int ffgpu_launch_yolo(const YoloHead &hd, int N, int netw, int neth, BBOX *cand, int *cand_key, int *ncand, int cap, int *ring_ctr, hipStream_t s)
{
    const long total = (long)N * 3 * hd.w * hd.h;
    hipLaunchKernelGGL(k_yolo, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, hd, N, netw, neth, cand, cand_key, ncand, cap, ring_ctr);
    LAUNCH_OK("yolo");
    return 0;
}

bool ffgpu_nms_in_lds(int cap_pow2)
{
    if (cap_pow2 > FFGPU_NMS_LDS_CAP) return false;
    int dev = 0, lds = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || lds <= 0) { (void)hipGetLastError(); lds = 64 * 1024; }
    return (size_t)13 * cap_pow2 <= (size_t)lds;      // (gfx950: 160 KB -- up to 8192 slots; a 64 KB part: 4096)
}

int ffgpu_launch_nms(const BBOX *cand, const int *cand_key, int *ncand, int cap, BBOX *full, void *scratch,
                     ffgpu_frame_dets *dets, ffgpu_frame_dets *dets_host, const int *ring_ctr, int N,
                     float thresh, int use_min, const ExecParams *prm, hipStream_t s)
{
    int p2 = 1;
    while (p2 < cap) p2 <<= 1;
    if (!ffgpu_nms_in_lds(p2)) {
        if (!scratch) { ffgpu_set_error("nms: %d candidate slots per frame need the global scratch buffer", cap); return -1; }
        hipLaunchKernelGGL(k_nms<true>, dim3(N), dim3(256), 0, s, cand, cand_key, ncand, cap, p2, full, (unsigned char *)scratch,
                           dets, dets_host, ring_ctr, thresh, use_min, prm);
    } else {
        const size_t lds = (size_t)13 * p2;
        if (lds_allow((const void *)k_nms<false>, lds > 64 * 1024 ? (size_t)13 * FFGPU_NMS_LDS_CAP : lds, "nms")) return -1;
        hipLaunchKernelGGL(k_nms<false>, dim3(N), dim3(256), lds, s, cand, cand_key, ncand, cap, p2, full, nullptr,
                           dets, dets_host, ring_ctr, thresh, use_min, prm);
    }
    LAUNCH_OK("nms");
    return 0;
}

int ffgpu_launch_set_params(ExecParams *d_prm, const ExecParams &v, hipStream_t s)
{
    hipLaunchKernelGGL(k_set_params, dim3(1), dim3(1), 0, s, d_prm, v);
    LAUNCH_OK("set_params");
    return 0;
}

int ffgpu_launch_clear(int *ncand, int N, int *ring_ctr, hipStream_t s)
{
    hipLaunchKernelGGL(k_clear, dim3((N + 255) / 256), dim3(256), 0, s, ncand, N, ring_ctr);
    LAUNCH_OK("clear");
    return 0;
}

// ---- compact form of the detection records for the multi-GPU gather: a step's `batch` fixed-size records (3088 bytes
// each, almost all of it unused box slots) become
//     int total, over, batch, cap | { int count, ncand, overflow, nfull } x batch | BBOX box[cap]
// with each frame's boxes packed behind each other (frame order: a frame's first box sits at the sum of the counts before it).  A step whose frames
// hold more than `cap` boxes together keeps the first cap of them and says so (`over`, and `overflow |= 2` on the frames
// that lost boxes).  One workgroup per step; 25 KB instead of 198 KB per step at batch 64, cap 1024.
__global__ void __launch_bounds__(256) k_pack_records(const ffgpu_frame_dets *recs, long slot_stride_recs, int batch, int cap, unsigned char *out, long out_stride)
{
    __shared__ int s_first[1025];
    const ffgpu_frame_dets *r = recs + (long)blockIdx.x * slot_stride_recs;
    unsigned char *o = out + (long)blockIdx.x * out_stride;
    int *hdr = reinterpret_cast<int *>(o);
    int *fr = hdr + 4;
    BBOX *box = reinterpret_cast<BBOX *>(fr + 4 * batch);
    if (threadIdx.x == 0) {                                       // batch <= 1024: a serial prefix sum is a microsecond
        int t = 0;
        for (int n = 0; n < batch; n++) { s_first[n] = t; t += r[n].count; }
        s_first[batch] = t;
        hdr[0] = min(t, cap); hdr[1] = t > cap; hdr[2] = batch; hdr[3] = cap;
    }
    __syncthreads();
    for (int n = threadIdx.x; n < batch; n += blockDim.x) {
        const int first = s_first[n], cnt = r[n].count, kept = max(0, min(cnt, cap - first));
        fr[4 * n] = kept; fr[4 * n + 1] = r[n].ncand; fr[4 * n + 2] = r[n].overflow | (kept < cnt ? 2 : 0); fr[4 * n + 3] = r[n].nfull;
    }
    // boxes: 6 floats each, copied as 32-bit words by the whole workgroup
    for (int n = 0; n < batch; n++) {
        const int first = s_first[n], kept = max(0, min(r[n].count, cap - first));
        const float *src = reinterpret_cast<const float *>(r[n].box);
        float *dst = reinterpret_cast<float *>(box + first);
        for (int i = threadIdx.x; i < kept * 6; i += blockDim.x) dst[i] = src[i];
    }
}

extern "C" size_t ffgpu_packed_records_bytes(int batch, int cap) { return ((size_t)16 + 16 * (size_t)batch + sizeof(BBOX) * (size_t)cap + 15) & ~(size_t)15; }

extern "C" int ffgpu_pack_records(const void *d_records, int nslots, long slot_stride_records, int batch, int cap, void *d_out, void *stream)
{
    if (!d_records || !d_out || nslots < 1 || batch < 1 || batch > 1024 || cap < 1) { ffgpu_set_error("pack_records: bad arguments"); return -1; }
    hipLaunchKernelGGL(k_pack_records, dim3(nslots), dim3(256), 0, (hipStream_t)stream, (const ffgpu_frame_dets *)d_records, slot_stride_records,
                       batch, cap, (unsigned char *)d_out, (long)ffgpu_packed_records_bytes(batch, cap));
    if (hipGetLastError() != hipSuccess) { ffgpu_set_error("pack_records: launch failed"); return -1; }
    return 0;
}


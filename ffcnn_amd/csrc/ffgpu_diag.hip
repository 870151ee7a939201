// ffgpu_diag.hip -- lab equipment (include/ffcnn_hip_diag.h): HBM stream calibration and pipe probes.
// Built into its OWN library, libffcnn_hip_diag.so; the product library exports none of this.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "ffcnn_hip_diag.h"

static thread_local char g_diag_err[256] = "";
static void ffgpu_set_error(const char *fmt, const char *a = "") { snprintf(g_diag_err, sizeof g_diag_err, fmt, a); fprintf(stderr, "ffgpu_diag: %s\n", g_diag_err); }

// ---------------------------------------------------------------------------
// HBM stream calibration (ffgpu_membench)
typedef float mb4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(256) k_membench(mb4 *dst, const mb4 *src, long n4)
{
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x, gsz = (long)gridDim.x * blockDim.x;
    if (MODE == 0) for (long i = gid; i < n4; i += gsz) dst[i] = src[i];
    if (MODE == 1) for (long i = gid; i < n4; i += gsz) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
    if (MODE == 2) {
        mb4 a = { 0.f, 0.f, 0.f, 0.f };
        for (long i = gid; i < n4; i += gsz) a += src[i];
        if (a.x + a.y + a.z + a.w == 12345.678f) dst[gid] = a;      // keep the loads alive
    }
    if (MODE == 3) { const mb4 v = { 1.f, 2.f, 3.f, 4.f }; for (long i = gid; i < n4; i += gsz) dst[i] = v; }
    if (MODE == 5 || MODE == 6 || MODE == 7) {
        // MODE 5: every WAVE owns one contiguous span and walks it with 4 x 1 KiB pieces in flight
        // MODE 6: every BLOCK owns one contiguous span; its waves interleave at 1 KiB granularity
        // MODE 7: as 5 with non-temporal loads and stores
        const int lane = threadIdx.x & 63;
        const long nw = (long)gridDim.x * (blockDim.x >> 6), w = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        long b, e, step;
        if (MODE == 6) {
            const long per = ((n4 + gridDim.x - 1) / gridDim.x + 255) & ~255L;
            b = blockIdx.x * per + (threadIdx.x >> 6) * 64; e = min(b - (threadIdx.x >> 6) * 64 + per, n4); step = blockDim.x;
        } else {
            const long per = ((n4 + nw - 1) / nw + 63) & ~63L;
            b = w * per; e = min(b + per, n4); step = 64;
        }
        long i = b + lane;
        for (; i + 3 * step < e; i += 4 * step) {
            mb4 a0, a1, a2, a3;
            if (MODE == 7) { a0 = __builtin_nontemporal_load(src + i); a1 = __builtin_nontemporal_load(src + i + step);
                             a2 = __builtin_nontemporal_load(src + i + 2 * step); a3 = __builtin_nontemporal_load(src + i + 3 * step); }
            else { a0 = src[i]; a1 = src[i + step]; a2 = src[i + 2 * step]; a3 = src[i + 3 * step]; }
            if (MODE == 7) { __builtin_nontemporal_store(a0, dst + i); __builtin_nontemporal_store(a1, dst + i + step);
                             __builtin_nontemporal_store(a2, dst + i + 2 * step); __builtin_nontemporal_store(a3, dst + i + 3 * step); }
            else { dst[i] = a0; dst[i + step] = a1; dst[i + 2 * step] = a2; dst[i + 3 * step] = a3; }
        }
        for (; i < e; i += step) dst[i] = src[i];
    }
    if (MODE == 4) {
        long i = gid;
        for (; i + 3 * gsz < n4; i += 4 * gsz) {
            const mb4 a = src[i], b = src[i + gsz], c = src[i + 2 * gsz], d = src[i + 3 * gsz];
            dst[i] = a; dst[i + gsz] = b; dst[i + 2 * gsz] = c; dst[i + 3 * gsz] = d;
        }
        for (; i < n4; i += gsz) dst[i] = src[i];
    }
}

// ---------------------------------------------------------------------------
// pipe probe (diagnostics): how the matrix cores and the vector ALU of a SIMD share time.  Every wave runs `iters` trips
// of [NM independent v_mfma_f32_16x16x4_f32] + [NV independent v_fma_f32]; blocks * 4 waves are launched so the
// caller controls waves per SIMD.  Result: microseconds per launch.
template <int NM, int NV>
__global__ void __launch_bounds__(256) k_pipe_probe(float *out, int iters)
{
    typedef float pv4 __attribute__((ext_vector_type(4)));
    pv4 acc[16];
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { acc[i] = (pv4){ 0.f, 0.f, 0.f, 0.f }; v[i] = (float)threadIdx.x + i; }
    const float a = 1.0f + threadIdx.x * 1e-6f, b = 0.999f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NM; i++) acc[i & 15] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i & 15], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NV; i++) v[i & 15] = fmaf(v[i & 15], b, a);
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) r += acc[i].x + acc[i].y + acc[i].z + acc[i].w + v[i];
    if (r == 12345.678f) out[threadIdx.x] = r;                  // keep the work alive
}

// ---- does vector-ALU work hide in the shadow of an MFMA?  Hand-placed instruction streams (inline asm, nothing for the
// compiler to repack or reorder): MODE 0: 16 x MFMA; 1: 16 x (MFMA, NS plain v_fma_f32); 2: 16 x (MFMA, NS/2 v_pk_fma_f32);
// 3: 16 x NS v_fma_f32 alone; 4: 16 x NS/2 v_pk_fma_f32 alone.  All operands independent of each other.
template <int MODE, int NS>
__global__ void __launch_bounds__(256) k_pipe_probe2(float *out, int iters)
{
    typedef float pv4 __attribute__((ext_vector_type(4)));
    typedef float pv2 __attribute__((ext_vector_type(2)));
    pv4 acc[16];
    float v[8];
    pv2 w[4];
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = (pv4){ 0.f, 0.f, 0.f, 0.f };
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = (float)threadIdx.x + i;
#pragma unroll
    for (int i = 0; i < 4; i++) w[i] = (pv2){ (float)threadIdx.x + i, 1.f };
    const float a = 1.0f + threadIdx.x * 1e-6f, b = 0.999f;
    const pv2 b2 = { b, b }, a2 = { a, a };
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (MODE <= 2) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            if (MODE == 1 || MODE == 3) {
#pragma unroll
                for (int j = 0; j < NS; j++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(b), "v"(a));
            }
            if (MODE == 2 || MODE == 4) {
#pragma unroll
                for (int j = 0; j < NS / 2; j++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(w[j & 3]) : "v"(b2), "v"(a2));
            }
        }
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) r += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
#pragma unroll
    for (int i = 0; i < 8; i++) r += v[i];
#pragma unroll
    for (int i = 0; i < 4; i++) r += w[i].x + w[i].y;
    if (r == 12345.678f) out[threadIdx.x] = r;
}

extern "C" float ffgpu_pipe_probe2(int mode, int ns, int blocks, int iters, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    static float *d_out = nullptr;
    if (!d_out && hipMalloc(&d_out, 256 * sizeof(float)) != hipSuccess) return -1.f;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.f;
    for (int rep = 0; rep < 3; rep++) {
        if (rep == 1) (void)hipEventRecord(e0, s);
#define PP2(M, N) if (mode == M && ns == N) hipLaunchKernelGGL((k_pipe_probe2<M, N>), dim3(blocks), dim3(256), 0, s, d_out, iters); else
        PP2(0, 0) PP2(1, 2) PP2(1, 4) PP2(1, 6) PP2(1, 8) PP2(2, 2) PP2(2, 4) PP2(2, 6) PP2(2, 8) PP2(3, 4) PP2(3, 8) PP2(4, 4) PP2(4, 8)
        { ffgpu_set_error("pipe_probe2: unsupported mix"); return -1.f; }
#undef PP2
    }
    (void)hipEventRecord(e1, s);
    if (hipEventSynchronize(e1) != hipSuccess) return -1.f;
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return ms * 1000.f / 2;
}

// ---- the same question for the BF16 matrix cores (round 5): do v_mfma_f32_16x16x32_bf16 and vector-ALU work overlap -- inside a wave, between the
// waves of a SIMD?  MODE 0: 16 x MFMA; 2: 16 x (MFMA, NS/2 v_pk_fma_f32); 4: the packed FMAs alone; 5: waves 0-3 of the workgroup run MODE 0's
// stream, waves 4-7 MODE 4's (cross-wave overlap: two waves of different kinds per SIMD at 8 waves per workgroup); 6: as 5 with the fp32 MFMA.
template <int MODE, int NS>
__global__ void __launch_bounds__(512) k_pipe_probe3(float *out, int iters)
{
    typedef float pv4 __attribute__((ext_vector_type(4)));
    typedef float pv2 __attribute__((ext_vector_type(2)));
    typedef __bf16 pb8 __attribute__((ext_vector_type(8)));
    typedef unsigned pu4 __attribute__((ext_vector_type(4)));
    pv4 acc[16];
    pv2 w[4];
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = (pv4){ 0.f, 0.f, 0.f, 0.f };
#pragma unroll
    for (int i = 0; i < 4; i++) w[i] = (pv2){ (float)threadIdx.x + i, 1.f };
    const float a = 1.0f + threadIdx.x * 1e-6f, b = 0.999f;
    const pv2 b2 = { b, b }, a2 = { a, a };
    const unsigned h = 0x3f803f80u ^ (threadIdx.x * 0x00010001u & 0x007f007fu);
    const pb8 A8 = __builtin_bit_cast(pb8, (pu4){ h, h ^ 0x00110022u, h ^ 0x00330044u, h ^ 0x00550066u }), B8 = __builtin_bit_cast(pb8, (pu4){ h ^ 0x00010001u, h, h ^ 0x00070003u, h });
    // (waves 0-3 / 4-7 of a 512-thread workgroup: one of each kind per SIMD; the role is a scalar and decided ONCE, outside the loops)
    const int role = (MODE == 5 || MODE == 6) ? (__builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) & 1) : 0;
    auto mfmas = [&]() {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (MODE == 6) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(A8), "v"(B8));
        }
    };
    auto pks = [&]() {
#pragma unroll
        for (int i = 0; i < 16; i++)
#pragma unroll
            for (int j = 0; j < NS / 2; j++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(w[(i * (NS / 2) + j) & 3]) : "v"(b2), "v"(a2));
    };
    if (MODE == 0) { for (int it = 0; it < iters; it++) mfmas(); }
    else if (MODE == 4) { for (int it = 0; it < iters; it++) pks(); }
    else if (MODE == 2) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(A8), "v"(B8));
#pragma unroll
                for (int j = 0; j < NS / 2; j++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(w[(i * (NS / 2) + j) & 3]) : "v"(b2), "v"(a2));
            }
        }
    } else if (role == 0) { for (int it = 0; it < iters; it++) mfmas(); }
    else { for (int it = 0; it < iters; it++) pks(); }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) r += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
#pragma unroll
    for (int i = 0; i < 4; i++) r += w[i].x + w[i].y;
    if (r == 12345.678f) out[threadIdx.x] = r;
}

extern "C" float ffgpu_pipe_probe3(int mode, int ns, int blocks, int threads, int iters, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    static float *d_out = nullptr;
    if (!d_out && hipMalloc(&d_out, 512 * sizeof(float)) != hipSuccess) return -1.f;
    if (threads != 256 && threads != 512) { ffgpu_set_error("pipe_probe3: 256 or 512 threads"); return -1.f; }
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.f;
    for (int rep = 0; rep < 3; rep++) {
        if (rep == 1) (void)hipEventRecord(e0, s);
#define PP3(M, N) if (mode == M && ns == N) hipLaunchKernelGGL((k_pipe_probe3<M, N>), dim3(blocks), dim3(threads), 0, s, d_out, iters); else
        PP3(0, 0) PP3(2, 2) PP3(2, 4) PP3(2, 8) PP3(2, 16) PP3(4, 2) PP3(4, 4) PP3(4, 8) PP3(4, 16) PP3(5, 4) PP3(5, 8) PP3(5, 16) PP3(6, 4) PP3(6, 8) PP3(6, 16)
        { ffgpu_set_error("pipe_probe3: unsupported mix"); return -1.f; }
#undef PP3
    }
    (void)hipEventRecord(e1, s);
    if (hipEventSynchronize(e1) != hipSuccess) return -1.f;
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return ms * 1000.f / 2;
}

// ---- MFMA-only floor of BASELINE config[2]'s split-bf16 GEMM (round 6): the 48-MFMA chunk pattern of k_pw_x3t (ffgpu_pw_x3t.inc:
// v_mfma_f32_32x32x16_bf16, 2 x 4 accumulators = 128 registers, six weight and twelve input fragments per chunk) with NOTHING else -- no loads, no split,
// no LDS, no barrier, no stores; the fragments are read once from `frag` ([18][256] x 16 bytes: real split operands, so the matrix cores toggle the way
// they do in the kernel).  Wrong results by construction: it is the time the chip needs for the launch's MFMAs alone at the clock its power limit grants.
__global__ void __launch_bounds__(256, 2) k_mfma_floor(const uint4 *frag, float *out, int trips)
{
    typedef __bf16 fb8 __attribute__((ext_vector_type(8)));
    typedef float f16v __attribute__((ext_vector_type(16)));
    fb8 fa[2][3], fb[4][3];
#pragma unroll
    for (int i = 0; i < 6; i++) fa[i / 3][i % 3] = __builtin_bit_cast(fb8, frag[i * 256 + threadIdx.x]);
#pragma unroll
    for (int i = 0; i < 12; i++) fb[i / 3][i % 3] = __builtin_bit_cast(fb8, frag[(6 + i) * 256 + threadIdx.x]);
    f16v acc[2][4];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[r][j][e] = 0.f;
    constexpr int WP[6] = { 0, 1, 2, 0, 1, 0 }, XP[6] = { 2, 1, 0, 1, 0, 0 };
    for (int t = 0; t < trips; t++) {
#pragma unroll
        for (int jp = 0; jp < 4; jp += 2)
#pragma unroll
            for (int m = 0; m < 6; m++)
#pragma unroll
                for (int jj = 0; jj < 2; jj++)
#pragma unroll
                    for (int r = 0; r < 2; r++)
                        acc[r][jp + jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[r][WP[m]], fb[jp + jj][XP[m]], acc[r][jp + jj], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 6; i++) asm volatile("" : "+v"(fa[i / 3][i % 3]));          // (the operands stay opaque: nothing is hoisted or folded across trips)
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) s += acc[r][j][e];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

extern "C" float ffgpu_mfma_floor(const void *d_frag, int trips, int blocks, int iters, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    static float *d_out = nullptr;
    if (!d_out && hipMalloc(&d_out, 256 * sizeof(float)) != hipSuccess) return -1.f;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.f;
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL(k_mfma_floor, dim3(blocks), dim3(256), 0, s, (const uint4 *)d_frag, d_out, trips);     // warm-up: the clock settles under the load
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL(k_mfma_floor, dim3(blocks), dim3(256), 0, s, (const uint4 *)d_frag, d_out, trips);
    (void)hipEventRecord(e1, s);
    if (hipEventSynchronize(e1) != hipSuccess) return -1.f;
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return ms * 1000.f / iters;
}

extern "C" float ffgpu_pipe_probe(int n_mfma, int n_valu, int blocks, int iters, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    static float *d_out = nullptr;
    if (!d_out && hipMalloc(&d_out, 256 * sizeof(float)) != hipSuccess) return -1.f;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.f;
    for (int rep = 0; rep < 3; rep++) {
        if (rep == 1) (void)hipEventRecord(e0, s);
        if (n_mfma == 16 && n_valu == 0)       hipLaunchKernelGGL((k_pipe_probe<16, 0>), dim3(blocks), dim3(256), 0, s, d_out, iters);
        else if (n_mfma == 0 && n_valu == 64)  hipLaunchKernelGGL((k_pipe_probe<0, 64>), dim3(blocks), dim3(256), 0, s, d_out, iters);
        else if (n_mfma == 16 && n_valu == 64) hipLaunchKernelGGL((k_pipe_probe<16, 64>), dim3(blocks), dim3(256), 0, s, d_out, iters);
        else if (n_mfma == 16 && n_valu == 128) hipLaunchKernelGGL((k_pipe_probe<16, 128>), dim3(blocks), dim3(256), 0, s, d_out, iters);
        else if (n_mfma == 0 && n_valu == 128) hipLaunchKernelGGL((k_pipe_probe<0, 128>), dim3(blocks), dim3(256), 0, s, d_out, iters);
        else { ffgpu_set_error("pipe_probe: unsupported mix"); return -1.f; }
    }
    (void)hipEventRecord(e1, s);
    if (hipEventSynchronize(e1) != hipSuccess) return -1.f;
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return ms * 1000.f / 2;
}

extern "C" float ffgpu_membench(void *d_dst, const void *d_src, size_t bytes, int mode, int blocks, int iters, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    const long n4 = (long)(bytes / 16);
    if (!d_dst || !d_src || n4 < 1 || blocks < 1 || iters < 1) { ffgpu_set_error("membench: bad arguments"); return -1.f; }
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.f;
    for (int it = 0; it < iters + 2; it++) {
        if (it == 2) (void)hipEventRecord(e0, s);
        switch (mode) {
        case 0: hipLaunchKernelGGL(k_membench<0>, dim3(blocks), dim3(256), 0, s, (mb4 *)d_dst, (const mb4 *)d_src, n4); break;
        case 1: hipLaunchKernelGGL(k_membench<1>, dim3(blocks), dim3(256), 0, s, (mb4 *)d_dst, (const mb4 *)d_src, n4); break;
        case 2: hipLaunchKernelGGL(k_membench<2>, dim3(blocks), dim3(256), 0, s, (mb4 *)d_dst, (const mb4 *)d_src, n4); break;
        case 3: hipLaunchKernelGGL(k_membench<3>, dim3(blocks), dim3(256), 0, s, (mb4 *)d_dst, (const mb4 *)d_src, n4); break;
        case 5: hipLaunchKernelGGL(k_membench<5>, dim3(blocks), dim3(256), 0, s, (mb4 *)d_dst, (const mb4 *)d_src, n4); break;
        case 6: hipLaunchKernelGGL(k_membench<6>, dim3(blocks), dim3(256), 0, s, (mb4 *)d_dst, (const mb4 *)d_src, n4); break;
        case 7: hipLaunchKernelGGL(k_membench<7>, dim3(blocks), dim3(256), 0, s, (mb4 *)d_dst, (const mb4 *)d_src, n4); break;
        default: hipLaunchKernelGGL(k_membench<4>, dim3(blocks), dim3(256), 0, s, (mb4 *)d_dst, (const mb4 *)d_src, n4); break;
        }
    }
    (void)hipEventRecord(e1, s);
    if (hipEventSynchronize(e1) != hipSuccess) { ffgpu_set_error("membench: %s", hipGetErrorString(hipGetLastError())); return -1.f; }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return ms * 1000.f / iters;
}

// ---- the slot tables of the split-bf16 expand GEMM (ffgpu_x3_terms.h), readable from the host: tests/test_x3_tables.py checks that
// every partial product (weight part i, input part j, channel pair) with i + j <= 2 occurs exactly once and nothing else does
#include "ffgpu_x3_terms.h"
// Clock probe: one wave that does nothing but read the constant 100 MHz counter (s_memrealtime) and the shader-clock counter (s_memtime) `samples`
// times, `gap` sleep units apart.  Launched on its own stream beside a workload it reports the shader clock the chip HOLDS under that workload
// (the issue-bound model of the full net has to be priced at that clock, not at the 2.4 GHz of the peak table).
__global__ void k_clock_probe(unsigned long long *out, int samples, int gap)
{
    for (int i = 0; i < samples; i++) {
        const unsigned long long rt = __builtin_amdgcn_s_memrealtime(), st = __builtin_amdgcn_s_memtime();
        if (threadIdx.x == 0) { out[2 * i] = rt; out[2 * i + 1] = st; }
        for (int g = 0; g < gap; g++) __builtin_amdgcn_s_sleep(127);
    }
}
extern "C" int ffgpu_clock_probe(unsigned long long *d_out, int samples, int gap, void *stream)
{
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, d_out, samples, gap);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int ffgpu_diag_x3_term(int ks1, int m, int d, int out[3])
{
    if ((ks1 != 2 && ks1 != 4 && ks1 != 6 && ks1 != 12) || m < 0 || m >= irbw_x3_nm(ks1) || d < 0 || d > 3) return -1;
    const IrbwX3Term t = irbw_x3_term(ks1, m, d);
    out[0] = t.wp; out[1] = t.xp; out[2] = t.pair;
    return irbw_x3_nm(ks1);
}
extern "C" int ffgpu_diag_xl_op(int m) { return irbw_xl_op(m); }


/*
 * ffcnn_hip_diag.h -- lab equipment, NOT part of the product ABI.
 *
 * These entry points live in their own library, ffcnn_amd/lib/libffcnn_hip_diag.so (built by the same Makefile from
 * ffgpu_diag.hip): HBM stream calibration and the matrix-core / vector-ALU pipe probes that DESIGN.md quotes.  bench.py
 * loads it (if present) to report the same-box copy rate next to the 8 TB/s spec; tools/ use the probes.
 * libffcnn_hip.so exports none of them.  The launch-dropping switches FFGPU_DBG_SKIP / FFGPU_DBG_KEEP
 * (tools/ablate_layers.py) exist only in a `make DIAG=1` build of the product library.
 */
#ifndef FFCNN_AMD_FFCNN_HIP_DIAG_H
#define FFCNN_AMD_FFCNN_HIP_DIAG_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* HBM stream calibration on this GPU: mean microseconds per pass over `bytes`
 * (16-byte lanes, grid-stride, `blocks` workgroups of 256).  mode 0: copy,
 * 1: copy with non-temporal loads+stores, 2: read only, 3: write only,
 * 4: copy, 4 independent 16-byte loads in flight per lane.  Used by bench.py
 * to report the measured copy ceiling next to the 8 TB/s spec. */
float ffgpu_membench(void *d_dst, const void *d_src, size_t bytes, int mode, int blocks, int iters, void *stream);
/* Pipe probe: every wave of `blocks` x 4 runs `iters` trips of n_mfma independent v_mfma_f32_16x16x4_f32 plus n_valu
 * independent v_fma_f32 (supported mixes: 16/0, 0/64, 16/64, 16/128, 0/128); returns microseconds per launch.  Shows
 * whether matrix-core and vector-ALU work of one wave / of several waves of a SIMD overlap. */
float ffgpu_pipe_probe(int n_mfma, int n_valu, int blocks, int iters, void *stream);
/* the same question with hand-placed instruction streams: mode 0 = 16 MFMAs per trip; 1 = each followed by `ns` plain
 * v_fma_f32; 2 = by ns/2 v_pk_fma_f32; 3 / 4 = the vector instructions alone.  Microseconds per launch. */
float ffgpu_pipe_probe2(int mode, int ns, int blocks, int iters, void *stream);
/* the bf16 matrix cores (v_mfma_f32_16x16x32_bf16): mode 0 = 16 MFMAs per trip; 2 = each followed by ns/2 v_pk_fma_f32; 4 = the packed FMAs alone;
 * 5 = waves 0-3 of a 512-thread workgroup run mode 0's stream, waves 4-7 mode 4's (cross-wave overlap); 6 = as 5 with v_mfma_f32_16x16x4_f32.
 * `threads` = 256 or 512 per workgroup.  Microseconds per launch. */
float ffgpu_pipe_probe3(int mode, int ns, int blocks, int threads, int iters, void *stream);
/* MFMA-only floor of k_pw_x3t (round 6): `blocks` x 4 waves each run `trips` x the kernel's 48-MFMA chunk pattern (v_mfma_f32_32x32x16_bf16, 128 accumulator
 * registers) on the 18 fragments in d_frag ([18][256] x 16 bytes) and nothing else; microseconds per launch after `iters` warm-up launches.  Wrong results by
 * construction -- it prices the launch's MFMAs alone at the clock the power limit grants (tools/mfma_floor.py). */
float ffgpu_mfma_floor(const void *d_frag, int trips, int blocks, int iters, void *stream);
/* host only: the slot tables of the split-bf16 ("X3") expand GEMM of the fused blocks (ffcnn_amd/csrc/ffgpu_x3_terms.h): dword d of MFMA m
 * for a lane with ks1 input channels holds weight part out[0] (-1: empty), input part out[1], channel pair out[2]; returns the number
 * of MFMAs per (strip, pixel) or -1.  ffgpu_diag_xl_op: which 16-byte window of the LDS image MFMA m of the 48-channel form reads. */
int ffgpu_diag_x3_term(int ks1, int m, int d, int out[3]);
int ffgpu_diag_xl_op(int m);
/* Clock probe: one wave writes `samples` pairs (s_memrealtime: constant 100 MHz, s_memtime: shader clock) into d_out (2 x samples u64), `gap` x ~8 k cycles of
 * s_sleep apart.  On its own stream beside a workload: the shader clock the chip holds under it = d(s_memtime) / d(s_memrealtime) x 100 MHz. */
int ffgpu_clock_probe(unsigned long long *d_out, int samples, int gap, void *stream);


#ifdef __cplusplus
}
#endif
#endif

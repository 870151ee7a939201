/*
 * conv.h -- Level-1 operator plug-in of the ffcnn forward path.
 *
 * The reference selects its convolution implementation at LINK time: exactly
 * one conv-vN.c object provides the single symbol below (reference conv.h:4-7,
 * build.sh:48) and ffcnn.c:374-379 is its only caller.  libffcnn_hip.so
 * exports the same symbol, so `gcc ffcnn.c bmpfile.c -lffcnn_hip` links the
 * unmodified reference net code against the MI355X kernels.
 *
 * Contract (identical to every conv-vN.c):
 *   in      host fp32, ic x ih x iw planar; group g reads channels [g*ic/ig, ...)
 *   filt    host fp32, fn rows of ALIGN(fs*fs*ic/ig,4)+4 floats; per row the taps
 *           in [ci][ky][kx] order, zero pad, then scale' at +K4 and bias' at +K4+1
 *   out     host fp32, oc x oh x ow planar, fully overwritten;
 *           out = act(scale' * sum(in * tap) + bias'), zero padding of `ipad`
 *   act     0 linear, 1 relu, 2 leaky(0.1), 3 sigmoid, anything else linear
 *   scratch/scratch_floats  callee-grown, caller-freed buffer of the reference
 *           (NET.cnntempbuf / cnnbufsize).  This implementation stages through
 *           device memory it owns and never touches *scratch.
 * Host pointers force an H2D + D2H round trip per call: this entry point is for
 * drop-in use and per-layer parity, not for throughput (use ffcnn_hip.h).
 * On a HIP failure it prints a diagnostic to stderr and leaves `out` untouched,
 * mirroring the reference's printf-and-return error path (conv-v6.c:509).
 *
 * Arithmetic: fp32 results within the repo's stated tolerance of the reference's
 * k-ordered chain (conv-v0.c:7-31) for every finite input.  The kernels AUTO
 * picks for dense 3x3 layers and large 1x1 layers compute each product from
 * three exact bf16 parts per operand on the bf16 matrix cores (24 significand
 * bits kept, fp32 accumulation: a summation ORDER, not a precision).
 * Non-finite inputs behave as in the reference (round 6): a +-Inf input gives
 * +-Inf with the sign of w * Inf where it meets non-zero weights, NaN where it
 * meets a zero weight or an opposite Inf; a NaN input gives NaN; no other output
 * is affected (tests/test_gpu_round5.py::test_x3_non_finite_lanes compares the
 * pattern with the oracle's output by output).  Sub-normal inputs and magnitudes
 * spread over 2^-30 .. 2^30 in one sum behave like fp32.
 */
#ifndef FFCNN_AMD_CONV_H
#define FFCNN_AMD_CONV_H

#ifdef __cplusplus
extern "C" {
#endif

void groupconv(float *in, float *filt, float *out,
               int iw, int ih, int ic, int ig, int ipad, int istride,
               int fs, int fn, int ow, int oh, int oc, int act,
               float **scratch, int *scratch_floats);

#ifdef __cplusplus
}
#endif
#endif /* FFCNN_AMD_CONV_H */

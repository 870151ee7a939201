// ffgpu_x3_terms.h -- slot tables of the split-bf16 ("X3") expand GEMM of the fused blocks (ffgpu_irb_wave.inc explains the scheme):
// which (weight part, input part, channel pair) sits in dword d of MFMA m for a lane that holds ks1 input channels.  Plain constexpr
// functions: shared by the kernels, the weight-packing kernel and the diagnostics library (tests check the tables on the host).
#ifndef FFGPU_X3_TERMS_H
#define FFGPU_X3_TERMS_H
#ifndef __host__
#define __host__
#define __device__
#endif
struct IrbwX3Term { int wp, xp, pair; };         // dword d (slots 2d, 2d + 1) of MFMA m: weight part, x part, channel pair (wp < 0: empty)
__host__ __device__ constexpr IrbwX3Term irbw_x3_term(int ks1, int m, int d)
{
    if (ks1 == 12) {                              // 36 dword slots in 9 MFMAs over FIVE operand windows of the lane's 20-dword image (irbw_xl_*)
        constexpr IrbwX3Term T12[9][4] = {
            { { 0, 0, 0 }, { 0, 0, 1 }, { 0, 0, 2 }, { 0, 0, 3 } },      // O0 = x0 a-d   x w0
            { { 1, 0, 0 }, { 1, 0, 1 }, { 1, 0, 2 }, { 1, 0, 3 } },      // O0            x w1
            { { 2, 0, 0 }, { 2, 0, 1 }, { 2, 0, 2 }, { 2, 0, 3 } },      // O0            x w2
            { { 0, 0, 4 }, { 0, 0, 5 }, { 0, 1, 0 }, { 0, 1, 1 } },      // O1 = x0 e f | x1 a b   x w0
            { { 1, 0, 4 }, { 1, 0, 5 }, { 1, 1, 0 }, { 1, 1, 1 } },      // O1            x w1
            { { 0, 1, 2 }, { 0, 1, 3 }, { 0, 1, 4 }, { 0, 1, 5 } },      // O2 = x1 c-f   x w0
            { { 1, 1, 2 }, { 1, 1, 3 }, { 1, 1, 4 }, { 1, 1, 5 } },      // O2            x w1
            { { 0, 2, 0 }, { 0, 2, 1 }, { 0, 2, 2 }, { 0, 2, 3 } },      // O3 = x2 a-d   x w0
            { { 0, 2, 4 }, { 0, 2, 5 }, { 2, 0, 4 }, { 2, 0, 5 } } };    // O4 = x2 e f | x0 e f   x (w0 | w2)
        return T12[m][d];
    }
    if (ks1 == 6) {                               // pairs a, b, c = 0, 1, 2 per part; 18 dword slots in 5 MFMAs (the last one half empty)
        constexpr IrbwX3Term T6[5][4] = {
            { { 0, 0, 0 }, { 0, 0, 1 }, { 0, 0, 2 }, { 0, 1, 0 } },
            { { 1, 0, 0 }, { 1, 0, 1 }, { 1, 0, 2 }, { 1, 1, 0 } },
            { { 0, 1, 1 }, { 0, 1, 2 }, { 0, 2, 0 }, { 0, 2, 1 } },
            { { 1, 1, 1 }, { 1, 1, 2 }, { 2, 0, 0 }, { 2, 0, 1 } },
            { { 2, 0, 2 }, { 0, 2, 2 }, { -1, 0, 0 }, { -1, 0, 0 } } };
        return T6[m][d];
    }
    if (ks1 == 4) {
        const int pair = d & 1, half = d >> 1;
        if (m == 0) return { 0, half, pair };
        if (m == 1) return { 1, half, pair };
        return half == 0 ? IrbwX3Term{ 2, 0, pair } : IrbwX3Term{ 0, 2, pair };
    }
    if (m == 0) return d == 0 ? IrbwX3Term{ 0, 0, 0 } : (d == 1 ? IrbwX3Term{ 1, 0, 0 } : (d == 2 ? IrbwX3Term{ 2, 0, 0 } : IrbwX3Term{ 0, 1, 0 }));
    return d == 0 ? IrbwX3Term{ 1, 1, 0 } : (d == 1 ? IrbwX3Term{ 0, 2, 0 } : IrbwX3Term{ -1, 0, 0 });
}
__host__ __device__ constexpr int irbw_x3_nm(int ks1) { return (ks1 * 6 + 7) / 8; }
// "XL" (48 input channels): the split tile does not fit the registers beside the accumulators, and every wave of a group-split workgroup
// holds the SAME tile -- so it lives once per workgroup in LDS, as ready-made B operands: per (strip, pixel, lane) 20 dwords =
// x0 a-f | x1 a-f | x2 a-f | x0 e f again, read as five 16-byte windows O0..O4; MFMA m of irbw_x3_term(12, m, .) multiplies window irbw_xl_op(m)
__host__ __device__ constexpr int irbw_xl_op(int m) { return m < 3 ? 0 : (m < 5 ? 1 : (m < 7 ? 2 : (m == 7 ? 3 : 4))); }
constexpr int IRBW_XL_DW = 20;
#endif

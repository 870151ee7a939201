// hazard_controls.hip -- POSITIVE CONTROLS for the two gfx950 hazards the product works around.  Lab equipment: built into
// ffcnn_amd/lib/libffcnn_hazard_lab.so by ffcnn_amd/csrc/Makefile, run by tests/test_gpu_hazard_controls.py, which EXPECTS the wrong results (xfail,
// non-strict, counts printed): a box / ROCm release on which a control stops failing is visible, and "hardware, not compiler" has a file anybody can run --
// both instruction pairs are written in inline assembly, the compiler places nothing between them.
//
//   (A) DESIGN.md 5.12 (b), profiles/r05_e_buffer_store_hazard.txt, tools/isa_lint.py rule 4:
//         buffer_store_dwordx4 v[2:5], voff, s[rsrc], s_off offen        <- SGPR soffset
//         v_mov_b32 v3, poison                                            <- overwrites data register 1, zero wait states
//       the store must still write the OLD v3.  LLVM's hazard recogniser puts the "VMEM store of more than 8 bytes" wait state there for global stores and for
//       buffer stores with an IMMEDIATE soffset only.  mode 1 = the same pair with `s_nop 0` between (the wait state): the negative control.
//   (B) DESIGN.md 5.10 / 5.13, tools/isa_lint.py rule 1:
//         v_mfma_f32_16x16x32_bf16 ...  (a stream of them, this wave's or a neighbour's on the SIMD)
//         v_pk_fma_f32 v[6:7], v[2:3], v[0:1], v[0:1] op_sel:[0,0,1] op_sel_hi:[1,0,1]     <- one register pair in two source slots, op_sel broadcast
//       must give (a.lo * s + b, a.hi * s + b); observed: a.lo * s + 0 on lanes 48-63, about once in 10^4 tiles, back-to-back launches only.
//       mode = place | 4 x plain: place 0 = behind 8 MFMAs of its own wave, 2 = between them, 3 = waves 2 / 3 of a workgroup issue only the FMAs, waves 0 / 1 -- their
//       neighbours on the SIMDs -- only the MFMAs; plain = two plain v_fma_f32 instead (what the product's epilogues use: pw_fma4_apart): the negative controls.
#include <hip/hip_runtime.h>
#include <cstdint>

typedef unsigned hz_u4 __attribute__((ext_vector_type(4)));
typedef float hz_f4 __attribute__((ext_vector_type(4)));
typedef float hz_f2 __attribute__((ext_vector_type(2)));
typedef __bf16 hz_b8 __attribute__((ext_vector_type(8)));

#define HZ_POISON 0xDEADBEEFu
__device__ __forceinline__ unsigned hz_val(unsigned row, unsigned lane, unsigned e) { return (row * 2654435761u) ^ (lane * 40503u) ^ (e * 0x9E3779B9u) | 1u; }

// (A) every lane stores `rows` 16-byte records: record (block, row) of lane l at dword ((block * rows + row) * 256 + l) * 4
template <int MODE>
__global__ void __launch_bounds__(256) k_hz_store(unsigned *out, int rows)
{
    const unsigned lane = threadIdx.x;
    const hz_u4 rs = { (unsigned)(uintptr_t)out, (unsigned)((uintptr_t)out >> 32) & 0xffffu, 0x7fffffffu, 0x00020000u };
    const unsigned voff = lane * 16u;
    for (int r = 0; r < rows; r++) {
        const unsigned row = blockIdx.x * (unsigned)rows + (unsigned)r;
        const unsigned soff = __builtin_amdgcn_readfirstlane(row * 4096u);       // an SGPR, not an immediate
        hz_u4 d = { hz_val(row, lane, 0), hz_val(row, lane, 1), hz_val(row, lane, 2), hz_val(row, lane, 3) };
        if (MODE == 0)
            asm volatile("buffer_store_dwordx4 v[2:5], %1, %2, %3 offen\n\tv_mov_b32 v3, %4\n\ts_nop 4" : "+{v[2:5]}"(d) : "v"(voff), "s"(rs), "s"(soff), "v"(HZ_POISON) : "memory");
        else
            asm volatile("buffer_store_dwordx4 v[2:5], %1, %2, %3 offen\n\ts_nop 0\n\tv_mov_b32 v3, %4\n\ts_nop 4" : "+{v[2:5]}"(d) : "v"(voff), "s"(rs), "s"(soff), "v"(HZ_POISON) : "memory");
    }
}

// (B) every wave: `trips` x (8 bf16 MFMAs, then the packed FMA on known operands); mismatches against plain fmaf are counted per (lane group of 16, half)
template <int MODE>
__global__ void __launch_bounds__(256) k_hz_pkfma(unsigned *bad, int trips)
{
    const unsigned lane = threadIdx.x & 63;
    const unsigned h = 0x3f803f80u ^ (threadIdx.x * 0x00010001u & 0x007f007fu);
    const hz_b8 A = __builtin_bit_cast(hz_b8, (hz_u4){ h, h ^ 0x00110022u, h ^ 0x00330044u, h ^ 0x00550066u }), B = __builtin_bit_cast(hz_b8, (hz_u4){ h ^ 0x00010001u, h, h ^ 0x00070003u, h });
    hz_f4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = (hz_f4){ 0.f, 0.f, 0.f, 0.f };
    unsigned nbad_lo = 0, nbad_hi = 0;
    for (int t = 0; t < trips; t++) {
        const float s = 1.25f + 0.001f * (float)((t + lane) & 31), b = 3.0f + (float)(t & 7);
        hz_f2 a = { 0.5f + 0.01f * (float)lane + (float)(t & 3), 2.0f - 0.02f * (float)lane };
        float want_lo, want_hi;                                                  // plain v_fma_f32, IN FRONT of the MFMAs (the compiler would pack and sink them)
        asm volatile("v_fma_f32 %0, %2, %4, %5\n\tv_fma_f32 %1, %3, %4, %5" : "=&v"(want_lo), "=&v"(want_hi) : "v"(a.x), "v"(a.y), "v"(s), "v"(b));
        hz_f2 sb = { s, b }, d;
        constexpr int PLACE = MODE & 3;                                         // 0: behind 8 MFMAs of its own wave; 2: between them; 3: waves 2 / 3 issue the FMAs, waves 0 / 1 -- their neighbours on the SIMDs -- the MFMAs
        constexpr bool PLAIN = (MODE & 4) != 0;                                 // two plain v_fma_f32 instead of the packed form: the negative control (the product's pw_fma4_apart)
        auto fma_under_test = [&]() {
            if (!PLAIN) asm volatile("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(sb));
            else { float d0, d1; asm volatile("v_fma_f32 %0, %2, %4, %5\n\tv_fma_f32 %1, %3, %4, %5" : "=&v"(d0), "=&v"(d1) : "v"(a.x), "v"(a.y), "v"(s), "v"(b)); d = (hz_f2){ d0, d1 }; }
        };
        const bool mfma_wave = PLACE != 3 || (threadIdx.x >> 6) < 2;
        if (PLACE == 2) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(A), "v"(B));
                fma_under_test();
                nbad_lo += d.x != want_lo;
                nbad_hi += d.y != want_hi;
            }
            continue;
        }
        if (mfma_wave) {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(A), "v"(B));
        }
        if (PLACE == 3 && mfma_wave) continue;
        fma_under_test();
        nbad_lo += d.x != want_lo;
        nbad_hi += d.y != want_hi;
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) r += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (r == 12345.678f) bad[15] = 1;                                            // (keeps the MFMAs alive)
    if (nbad_lo) atomicAdd(&bad[(lane >> 4) * 2], nbad_lo);
    if (nbad_hi) atomicAdd(&bad[(lane >> 4) * 2 + 1], nbad_hi);
}

// (A): launches `launches` x `blocks` workgroups x 256 lanes x `rows` records into d_out (blocks * rows * 4096 bytes; refilled with zeros by the caller between launches if wanted);
// the caller checks the records (value(row, lane, e), poison = 0xDEADBEEF).  Returns 0 / -1.
extern "C" int ffhz_store(void *d_out, int blocks, int rows, int mode, int launches, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    for (int i = 0; i < launches; i++) {
        if (mode == 0) hipLaunchKernelGGL(k_hz_store<0>, dim3(blocks), dim3(256), 0, s, (unsigned *)d_out, rows);
        else           hipLaunchKernelGGL(k_hz_store<1>, dim3(blocks), dim3(256), 0, s, (unsigned *)d_out, rows);
    }
    return hipStreamSynchronize(s) == hipSuccess ? 0 : -1;
}
extern "C" unsigned ffhz_store_value(unsigned row, unsigned lane, unsigned e) { return (row * 2654435761u) ^ (lane * 40503u) ^ (e * 0x9E3779B9u) | 1u; }

// (B): d_bad = 16 zeroed counters: [lane group of 16][lo, hi]; launches back to back, no host work between them
extern "C" int ffhz_pkfma(void *d_bad, int blocks, int trips, int mode, int launches, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    for (int i = 0; i < launches; i++) {
#define HZ_L(M) case M: hipLaunchKernelGGL(k_hz_pkfma<M>, dim3(blocks), dim3(256), 0, s, (unsigned *)d_bad, trips); break;
        switch (mode) { HZ_L(0) HZ_L(2) HZ_L(3) HZ_L(4) HZ_L(6) HZ_L(7) default: return -1; }
#undef HZ_L
    }
    return hipStreamSynchronize(s) == hipSuccess ? 0 : -1;
}


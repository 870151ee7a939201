// Lab: where does the dispatcher put the workgroups of a 256-thread / 74 KB-LDS launch (two resident per CU)?  Prints, per XCD-local workgroup index,
// the CU it ran on and its start time.  hipcc --offload-arch=gfx950 -O2 -o tools/lab/bin/dispatch_probe tools/lab/dispatch_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(256) probe(unsigned *hw, unsigned long long *t0, int spin)
{
    __shared__ unsigned pad[74 * 256];
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long t = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { hw[blockIdx.x] = id | (xcc << 28); t0[blockIdx.x] = t; }
    pad[threadIdx.x] = id;
    for (int i = 0; i < spin; i++) __builtin_amdgcn_s_sleep(100);
    if (pad[(threadIdx.x + 1) & 255] == 0xdeadbeef) hw[0] = 0;
}
int main()
{
    const int n = 1600;
    unsigned *hw; unsigned long long *t0;
    hipMalloc(&hw, n * 4); hipMalloc(&t0, n * 8);
    probe<<<n, 256>>>(hw, t0, 40);
    hipDeviceSynchronize();
    std::vector<unsigned> h(n); std::vector<unsigned long long> t(n);
    hipMemcpy(h.data(), hw, n * 4, hipMemcpyDeviceToHost); hipMemcpy(t.data(), t0, n * 8, hipMemcpyDeviceToHost);
    unsigned long long tmin = t[0];
    for (int i = 0; i < n; i++) if (t[i] < tmin) tmin = t[i];
    // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
    for (int i = 0; i < n; i += 8) {
        if ((i >> 3) < 72 || (i >> 3) % 16 == 0)
            printf("wg %4d (xcd-local %3d): xcc %u se %u sh %u cu %2u simd %u  t0 %6.2f us\n", i, i >> 3, h[i] >> 28, (h[i] >> 13) & 7, (h[i] >> 12) & 1, (h[i] >> 8) & 15, (h[i] >> 4) & 3, (t[i] - tmin) / 100.0);
    }
    return 0;
}

/* LD_PRELOAD shim: log selected HIP calls (who configures what at start-up -- e.g. what ncclCommInitAll does to the process).
 *   gcc -shared -fPIC -O1 -o gpurun_out/hipspy.so tools/hipspy.c -ldl
 * (lab equipment; not part of the product) */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stddef.h>

#define REAL(name) static int (*real)() = 0; if (!real) real = (int (*)())dlsym(RTLD_NEXT, #name)

int hipDeviceSetLimit(int limit, size_t value) { REAL(hipDeviceSetLimit); int r = real(limit, value); fprintf(stderr, "[hipspy] hipDeviceSetLimit(%d, %zu) -> %d\n", limit, value, r); return r; }
int hipStreamCreateWithFlags(void **s, unsigned flags) { REAL(hipStreamCreateWithFlags); int r = real(s, flags); fprintf(stderr, "[hipspy] hipStreamCreateWithFlags(flags %u) -> %p\n", flags, *s); return r; }
int hipStreamCreateWithPriority(void **s, unsigned flags, int prio) { REAL(hipStreamCreateWithPriority); int r = real(s, flags, prio); fprintf(stderr, "[hipspy] hipStreamCreateWithPriority(flags %u, prio %d) -> %p\n", flags, prio, *s); return r; }
int hipExtMallocWithFlags(void **p, size_t n, unsigned flags) { REAL(hipExtMallocWithFlags); int r = real(p, n, flags); fprintf(stderr, "[hipspy] hipExtMallocWithFlags(%zu, flags %u)\n", n, flags); return r; }
int hipHostMalloc(void **p, size_t n, unsigned flags) { REAL(hipHostMalloc); int r = real(p, n, flags); fprintf(stderr, "[hipspy] hipHostMalloc(%zu, flags %u)\n", n, flags); return r; }
int hipSetDeviceFlags(unsigned flags) { REAL(hipSetDeviceFlags); int r = real(flags); fprintf(stderr, "[hipspy] hipSetDeviceFlags(%u) -> %d\n", flags, r); return r; }
int hipFuncSetAttribute(const void *f, int attr, int v) { REAL(hipFuncSetAttribute); int r = real(f, attr, v); fprintf(stderr, "[hipspy] hipFuncSetAttribute(attr %d, %d)\n", attr, v); return r; }
int hipMemPoolCreate(void *a, void *b) { REAL(hipMemPoolCreate); int r = real(a, b); fprintf(stderr, "[hipspy] hipMemPoolCreate\n"); return r; }
int hipDeviceEnablePeerAccess(int d, unsigned f) { REAL(hipDeviceEnablePeerAccess); int r = real(d, f); fprintf(stderr, "[hipspy] hipDeviceEnablePeerAccess(%d)\n", d); return r; }
int hipHostRegister(void *p, size_t n, unsigned f) { REAL(hipHostRegister); int r = real(p, n, f); fprintf(stderr, "[hipspy] hipHostRegister(%zu, flags %u)\n", n, f); return r; }
int hipMalloc(void **p, size_t n) { REAL(hipMalloc); int r = real(p, n); if (n >= (1u << 20)) fprintf(stderr, "[hipspy] hipMalloc(%zu)\n", n); return r; }
int hipLaunchKernel(const void *f, unsigned long long g0, unsigned g1, unsigned long long b0, unsigned b1, void **args, size_t shm, void *st)
{ static int (*real)(const void *, unsigned long long, unsigned, unsigned long long, unsigned, void **, size_t, void *) = 0; if (!real) real = dlsym(RTLD_NEXT, "hipLaunchKernel");
  static int n = 0; if (n++ < 3) fprintf(stderr, "[hipspy] hipLaunchKernel #%d stream %p\n", n, st); return real(f, g0, g1, b0, b1, args, shm, st); }
int hipExtLaunchKernel(const void *f, unsigned long long g0, unsigned g1, unsigned long long b0, unsigned b1, void **args, size_t shm, void *st, void *e0, void *e1, int fl)
{ static int (*real)(const void *, unsigned long long, unsigned, unsigned long long, unsigned, void **, size_t, void *, void *, void *, int) = 0; if (!real) real = dlsym(RTLD_NEXT, "hipExtLaunchKernel");
  fprintf(stderr, "[hipspy] hipExtLaunchKernel stream %p\n", st); return real(f, g0, g1, b0, b1, args, shm, st, e0, e1, fl); }

// Microbenchmark: do 256 workgroups that stream slices a power of two apart (patients of 2^k states, all in step) collide on
// HBM channels?  Read rate with the slices 4 MB apart against 4 MB + pad.  hipcc -O3 --offload-arch=gfx950 -o build_ab/stride scripts/micro/stride.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(1024) void k_rw(d2* __restrict__ a, size_t per16, size_t stride16, int reps, int wr, double* out) {
  d2* s = a + (size_t)blockIdx.x * stride16;
  d2 acc = {0, 0};
  for (int r = 0; r < reps; ++r)
    for (size_t i = threadIdx.x; i < per16; i += 1024) {
      d2 v = s[i]; acc += v; asm volatile("" : "+v"(acc));
      if (wr) s[i + per16] = v;                       // (second half of the slice: written)
    }
  if (acc[0] + acc[1] == 1.2345) out[0] = acc[0];
}
int main() {
  const size_t TOT = 6ull << 30;
  d2* a; double* o;
  CK(hipMalloc(&a, TOT)); CK(hipMalloc(&o, 8)); CK(hipMemset(a, 0, TOT));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int WG = 256;
  for (int wr = 0; wr < 2; ++wr)
    for (size_t padKB : {0, 4, 20, 68, 132, 260, 1028}) {
      const size_t perKB = 4096, per16 = perKB * 1024 / 16, stride16 = (2 * perKB + padKB) * 1024 / 16;
      const int reps = 4;
      for (int it = 0; it < 2; ++it) {
        CK(hipEventRecord(e0));
        k_rw<<<WG, 1024>>>(a, per16, stride16, reps, wr, o);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it == 1)
          printf("%s slices of %zu KB, %zu KB + %4zu KB apart: %7.2f ms  %8.1f GB/s\n", wr ? "read+write" : "read      ", perKB, 2 * perKB, padKB, ms,
                 (double)perKB * 1024 * WG * reps * (1 + wr) / ms / 1e6);
      }
    }
  return 0;
}

// Microbenchmark: does a re-read of a recently read / written footprint come back faster than HBM?
// (memory-side cache, 256 MB on MI355X).  Build: hipcc -O3 --offload-arch=gfx950 -o build_ab/mall scripts/micro/mall.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
// every workgroup sweeps its own slice of `per` bytes `reps` times
__global__ __launch_bounds__(1024) void k_read(const d2* __restrict__ a, size_t per16, int reps, double* out) {
  const d2* s = a + (size_t)blockIdx.x * per16;
  d2 acc = {0, 0};
  for (int r = 0; r < reps; ++r)
    for (size_t i = threadIdx.x; i < per16; i += 1024) { d2 v = __builtin_nontemporal_load(s + i); acc += v; }
  if (acc[0] + acc[1] == 1.2345) out[0] = acc[0];
}
__global__ __launch_bounds__(1024) void k_read_plain(const d2* __restrict__ a, size_t per16, int reps, double* out) {
  const d2* s = a + (size_t)blockIdx.x * per16;
  d2 acc = {0, 0};
  for (int r = 0; r < reps; ++r)
    for (size_t i = threadIdx.x; i < per16; i += 1024) { d2 v = s[i]; acc += v; asm volatile("" : "+v"(acc)); }
  if (acc[0] + acc[1] == 1.2345) out[0] = acc[0];
}
// write the slice, then read it back `reps` times
__global__ __launch_bounds__(1024) void k_wr(d2* __restrict__ a, size_t per16, int reps, double* out) {
  d2* s = a + (size_t)blockIdx.x * per16;
  for (size_t i = threadIdx.x; i < per16; i += 1024) s[i] = d2{1.0, 2.0};
  __syncthreads();
  d2 acc = {0, 0};
  for (int r = 0; r < reps; ++r)
    for (size_t i = threadIdx.x; i < per16; i += 1024) { d2 v = s[i]; acc += v; asm volatile("" : "+v"(acc)); }
  if (acc[0] + acc[1] == 1.2345) out[0] = acc[0];
}
int main() {
  const size_t TOT = 4ull << 30;
  d2* a; double* o;
  CK(hipMalloc(&a, TOT)); CK(hipMalloc(&o, 8)); CK(hipMemset(a, 0, TOT));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int WG = 256;
  for (int mode = 0; mode < 2; ++mode)
    for (size_t perKB : {64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384}) {
      const size_t per16 = perKB * 1024 / 16;
      const int reps = (int)(16384 / perKB) * 8;
      for (int it = 0; it < 2; ++it) {
        CK(hipEventRecord(e0));
        if (mode == 0) k_read_plain<<<WG, 1024>>>(a, per16, reps, o); else k_wr<<<WG, 1024>>>(a, per16, reps, o);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it == 1)
          printf("%s slice %6zu KB x %d WGs = footprint %6.0f MB, %3d sweeps: %7.2f ms  %8.1f GB/s\n", mode ? "write+read" : "read      ", perKB, WG,
                 perKB * WG / 1024.0, reps, ms, (double)perKB * 1024 * WG * (reps + (mode ? 1 : 0)) / ms / 1e6);
      }
    }
  return 0;
}

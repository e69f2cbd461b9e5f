// Host cost of HIP launches, event records and cross-stream waits on the GPU box (small cohorts):
//   hipcc -O2 --offload-arch=gfx950 -Wno-unused-value -o build_ab/launchcost scripts/launchcost.hip
//   gpurun -- './build_ab/launchcost'
// Measured (MI355X box, ROCm 7.0.2): launch 3 - 5 us, hipMemsetAsync 4 us, timing-event pair around a launch +10 us,
// fork / launch / join / launch across two streams (4 event operations) 33 - 36 us.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_null(int* p) { if (p && threadIdx.x == 12345) *p = 1; }
__global__ void k_args(const int* a, const int* b, const int* c, const double* d, const double* e, double* f, double* g, double* h, double* i, double* j, int k, int l, int m) { if (threadIdx.x == 12345) *f = k + l + m; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t s0, s1; hipStreamCreateWithFlags(&s0, hipStreamNonBlocking); hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  hipEvent_t e0, e1, t0e, t1e; hipEventCreateWithFlags(&e0, hipEventDisableTiming); hipEventCreateWithFlags(&e1, hipEventDisableTiming);
  hipEventCreate(&t0e); hipEventCreate(&t1e);
  double* buf; hipMalloc(&buf, 1 << 20);
  const int R = 2000;
  for (int w = 0; w < 2; ++w) {
    double t = now();
    for (int i = 0; i < R; ++i) hipLaunchKernelGGL(k_null, dim3(1), dim3(64), 0, s0, nullptr);
    double ti = now() - t; hipStreamSynchronize(s0); double tt = now() - t;
    printf("null launch: issue %.2f us/launch, complete %.2f us/launch\n", ti / R, tt / R);
    t = now();
    for (int i = 0; i < R; ++i) hipLaunchKernelGGL(k_args, dim3(1), dim3(64), 0, s0, nullptr, nullptr, nullptr, nullptr, nullptr, buf, buf, buf, buf, buf, 1, 2, 3);
    ti = now() - t; hipStreamSynchronize(s0); tt = now() - t;
    printf("13-arg launch: issue %.2f us, complete %.2f us\n", ti / R, tt / R);
    t = now();
    for (int i = 0; i < R; ++i) hipMemsetAsync(buf, 0, 4096, s0);
    ti = now() - t; hipStreamSynchronize(s0); tt = now() - t;
    printf("memsetAsync 4 KB: issue %.2f us, complete %.2f us\n", ti / R, tt / R);
    t = now();
    for (int i = 0; i < R; ++i) { hipEventRecord(t0e, s0); hipLaunchKernelGGL(k_null, dim3(1), dim3(64), 0, s0, nullptr); hipEventRecord(t1e, s0); }
    ti = now() - t; hipStreamSynchronize(s0); tt = now() - t;
    printf("timing-event record + launch + record: issue %.2f us, complete %.2f us\n", ti / R, tt / R);
    t = now();
    for (int i = 0; i < R; ++i) {
      hipEventRecord(e0, s0); hipStreamWaitEvent(s1, e0, 0);
      hipLaunchKernelGGL(k_null, dim3(1), dim3(64), 0, s1, nullptr);
      hipEventRecord(e1, s1); hipStreamWaitEvent(s0, e1, 0);
      hipLaunchKernelGGL(k_null, dim3(1), dim3(64), 0, s0, nullptr);
    }
    ti = now() - t; hipStreamSynchronize(s0); tt = now() - t;
    printf("fork/launch/join/launch (6 calls): issue %.2f us, complete %.2f us\n", ti / R, tt / R);
  }
  return 0;
}

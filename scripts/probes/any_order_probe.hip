// Does hipExtAnyOrderLaunch let two INDEPENDENT kernels of one stream overlap on this part (gfx950)?  Two spin kernels of 64 workgroups each
// (a quarter of the CUs), ~200 us; back to back they take 2 x T when serialised by the packet barrier bit and ~T when the second may start early.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ __launch_bounds__(256) void spin(unsigned* sink, int iters) {
  unsigned v = threadIdx.x;
  for (int i = 0; i < iters; ++i) v = v * 1664525u + 1013904223u;
  if (v == 0x12345678u) *sink = v;
}
int main() {
  unsigned* sink; CK(hipMalloc(&sink, 4));
  hipStream_t st, st2; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 100000;
  auto time = [&](int mode) {
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
      CK(hipEventRecord(e0, st));
      for (int rep = 0; rep < 8; ++rep) {
        hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st, sink, iters);
        if (mode == 0) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st, sink, iters);
        if (mode == 1) hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, sink, iters);
      }
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    return best;
  };
  const float t_single = time(2), t_plain = time(0), t_any = time(1);
  printf("8 x one kernel                         : %.3f ms\n", t_single);
  printf("8 x two kernels, plain launches        : %.3f ms\n", t_plain);
  printf("8 x two kernels, second any-order      : %.3f ms   (%s)\n", t_any, t_any < 0.75f * t_plain ? "OVERLAPS" : "serialised: flag not honoured");
  return 0;
}

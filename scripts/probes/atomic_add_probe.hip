// fp32 atomic-add throughput on gfx950 for the access pattern of a single-pass attention backward: every workgroup adds 64 x 64 fp32 tiles
// (16 KiB, 256 B per wave instruction) into a [tiles][4096] buffer.  (a) disjoint tiles per workgroup; (b) the 12 workgroups of a "head" (consecutive block ids after
// the XCD remap, i.e. one XCD) walk the SAME 24 tiles, staggered by their index; (c) the same without stagger.  unsafeAtomicAdd = global_atomic_add_f32, no return.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ __launch_bounds__(256) void adds(float* buf, int mode, int iters) {
  // XCD-contiguous logical id, as the attention kernels remap it (8 XCDs, round-robin dispatch)
  const int n = gridDim.x, bid = (blockIdx.x & 7) * (n >> 3) + (blockIdx.x >> 3);
  const int head = bid / 12, kb = bid % 12;
  for (int it = 0; it < iters; ++it) {
    const int qt = mode == 0 ? it % 24 : (mode == 1 ? (it + 2 * kb) % 24 : it % 24);
    float* tile = buf + ((size_t)(mode == 0 ? bid : head) * 24 + qt) * 4096;
#pragma unroll
    for (int j = 0; j < 16; ++j) unsafeAtomicAdd(tile + j * 256 + threadIdx.x, 1.0f);
  }
}
int main() {
  const int grid = 3072, iters = 480;  // 256 heads x 12 key blocks; 20 passes over the 24 query tiles
  float* buf;
  const size_t bytes = (size_t)grid * 24 * 4096 * 4;
  CK(hipMalloc(&buf, bytes));
  CK(hipMemset(buf, 0, bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const char* names[3] = {"disjoint tiles per workgroup", "12 workgroups of one XCD share 24 tiles, staggered", "12 workgroups of one XCD share 24 tiles, same order"};
  for (int mode = 0; mode < 3; ++mode) {
    float best = 1e9f;
    for (int r = 0; r < 4; ++r) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(adds, dim3(grid), dim3(256), 0, 0, buf, mode, iters);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r && ms < best) best = ms;
    }
    const double b = (double)grid * iters * 16384.0;
    printf("%-55s %7.3f ms  %6.2f TB/s of fp32 atomic adds (%.1f G adds/s)\n", names[mode], best, b / best / 1e9, b / 4 / best / 1e6);
  }
  float h[4];
  CK(hipMemcpy(h, buf, 16, hipMemcpyDeviceToHost));
  printf("(sanity: first element %.0f)\n", h[0]);
  return 0;
}

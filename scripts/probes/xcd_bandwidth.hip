#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// participants stream disjoint slices of a buffer (16 B per lane, 8 loads in flight), result folded so nothing is optimised away
__global__ __launch_bounds__(256) void rd(const u32x4* __restrict__ src, size_t n16, unsigned* sink, int mask, int nwg, int wg_per_cu) {
  if (blockIdx.x & mask) return;
  const int me = mask ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const size_t per = n16 / nwg;
  const u32x4* p = src + (size_t)me * per;
  unsigned acc = 0;
  for (size_t i = threadIdx.x; i + 7 * 256 < per; i += 8 * 256) {
    u32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(p + i + j * 256);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc ^= v[j][0] ^ v[j][1] ^ v[j][2] ^ v[j][3];
  }
  if (acc == 0x12345678u) *sink = acc;
}
int main() {
  const size_t bytes = 512ull << 20;
  u32x4* src; unsigned* sink;
  CK(hipMalloc(&src, bytes)); CK(hipMalloc(&sink, 4)); CK(hipMemset(src, 1, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct { const char* name; int grid, mask, nwg; } cfg[] = {
      {"all XCDs, 256 workgroups", 256, 0, 256}, {"all XCDs, 1024 workgroups", 1024, 0, 1024},
      {"XCD 0 only, 32 workgroups (1 per CU)", 256, 7, 32}, {"XCD 0 only, 128 workgroups (4 per CU)", 1024, 7, 128},
      {"XCD 0 only, 256 workgroups (8 per CU)", 2048, 7, 256}};
  for (auto& c : cfg) {
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(rd, dim3(c.grid), dim3(256), 0, 0, src, bytes / 16, sink, c.mask, c.nwg, 1);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-45s %8.1f GB/s (%.3f ms for %zu MB)\n", c.name, bytes / best / 1e6, best, bytes >> 20);
  }
  return 0;
}

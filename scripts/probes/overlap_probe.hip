// Does VALU work of one wave overlap MFMA work of another wave on the same SIMD?  (gfx950 probe)
// mode 0: MFMA blocks only; 1: VALU blocks only; 2: each wave alternates MFMA block / VALU block, both waves of a SIMD in
// phase; 3: same, but every second wave starts with the VALU block (complementary phases); 4 / 5: as 3 / 2 with s_setprio 1
// around the MFMA blocks.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int VK>  // VK 0: exp + fma mix, 1: fma only, 2: exp only
__global__ __launch_bounds__(512, 2) void k(float* out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (__bf16)(threadIdx.x * 0.001f + i);
    b[i] = (__bf16)(threadIdx.x * 0.002f - i);
  }
  float v[32];
  for (int i = 0; i < 32; ++i) v[i] = threadIdx.x * 0.01f + i;
  const int wave = threadIdx.x >> 6;
  const bool flip = (MODE == 3 || MODE == 4) && (wave >= 4);  // waves 4-7 share SIMDs with waves 0-3
  for (int it = 0; it < iters; ++it) {
    for (int half = 0; half < 2; ++half) {
      const bool do_mfma = (MODE == 0) || ((MODE >= 2) && ((half == 0) != flip));
      const bool do_valu = (MODE == 1) || ((MODE >= 2) && ((half == 1) != flip));
      if (do_mfma) {
        if (MODE >= 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j & 3], 0, 0, 0);
        if (MODE >= 4) __builtin_amdgcn_s_setprio(0);
      }
      if (do_valu) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
          for (int i = 0; i < 32; ++i) {
            if (VK == 0) v[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(v[i], 0.5f, -1.0f)) + v[(i + 1) & 31] * 0.25f;
            if (VK == 1) v[i] = __builtin_fmaf(__builtin_fmaf(v[i], 0.5f, -1.0f), 0.99f, v[(i + 1) & 31] * 0.25f);
            if (VK == 2) v[i] = __builtin_amdgcn_exp2f(v[i]) ;
          }
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 32; ++i) s += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE, int VK>
float run(float* out, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, VK>), dim3(256), dim3(512), 0, 0, out, iters);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<MODE, VK>), dim3(256), dim3(512), 0, 0, out, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * 4);
  const int iters = 2000;
  {
    const float m = run<0, 0>(out, iters) / 2;
    printf("mfma block %.3f ms\n", m);
    const float a0 = run<1, 0>(out, iters) / 2, a2 = run<2, 0>(out, iters), a3 = run<3, 0>(out, iters);
    printf("valu = exp+fma mix : block %.3f | in phase %.3f | complementary %.3f | sum %.3f max %.3f\n", a0, a2, a3, m + a0, m > a0 ? m : a0);
    const float b0 = run<1, 1>(out, iters) / 2, b2 = run<2, 1>(out, iters), b3 = run<3, 1>(out, iters);
    printf("valu = fma only    : block %.3f | in phase %.3f | complementary %.3f | sum %.3f max %.3f\n", b0, b2, b3, m + b0, m > b0 ? m : b0);
    const float c0 = run<1, 2>(out, iters) / 2, c2 = run<2, 2>(out, iters), c3 = run<3, 2>(out, iters);
    printf("valu = exp only    : block %.3f | in phase %.3f | complementary %.3f | sum %.3f max %.3f\n", c0, c2, c3, m + c0, m > c0 ? m : c0);
  }
  return 0;
}

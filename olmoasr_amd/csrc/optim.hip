// Fused GradScaler.unscale_ + clip_grad_norm_ + AdamW over flat fp32 arenas (gfx950, HBM-bound: 16 B read +
// 12 B (+2 B bf16 shadow) written per parameter).
// Reference: scaler.unscale_(optimizer); clip_grad_norm_(params, 1.0); scaler.step(optimizer)
// (scripts/training/train_timestamps.py:1509-1512) with AdamW(lr, betas=(0.9,0.98), eps=1e-6, weight_decay=0.1)
// over ONE param group holding every tensor (train_timestamps.py:727-733).
#include "kernels.h"

namespace {

__global__ __launch_bounds__(256) void grad_stats_kernel(const float* __restrict__ g, long n, double* __restrict__ partial,
                                                         float* __restrict__ stats) {
  __shared__ double red[4];
  __shared__ int bad;
  if (threadIdx.x == 0) bad = 0;
  __syncthreads();
  double s = 0.0;
  int nf = 0;
  const long n4 = n >> 2;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4_t v = ((const f32x4_t*)g)[i];
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      q += v[j] * v[j];
      nf |= !(fabsf(v[j]) <= 3.4028234e38f);  // inf or nan
    }
    s += (double)q;
  }
  if (blockIdx.x == 0)
    for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) {
      s += (double)g[i] * g[i];
      nf |= !(fabsf(g[i]) <= 3.4028234e38f);
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  if (nf) bad = 1;
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    if (bad) atomicAdd(stats + 1, 1.0f);
  }
}

__global__ __launch_bounds__(256) void grad_stats_final_kernel(const double* __restrict__ partial, int nb, float* __restrict__ stats) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) s += partial[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) stats[0] = (float)((red[0] + red[1]) + (red[2] + red[3]));
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16_t* __restrict__ shadow, long n,
                                                    const float* __restrict__ stats, float inv_scale, float max_norm, float lr,
                                                    float b1, float b2, float eps, float wd, float bc1, float bc2) {
  if (stats[1] != 0.f) return;  // GradScaler: skip the step when any gradient is inf/nan
  // total_norm of the UNSCALED grads; clip coefficient exactly as torch.nn.utils.clip_grad_norm_
  const float total_norm = sqrtf(stats[0]) * inv_scale;
  float coef = max_norm / (total_norm + 1e-6f);
  coef = coef > 1.0f ? 1.0f : coef;
  const float gmul = coef * inv_scale;
  const float step_size = lr / bc1;
  const float inv_sqrt_bc2 = rsqrtf(bc2);
  const float decay = 1.0f - lr * wd;
  const long n4 = n >> 2;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4_t pp = ((f32x4_t*)p)[i], mm = ((f32x4_t*)m)[i], vv = ((f32x4_t*)v)[i];
    const f32x4_t gg = ((const f32x4_t*)g)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gr = gg[j] * gmul;
      pp[j] *= decay;
      mm[j] = mm[j] * b1 + gr * (1.0f - b1);
      vv[j] = vv[j] * b2 + gr * gr * (1.0f - b2);
      const float denom = sqrtf(vv[j]) * inv_sqrt_bc2 + eps;
      pp[j] -= step_size * (mm[j] / denom);
    }
    ((f32x4_t*)p)[i] = pp;
    ((f32x4_t*)m)[i] = mm;
    ((f32x4_t*)v)[i] = vv;
    if (shadow) {
      u32x2_t o;
      o[0] = pack_bf2(pp[0], pp[1]);
      o[1] = pack_bf2(pp[2], pp[3]);
      ((u32x2_t*)shadow)[i] = o;
    }
  }
}

}  // namespace

int launch_grad_stats(const float* g, long n, double* partial, float* stats, hipStream_t s) {
  OASR_REQUIRE(g && partial && stats && n > 0 && (n % 4) == 0, "grad_stats: bad args (n must be a multiple of 4)");
  OASR_CHECK_HIP(hipMemsetAsync(stats, 0, 2 * sizeof(float), s));
  const int nb = 1024;
  hipLaunchKernelGGL(grad_stats_kernel, dim3(nb), dim3(256), 0, s, g, n, partial, stats);
  OASR_LAUNCH_CHECK();
  hipLaunchKernelGGL(grad_stats_final_kernel, dim3(1), dim3(256), 0, s, partial, nb, stats);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}

int launch_adamw(float* p, const float* g, float* m, float* v, bf16_t* shadow, long n, const float* stats, float inv_scale,
                 float max_norm, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2, hipStream_t s) {
  OASR_REQUIRE(p && g && m && v && stats && n > 0 && (n % 4) == 0, "adamw: bad args (n must be a multiple of 4)");
  hipLaunchKernelGGL(adamw_kernel, dim3(2048), dim3(256), 0, s, p, g, m, v, shadow, n, stats, inv_scale, max_norm, lr, b1, b2, eps,
                     wd, bc1, bc2);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}

// LayerNorm forward/backward for gfx950 (HBM-bound; one wave64 per row, 16-byte bf16 vector accesses).
// Reference: olmoasr/model.py:14-39 -- F.layer_norm in fp32 (eps 1e-5), cast back to the input dtype.
// d <= 2048, d % 8 == 0 (reference widths: 384, 512, 768, 1024, 1280).
#include "kernels.h"

namespace {

constexpr int MAXC = 4;  // 16-byte chunks per lane: 64 lanes * 4 * 8 = 2048 columns

__device__ __forceinline__ void unpack8(const u32x4_t& p, float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = bf_lo(p[i]);
    f[2 * i + 1] = bf_hi(p[i]);
  }
}
__device__ __forceinline__ u32x4_t pack8(const float (&f)[8]) {
  u32x4_t p;
#pragma unroll
  for (int i = 0; i < 4; ++i) p[i] = pack_bf2(f[2 * i], f[2 * i + 1]);
  return p;
}

__global__ __launch_bounds__(256) void ln_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out, long rows,
                                                     int d) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nchunk = d >> 3;
  float v[MAXC][8];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nchunk) {
      unpack8(*(const u32x4_t*)(x + row * d + ch * 8), v[c]);
#pragma unroll
      for (int i = 0; i < 8; ++i) s += v[c][i];
    }
  }
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nchunk) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float t = v[c][i] - mean;
        q += t * t;
      }
    }
  }
  const float var = wave_sum(q) / (float)d;
  const float rstd = rsqrtf(var + 1e-5f);
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nchunk) {
      float o[8];
      const f32x4_t g0 = *(const f32x4_t*)(gamma + ch * 8), g1 = *(const f32x4_t*)(gamma + ch * 8 + 4);
      const f32x4_t b0 = *(const f32x4_t*)(beta + ch * 8), b1 = *(const f32x4_t*)(beta + ch * 8 + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        o[i] = (v[c][i] - mean) * rstd * g0[i] + b0[i];
        o[4 + i] = (v[c][4 + i] - mean) * rstd * g1[i] + b1[i];
      }
      *(u32x4_t*)(y + row * d + ch * 8) = pack8(o);
    }
  }
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
}

// dx = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat)) (+ dres);  dgamma += dy*xhat, dbeta += dy
// NCH = 16-byte chunks per lane (d <= 512 * NCH).  Each wave walks its rows two at a time: the loads of both rows (and
// of the residual gradient) are issued before either is reduced, so twice the bytes are in flight per wave (the
// one-row version was latency bound at 2.6 TB/s: a wave's next row was not requested until the previous was stored).
template <int NCH>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const bf16_t* __restrict__ dres,
                                                     bf16_t* __restrict__ dx, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, float* __restrict__ dsum, long rows, int d) {
  __shared__ float red[4][512 * NCH];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = d >> 3;
  float gw[NCH][8], dg[NCH][8], db[NCH][8], dsx[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = lane + 64 * c;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      dg[c][i] = 0.f;
      db[c][i] = 0.f;
      dsx[c][i] = 0.f;
      gw[c][i] = (ch < nchunk) ? gamma[ch * 8 + i] : 0.f;
    }
  }
  const float inv_d = 1.0f / (float)d;
  const long stride = (long)gridDim.x * 4;
  for (long row0 = (long)blockIdx.x * 4 + wave; row0 < rows; row0 += 2 * stride) {
    u32x4_t px[2][NCH], pd[2][NCH], pr[2][NCH];
    bool live[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const long row = row0 + r * stride;
      live[r] = row < rows;
      const long rr = live[r] ? row : row0;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nchunk) {
          px[r][c] = *(const u32x4_t*)(x + rr * d + ch * 8);
          pd[r][c] = *(const u32x4_t*)(dy + rr * d + ch * 8);
          if (dres) pr[r][c] = *(const u32x4_t*)(dres + rr * d + ch * 8);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (!live[r]) continue;
      const long row = row0 + r * stride;
      const float mu = mean[row], rs = rstd[row];
      float xh[NCH][8], gy[NCH][8];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nchunk) {
          float xv[8], dv[8];
          unpack8(px[r][c], xv);
          unpack8(pd[r][c], dv);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            xh[c][i] = (xv[i] - mu) * rs;
            gy[c][i] = dv[i] * gw[c][i];
            s1 += gy[c][i];
            s2 += gy[c][i] * xh[c][i];
            dg[c][i] += dv[i] * xh[c][i];
            db[c][i] += dv[i];
          }
        }
      }
      s1 = wave_sum(s1) * inv_d;
      s2 = wave_sum(s2) * inv_d;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nchunk) {
          float o[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] = rs * (gy[c][i] - s1 - xh[c][i] * s2);
          if (dres) {
            float rv[8];
            unpack8(pr[r][c], rv);
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = bf_round(o[i]) + rv[i];
          }
          const u32x4_t packed = pack8(o);
          *(u32x4_t*)(dx + row * d + ch * 8) = packed;
          if (dsum) {  // column sum of the stored (bf16) gradient = bias gradient of the Linear that produced this stream
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              dsx[c][2 * i] += bf_lo(packed[i]);
              dsx[c][2 * i + 1] += bf_hi(packed[i]);
            }
          }
        }
      }
    }
  }
  // block reduction of the column partials, then one atomic per column per block
  for (int pass = 0; pass < (dsum ? 3 : 2); ++pass) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nchunk) {
#pragma unroll
        for (int i = 0; i < 8; ++i) red[wave][ch * 8 + i] = pass == 0 ? dg[c][i] : (pass == 1 ? db[c][i] : dsx[c][i]);
      }
    }
    __syncthreads();
    float* dst = pass == 0 ? dgamma : (pass == 1 ? dbeta : dsum);
    for (int j = threadIdx.x; j < d; j += 256) {
      const float t = red[0][j] + red[1][j] + red[2][j] + red[3][j];
      unsafeAtomicAdd(dst + j, t);
    }
    __syncthreads();
  }
}

}  // namespace

int launch_layernorm_fwd(const bf16_t* x, const float* gamma, const float* beta, bf16_t* y, float* mean, float* rstd,
                         long rows, int d, hipStream_t s) {
  OASR_REQUIRE(x && gamma && beta && y, "layernorm_fwd: null pointer");
  OASR_REQUIRE(d % 8 == 0 && d <= 2048 && d > 0, "layernorm: d=%d must be a multiple of 8 and <= 2048", d);
  if (rows <= 0) return OASR_OK;
  hipLaunchKernelGGL(ln_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, gamma, beta, y, mean, rstd, rows, d);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}

int launch_layernorm_bwd(const bf16_t* dy, const bf16_t* x, const float* gamma, const float* mean, const float* rstd,
                         const bf16_t* dres, bf16_t* dx, float* dgamma, float* dbeta, float* dsum, long rows, int d, hipStream_t s) {
  OASR_REQUIRE(dy && x && gamma && mean && rstd && dx && dgamma && dbeta, "layernorm_bwd: null pointer");
  OASR_REQUIRE(d % 8 == 0 && d <= 2048 && d > 0, "layernorm: d=%d must be a multiple of 8 and <= 2048", d);
  if (rows <= 0) return OASR_OK;
  long blocks = (rows + 3) / 4;
  if (blocks > 512) blocks = 512;  // 2 workgroups per CU; more only adds column atomics (measured: scripts/ln_bench.py history)
  const dim3 grid((unsigned)blocks);
  if (d <= 512)
    hipLaunchKernelGGL(ln_bwd_kernel<1>, grid, dim3(256), 0, s, dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, dsum, rows, d);
  else if (d <= 1024)
    hipLaunchKernelGGL(ln_bwd_kernel<2>, grid, dim3(256), 0, s, dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, dsum, rows, d);
  else if (d <= 1536)
    hipLaunchKernelGGL(ln_bwd_kernel<3>, grid, dim3(256), 0, s, dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, dsum, rows, d);
  else
    hipLaunchKernelGGL(ln_bwd_kernel<4>, grid, dim3(256), 0, s, dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, dsum, rows, d);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}

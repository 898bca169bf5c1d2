// LayerNorm-in-the-operand-load projection for the KV-cached decoder step on gfx950 (TextDecoder.forward for ONE new token per
// sequence, olmoasr/model.py:786-817 with the kv_cache hooks of :925-964).
//
// A decode step is latency-bound: ~11 dependent launches per layer, each a few microseconds of work.  For a handful of sequences
// (B <= 4: the timestamp-mode transcribe loop decodes ONE window at a time) the LayerNorm in front of a projection is folded into
// the projection's operand load: every workgroup recomputes the B row statistics (B rows of d) and normalises the rows on their
// way into the MFMA operands (8 launches per layer instead of 11; bit-identical to the separate kernels: same rounding points --
// bf16 LN output, bf16 Linear output before GELU / residual -- same skinny-GEMM accumulation scheme as gemm.hip: 32 output columns
// per workgroup, K split over the 4 waves, weights streamed once straight into MFMA operands).  From B = 16 the recomputed
// statistics cost more than the launches they save (+20 %, profiles/r02_decode_step.txt), so larger batches keep the separate
// LayerNorm kernels.  (Round 2 also carried a ONE-launch step built on this phase body with device-wide barriers between phases;
// it was 1.7x slower on the 8-XCD part -- an L2 write-back + invalidate per barrier per workgroup -- and was removed in round 3.)
#include "decode_shared.h"

namespace {

using dec::MAXC;
using dec::unpack8;
using dec::Epi;

struct ProjSmem {
  float red[4][16][64];  // K-split partial accumulators of a projection tile
  float mean[32], rstd[32];
};
// Row statistics of x [M][d] (bf16) into LDS: wave w takes rows w, w + 4, ...  Same arithmetic as ln_fwd_kernel (norm.hip).
__device__ __forceinline__ void ln_stats(const bf16_t* x, int M, int d, ProjSmem& sm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = d >> 3;
  for (int row = wave; row < M; row += 4) {
    u32x4_t raw[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nchunk) raw[c] = *(const u32x4_t*)(x + (long)row * d + ch * 8);
    }
    float mean, rstd;
    dec::row_stats(raw, lane, nchunk, d, mean, rstd);
    if (lane == 0) {
      sm.mean[row] = mean;
      sm.rstd[row] = rstd;
    }
  }
  __syncthreads();
}

// out[M][N] = epi( LN?(x)[M][K] . W[N][K]^T ) over the work items (32-column tiles) of this workgroup.
template <bool LN>
__device__ __forceinline__ void proj_phase(const bf16_t* x, int M, int K, const bf16_t* W, int N, const float* g, const float* bta,
                                           const Epi& e, ProjSmem& sm) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  if (LN) ln_stats(x, M, K, sm);
  int row = lane & 31;
  row = row < M ? row : M - 1;
  const bf16_t* xp = x + (long)row * K + h * 8;
  const float mu = LN ? sm.mean[row] : 0.f, rs = LN ? sm.rstd[row] : 1.f;
  const int ntile = (N + 31) >> 5;
  const int kq = K >> 2;  // K % 64 == 0: every wave's share is a multiple of 16
  const int k_begin = wave * kq, k_end = k_begin + kq;
  for (int t = blockIdx.x; t < ntile; t += gridDim.x) {
    const int n0 = t << 5;
    int col = n0 + (lane & 31);
    col = col < N ? col : N - 1;
    const bf16_t* wp = W + (long)col * K + h * 8;
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    int k = k_begin;
    for (; k + 64 <= k_end; k += 64) {
      u32x4_t wq[4], xq[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        wq[j] = __builtin_nontemporal_load((const u32x4_t*)(wp + k + 16 * j));  // streamed once per step
        xq[j] = *(const u32x4_t*)(xp + k + 16 * j);
      }
      if (LN) {
#pragma unroll
        for (int j = 0; j < 4; ++j) xq[j] = dec::ln_apply8(xq[j], mu, rs, g, bta, k + 16 * j + h * 8);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)  // D'[n][m]: lane owns output row m = lane & 31
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wq[j]), __builtin_bit_cast(bf16x8_t, xq[j]), acc, 0, 0, 0);
    }
    for (; k < k_end; k += 16) {
      const u32x4_t wq = __builtin_nontemporal_load((const u32x4_t*)(wp + k));
      u32x4_t xq = *(const u32x4_t*)(xp + k);
      if (LN) xq = dec::ln_apply8(xq, mu, rs, g, bta, k + h * 8);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wq), __builtin_bit_cast(bf16x8_t, xq), acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) sm.red[wave][r][lane] = acc[r];
    __syncthreads();
    // wave w finishes register group w: columns n0 + 8w + 4h .. +3 of output row m = lane & 31
    const int n = n0 + 8 * wave + 4 * h, m = lane & 31;
    if (m < M && n < N) {
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        v[i] = sm.red[0][wave * 4 + i][lane] + sm.red[1][wave * 4 + i][lane] + sm.red[2][wave * 4 + i][lane] + sm.red[3][wave * 4 + i][lane];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (n + i < N) {
          const float bias = e.bias ? e.bias[n + i] : 0.f;
          if (e.out_f32) e.out_f32[(long)m * e.ldf + n + i] = dec::epi_logit(v[i], bias);
          if (e.out)
            e.out[(long)m * e.ldc + n + i] =
                f2bf_dev(dec::epi_value(v[i], bias, e.gelu != 0, e.resid != nullptr, e.resid ? bf2f(e.resid[(long)m * e.ldr + n + i]) : 0.f));
        }
      }
    }
    __syncthreads();  // red is reused by the next tile
  }
}

// LayerNorm + Linear (+ GELU / residual / fp32 logits) of a handful of token rows in ONE kernel, one 32-column tile per workgroup.
struct ProjArgs {
  const bf16_t* x;
  int M, K, N;
  const bf16_t* W;
  const float *ln_g, *ln_b;
  Epi e;
};
template <bool LN>
__global__ __launch_bounds__(256) void decode_proj_kernel(ProjArgs a) {
  __shared__ ProjSmem sm;
  proj_phase<LN>(a.x, a.M, a.K, a.W, a.N, a.ln_g, a.ln_b, a.e, sm);
}

}  // namespace

int launch_decode_proj(const bf16_t* x, int M, int K, const bf16_t* W, int N, const float* ln_g, const float* ln_b, const float* bias,
                       int gelu, const bf16_t* resid, long ldr, bf16_t* out, long ldc, float* out_f32, long ldf, hipStream_t s) {
  OASR_REQUIRE(x && W && (out || out_f32) && M > 0 && M <= 32 && K % 64 == 0 && K <= 8192 && N > 0, "decode_proj: bad args (M=%d K=%d N=%d)", M, K, N);
  OASR_REQUIRE(!ln_g || (ln_b && K <= 2048), "decode_proj: LayerNorm prologue needs beta and K <= 2048");
  ProjArgs a{x, M, K, N, W, ln_g, ln_b, Epi{bias, gelu, resid, ldr, out, ldc, out_f32, ldf}};
  const dim3 grid((N + 31) / 32);
  if (ln_g)
    hipLaunchKernelGGL(decode_proj_kernel<true>, grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(decode_proj_kernel<false>, grid, dim3(256), 0, s, a);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}

// The attention SCORE matrix on request: ``qk`` of MultiHeadAttention.qkv_attention (olmoasr/model.py:347-442; inf_model.py:172-196),
//     qk[b, h, i, j] = (q[b, i, h, :] * 64^-1/4) . (k[b, j, h, :] * 64^-1/4)  (+ mask),   fp32, pre-softmax,
// which the reference returns from the manual attention path (a 2-D causal mask: the eval loop's decoder self-attention, model.py:316-327) and
// which word-level timestamp alignment reads from the cross-attention (whisper.timing.find_alignment hooks cross_attn and takes outs[-1]).
// The training / decoding kernels (attention.hip) never form this matrix; this one is for the few calls that want to look at it, so it is
// a plain LDS-tiled fp32 kernel: 64 x 64 scores per workgroup, operands converted to fp32 on the way into LDS, 16 B/lane stores, masked entries
// -inf (causal: j > i; key padding: j >= kv_len[b]) exactly as the additive mask of the reference leaves them.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace {

template <typename T>
__device__ __forceinline__ float to_f32(T v);
template <>
__device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return bf2f(v); }

template <typename T>
__global__ __launch_bounds__(256) void attn_scores_kernel(const T* __restrict__ q, const T* __restrict__ k, long ldq, long ldk, long bsq, long bsk,
                                                          const int32_t* __restrict__ kv_len, int H, int Tq, int Tk, int causal,
                                                          float* __restrict__ out) {
  __shared__ float sq[64][65], sk[64][65];  // [row][channel], odd stride: conflict-free column walks
  const int tid = threadIdx.x, bh = blockIdx.z, b = bh / H, h = bh - b * H;
  const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  const T* qb = q + (long)b * bsq + (long)h * 64;
  const T* kb = k + (long)b * bsk + (long)h * 64;
  for (int e = tid; e < 64 * 64; e += 256) {
    const int r = e >> 6, c = e & 63;
    sq[r][c] = i0 + r < Tq ? to_f32<T>(qb[(long)(i0 + r) * ldq + c]) : 0.f;
    sk[r][c] = j0 + r < Tk ? to_f32<T>(kb[(long)(j0 + r) * ldk + c]) : 0.f;
  }
  __syncthreads();
  const int ty = tid >> 4, tx = tid & 15;  // rows ty*4 .. +3, columns tx*4 .. +3
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
#pragma unroll 8
  for (int c = 0; c < 64; ++c) {
    float a[4], bb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = sq[ty * 4 + i][c], bb[i] = sk[tx * 4 + i][c];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], bb[j], acc[i][j]);
  }
  const int klim = kv_len ? min(kv_len[b], Tk) : Tk;
  const float ninf = -__builtin_huge_valf();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int qi = i0 + ty * 4 + i;
    if (qi >= Tq) continue;
    float* row = out + ((long)bh * Tq + qi) * Tk;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kj = j0 + tx * 4 + j;
      if (kj >= Tk) continue;
      const bool masked = kj >= klim || (causal && kj > qi);
      row[kj] = masked ? ninf : acc[i][j] * 0.125f;  // 64^-1/4 on q and on k
    }
  }
}

template <typename T>
int launch_scores(const AttnArgsT<T>& a, float* out, hipStream_t s) {
  OASR_REQUIRE(a.q && a.k && out && a.B > 0 && a.H > 0 && a.Tq > 0 && a.Tk > 0, "attention_scores: bad arguments");
  OASR_REQUIRE(!a.q_rows && !a.k_rows, "attention_scores: chunked token rows are a training-step layout; not supported here");
  OASR_REQUIRE((long)a.B * a.H <= 65535, "attention_scores: B * H = %ld exceeds the grid's z extent", (long)a.B * a.H);
  dim3 grid((a.Tk + 63) / 64, (a.Tq + 63) / 64, a.B * a.H);
  hipLaunchKernelGGL(attn_scores_kernel<T>, grid, dim3(256), 0, s, a.q, a.k, a.ldq, a.ldk, a.bsq, a.bsk, a.kv_len, a.H, a.Tq, a.Tk, a.causal, out);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}

}  // namespace

int launch_attention_scores(const AttnArgs& a, float* out, hipStream_t s) { return launch_scores<bf16_t>(a, out, s); }
int launch_attention_scores(const AttnArgsF& a, float* out, hipStream_t s) { return launch_scores<float>(a, out, s); }

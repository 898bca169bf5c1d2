// Arithmetic shared by the KV-cached decoder step's engines (multi-launch: decode_proj.hip + attention.hip::attn_decode_kernel; one launch:
// decode_xcd.hip).  The engines must agree to the last bit (tests/test_gpu_decode_step.py), so every expression whose fp32 rounding
// depends on how hipcc contracts it (a * b + c -> fma) lives HERE, once, and is inlined into each engine: same source expression,
// same contraction, same result.  Reference semantics: TextDecoder.forward for one new token with the kv_cache hooks,
// olmoasr/model.py:786-817, 925-964; MultiHeadAttention.qkv_attention :347-442.
#pragma once
#include "kernels.h"

namespace dec {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float SCALE = 0.125f;  // 1/sqrt(64)
constexpr float NEG = -1.0e30f;

// ---- LayerNorm of a handful of rows ------------------------------------------------------------------------------------------
constexpr int MAXC = 4;  // 16-byte chunks per lane (d <= 2048)

__device__ __forceinline__ void unpack8(const u32x4_t& p, float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = bf_lo(p[i]);
    f[2 * i + 1] = bf_hi(p[i]);
  }
}

// One wave, one row of d bf16 values held RAW as chunks raw[c] = elements (lane + 64 c) * 8 .. + 7 (chunks past d/8 unused): mean and
// 1/sqrt(var + eps) exactly as ln_fwd_kernel (norm.hip) computes them.  (The values are unpacked twice instead of being kept as 32
// floats: the unpacking is exact, and the one-launch engine has no registers to spare.)
__device__ __forceinline__ void row_stats(const u32x4_t (&raw)[MAXC], int lane, int nchunk, int d, float& mean_out, float& rstd_out) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    if (lane + 64 * c < nchunk) {
      float v[8];
      unpack8(raw[c], v);
#pragma unroll
      for (int i = 0; i < 8; ++i) s += v[i];
    }
  }
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    if (lane + 64 * c < nchunk) {
      float v[8];
      unpack8(raw[c], v);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float t = v[i] - mean;
        q += t * t;
      }
    }
  }
  const float var = wave_sum(q) / (float)d;
  mean_out = mean;
  rstd_out = rsqrtf(var + 1e-5f);
}

// 8 consecutive elements k .. k + 7 of a row, normalised and rounded to bf16 (the LayerNorm's autocast output)
__device__ __forceinline__ u32x4_t ln_apply8(const u32x4_t& x8, float mu, float rs, const float* g, const float* bta, int k) {
  const f32x4_t g0 = *(const f32x4_t*)(g + k), g1 = *(const f32x4_t*)(g + k + 4);
  const f32x4_t b0 = *(const f32x4_t*)(bta + k), b1 = *(const f32x4_t*)(bta + k + 4);
  float v[8];
  unpack8(x8, v);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[i] = (v[i] - mu) * rs * g0[i] + b0[i];
    v[4 + i] = (v[4 + i] - mu) * rs * g1[i] + b1[i];
  }
  u32x4_t o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = pack_bf2(v[2 * i], v[2 * i + 1]);
  return o;
}

// ---- projection epilogue: one output value (the Linear's fp32 accumulation `acc` of row m, column n) ----------------------------
struct Epi {
  const float* bias;    // [N] or null
  int gelu;             // GELU after the bf16 rounding of the Linear output
  const bf16_t* resid;  // [M][ldr] or null: added after the rounding
  long ldr;
  bf16_t* out;          // bf16 [M][ldc] or null
  long ldc;
  float* out_f32;       // fp32 [M][ldf] or null (logits)
  long ldf;
};
// returns the bf16 output value (as float); `resid_val` is the residual element (ignored unless has_resid)
__device__ __forceinline__ float epi_value(float acc, float bias, bool gelu, bool has_resid, float resid_val) {
  float y = acc + bias;
  y = bf_round(y);  // the Linear's bf16 output
  if (gelu) y = gelu_f(y);
  if (has_resid) y = bf_round(y) + resid_val;
  return y;
}
__device__ __forceinline__ float epi_logit(float acc, float bias) { return bf_round(acc + bias); }  // (the bf16 logits of the autocast Linear, widened)

// ---- one query row against a SEGMENT of cached keys (Tq == 1) --------------------------------------------------------------------
// 256 threads: thread (grp = tid >> 3, l8 = tid & 7) owns dims l8*8 .. +7 of the head and the keys t == grp (mod 32) of the segment.
// Keys are processed in segments of <= SEG_KEYS (the scores of a segment live in LDS); a segment yields (m_s, l_s, o_s[64]) with
// its own maximum, segments are merged in order (merge_segments).  One segment (self-attention: <= 448 keys) reduces to the plain
// two-pass softmax.  Numerics as the tiled kernels: fp32 scores and normaliser, P rounded to bf16 before it multiplies V.
constexpr int SEG_KEYS = 768;
__host__ __device__ __forceinline__ int n_segments(int Tk) { return (Tk + SEG_KEYS - 1) / SEG_KEYS; }

__device__ __forceinline__ void load_q8(const u32x4_t& q4, float (&qv)[8]) { unpack8(q4, qv); }

// value of lane (lane ^ X), X in {1, 2, 4}: __shfl_xor through the DPP cross-lane path (register to register) instead of the
// ds_bpermute hipcc emits for it (an LDS-pipe round trip per step, three dependent ones per key in score8)
template <int X>
__device__ __forceinline__ float xor_lane(float v) {
  const int x = __float_as_int(v);
  int r;
  if (X == 1) {
    r = __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false);  // quad_perm [1, 0, 3, 2]
  } else if (X == 2) {
    r = __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false);  // quad_perm [2, 3, 0, 1]
  } else {
    r = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xF, 0x5, false);  // row_shl:4 into lanes 0-3, 8-11 of every row: lane i <- lane i + 4
    r = __builtin_amdgcn_update_dpp(r, x, 0x114, 0xF, 0xA, false);  // row_shr:4 into lanes 4-7, 12-15:           lane i <- lane i - 4
  }
  return __int_as_float(r);
}

// score of one key (all 8 lanes of the group return it), in the exp2 domain
__device__ __forceinline__ float score8(const float (&qv)[8], const u32x4_t& k4) {
  float d = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) d += qv[2 * i] * bf_lo(k4[i]) + qv[2 * i + 1] * bf_hi(k4[i]);
  d += xor_lane<1>(d);
  d += xor_lane<2>(d);
  d += xor_lane<4>(d);
  return d * (SCALE * LOG2E);
}
// p = exp2(score - m) (0 for a key outside the segment): normaliser and P.V accumulation of one key
__device__ __forceinline__ void accum_pv(float p, const u32x4_t& v4, float& l, float (&o)[8]) {
  l += p;
  const float pb = bf_round(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[2 * i] += pb * bf_lo(v4[i]);
    o[2 * i + 1] += pb * bf_hi(v4[i]);
  }
}
// thread tid < 64 (one head dimension each): sum of the 32 groups' partial rows, in group order
__device__ __forceinline__ void reduce_groups(const float (*red)[64], const float* lsum, int tid, float& acc, float& lt) {
  acc = 0.f;
  lt = 0.f;
#pragma unroll 8
  for (int g = 0; g < 32; ++g) {
    acc += red[g][tid];
    lt += lsum[g];
  }
}
// merge of the segments' (m_s, l_s, o_s) in segment order; returns the attention output value of this thread's dimension
constexpr int MAX_SEG = 2;  // Tk <= 1536
__device__ __forceinline__ float merge_segments(const float (&m_s)[MAX_SEG], const float (&l_s)[MAX_SEG], const float (&o_s)[MAX_SEG], int ns,
                                                float& m_out, float& l_out) {
  float m = m_s[0];
#pragma unroll
  for (int s = 1; s < MAX_SEG; ++s)
    if (s < ns) m = fmaxf(m, m_s[s]);
  float L = 0.f, O = 0.f;
#pragma unroll
  for (int s = 0; s < MAX_SEG; ++s) {
    if (s < ns) {
      const float w = __builtin_amdgcn_exp2f(m_s[s] - m);
      L += l_s[s] * w;
      O += o_s[s] * w;
    }
  }
  m_out = m;
  l_out = L;
  return L > 0.f ? O / L : 0.f;
}

}  // namespace dec

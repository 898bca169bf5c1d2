// fp32 VALIDATION kernels: the float overloads of the launchers in kernels.h.
//
// compute_dtype = OASR_DTYPE_F32 runs the SAME engine schedule (engine.hip: forward, loss, hand-written backward,
// gradient arena, optimizer) with fp32 activations, fp32 operands, fp32 accumulation and exact-erf GELU -- the
// reference's precision="float32" path (scripts/training/train_timestamps.py:2128,2220-2224; olmoasr/model.py:39,97-101).
// Its purpose is parity, not speed: it separates "bf16 rounding" from "bug" by holding logits to 1e-3 abs and gradients
// to 1e-3 rel against the fp32 CPU oracle (BASELINE.json north_star; tests/test_gpu_fp32_mode.py).  The kernels are
// deliberately plain (LDS-tiled VALU FMA, one wave per attention row, scalar accesses so no alignment or padding rule
// applies) -- a few TFLOP/s, enough for medium at batch 1-2 in about a second.
#include <math.h>

#include "kernels.h"

namespace {

inline unsigned grid_for(long work_items, int per_block = 256, int cap = 4096) {
  long b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (unsigned)b;
}

__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_exact(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * expf(-0.5f * x * x) * 0.3989422804014327f;
}

// ---- GEMM ----------------------------------------------------------------------------------------------------
// Element (row i, column j) of an operand view whose rows are `i` (see OperandView in kernels.h): plain or conv windows.
__device__ __forceinline__ float view_elem(const OperandViewF& v, long i, long j) {
  if (v.rpb) {
    const long b = i / v.rpb, t = i - b * v.rpb;
    if (j >= v.kvalid || (t == 0 && j < v.lead) || (t == v.rpb - 1 && j >= v.trail_from)) return 0.f;
    return v.ptr[b * v.bstride + t * v.ld - v.lead + j];
  }
  return v.ptr[i * v.ld + j];
}
// A(m, k): ta == 0 -> view rows are m; ta == 1 -> view rows are k (stored [K][M])
__device__ __forceinline__ float operand(const OperandViewF& v, int trans, long r, long k) {
  return trans ? view_elem(v, k, r) : view_elem(v, r, k);
}

constexpr int GT = 64, GK = 16;

__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgsF p) {
  __shared__ float As[GK][GT + 1], Bs[GK][GT + 1];
  __shared__ float csum[GT];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const long m0 = (long)blockIdx.y * GT, n0 = (long)blockIdx.x * GT;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  if (threadIdx.x < GT) csum[threadIdx.x] = 0.f;
  for (long k0 = 0; k0 < p.K; k0 += GK) {
    for (int e = threadIdx.x; e < GT * GK; e += 256) {
      // consecutive threads walk the contiguous direction of each operand
      int r, kk;
      if (p.ta) { r = e % GT; kk = e / GT; } else { kk = e % GK; r = e / GK; }
      const long m = m0 + r, k = k0 + kk;
      As[kk][r] = (m < p.M && k < p.K) ? operand(p.A, p.ta, m, k) : 0.f;
      if (p.tb) { r = e % GT; kk = e / GT; } else { kk = e % GK; r = e / GK; }
      const long n = n0 + r, k2 = k0 + kk;
      Bs[kk][r] = (n < p.N && k2 < p.K) ? operand(p.B, p.tb, n, k2) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < GK; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
  // epilogue: same order of operations as the bf16 kernels (kernels.h), without the bf16 rounding points
  float colpart[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long m = m0 + ty * 4 + i;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long n = n0 + tx * 4 + j;
      if (n >= p.N) continue;
      float v = p.alpha * acc[i][j];
      if (p.bias) v += p.bias[n];
      if (p.out_pre) p.out_pre[m * p.ldc + n] = p.act == 2 ? dgelu_exact(v) : v;
      if (p.act) v = gelu_exact(v);
      if (p.pos) v += p.pos[(m % p.pos_period) * p.N + n];
      if (p.dgelu_u) v *= p.dgelu_deriv ? p.dgelu_u[m * p.ldu + n] : dgelu_exact(p.dgelu_u[m * p.ldu + n]);
      if (p.resid) v += p.resid[m * p.ldr + n];
      if (p.out) p.out[m * p.ldc + n] = v;
      if (p.out_f32) {
        float* dst = p.out_f32 + m * p.ldc32 + n;
        if (p.atomic) atomicAdd(dst, v);
        else *dst = (p.beta != 0.f ? p.beta * *dst : 0.f) + v;
      }
      colpart[j] += v;
    }
  }
  if (p.colsum) {
#pragma unroll
    for (int j = 0; j < 4; ++j) atomicAdd(&csum[tx * 4 + j], colpart[j]);
    __syncthreads();
    if (threadIdx.x < GT && n0 + threadIdx.x < p.N) atomicAdd(p.colsum + n0 + threadIdx.x, csum[threadIdx.x]);
  }
}

// ---- attention (head_dim 64, one wave per query / key row) -----------------------------------------------------
constexpr int MAX_T = 1536;
constexpr float ATT_SCALE = 0.125f;

__device__ __forceinline__ int visible_keys(const AttnArgsF& a, int b, int i) {  // keys [0, n) are visible to query i
  int n = a.Tk;
  if (a.kv_len) n = min(n, a.kv_len[b]);
  if (a.causal) n = min(n, i + 1);
  return max(n, 0);
}

// element offset of token row t of sample b: chunk-row table (kernels.h) or batch stride
__device__ __forceinline__ long tok_off(const int32_t* rows, int b, int t, long ld, long bs) {
  return rows ? ((long)rows[b * OASR_ROWTAB + (t >> 6)] + (t & 63)) * ld : (long)b * bs + (long)t * ld;
}

__device__ __forceinline__ float dot64(const float* __restrict__ row, const float* __restrict__ vec_lds) {
  float s = 0.f;
#pragma unroll 16
  for (int c = 0; c < 64; ++c) s = fmaf(row[c], vec_lds[c], s);
  return s;
}

__global__ __launch_bounds__(64) void attn_fwd_f32_kernel(AttnArgsF a) {
  __shared__ float qs[64], ps[MAX_T];
  const int i = blockIdx.x, h = blockIdx.y, b = blockIdx.z, lane = threadIdx.x;
  if (a.q_span && i >= a.q_span[b]) return;  // span-limited forward: this query row is not computed
  qs[lane] = a.q[tok_off(a.q_rows, b, i, a.ldq, a.bsq) + h * 64 + lane];
  __syncthreads();
  const int nk = visible_keys(a, b, i);
  const float* K = a.k + h * 64;
  const float* V = a.v + h * 64;
  float mx = -INFINITY;
  for (int j = lane; j < nk; j += 64) {
    const float s = ATT_SCALE * dot64(K + tok_off(a.k_rows, b, j, a.ldk, a.bsk), qs);
    ps[j] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < nk; j += 64) {
    const float e = expf(ps[j] - mx);
    ps[j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  __syncthreads();
  float o = 0.f;
#pragma unroll 8
  for (int j = 0; j < nk; ++j) o = fmaf(ps[j], V[tok_off(a.k_rows, b, j, a.ldv, a.bsv) + lane], o);
  a.o[tok_off(a.q_rows, b, i, a.ldo, a.bso) + h * 64 + lane] = nk ? o / sum : 0.f;
  if (lane == 0) a.lse[((long)b * a.H + h) * a.Tq + i] = nk ? mx + logf(sum) : -INFINITY;
}

// per query row: delta_i, dS_ij, dQ_i
__global__ __launch_bounds__(64) void attn_bwd_q_f32_kernel(AttnArgsF a) {
  __shared__ float qs[64], dos[64], ds[MAX_T];
  const int i = blockIdx.x, h = blockIdx.y, b = blockIdx.z, lane = threadIdx.x;
  if (a.q_span && i >= a.q_span[b]) return;  // no gradient at this query position: d_o row not read, dq row not written
  const long orow = tok_off(a.q_rows, b, i, a.ldo, a.bso) + h * 64 + lane;
  qs[lane] = a.q[tok_off(a.q_rows, b, i, a.ldq, a.bsq) + h * 64 + lane];
  const float dov = a.d_o[orow];
  dos[lane] = dov;
  const float delta = wave_sum(dov * a.o[orow]);
  __syncthreads();
  const long sidx = ((long)b * a.H + h) * a.Tq + i;
  if (lane == 0) a.delta[sidx] = delta;
  const int nk = visible_keys(a, b, i);
  const float lse = a.lse[sidx];
  const float* K = a.k + h * 64;
  const float* V = a.v + h * 64;
  for (int j = lane; j < nk; j += 64) {
    const float s = ATT_SCALE * dot64(K + tok_off(a.k_rows, b, j, a.ldk, a.bsk), qs);
    const float pj = expf(s - lse);
    const float dp = dot64(V + tok_off(a.k_rows, b, j, a.ldv, a.bsv), dos);
    ds[j] = pj * (dp - delta);
  }
  __syncthreads();
  float dq = 0.f;
#pragma unroll 8
  for (int j = 0; j < nk; ++j) dq = fmaf(ds[j], K[tok_off(a.k_rows, b, j, a.ldk, a.bsk) + lane], dq);
  a.dq[tok_off(a.q_rows, b, i, a.ldq, a.bsq) + h * 64 + lane] = ATT_SCALE * dq;
  if (a.dq_colsum) atomicAdd(a.dq_colsum + h * 64 + lane, ATT_SCALE * dq);
}

// per key row: dK_j, dV_j (delta must have been written by attn_bwd_q_f32_kernel)
__global__ __launch_bounds__(64) void attn_bwd_kv_f32_kernel(AttnArgsF a) {
  __shared__ float ks[64], vs[64], ps[MAX_T], ds[MAX_T];
  const int j = blockIdx.x, h = blockIdx.y, b = blockIdx.z, lane = threadIdx.x;
  const int q_end = a.q_span ? min(a.q_span[b], a.Tq) : a.Tq;  // queries past it carry no gradient (their d_o rows are not read)
  if (a.k_rows && a.q_span && j >= q_end) return;  // chunked self-attention: this key's dk / dv rows are outside the active span
  ks[lane] = a.k[tok_off(a.k_rows, b, j, a.ldk, a.bsk) + h * 64 + lane];
  vs[lane] = a.v[tok_off(a.k_rows, b, j, a.ldv, a.bsv) + h * 64 + lane];
  __syncthreads();
  const bool key_ok = !a.kv_len || j < a.kv_len[b];
  const int i0 = a.causal ? j : 0;  // queries i >= i0 see key j
  const float* Q = a.q + h * 64;
  const float* dO = a.d_o + h * 64;
  const long sbase = ((long)b * a.H + h) * a.Tq;
  if (key_ok) {
    for (int i = i0 + lane; i < q_end; i += 64) {
      const float s = ATT_SCALE * dot64(Q + tok_off(a.q_rows, b, i, a.ldq, a.bsq), ks);
      const float lse = a.lse[sbase + i];
      const float pi = expf(s - lse);  // lse = -inf cannot happen for a row that sees key j
      const float dp = dot64(dO + tok_off(a.q_rows, b, i, a.ldo, a.bso), vs);
      ps[i] = pi;
      ds[i] = pi * (dp - a.delta[sbase + i]);
    }
  }
  __syncthreads();
  float dk = 0.f, dv = 0.f;
  if (key_ok) {
#pragma unroll 4
    for (int i = i0; i < q_end; ++i) {
      dv = fmaf(ps[i], dO[tok_off(a.q_rows, b, i, a.ldo, a.bso) + lane], dv);
      dk = fmaf(ds[i], Q[tok_off(a.q_rows, b, i, a.ldq, a.bsq) + lane], dk);
    }
  }
  a.dk[tok_off(a.k_rows, b, j, a.ldk, a.bsk) + h * 64 + lane] = ATT_SCALE * dk;
  a.dv[tok_off(a.k_rows, b, j, a.ldv, a.bsv) + h * 64 + lane] = dv;
  if (a.dv_colsum) atomicAdd(a.dv_colsum + h * 64 + lane, dv);
}

// ---- LayerNorm (one wave per row, d <= 2048) --------------------------------------------------------------------
constexpr int LN_MAXC = 32;

__global__ __launch_bounds__(256) void ln_fwd_f32_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y, float* __restrict__ mean,
                                                        float* __restrict__ rstd, long rows, int d) {
  const int lane = threadIdx.x & 63;
  const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * d;
  float v[LN_MAXC];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAXC; ++k) {
    const int c = lane + 64 * k;
    v[k] = c < d ? xr[c] : 0.f;
    s += v[k];
  }
  const float mu = wave_sum(s) / d;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAXC; ++k) {
    const int c = lane + 64 * k;
    const float t = c < d ? v[k] - mu : 0.f;
    q += t * t;
  }
  const float rs = rsqrtf(wave_sum(q) / d + 1e-5f);
#pragma unroll
  for (int k = 0; k < LN_MAXC; ++k) {
    const int c = lane + 64 * k;
    if (c < d) y[row * d + c] = (v[k] - mu) * rs * gamma[c] + beta[c];
  }
  if (lane == 0) {
    mean[row] = mu;
    rstd[row] = rs;
  }
}

__global__ __launch_bounds__(256) void ln_bwd_f32_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                        const float* __restrict__ gamma, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, const float* __restrict__ dres,
                                                        float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                        float* __restrict__ dsum, long rows, int d) {
  const int lane = threadIdx.x & 63;
  float ag[LN_MAXC], ab[LN_MAXC], as_[LN_MAXC];
#pragma unroll
  for (int k = 0; k < LN_MAXC; ++k) ag[k] = ab[k] = as_[k] = 0.f;
  for (long row = blockIdx.x * 4L + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4L) {
    const float mu = mean[row], rs = rstd[row];
    float gdy[LN_MAXC], xh[LN_MAXC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXC; ++k) {
      const int c = lane + 64 * k;
      if (c < d) {
        const float g = dy[row * d + c];
        xh[k] = (x[row * d + c] - mu) * rs;
        gdy[k] = g * gamma[c];
        ag[k] += g * xh[k];
        ab[k] += g;
        s1 += gdy[k];
        s2 += gdy[k] * xh[k];
      } else {
        xh[k] = gdy[k] = 0.f;
      }
    }
    s1 = wave_sum(s1) / d;
    s2 = wave_sum(s2) / d;
#pragma unroll
    for (int k = 0; k < LN_MAXC; ++k) {
      const int c = lane + 64 * k;
      if (c < d) {
        float v = rs * (gdy[k] - s1 - xh[k] * s2);
        if (dres) v += dres[row * d + c];
        dx[row * d + c] = v;
        as_[k] += v;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < LN_MAXC; ++k) {
    const int c = lane + 64 * k;
    if (c < d) {
      atomicAdd(dgamma + c, ag[k]);
      atomicAdd(dbeta + c, ab[k]);
      if (dsum) atomicAdd(dsum + c, as_[k]);
    }
  }
}

// ---- glue ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_conv_f32_kernel(const float* __restrict__ w, float* __restrict__ dst, int co, int ci, int ldk) {
  const long total = (long)co * ldk;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int o = (int)(i / ldk), k = (int)(i - (long)o * ldk);
    float v = 0.f;
    if (k < 3 * ci) {
      const int kk = k / ci, c = k - kk * ci;
      v = w[((long)o * ci + c) * 3 + kk];
    }
    dst[i] = v;
  }
}
__global__ __launch_bounds__(256) void mel_tm_f32_kernel(const float* __restrict__ mel, float* __restrict__ out, int C, int T, long total,
                                                        const float* __restrict__ clip_max) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {  // i over [B][T][C]
    const int c = (int)(i % C);
    const long bt = i / C;
    const int t = (int)(bt % T);
    const long b = bt / T;
    float v = mel[(b * C + c) * T + t];
    if (clip_max) v = (fmaxf(v, clip_max[b] - 8.0f) + 4.0f) * 0.25f;  // un-finalized log-mel (kernels.h)
    out[i] = v;
  }
}
__global__ __launch_bounds__(256) void embedding_fwd_f32_kernel(const int64_t* __restrict__ tok, const float* __restrict__ E,
                                                               const float* __restrict__ pos, float* __restrict__ x, int S, int d,
                                                               long total, long n_embed, const int32_t* __restrict__ rowtab) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / d;  // logical row (b, s)
    const int c = (int)(i - r * d);
    const long t = tok[r];
    const float e = (t >= 0 && t < n_embed) ? E[t * d + c] : 0.f;
    const int s = (int)(r % S);
    const long xr = rowtab ? (long)rowtab[(r / S) * OASR_ROWTAB + (s >> 6)] + (s & 63) : r;
    x[xr * d + c] = e + pos[(long)s * d + c];
  }
}
__global__ __launch_bounds__(256) void embedding_bwd_f32_kernel(const int64_t* __restrict__ tok, const float* __restrict__ dx,
                                                               float* __restrict__ dE, float* __restrict__ dpos, int B, int S, int d,
                                                               long pad_id, long n_embed, const int32_t* __restrict__ rowtab,
                                                               const int32_t* __restrict__ span) {
  const int s = blockIdx.x;
  for (int c = threadIdx.x; c < d; c += 256) {
    float acc = 0.f;
    for (int b = 0; b < B; ++b) {
      if (span && s >= span[b]) continue;
      const long r = rowtab ? (long)rowtab[b * OASR_ROWTAB + (s >> 6)] + (s & 63) : (long)b * S + s;
      const float g = dx[r * d + c];
      acc += g;
      const long t = tok[(long)b * S + s];
      if (t != pad_id && t >= 0 && t < n_embed) atomicAdd(dE + t * d + c, g);
    }
    dpos[(long)s * d + c] += acc;
  }
}
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ x, long ld, long M, int ncols, float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= ncols) return;
  float acc = 0.f;
  for (long m = blockIdx.y; m < M; m += gridDim.y) acc += x[m * ld + c];
  atomicAdd(out + c, acc);
}
__global__ __launch_bounds__(256) void col2im_dgelu_f32_kernel(const float* __restrict__ dA, const float* __restrict__ u1,
                                                              float* __restrict__ dpre1, int T1, int d, long total) {
  const int T2 = T1 >> 1;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / d;  // (b, t)
    const int c = (int)(i - r * d);
    const long b = r / T1;
    const int t = (int)(r - b * T1);
    // conv2 (k3, s2, p1) window t' covers input rows 2t' - 1 + kk: row t is tap 1 of t/2 (t even); tap 2 of t/2 and tap 0 of t/2 + 1 (t odd)
    float g;
    if ((t & 1) == 0) {
      g = dA[((b * T2 + (t >> 1)) * 3 + 1) * d + c];
    } else {
      g = dA[((b * T2 + (t >> 1)) * 3 + 2) * d + c];
      if ((t >> 1) + 1 < T2) g += dA[((b * T2 + (t >> 1) + 1) * 3 + 0) * d + c];
    }
    dpre1[i] = g * dgelu_exact(u1[i]);
  }
}
__global__ __launch_bounds__(256) void dgelu_mul_f32_kernel(const float* __restrict__ dy, const float* __restrict__ u, float* __restrict__ out,
                                                           long n) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = dy[i] * dgelu_exact(u[i]);
}
__global__ __launch_bounds__(256) void dlogits_copy_f32_kernel(const float* __restrict__ src, int V, long ld, float* __restrict__ dst, long total) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / ld;
    const int col = (int)(i - r * ld);
    dst[i] = col < V ? src[r * V + col] : 0.f;
  }
}
__global__ __launch_bounds__(256) void logits_copy_f32_kernel(const float* __restrict__ lg, long ld, int V, float* __restrict__ out, long total) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / V;
    out[i] = lg[r * ld + (i - r * V)];
  }
}

// ---- cross-entropy: one workgroup per row, logits fp32 overwritten by the gradient --------------------------------
__device__ __forceinline__ float block_reduce_f(float v, float* red, bool is_max) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = is_max ? wave_max(v) : wave_sum(v);
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return is_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void ce_f32_kernel(float* __restrict__ logits, long ld, int V, const int64_t* __restrict__ targets,
                                                    long ignore, float gscale, const int32_t* __restrict__ n_valid_dev,
                                                    float* __restrict__ row_loss, int write_grad) {
  __shared__ float red[4];
  const long row = blockIdx.x;
  float* lr = logits + row * ld;
  const long tgt = targets[row];
  if (tgt == ignore || tgt < 0 || tgt >= V) {
    if (write_grad)
      for (int c = threadIdx.x; c < ld; c += 256) lr[c] = 0.f;
    if (threadIdx.x == 0) row_loss[row] = 0.f;
    return;
  }
  float mx = -INFINITY;
  for (int c = threadIdx.x; c < V; c += 256) mx = fmaxf(mx, lr[c]);
  mx = block_reduce_f(mx, red, true);
  float sum = 0.f;
  for (int c = threadIdx.x; c < V; c += 256) sum += expf(lr[c] - mx);
  sum = block_reduce_f(sum, red, false);
  const float lse = mx + logf(sum);
  if (threadIdx.x == 0) row_loss[row] = lse - lr[tgt];
  if (!write_grad) return;
  __syncthreads();
  const float g = gscale / (float)max(1, *n_valid_dev);
  for (int c = threadIdx.x; c < ld; c += 256) lr[c] = c < V ? (expf(lr[c] - lse) - (c == tgt ? 1.f : 0.f)) * g : 0.f;
}

}  // namespace

// ================================================ launchers =======================================================
int launch_gemm(const GemmArgsF& a, hipStream_t stream) {
  OASR_REQUIRE(a.A.ptr && a.B.ptr && a.M > 0 && a.N > 0 && a.K > 0, "gemm(f32): bad args (M=%d N=%d K=%d)", a.M, a.N, a.K);
  OASR_REQUIRE(a.out || a.out_pre || a.out_f32, "gemm(f32): no output");
  OASR_REQUIRE(!a.pos || a.pos_period > 0, "gemm(f32): pos needs pos_period");
  hipLaunchKernelGGL(gemm_f32_kernel, dim3(cdiv(a.N, GT), cdiv(a.M, GT)), dim3(256), 0, stream, a);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}

static int check_attn(const AttnArgsF& a, bool bwd) {
  OASR_REQUIRE(a.q && a.k && a.v && a.o && a.lse && a.B > 0 && a.H > 0 && a.Tq > 0 && a.Tk > 0, "attention(f32): bad args");
  OASR_REQUIRE(a.Tq <= MAX_T && a.Tk <= MAX_T, "attention(f32): Tq/Tk up to %d", MAX_T);
  if (bwd) OASR_REQUIRE(a.d_o && a.delta && a.dq && a.dk && a.dv, "attention_bwd(f32): null gradient pointers");
  return OASR_OK;
}
int launch_attention_fwd(const AttnArgsF& a, hipStream_t s) {
  int rc = check_attn(a, false);
  if (rc) return rc;
  hipLaunchKernelGGL(attn_fwd_f32_kernel, dim3(a.Tq, a.H, a.B), dim3(64), 0, s, a);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}
int launch_attention_bwd(const AttnArgsF& a, hipStream_t s) {
  int rc = check_attn(a, true);
  if (rc) return rc;
  hipLaunchKernelGGL(attn_bwd_q_f32_kernel, dim3(a.Tq, a.H, a.B), dim3(64), 0, s, a);
  hipLaunchKernelGGL(attn_bwd_kv_f32_kernel, dim3(a.Tk, a.H, a.B), dim3(64), 0, s, a);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}

int launch_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, long rows, int d,
                         hipStream_t s) {
  OASR_REQUIRE(x && gamma && beta && y && mean && rstd && d > 0 && d <= 64 * LN_MAXC, "layernorm_fwd(f32): bad args (d=%d)", d);
  if (rows <= 0) return OASR_OK;
  hipLaunchKernelGGL(ln_fwd_f32_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, gamma, beta, y, mean, rstd, rows, d);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}
int launch_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, const float* dres,
                         float* dx, float* dgamma, float* dbeta, float* dsum, long rows, int d, hipStream_t s) {
  OASR_REQUIRE(dy && x && gamma && mean && rstd && dx && dgamma && dbeta && d > 0 && d <= 64 * LN_MAXC, "layernorm_bwd(f32): bad args");
  if (rows <= 0) return OASR_OK;
  long nb = (rows + 3) / 4;
  if (nb > 512) nb = 512;
  hipLaunchKernelGGL(ln_bwd_f32_kernel, dim3((unsigned)nb), dim3(256), 0, s, dy, x, gamma, mean, rstd, dres, dx, dgamma, dbeta, dsum, rows, d);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}

int launch_pack_conv_weight(const float* w, float* dst, int co, int ci, int ldk, hipStream_t s) {
  OASR_REQUIRE(w && dst && ldk >= 3 * ci, "pack_conv_weight(f32): bad args");
  hipLaunchKernelGGL(pack_conv_f32_kernel, dim3(grid_for((long)co * ldk)), dim3(256), 0, s, w, dst, co, ci, ldk);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}
int launch_mel_to_time_major(const float* mel, float* out, int B, int n_mels, int T, hipStream_t s, const float* clip_max) {
  OASR_REQUIRE(mel && out, "mel_to_time_major(f32): bad args");
  const long total = (long)B * T * n_mels;
  hipLaunchKernelGGL(mel_tm_f32_kernel, dim3(grid_for(total)), dim3(256), 0, s, mel, out, n_mels, T, total, clip_max);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}
int launch_embedding_fwd(const int64_t* tok, const float* E, const float* pos, float* x, int B, int S, int d, long n_embed, hipStream_t s,
                         const int32_t* rows_tab) {
  OASR_REQUIRE(tok && E && pos && x, "embedding_fwd(f32): bad args");
  const long total = (long)B * S * d;
  hipLaunchKernelGGL(embedding_fwd_f32_kernel, dim3(grid_for(total)), dim3(256), 0, s, tok, E, pos, x, S, d, total, n_embed, rows_tab);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}
int launch_embedding_bwd(const int64_t* tok, const float* dx, float* dE, float* dpos, int B, int S, int d, long pad_id, long n_embed,
                         hipStream_t s, const int32_t* rows_tab, const int32_t* span) {
  OASR_REQUIRE(tok && dx && dE && dpos, "embedding_bwd(f32): bad args");
  hipLaunchKernelGGL(embedding_bwd_f32_kernel, dim3(S), dim3(256), 0, s, tok, dx, dE, dpos, B, S, d, pad_id, n_embed, rows_tab, span);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}
int launch_colsum_accum(const float* x, long ld, long M, int ncols, float* out, hipStream_t s) {
  OASR_REQUIRE(x && out && ncols > 0, "colsum(f32): bad args");
  if (M <= 0) return OASR_OK;
  long gy = M < 256 ? M : 256;
  hipLaunchKernelGGL(colsum_f32_kernel, dim3(cdiv(ncols, 256), (unsigned)gy), dim3(256), 0, s, x, ld, M, ncols, out);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}
int launch_conv2_col2im_dgelu(const float* dA, const float* u1, float* dpre1, int B, int T1, int d, hipStream_t s) {
  OASR_REQUIRE(dA && u1 && dpre1 && T1 % 2 == 0, "col2im(f32): bad args");
  const long total = (long)B * T1 * d;
  hipLaunchKernelGGL(col2im_dgelu_f32_kernel, dim3(grid_for(total)), dim3(256), 0, s, dA, u1, dpre1, T1, d, total);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}
int launch_dgelu_mul(const float* dy, const float* u, float* out, long n, hipStream_t s) {
  OASR_REQUIRE(dy && u && out, "dgelu_mul(f32): bad args");
  hipLaunchKernelGGL(dgelu_mul_f32_kernel, dim3(grid_for(n)), dim3(256), 0, s, dy, u, out, n);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}
int launch_dlogits_from_f32(const float* src, int V, long rows, long ld, float* dst, hipStream_t s) {
  OASR_REQUIRE(src && dst && V <= ld, "dlogits_from_f32(f32): bad args");
  const long total = rows * ld;
  hipLaunchKernelGGL(dlogits_copy_f32_kernel, dim3(grid_for(total)), dim3(256), 0, s, src, V, ld, dst, total);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}
int launch_logits_to_f32(const float* logits, long ld, long rows, int V, float* out, hipStream_t s) {
  const long total = rows * V;
  hipLaunchKernelGGL(logits_copy_f32_kernel, dim3(grid_for(total)), dim3(256), 0, s, logits, ld, V, out, total);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}
int launch_cross_entropy(float* logits, long ld, int V, const int64_t* targets, long rows, long ignore, float gscale,
                         const int32_t* n_valid_dev, float* row_loss, int write_grad, hipStream_t s) {
  OASR_REQUIRE(logits && targets && n_valid_dev && row_loss && V <= ld, "cross_entropy(f32): bad args");
  if (rows <= 0) return OASR_OK;
  hipLaunchKernelGGL(ce_f32_kernel, dim3((unsigned)rows), dim3(256), 0, s, logits, ld, V, targets, ignore, gscale, n_valid_dev, row_loss,
                     write_grad);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}

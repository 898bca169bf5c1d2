// HBM-bound glue kernels (casts, layout packs, embedding gather/scatter, column sums, conv fold).
// All vectorised to 16-byte accesses where the layout allows; grid-stride, <= 2048 workgroups.
#include "kernels.h"

namespace {

inline unsigned grid_for(long work_items, int per_block = 256, int cap = 2048) {
  long b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (unsigned)b;
}

__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n) {
  const long n8 = n >> 3;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const f32x4_t a = ((const f32x4_t*)src)[2 * i], b = ((const f32x4_t*)src)[2 * i + 1];
    u32x4_t o;
    o[0] = pack_bf2(a[0], a[1]);
    o[1] = pack_bf2(a[2], a[3]);
    o[2] = pack_bf2(b[0], b[1]);
    o[3] = pack_bf2(b[2], b[3]);
    ((u32x4_t*)dst)[i] = o;
  }
  if (blockIdx.x == 0)
    for (long i = (n8 << 3) + threadIdx.x; i < n; i += 256) dst[i] = f2bf_dev(src[i]);
}

// w [co][ci][3] -> dst [co][ldk], k = kk*ci_n + ci
__global__ __launch_bounds__(256) void pack_conv_kernel(const float* __restrict__ w, bf16_t* __restrict__ dst, int co, int ci,
                                                        int ldk) {
  const long total = (long)co * ldk;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int o = (int)(i / ldk), k = (int)(i - (long)o * ldk);
    float v = 0.f;
    if (k < 3 * ci) {
      const int kk = k / ci, c = k - kk * ci;
      v = w[((long)o * ci + c) * 3 + kk];
    }
    dst[i] = f2bf_dev(v);
  }
}

__global__ __launch_bounds__(256) void unpack_conv_grad_kernel(const float* __restrict__ g, float* __restrict__ dw, int co, int ci,
                                                               int ldk) {
  const long total = (long)co * ci * 3;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int kk = (int)(i % 3);
    const long oc = i / 3;
    const int c = (int)(oc % ci), o = (int)(oc / ci);
    dw[i] += g[(long)o * ldk + kk * ci + c];
  }
}

__global__ __launch_bounds__(256) void pack_embedding_kernel(const float* __restrict__ e, bf16_t* __restrict__ dst, long n_valid,
                                                             long n_total) {
  const long n8 = n_total >> 3;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    u32x4_t o = {0u, 0u, 0u, 0u};
    if (i * 8 < n_valid) {  // rows are multiples of 8 elements wide, so a chunk is entirely valid or entirely pad
      const f32x4_t a = ((const f32x4_t*)e)[2 * i], b = ((const f32x4_t*)e)[2 * i + 1];
      o[0] = pack_bf2(a[0], a[1]);
      o[1] = pack_bf2(a[2], a[3]);
      o[2] = pack_bf2(b[0], b[1]);
      o[3] = pack_bf2(b[2], b[3]);
    }
    ((u32x4_t*)dst)[i] = o;
  }
}

// mel [B][C][T] f32 -> out [B][T][C] bf16 through a 32(t) x C LDS tile so both sides stay coalesced
// clip_max (optional, [B]): `mel` is the UN-finalized log10 mel power (oasr_log_mel_raw) and whisper's last two lines are applied here,
// on the way through: max(x, clip_max[b] - 8), then (x + 4) / 4 -- the same fp32 operations logmel_finalize performs
__global__ __launch_bounds__(256) void mel_tm_kernel(const float* __restrict__ mel, bf16_t* __restrict__ out, int C, int T,
                                                     const float* __restrict__ clip_max) {
  __shared__ float tile[128][33];
  const int b = blockIdx.y, t0 = blockIdx.x * 32;
  const float* src = mel + (long)b * C * T;
  const float floor_v = clip_max ? clip_max[b] - 8.0f : 0.f;
  for (int i = threadIdx.x; i < C * 32; i += 256) {
    const int c = i >> 5, tt = i & 31;
    float v = (t0 + tt < T) ? src[(long)c * T + t0 + tt] : 0.f;
    if (clip_max) v = (fmaxf(v, floor_v) + 4.0f) * 0.25f;
    tile[c][tt] = v;
  }
  __syncthreads();
  bf16_t* dst = out + ((long)b * T + t0) * C;
  for (int i = threadIdx.x; i < C * 32; i += 256) {
    const int tt = i / C, c = i - tt * C;
    if (t0 + tt < T) dst[(long)tt * C + c] = f2bf_dev(tile[c][tt]);
  }
}

__global__ __launch_bounds__(256) void embedding_fwd_kernel(const int64_t* __restrict__ tok, const float* __restrict__ E,
                                                            const float* __restrict__ pos, bf16_t* __restrict__ x, int S, int d,
                                                            long rows, long n_embed, const int32_t* __restrict__ rowtab) {
  const int cpr = d >> 3;  // 8-element chunks per row
  const long total = rows * cpr;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long lr = i / cpr;  // logical row (b, s)
    const int ch = (int)(i - lr * cpr);
    const long t = tok[lr];
    const int s = (int)(lr % S);
    const long r = rowtab ? (long)rowtab[(lr / S) * OASR_ROWTAB + (s >> 6)] + (s & 63) : lr;  // row of x
    // ids outside the table (nn.Embedding would raise; e.g. the pad id fed to the pad-row-less inference model) read as
    // a zero row instead of out-of-bounds memory
    const bool ok = t >= 0 && t < n_embed;
    const f32x4_t z4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4_t a0 = ok ? *(const f32x4_t*)(E + t * d + ch * 8) : z4, a1 = ok ? *(const f32x4_t*)(E + t * d + ch * 8 + 4) : z4;
    const f32x4_t p0 = *(const f32x4_t*)(pos + (long)s * d + ch * 8), p1 = *(const f32x4_t*)(pos + (long)s * d + ch * 8 + 4);
    u32x4_t o;
    o[0] = pack_bf2(a0[0] + p0[0], a0[1] + p0[1]);
    o[1] = pack_bf2(a0[2] + p0[2], a0[3] + p0[3]);
    o[2] = pack_bf2(a1[0] + p1[0], a1[1] + p1[1]);
    o[3] = pack_bf2(a1[2] + p1[2], a1[3] + p1[3]);
    *(u32x4_t*)(x + r * d + ch * 8) = o;
  }
}

// one workgroup per sequence position s: dpos[s] += sum_b dx[b,s,:] (plain RMW, each s owned by one block);
// dE[tok[b,s]] += dx[b,s,:] with fp32 atomics (tokens repeat across the batch)
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const int64_t* __restrict__ tok, const bf16_t* __restrict__ dx,
                                                            float* __restrict__ dE, float* __restrict__ dpos, int B, int S, int d,
                                                            long pad_id, long n_embed, const int32_t* __restrict__ rowtab,
                                                            const int32_t* __restrict__ span) {
  const int s = blockIdx.x;
  for (int c = threadIdx.x; c < d; c += 256) {
    float acc = 0.f;
    for (int b = 0; b < B; ++b) {
      if (span && s >= span[b]) continue;  // no gradient at this position (and its dx row was never written)
      const long r = rowtab ? (long)rowtab[b * OASR_ROWTAB + (s >> 6)] + (s & 63) : (long)b * S + s;
      const float g = bf2f(dx[r * d + c]);
      acc += g;
      const long t = tok[(long)b * S + s];
      if (t != pad_id && t >= 0 && t < n_embed) unsafeAtomicAdd(dE + t * d + c, g);  // ids outside the table: no scatter (fwd read zeros)
    }
    dpos[(long)s * d + c] += acc;
  }
}

// out[n] += sum_m x[m][n]; block = 64 lanes x 8 columns (16-byte loads) x 4 row-lanes, rows strided over gridDim.y
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ x, long ld, long M, int ncols, float* __restrict__ out) {
  __shared__ float red[4][512];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int col = (blockIdx.x * 64 + cl) * 8;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  if (col < ncols) {
    const long step = (long)gridDim.y * 4;
    long m = (long)blockIdx.y * 4 + rl;
    for (; m + 3 * step < M; m += 4 * step) {  // four independent 16-byte loads in flight per lane
      u32x4_t p[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) p[u] = *(const u32x4_t*)(x + (m + u * step) * ld + col);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[2 * j] += bf_lo(p[u][j]);
          acc[2 * j + 1] += bf_hi(p[u][j]);
        }
    }
    for (; m < M; m += step) {
      const u32x4_t p = *(const u32x4_t*)(x + m * ld + col);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[2 * j] += bf_lo(p[j]);
        acc[2 * j + 1] += bf_hi(p[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[rl][cl * 8 + j] = acc[j];
  __syncthreads();
  for (int j = threadIdx.x; j < 512; j += 256) {
    const int c = blockIdx.x * 512 + j;
    if (c < ncols) unsafeAtomicAdd(out + c, red[0][j] + red[1][j] + red[2][j] + red[3][j]);
  }
}

__global__ __launch_bounds__(256) void col2im_dgelu_kernel(const bf16_t* __restrict__ dA, const bf16_t* __restrict__ u1,
                                                           bf16_t* __restrict__ dpre1, int T1, int d, long total8) {
  const int cpr = d >> 3, T2 = T1 >> 1;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total8; i += (long)gridDim.x * 256) {
    const long r = i / cpr;  // (b, t)
    const int ch = (int)(i - r * cpr);
    const long b = r / T1;
    const int t = (int)(r - b * T1);
    float g[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = 0.f;
    auto add = [&](long row2, int kk) {
      const u32x4_t p = *(const u32x4_t*)(dA + (row2 * 3 + kk) * d + ch * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        g[2 * j] += bf_lo(p[j]);
        g[2 * j + 1] += bf_hi(p[j]);
      }
    };
    // conv2 window t' covers input rows 2t'-1+kk
    if ((t & 1) == 0) {
      add(b * T2 + (t >> 1), 1);
    } else {
      add(b * T2 + (t >> 1), 2);
      if ((t >> 1) + 1 < T2) add(b * T2 + (t >> 1) + 1, 0);
    }
    const u32x4_t u = *(const u32x4_t*)(u1 + r * d + ch * 8);
    u32x4_t o;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      o[j] = pack_bf2(bf_round(g[2 * j]) * dgelu_f(bf_lo(u[j])), bf_round(g[2 * j + 1]) * dgelu_f(bf_hi(u[j])));
    *(u32x4_t*)(dpre1 + r * d + ch * 8) = o;
  }
}

__global__ __launch_bounds__(256) void logits_to_f32_kernel(const bf16_t* __restrict__ lg, long ld, int V, float* __restrict__ out,
                                                           long total) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / V;
    const int col = (int)(i - r * V);
    out[i] = bf2f(lg[r * ld + col]);
  }
}
// d(loss)/d(logits) handed in by torch.autograd (fp32 [rows, V]) -> the engine's padded bf16 [rows, ld] buffer (padding columns zero)
__global__ __launch_bounds__(256) void dlogits_from_f32_kernel(const float* __restrict__ src, int V, long ld, bf16_t* __restrict__ dst, long total) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / ld;
    const int col = (int)(i - r * ld);
    dst[i] = col < V ? f2bf_dev(src[r * V + col]) : (bf16_t)0;
  }
}
__global__ __launch_bounds__(256) void dgelu_mul_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ u, bf16_t* __restrict__ out,
                                                       long n8) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const u32x4_t a = ((const u32x4_t*)dy)[i], b = ((const u32x4_t*)u)[i];
    u32x4_t o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = pack_bf2(bf_lo(a[j]) * dgelu_f(bf_lo(b[j])), bf_hi(a[j]) * dgelu_f(bf_hi(b[j])));
    ((u32x4_t*)out)[i] = o;
  }
}

__global__ __launch_bounds__(256) void axpy_kernel(const float* __restrict__ src, float* __restrict__ dst, long n, float a) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] += a * src[i];
}

}  // namespace

int launch_cast_f32_bf16(const float* src, bf16_t* dst, long n, hipStream_t s) {
  OASR_REQUIRE(src && dst && n >= 0, "cast: bad args");
  if (n == 0) return OASR_OK;
  hipLaunchKernelGGL(cast_kernel, dim3(grid_for(n / 8 + 1)), dim3(256), 0, s, src, dst, n);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}
int launch_pack_conv_weight(const float* w, bf16_t* dst, int co, int ci, int ldk, hipStream_t s) {
  OASR_REQUIRE(w && dst && ldk >= 3 * ci, "pack_conv_weight: bad args");
  hipLaunchKernelGGL(pack_conv_kernel, dim3(grid_for((long)co * ldk)), dim3(256), 0, s, w, dst, co, ci, ldk);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}
int launch_unpack_conv_grad(const float* g, float* dw, int co, int ci, int ldk, hipStream_t s) {
  OASR_REQUIRE(g && dw && ldk >= 3 * ci, "unpack_conv_grad: bad args");
  hipLaunchKernelGGL(unpack_conv_grad_kernel, dim3(grid_for((long)co * ci * 3)), dim3(256), 0, s, g, dw, co, ci, ldk);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}
int launch_pack_embedding(const float* e, bf16_t* dst, int rows, int rows_pad, int d, hipStream_t s) {
  OASR_REQUIRE(e && dst && rows_pad >= rows && d % 8 == 0, "pack_embedding: bad args");
  hipLaunchKernelGGL(pack_embedding_kernel, dim3(grid_for((long)rows_pad * d / 8)), dim3(256), 0, s, e, dst, (long)rows * d,
                     (long)rows_pad * d);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}
int launch_mel_to_time_major(const float* mel, bf16_t* out, int B, int n_mels, int T, hipStream_t s, const float* clip_max) {
  OASR_REQUIRE(mel && out && n_mels <= 128, "mel_to_time_major: bad args");
  hipLaunchKernelGGL(mel_tm_kernel, dim3(cdiv(T, 32), B), dim3(256), 0, s, mel, out, n_mels, T, clip_max);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}
int launch_embedding_fwd(const int64_t* tok, const float* E, const float* pos, bf16_t* x, int B, int S, int d, long n_embed,
                         hipStream_t s, const int32_t* rows_tab) {
  OASR_REQUIRE(tok && E && pos && x && d % 8 == 0, "embedding_fwd: bad args");
  OASR_REQUIRE(!rows_tab || (S % 64 == 0 && S <= 64 * OASR_ROWTAB), "embedding_fwd: chunk rows need S %% 64 == 0");
  const long rows = (long)B * S;
  hipLaunchKernelGGL(embedding_fwd_kernel, dim3(grid_for(rows * (d / 8))), dim3(256), 0, s, tok, E, pos, x, S, d, rows, n_embed, rows_tab);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}
int launch_embedding_bwd(const int64_t* tok, const bf16_t* dx, float* dE, float* dpos, int B, int S, int d, long pad_id, long n_embed,
                         hipStream_t s, const int32_t* rows_tab, const int32_t* span) {
  OASR_REQUIRE(tok && dx && dE && dpos, "embedding_bwd: bad args");
  OASR_REQUIRE(!rows_tab || (S % 64 == 0 && S <= 64 * OASR_ROWTAB), "embedding_bwd: chunk rows need S %% 64 == 0");
  hipLaunchKernelGGL(embedding_bwd_kernel, dim3(S), dim3(256), 0, s, tok, dx, dE, dpos, B, S, d, pad_id, n_embed, rows_tab, span);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}

// ---- supervised-span tables (engine.hip) ------------------------------------------------------------------------------------
namespace {
struct SpanArg {
  uint16_t span[512];
};
// one workgroup.  Chunk (b, c) = positions [64c, 64c + 64) of sample b; active iff 64c < span[b].  Row order: active chunks first,
// position-block-major (c outer, b inner), then the inactive ones in the same order.
__global__ __launch_bounds__(1024) void build_span_tables_kernel(SpanArg sp, int B, int S, const int64_t* __restrict__ targets, long ignore,
                                                                 int32_t* __restrict__ rows, int32_t* __restrict__ span_dev,
                                                                 int64_t* __restrict__ targets_phys) {
  __shared__ int cnt_act[OASR_ROWTAB + 1];  // active chunks in position blocks < c
  __shared__ int rows_s[512 * OASR_ROWTAB];
  const int nch = S >> 6, tid = threadIdx.x;
  if (tid <= OASR_ROWTAB) cnt_act[tid] = 0;
  __syncthreads();
  if (tid < nch) {  // active samples of position block tid
    int n = 0;
    for (int b = 0; b < B; ++b) n += (64 * tid < (int)sp.span[b]) ? 1 : 0;
    for (int c = tid + 1; c <= nch; ++c) atomicAdd(&cnt_act[c], n);
  }
  __syncthreads();
  const int total_act = cnt_act[nch];
  for (int i = tid; i < B * OASR_ROWTAB; i += 1024) {
    const int b = i / OASR_ROWTAB, c = i - b * OASR_ROWTAB;
    int row = 0x3fffffff;  // past every tensor: out of range for the buffer descriptors
    if (c < nch) {
      const bool act = 64 * c < (int)sp.span[b];
      // rank of (b, c) among the chunks of its kind: blocks before c, then samples before b inside block c
      int before = 0;
      for (int bb = 0; bb < b; ++bb) before += ((64 * c < (int)sp.span[bb]) == act) ? 1 : 0;
      const int blk_before = act ? cnt_act[c] : c * B - cnt_act[c];
      row = 64 * ((act ? 0 : total_act) + blk_before + before);
    }
    rows[i] = row;
    rows_s[i] = row;
  }
  for (int b = tid; b < B; b += 1024) span_dev[b] = ((int)sp.span[b] + 63) & ~63;
  __syncthreads();
  // targets in row order for the active rows (every target of an inactive chunk is `ignore` by the caller's contract)
  for (long i = tid; i < (long)B * S; i += 1024) {
    const int b = (int)(i / S), s = (int)(i - (long)b * S);
    if (s < (((int)sp.span[b] + 63) & ~63)) targets_phys[(long)rows_s[b * OASR_ROWTAB + (s >> 6)] + (s & 63)] = targets[i];
  }
  (void)ignore;
}
}  // namespace
int launch_build_span_tables(const int32_t* span_host, int B, int S, const int64_t* targets, long ignore, int32_t* rows, int32_t* span_dev,
                             int64_t* targets_phys, long* active_rows, hipStream_t s) {
  OASR_REQUIRE(span_host && targets && rows && span_dev && targets_phys && active_rows, "build_span_tables: null pointer");
  OASR_REQUIRE(B > 0 && B <= 512 && S > 0 && S % 64 == 0 && S <= 64 * OASR_ROWTAB, "build_span_tables: need B <= 512, S %% 64 == 0, S <= %d (B=%d S=%d)",
               64 * OASR_ROWTAB, B, S);
  SpanArg sp;
  long act = 0;
  for (int b = 0; b < B; ++b) {
    OASR_REQUIRE(span_host[b] >= 0 && span_host[b] <= S, "build_span_tables: span[%d] = %d outside [0, %d]", b, span_host[b], S);
    sp.span[b] = (uint16_t)span_host[b];
    act += (span_host[b] + 63) / 64 * 64;
  }
  for (int b = B; b < 512; ++b) sp.span[b] = 0;
  *active_rows = act;
  hipLaunchKernelGGL(build_span_tables_kernel, dim3(1), dim3(1024), 0, s, sp, B, S, targets, ignore, rows, span_dev, targets_phys);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}
int launch_colsum_accum(const bf16_t* x, long ld, long M, int ncols, float* out, hipStream_t s) {
  OASR_REQUIRE(x && out && ncols > 0 && ncols % 8 == 0 && ld % 8 == 0, "colsum: bad args");
  if (M <= 0) return OASR_OK;
  const int gx = cdiv(ncols, 512);
  long gy = 2048 / gx;
  if (gy < 1) gy = 1;
  if (gy > (M + 3) / 4) gy = (M + 3) / 4;
  hipLaunchKernelGGL(colsum_kernel, dim3(gx, (unsigned)gy), dim3(256), 0, s, x, ld, M, ncols, out);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}
int launch_conv2_col2im_dgelu(const bf16_t* dA, const bf16_t* u1, bf16_t* dpre1, int B, int T1, int d, hipStream_t s) {
  OASR_REQUIRE(dA && u1 && dpre1 && d % 8 == 0 && T1 % 2 == 0, "col2im: bad args");
  const long total8 = (long)B * T1 * (d / 8);
  hipLaunchKernelGGL(col2im_dgelu_kernel, dim3(grid_for(total8)), dim3(256), 0, s, dA, u1, dpre1, T1, d, total8);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}
int launch_axpy_f32(const float* src, float* dst, long n, float a, hipStream_t s) {
  if (n <= 0) return OASR_OK;
  hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, s, src, dst, n, a);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}
int launch_dgelu_mul(const bf16_t* dy, const bf16_t* u, bf16_t* out, long n, hipStream_t s) {
  OASR_REQUIRE(dy && u && out && n % 8 == 0, "dgelu_mul: bad args");
  hipLaunchKernelGGL(dgelu_mul_kernel, dim3(grid_for(n / 8)), dim3(256), 0, s, dy, u, out, n / 8);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}
int launch_dlogits_from_f32(const float* src, int V, long rows, long ld, bf16_t* dst, hipStream_t s) {
  OASR_REQUIRE(src && dst && V <= ld, "dlogits_from_f32: bad args");
  const long total = rows * ld;
  hipLaunchKernelGGL(dlogits_from_f32_kernel, dim3(grid_for(total, 256, 4096)), dim3(256), 0, s, src, V, ld, dst, total);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}
int launch_logits_to_f32(const bf16_t* logits, long ld, long rows, int V, float* out, hipStream_t s) {
  const long total = rows * V;
  hipLaunchKernelGGL(logits_to_f32_kernel, dim3(grid_for(total, 256, 4096)), dim3(256), 0, s, logits, ld, V, out, total);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}

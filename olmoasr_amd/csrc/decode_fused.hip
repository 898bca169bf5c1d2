// One-launch KV-cached decoder step for gfx950: every layer of TextDecoder.forward for ONE new token per sequence
// (olmoasr/model.py:786-817 with the kv_cache hooks of :925-964) in a single persistent kernel.
//
// Why: a decode step is latency-bound, not bandwidth-bound.  As separate launches it is 12 kernels per layer (146 for the
// 12-layer model), each a few microseconds of work behind a launch gap -- 1.12 ms of GPU time and 0.89 ms of host enqueue per
// step at B = 16 (scripts/decode_step_probe.py), against ~0.3 ms that its bytes need.  Here one workgroup per CU stays
// resident and walks the step's phases, separated by a device-wide barrier (one atomic counter in HBM/L2):
//   per layer:  [LN1 + q|k|v projection -> cache row]  | self-attention over the cached keys | out-proj + residual
//               [LN2 + cross query]                    | cross-attention over the encoder K/V | out-proj + residual
//               [LN3 + MLP1 + GELU]                    | MLP2 + residual
//   then        [final LN + logits against the token embedding] -> fp32 logits of the next position.
// LayerNorm never makes a pass of its own: every workgroup recomputes the B row statistics (B <= 32 rows of d) and
// normalises the rows on their way into the MFMA operands.  Projections are the skinny-GEMM scheme of gemm.hip
// (32 output columns per work item, K split over the 4 waves, weights streamed once straight into MFMA operands).
// Rounding points are those of the multi-launch path (bf16 LN output, bf16 Linear output before GELU / residual), so the
// two paths agree to bf16 rounding of the fp32 accumulation order; tests/test_gpu_decode_parity.py holds both to the oracle.
//
// The barrier spin is bounded: a workgroup that waits ~2^22 polls raises the error flag and every workgroup drains out,
// so a scheduling accident can cost a wrong step (reported as OASR_EHIP by the caller's check), never a hung GPU.
#include "kernels.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float SCALE = 0.125f;  // 1/sqrt(64)
constexpr float NEG = -1.0e30f;
constexpr int MAXC = 4;  // LayerNorm: 16-byte chunks per lane (d <= 2048)

struct ProjSmem {
  float red[4][16][64];  // K-split partial accumulators of a projection tile / attention output partials ([32][64] view)
  float mean[32], rstd[32];
};
struct Smem : ProjSmem {
  float sc[1536];  // attention scores of one (b, h)
  float lsum[32];
  float wmax[4];
};

__device__ __forceinline__ void unpack8(const u32x4_t& p, float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = bf_lo(p[i]);
    f[2 * i + 1] = bf_hi(p[i]);
  }
}

// Device-wide barrier between two phases.  `target` = (barriers passed so far + 1) * gridDim.x on a counter zeroed before
// the launch.  Release: every thread's stores are ordered before thread 0's atomic by __syncthreads + the fence; acquire:
// thread 0's fence after the spin invalidates this CU's view before the workgroup continues.
__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned* err, unsigned target) {
  __shared__ int bail;
  __syncthreads();
  if (threadIdx.x == 0) {
    // release once (write this CU's results back from its L2), poll with plain device-scope atomic loads, acquire once:
    // an acquire on every poll would invalidate the XCD's L2 hundreds of times per barrier (measured: 50 us per barrier)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    int b = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 1023u) == 0 && (spins > (1u << 22) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        b = 1;
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    bail = b;
  }
  __syncthreads();
  return bail != 0;
}

// Row statistics of x [M][d] (bf16) into LDS: wave w takes rows w, w + 4, ...  Same arithmetic as ln_fwd_kernel (norm.hip).
__device__ __forceinline__ void ln_stats(const bf16_t* x, int M, int d, ProjSmem& sm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = d >> 3;
  for (int row = wave; row < M; row += 4) {
    float v[MAXC][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nchunk) {
        unpack8(*(const u32x4_t*)(x + (long)row * d + ch * 8), v[c]);
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
    if (lane == 0) {
      sm.mean[row] = mean;
      sm.rstd[row] = rsqrtf(var + 1e-5f);
    }
  }
  __syncthreads();
}

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
        for (int j = 0; j < 4; ++j) {
          const int kk = k + 16 * j + h * 8;
          const f32x4_t g0 = *(const f32x4_t*)(g + kk), g1 = *(const f32x4_t*)(g + kk + 4);
          const f32x4_t b0 = *(const f32x4_t*)(bta + kk), b1 = *(const f32x4_t*)(bta + kk + 4);
          float v[8];
          unpack8(xq[j], v);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            v[i] = (v[i] - mu) * rs * g0[i] + b0[i];
            v[4 + i] = (v[4 + i] - mu) * rs * g1[i] + b1[i];
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) xq[j][i] = pack_bf2(v[2 * i], v[2 * i + 1]);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)  // D'[n][m]: lane owns output row m = lane & 31
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wq[j]), __builtin_bit_cast(bf16x8_t, xq[j]), acc, 0, 0, 0);
    }
    for (; k < k_end; k += 16) {
      const u32x4_t wq = __builtin_nontemporal_load((const u32x4_t*)(wp + k));
      u32x4_t xq = *(const u32x4_t*)(xp + k);
      if (LN) {
        const int kk = k + h * 8;
        const f32x4_t g0 = *(const f32x4_t*)(g + kk), g1 = *(const f32x4_t*)(g + kk + 4);
        const f32x4_t b0 = *(const f32x4_t*)(bta + kk), b1 = *(const f32x4_t*)(bta + kk + 4);
        float v[8];
        unpack8(xq, v);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v[i] = (v[i] - mu) * rs * g0[i] + b0[i];
          v[4 + i] = (v[4 + i] - mu) * rs * g1[i] + b1[i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) xq[i] = pack_bf2(v[2 * i], v[2 * i + 1]);
      }
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
          float y = v[i] + (e.bias ? e.bias[n + i] : 0.f);
          if (e.out_f32) e.out_f32[(long)m * e.ldf + n + i] = bf_round(y);  // (the bf16 logits of the autocast Linear, widened)
          if (e.out) {
            y = bf_round(y);  // the Linear's bf16 output
            if (e.gelu) y = gelu_f(y);
            if (e.resid) y = bf_round(y) + bf2f(e.resid[(long)m * e.ldr + n + i]);
            e.out[(long)m * e.ldc + n + i] = f2bf_dev(y);
          }
        }
      }
    }
    __syncthreads();  // red is reused by the next tile
  }
}

// One query token per (b, h) against Tk cached keys/values: attn_decode_kernel (attention.hip) as a work-item loop.
__device__ __forceinline__ void attn_phase(const bf16_t* q, long bsq, const bf16_t* kbase, const bf16_t* vbase, long ldkv, long bskv, int Tk,
                                           bf16_t* o, long bso, int B, int H, Smem& sm) {
  const int tid = threadIdx.x, l8 = tid & 7, grp = tid >> 3, wave = tid >> 6, lane = tid & 63;
  float (*red)[64] = (float (*)[64]) & sm.red[0][0][0];  // [32][64] view of the first 8 KiB
  for (int item = blockIdx.x; item < B * H; item += gridDim.x) {
    const int b = item / H, hh = item - b * H;
    const u32x4_t q4 = *(const u32x4_t*)(q + (long)b * bsq + hh * 64 + l8 * 8);
    float qv[8];
    unpack8(q4, qv);
    const bf16_t* kp = kbase + (long)b * bskv + hh * 64 + l8 * 8;
    const bf16_t* vp = vbase + (long)b * bskv + hh * 64 + l8 * 8;
    float mx = NEG;
    for (int t0 = grp; t0 < Tk; t0 += 32 * 8) {  // 8 independent 16-byte loads in flight per lane
      u32x4_t k4[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = t0 + 32 * u;
        k4[u] = *(const u32x4_t*)(kp + (long)(t < Tk ? t : Tk - 1) * ldkv);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = t0 + 32 * u;
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) d += qv[2 * i] * bf_lo(k4[u][i]) + qv[2 * i + 1] * bf_hi(k4[u][i]);
        d += __shfl_xor(d, 1, 64);
        d += __shfl_xor(d, 2, 64);
        d += __shfl_xor(d, 4, 64);
        const float s2 = d * (SCALE * LOG2E);
        if (t < Tk) {
          if (l8 == 0) sm.sc[t] = s2;
          mx = fmaxf(mx, s2);
        }
      }
    }
    mx = wave_max(mx);
    if (lane == 0) sm.wmax[wave] = mx;
    __syncthreads();
    const float m = fmaxf(fmaxf(sm.wmax[0], sm.wmax[1]), fmaxf(sm.wmax[2], sm.wmax[3]));
    float l = 0.f, ov[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) ov[j] = 0.f;
    for (int t0 = grp; t0 < Tk; t0 += 32 * 8) {
      u32x4_t v4[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = t0 + 32 * u;
        v4[u] = *(const u32x4_t*)(vp + (long)(t < Tk ? t : Tk - 1) * ldkv);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = t0 + 32 * u;
        const float p = t < Tk ? __builtin_amdgcn_exp2f(sm.sc[t] - m) : 0.f;
        l += p;
        const float pb = bf_round(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          ov[2 * i] += pb * bf_lo(v4[u][i]);
          ov[2 * i + 1] += pb * bf_hi(v4[u][i]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[grp][l8 * 8 + j] = ov[j];
    if (l8 == 0) sm.lsum[grp] = l;
    __syncthreads();
    if (tid < 64) {
      float acc = 0.f, lt = 0.f;
#pragma unroll 8
      for (int gg = 0; gg < 32; ++gg) {
        acc += red[gg][tid];
        lt += sm.lsum[gg];
      }
      const float val = lt > 0.f ? acc / lt : 0.f;
      const float nb = __shfl_xor(val, 1, 64);
      if ((tid & 1) == 0) *(uint32_t*)(o + (long)b * bso + hh * 64 + tid) = pack_bf2(val, nb);
    }
    __syncthreads();  // sc / red / lsum / wmax are reused by the next item
  }
}

__global__ __launch_bounds__(256) void decode_fused_kernel(FusedDecArgs a) {
  __shared__ Smem sm;
  const int B = a.B, d = a.d;
  unsigned nbar = 0;
#define OASR_GRID_BARRIER()                                                    \
  do {                                                                         \
    ++nbar;                                                                    \
    if (grid_barrier(a.counter, a.err, nbar * gridDim.x)) return;              \
  } while (0)

  // phase 0: token + positional embedding of position `pos` (ids outside the table read as a zero row, as embedding_fwd)
  {
    const int cpr = d >> 3;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < B * cpr; i += gridDim.x * 256) {
      const int r = i / cpr, ch = i - r * cpr;
      const long t = a.tok[r];
      const bool ok = t >= 0 && t < a.n_embed;
      const f32x4_t z4 = {0.f, 0.f, 0.f, 0.f};
      const f32x4_t a0 = ok ? *(const f32x4_t*)(a.E + t * d + ch * 8) : z4, a1 = ok ? *(const f32x4_t*)(a.E + t * d + ch * 8 + 4) : z4;
      const f32x4_t p0 = *(const f32x4_t*)(a.pos_emb + ch * 8), p1 = *(const f32x4_t*)(a.pos_emb + ch * 8 + 4);
      u32x4_t o;
      o[0] = pack_bf2(a0[0] + p0[0], a0[1] + p0[1]);
      o[1] = pack_bf2(a0[2] + p0[2], a0[3] + p0[3]);
      o[2] = pack_bf2(a1[0] + p1[0], a1[1] + p1[1]);
      o[3] = pack_bf2(a1[2] + p1[2], a1[3] + p1[3]);
      *(u32x4_t*)(a.r0 + (long)r * d + ch * 8) = o;
    }
  }
  OASR_GRID_BARRIER();
  const long cache_bs = (long)a.S_max * 3 * d;
  for (int l = 0; l < a.L; ++l) {
    const FusedDecLayer& P = a.layers[l];
    bf16_t* row = P.self_qkv + (long)a.pos * 3 * d;  // [b] stride cache_bs: q | k | v of this position
    {  // LN1 + fused q|k|v projection straight into the cache row
      const Epi e{P.bqkv, 0, nullptr, 0, row, cache_bs, nullptr, 0};
      proj_phase<true>(a.r0, B, d, P.wqkv, 3 * d, P.ln1_g, P.ln1_b, e, sm);
    }
    OASR_GRID_BARRIER();
    attn_phase(row, cache_bs, P.self_qkv + d, P.self_qkv + 2 * d, 3 * d, cache_bs, a.pos + 1, a.o, d, B, a.H, sm);
    OASR_GRID_BARRIER();
    {
      const Epi e{P.bo, 0, a.r0, d, a.r1, d, nullptr, 0};
      proj_phase<false>(a.o, B, d, P.wo, d, nullptr, nullptr, e, sm);
    }
    OASR_GRID_BARRIER();
    {  // LN2 + cross-attention query
      const Epi e{P.bcq, 0, nullptr, 0, a.q, d, nullptr, 0};
      proj_phase<true>(a.r1, B, d, P.wcq, d, P.ln2_g, P.ln2_b, e, sm);
    }
    OASR_GRID_BARRIER();
    attn_phase(a.q, d, P.cross_kv, P.cross_kv + d, 2 * d, (long)a.Te * 2 * d, a.Te, a.o, d, B, a.H, sm);
    OASR_GRID_BARRIER();
    {
      const Epi e{P.bco, 0, a.r1, d, a.r2, d, nullptr, 0};
      proj_phase<false>(a.o, B, d, P.wco, d, nullptr, nullptr, e, sm);
    }
    OASR_GRID_BARRIER();
    {  // LN3 + MLP1 + GELU
      const Epi e{P.b1, 1, nullptr, 0, a.hg, 4 * d, nullptr, 0};
      proj_phase<true>(a.r2, B, d, P.w1, 4 * d, P.ln3_g, P.ln3_b, e, sm);
    }
    OASR_GRID_BARRIER();
    {
      const Epi e{P.b2, 0, a.r2, d, a.r0, d, nullptr, 0};
      proj_phase<false>(a.hg, B, 4 * d, P.w2, d, nullptr, nullptr, e, sm);
    }
    OASR_GRID_BARRIER();
  }
  {  // final LN + logits (fp32, unpadded rows)
    const Epi e{nullptr, 0, nullptr, 0, nullptr, 0, a.logits, a.V};
    proj_phase<true>(a.r0, B, d, a.Wemb, a.V, a.lnf_g, a.lnf_b, e, sm);
  }
#undef OASR_GRID_BARRIER
}

// The same projection as a launch of its own (the multi-launch step): LayerNorm + Linear (+ GELU / residual / fp32 logits) of a
// handful of token rows in ONE kernel, one 32-column tile per workgroup.
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

int launch_decode_fused(const FusedDecArgs& a, hipStream_t s) {
  OASR_REQUIRE(a.B > 0 && a.B <= 32 && a.d % 64 == 0 && a.d <= 2048 && a.H * 64 == a.d, "decode_fused: B=%d d=%d H=%d unsupported", a.B, a.d, a.H);
  OASR_REQUIRE(a.pos >= 0 && a.pos < a.S_max && a.pos + 1 <= 1536 && a.Te <= 1536, "decode_fused: pos=%d Te=%d outside the score buffer", a.pos, a.Te);
  OASR_REQUIRE(a.tok && a.E && a.pos_emb && a.layers && a.counter && a.err && a.logits, "decode_fused: null pointer");
  static const int n_cu = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
    return n > 0 ? n : 256;
  }();
  // every workgroup must be resident for the device-wide barrier: one per CU (256 threads, ~31 KiB of LDS -- several fit)
  OASR_CHECK_HIP(hipMemsetAsync(a.counter, 0, 2 * sizeof(unsigned), s));  // counter | err are adjacent
  hipLaunchKernelGGL(decode_fused_kernel, dim3(n_cu), dim3(256), 0, s, a);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}

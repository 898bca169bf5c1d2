// bf16 MFMA GEMM for gfx950 with fp32 accumulation and a fused epilogue.
//
// Implements every Linear / Conv1d / tied-logits contraction of the reference (olmoasr/model.py:97-101 Linear,
// :104-195 Conv1d as window GEMMs, :768-770 logits) and their autograd backward (dgrad, wgrad), SURVEY.md §2.3
// K2,K3,K5,K9,K12,K14.
//
// Tiling: 128x128x64 per 256-thread workgroup, 2x2 waves, each wave a 64x64 sub-tile as 2x2
// v_mfma_f32_32x32x16_bf16 blocks (64 accumulator VGPRs).  Operands are staged HBM -> VGPR -> LDS with
// bounds-checked buffer loads (out-of-range rows / conv padding read as zero) and double-buffered in LDS (64 KiB,
// two workgroups per CU), one barrier per K-tile.  Operands whose reduction index is NOT contiguous in memory
// (dgrad's W[N][K], wgrad's dY[M][N] and X[M][K]) are kept in their natural layout and transposed on the way
// into the matrix core with ds_read_b64_tr_b16, so no transposed copies of weights or activations exist in HBM.
// LDS images are XOR-swizzled so both ds_read_b128 (k-contiguous tiles) and the transpose reads are
// bank-conflict free.  The MFMA is issued with swapped operands (D'[n][m]) so each lane ends up holding 4
// consecutive columns of one output row -> 8-byte bf16 / 16-byte fp32 row-contiguous stores.
#include <vector>

#include "kernels.h"

namespace {

// ---- optional per-launch timing (bench.py's live roofline measurement): HIP events on the launch stream --------
struct GemmProfile {
  bool on = false;
  std::vector<hipEvent_t> events;  // pairs (start, stop)
  struct Rec {
    int kind;
    double flops;
  };
  std::vector<Rec> recs;
};
GemmProfile g_prof;

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;
constexpr unsigned OOB = 0x80000000u;  // > num_records of every descriptor below -> hardware returns 0

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}

// Issue the 4 x 16-byte loads this thread contributes to one 128x64 (or 64x128) operand tile.
//   !TRANS: tile rows = output rows [row0, row0+128), cols = reduction [k0, k0+64)
//    TRANS: tile rows = reduction  [k0, k0+64),       cols = output rows [row0, row0+128)
template <bool TRANS>
__device__ __forceinline__ void issue_loads(const OperandView& v, int R, int K, int row0, int k0, int tid,
                                            u32x4_t (&regs)[4]) {
  if (!TRANS) {
    long base_el;
    if (v.rpb) {
      const int b0 = row0 / v.rpb, t0 = row0 - b0 * v.rpb;
      base_el = (long)b0 * v.bstride + (long)t0 * v.ld - v.lead + k0;
    } else {
      base_el = (long)row0 * v.ld + k0;
    }
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(v.ptr + base_el);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int id = tid + 256 * i;
      const int rl = id >> 3, c16 = id & 7;
      const int r = row0 + rl, k = k0 + c16 * 8;
      bool ok = (r < R) && (k < K);
      long off_el;
      if (v.rpb) {
        const int b = r / v.rpb, t = r - b * v.rpb;
        off_el = (long)b * v.bstride + (long)t * v.ld - v.lead + k - base_el;
        ok = ok && (k < v.kvalid) && !(t == 0 && k < v.lead) && !(t == v.rpb - 1 && k >= v.trail_from);
      } else {
        off_el = (long)rl * v.ld + c16 * 8;
      }
      const unsigned voff = ok ? (unsigned)(off_el * 2) : OOB;
      regs[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0);
    }
  } else {
    long base_el;
    if (v.rpb) {
      const int b0 = k0 / v.rpb, t0 = k0 - b0 * v.rpb;
      base_el = (long)b0 * v.bstride + (long)t0 * v.ld - v.lead + row0;
    } else {
      base_el = (long)k0 * v.ld + row0;
    }
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(v.ptr + base_el);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int id = tid + 256 * i;
      const int kl = id >> 4, c16 = id & 15;
      const int kk = k0 + kl, col = row0 + c16 * 8;
      bool ok = (kk < K) && (col < R);
      long off_el;
      if (v.rpb) {
        const int b = kk / v.rpb, t = kk - b * v.rpb;
        off_el = (long)b * v.bstride + (long)t * v.ld - v.lead + col - base_el;
        ok = ok && (col < v.kvalid) && !(t == 0 && col < v.lead) && !(t == v.rpb - 1 && col >= v.trail_from);
      } else {
        off_el = (long)kl * v.ld + c16 * 8;
      }
      const unsigned voff = ok ? (unsigned)(off_el * 2) : OOB;
      regs[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0);
    }
  }
}

template <bool TRANS>
__device__ __forceinline__ void store_tile(char* lds, int tid, const u32x4_t (&regs)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = tid + 256 * i;
    int addr;
    if (!TRANS) {
      const int rl = id >> 3, c16 = id & 7;
      addr = rl * 128 + ((c16 ^ ((rl >> 1) & 7)) << 4);
    } else {
      const int kl = id >> 4, c16 = id & 15;
      addr = kl * 256 + (((((c16 >> 2) ^ (kl & 3)) << 2) | (c16 & 3)) << 4);
    }
    *(u32x4_t*)(lds + addr) = regs[i];
  }
}

typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;

// MFMA 32x32x16 operand fragment for rows [sub, sub+32) of the tile, k-step ks (16 wide):
// lane l holds row (l & 31), k = ks*16 + (l >> 5)*8 + 0..7.
template <bool TRANS>
__device__ __forceinline__ bf16x8_t read_frag(const char* lds, int sub, int ks, int lane) {
  if (!TRANS) {
    const int row = sub + (lane & 31);
    const int c16 = ks * 2 + (lane >> 5);
    const int addr = row * 128 + ((c16 ^ ((row >> 1) & 7)) << 4);
    return *(const bf16x8_t*)(lds + addr);
  } else {
    const int G = lane >> 4, i = lane & 15;
    const int krow = ks * 16 + (G >> 1) * 8 + (i >> 2);
    const int col = sub + (G & 1) * 16 + (i & 3) * 4;
    const int c16 = col >> 3;
    const int addr = krow * 256 + (((((c16 >> 2) ^ (krow & 3)) << 2) | (c16 & 3)) << 4) + (col & 7) * 2;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lds + addr));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lds + addr + 4 * 256));
    const s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, v);
  }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int ntile = gridDim.x;
  const int bid = xcd_remap(blockIdx.x, ntile);
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int kt_total = (p.K + BK - 1) / BK;
  const int per = (kt_total + p.split_k - 1) / p.split_k;
  const int kt0 = blockIdx.y * per;
  const int kt1 = min(kt_total, kt0 + per);
  if (kt0 >= kt1) return;

  // stage s: A tile at smem + s*2*TILE_BYTES, B tile right behind it

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  u32x4_t ra[4], rb[4];
  issue_loads<TA>(p.A, p.M, p.K, m0, kt0 * BK, tid, ra);
  issue_loads<TB>(p.B, p.N, p.K, n0, kt0 * BK, tid, rb);
  store_tile<TA>(smem, tid, ra);
  store_tile<TB>(smem + TILE_BYTES, tid, rb);
  __syncthreads();

  for (int kt = kt0; kt < kt1; ++kt) {
    const int cur = (kt - kt0) & 1;
    const bool more = (kt + 1 < kt1);
    const char* cA = smem + cur * 2 * TILE_BYTES;
    const char* cB = cA + TILE_BYTES;
    char* nA = smem + (cur ^ 1) * 2 * TILE_BYTES;
    if (more) {
      issue_loads<TA>(p.A, p.M, p.K, m0, (kt + 1) * BK, tid, ra);
      issue_loads<TB>(p.B, p.N, p.K, n0, (kt + 1) * BK, tid, rb);
    }
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8_t af[2], bfr[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        af[t] = read_frag<TA>(cA, wm * 64 + t * 32, ks, lane);
        bfr[t] = read_frag<TB>(cB, wn * 64 + t * 32, ks, lane);
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[nt], af[mt], acc[mt][nt], 0, 0, 0);
    }
    if (more) {
      store_tile<TA>(nA, tid, ra);
      store_tile<TB>(nA + TILE_BYTES, tid, rb);
    }
    __syncthreads();
  }

  // ---- epilogue: lane holds, for row m, columns n = nbase + 8q + 4h + (0..3) --------------------------------
  const int h = lane >> 5;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int m = m0 + wm * 64 + mt * 32 + (lane & 31);
    if (m >= p.M) continue;
    const int pos_row = p.pos ? (m % p.pos_period) : 0;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn * 64 + nt * 32 + 8 * q + 4 * h;
        if (n >= p.N) continue;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = p.alpha * acc[mt][nt][q * 4 + i];
        if (p.bias) {
          const f32x4_t b4 = *(const f32x4_t*)(p.bias + n);
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] += b4[i];
        }
        if (p.out_pre) {
          u32x2_t o;
          o[0] = pack_bf2(v[0], v[1]);
          o[1] = pack_bf2(v[2], v[3]);
          *(u32x2_t*)(p.out_pre + (long)m * p.ldc + n) = o;
        }
        if (p.act == 1) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = gelu_f(bf_round(v[i]));
        }
        if (p.pos) {
          const f32x4_t p4 = *(const f32x4_t*)(p.pos + (long)pos_row * p.N + n);
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = bf_round(v[i]) + p4[i];
        }
        if (p.dgelu_u) {
          const u32x2_t u = *(const u32x2_t*)(p.dgelu_u + (long)m * p.ldu + n);
          v[0] = bf_round(v[0]) * dgelu_f(bf_lo(u[0]));
          v[1] = bf_round(v[1]) * dgelu_f(bf_hi(u[0]));
          v[2] = bf_round(v[2]) * dgelu_f(bf_lo(u[1]));
          v[3] = bf_round(v[3]) * dgelu_f(bf_hi(u[1]));
        }
        if (p.resid) {
          const u32x2_t r = *(const u32x2_t*)(p.resid + (long)m * p.ldr + n);
          v[0] = bf_round(v[0]) + bf_lo(r[0]);
          v[1] = bf_round(v[1]) + bf_hi(r[0]);
          v[2] = bf_round(v[2]) + bf_lo(r[1]);
          v[3] = bf_round(v[3]) + bf_hi(r[1]);
        }
        if (p.out) {
          u32x2_t o;
          o[0] = pack_bf2(v[0], v[1]);
          o[1] = pack_bf2(v[2], v[3]);
          *(u32x2_t*)(p.out + (long)m * p.ldc + n) = o;
        }
        if (p.out_f32) {
          float* dst = p.out_f32 + (long)m * p.ldc32 + n;
          if (p.atomic) {
#pragma unroll
            for (int i = 0; i < 4; ++i) unsafeAtomicAdd(dst + i, v[i]);
          } else {
            f32x4_t c4;
            if (p.beta != 0.f) {
              c4 = *(const f32x4_t*)dst;
#pragma unroll
              for (int i = 0; i < 4; ++i) c4[i] = p.beta * c4[i] + v[i];
            } else {
#pragma unroll
              for (int i = 0; i < 4; ++i) c4[i] = v[i];
            }
            *(f32x4_t*)dst = c4;
          }
        }
      }
    }
  }
}

template <bool TA, bool TB>
int launch_t(const GemmArgs& a, hipStream_t stream) {
  static bool attr = false;
  const int lds = 4 * TILE_BYTES;
  if (!attr) {
    OASR_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_kernel<TA, TB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr = true;
  }
  const int tiles = cdiv(a.M, BM) * cdiv(a.N, BN);
  dim3 grid(tiles, a.split_k);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (g_prof.on) {
    const size_t idx = g_prof.recs.size();
    while (g_prof.events.size() < 2 * (idx + 1)) {
      hipEvent_t e;
      OASR_CHECK_HIP(hipEventCreate(&e));
      g_prof.events.push_back(e);
    }
    e0 = g_prof.events[2 * idx];
    e1 = g_prof.events[2 * idx + 1];
    // algorithmic flops: conv windows count their real kernel width, not the zero padding
    const double kk = a.A.rpb ? (double)(a.ta ? a.K : a.A.kvalid) : (double)a.K;
    const double nn = (a.B.rpb && a.tb) ? (double)a.B.kvalid : (double)a.N;
    g_prof.recs.push_back({(TA ? 2 : 0) + (TB ? 1 : 0), 2.0 * (double)a.M * nn * kk});
    OASR_CHECK_HIP(hipEventRecord(e0, stream));
  }
  hipLaunchKernelGGL((gemm_kernel<TA, TB>), grid, dim3(256), lds, stream, a);
  OASR_LAUNCH_CHECK();
  if (e1) OASR_CHECK_HIP(hipEventRecord(e1, stream));
  return OASR_OK;
}

}  // namespace

int launch_gemm(const GemmArgs& a, hipStream_t stream) {
  OASR_REQUIRE(a.A.ptr && a.B.ptr, "gemm: null operand");
  OASR_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm: bad shape %d %d %d", a.M, a.N, a.K);
  OASR_REQUIRE((a.N % 4) == 0, "gemm: N (%d) must be a multiple of 4", a.N);
  OASR_REQUIRE((a.A.ld % 8) == 0 && (a.B.ld % 8) == 0, "gemm: operand leading dims must be multiples of 8 (16-byte loads)");
  OASR_REQUIRE(a.ta || (a.K % 8) == 0 || a.A.rpb, "gemm: K must be a multiple of 8 for k-contiguous A");
  OASR_REQUIRE(!a.ta || (a.M % 8) == 0 || true, "gemm");
  OASR_REQUIRE(a.split_k >= 1, "gemm: split_k");
  OASR_REQUIRE(a.split_k == 1 || (a.atomic && a.out_f32 && !a.out && !a.out_pre), "gemm: split_k > 1 needs atomic fp32 output only");
  OASR_REQUIRE(a.out || a.out_f32 || a.out_pre, "gemm: no output");
  if (!a.ta && !a.tb) return launch_t<false, false>(a, stream);
  if (!a.ta && a.tb) return launch_t<false, true>(a, stream);
  if (a.ta && !a.tb) return launch_t<true, false>(a, stream);
  return launch_t<true, true>(a, stream);
}

void gemm_profile_enable(int on) {
  g_prof.on = on != 0;
  if (on) g_prof.recs.clear();
}

// Sums elapsed ms / algorithmic flops / launch count per kernel variant (index = 2*ta + tb).  Synchronises.
int gemm_profile_collect(double ms[4], double flops[4], long count[4]) {
  for (int i = 0; i < 4; ++i) {
    ms[i] = 0;
    flops[i] = 0;
    count[i] = 0;
  }
  for (size_t i = 0; i < g_prof.recs.size(); ++i) {
    OASR_CHECK_HIP(hipEventSynchronize(g_prof.events[2 * i + 1]));
    float t = 0.f;
    OASR_CHECK_HIP(hipEventElapsedTime(&t, g_prof.events[2 * i], g_prof.events[2 * i + 1]));
    const int k = g_prof.recs[i].kind;
    ms[k] += t;
    flops[k] += g_prof.recs[i].flops;
    count[k] += 1;
  }
  g_prof.recs.clear();
  return OASR_OK;
}

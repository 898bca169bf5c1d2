// bf16 MFMA GEMM for gfx950 with fp32 accumulation and a fused epilogue.
//
// Implements every Linear / Conv1d / tied-logits contraction of the reference (olmoasr/model.py:97-101 Linear,
// :104-195 Conv1d as window GEMMs, :768-770 logits) and their autograd backward (dgrad, wgrad), SURVEY.md §2.3
// K2,K3,K5,K9,K12,K14.
//
// Four kernels; the first three share the operand views, LDS images and fragment readers:
//  * oasr_gemm_pp_kernel   256x256x64 "ping-pong": 8 waves, one workgroup per CU, direct-to-LDS (buffer_load ... lds)
//    staging that is never drained inside the K loop (one counted vmcnt per K-tile), two wave groups one barrier
//    apart so one of them is always inside an MFMA section.  Default for every bf16-output GEMM with more than half
//    a wave of tiles (all Linear forward / dgrad GEMMs of the training step).
//  * oasr_gemm_fast_kernel 256x128x64 (4 waves; 3 workgroups per CU for the split-K / atomic wgrad variant, 2 for
//    bf16 outputs) and 256x256x64 (8 waves, 2 stages): direct-to-LDS staging, one stage drained per barrier;
//    co-resident workgroups hide each other's drains.  wgrad (fp32 atomics, XCD-owned K ranges) and small problems.
//  * gemm_kernel           128x128x64, register-staged with bounds-checked buffer loads: conv window views (rpb != 0),
//    K not a multiple of 64, fp32 / positional-embedding / odd-stride outputs.
//  * gemm_skinny_kernel    M <= 64 rows (decode steps): 32 columns per workgroup, K split over the 4 waves, weights streamed
//    from L2/HBM straight into MFMA operands.
// Operands whose reduction index is NOT contiguous in memory (dgrad's W[N][K], wgrad's dY[M][N] and X[M][K]) are kept
// in their natural layout and transposed on the way into the matrix core with ds_read_b64_tr_b16, so no transposed
// copies of weights or activations exist in HBM.  LDS images are XOR-swizzled so both ds_read_b128 (k-contiguous
// tiles) and the transpose reads are bank-conflict free.  For bf16 outputs the MFMA is issued with swapped operands
// (D'[n][m]) so each lane holds 4 consecutive columns of one output row; results leave through a wave-private LDS
// tile so that every global access of the epilogue (stores, residual / dGELU-input reads) is 16 bytes per lane.
#include <math.h>
#include <stdlib.h>

#include <map>
#include <string>
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
    const char* name;  // kernel symbol as rocprofv3 prints it (template arguments included)
    int M = 0, N = 0, K = 0, epi = 0;  // shape + epilogue summary (bit 0 bias, 1 residual, 2 GELU, 3 GELU' input, 4 colsum, 5 atomic/split-K)
    int lane = 0;                      // 0 = the caller's stream with the chip to itself, 1 = a lowest-priority side stream of the span step,
                                       // 2 = the caller's stream while side-stream filler is in flight (gemm_profile_lane)
  };
  std::vector<Rec> recs;
  int lane = 0;
  // A side-stream launch is low-priority filler: its begin-to-end span (HIP events and rocprofv3 alike) includes the time it waits for
  // compute units behind the main stream's workgroups, so it is a queueing time, not a kernel time.  Records carry the lane they were
  // launched on and gemm_profile_collect() reports the lanes apart ("symbol [side]"; "symbol [shared]" = main-stream launches of the decoder
  // phases, which give up compute units to that filler and run longer for it): the roofline of a symbol is that of the launches that have
  // the chip to themselves (profiles/r05_side_streams.txt: mixing them turned 0.40 into 0.33 while the step got faster).
  void push(Rec r) {
    r.lane = lane;
    recs.push_back(r);
  }
};
GemmProfile g_prof;
bool g_force_general = false;
int g_stagger = 0, g_stagger_phases = 2;  // experiment hook (oasr_gemm_set_stagger)
int g_pp_dma_in_mma = -1;                 // ping-pong kernel variant (oasr_gemm_set_variant): -1 = default
int g_pp_persistent = -1;                 // ping-pong kernel launched persistent: -1 = default rule, 0 / 1 forced
int g_epi_flags = -1;                     // epilogue memory policy (GemmArgs::epi_flags): -1 = default
int g_fast_geometry = 0;  // 0 = heuristic, 1 = force 256x128 (4 waves), 2 = force 256x256 (8 waves, 2 stages)  // tests: run the register-staged general kernel even where the fast path applies

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

// Fused epilogue math for 4 consecutive columns n..n+3 of output row m (see GemmArgs in kernels.h for the order of
// ops): returns the pre-activation and the final value packed to bf16; v[] holds the final fp32 values on return.
__device__ __forceinline__ void epilogue_math(const GemmArgs& p, int m, int n, int pos_row, float (&v)[4], u32x2_t& pre, u32x2_t& fin) {
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] *= p.alpha;
  if (p.bias) {
    const f32x4_t b4 = *(const f32x4_t*)(p.bias + n);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] += b4[i];
  }
  pre[0] = pack_bf2(v[0], v[1]);
  pre[1] = pack_bf2(v[2], v[3]);
  if (p.act == 1) {
    v[0] = gelu_f(bf_lo(pre[0]));
    v[1] = gelu_f(bf_hi(pre[0]));
    v[2] = gelu_f(bf_lo(pre[1]));
    v[3] = gelu_f(bf_hi(pre[1]));
  } else if (p.act == 2) {
    float d[4];
    gelu_and_deriv(bf_lo(pre[0]), v[0], d[0]);
    gelu_and_deriv(bf_hi(pre[0]), v[1], d[1]);
    gelu_and_deriv(bf_lo(pre[1]), v[2], d[2]);
    gelu_and_deriv(bf_hi(pre[1]), v[3], d[3]);
    pre[0] = pack_bf2(d[0], d[1]);
    pre[1] = pack_bf2(d[2], d[3]);
  }
  if (p.pos) {
    const f32x4_t p4 = *(const f32x4_t*)(p.pos + (long)pos_row * p.N + n);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = bf_round(v[i]) + p4[i];
  }
  if (p.dgelu_u) {
    const u32x2_t u = *(const u32x2_t*)(p.dgelu_u + (long)m * p.ldu + n);
    v[0] = bf_round(v[0]) * (p.dgelu_deriv ? bf_lo(u[0]) : dgelu_f(bf_lo(u[0])));
    v[1] = bf_round(v[1]) * (p.dgelu_deriv ? bf_hi(u[0]) : dgelu_f(bf_hi(u[0])));
    v[2] = bf_round(v[2]) * (p.dgelu_deriv ? bf_lo(u[1]) : dgelu_f(bf_lo(u[1])));
    v[3] = bf_round(v[3]) * (p.dgelu_deriv ? bf_hi(u[1]) : dgelu_f(bf_hi(u[1])));
  }
  if (p.resid) {
    const u32x2_t r = *(const u32x2_t*)(p.resid + (long)m * p.ldr + n);
    v[0] = bf_round(v[0]) + bf_lo(r[0]);
    v[1] = bf_round(v[1]) + bf_hi(r[0]);
    v[2] = bf_round(v[2]) + bf_lo(r[1]);
    v[3] = bf_round(v[3]) + bf_hi(r[1]);
  }
  fin[0] = pack_bf2(v[0], v[1]);
  fin[1] = pack_bf2(v[2], v[3]);
}

__device__ __forceinline__ void epilogue_f32(const GemmArgs& p, int m, int n, const float (&v)[4]) {
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

// direct (8-byte) stores: general kernel
__device__ __forceinline__ void epilogue_store(const GemmArgs& p, int m, int n, int pos_row, float (&v)[4]) {
  u32x2_t pre, fin;
  epilogue_math(p, m, n, pos_row, v, pre, fin);
  if (p.out_pre) *(u32x2_t*)(p.out_pre + (long)m * p.ldc + n) = pre;
  if (p.out) *(u32x2_t*)(p.out + (long)m * p.ldc + n) = fin;
  if (p.out_f32) epilogue_f32(p, m, n, v);
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
        for (int i = 0; i < 4; ++i) v[i] = acc[mt][nt][q * 4 + i];
        epilogue_store(p, m, n, pos_row, v);
      }
    }
  }
}


// ====================================================================================================================
// Fast path (plain operands, K % 64 == 0): operand tiles go L2/HBM -> LDS directly with buffer_load_dwordx4 ... lds
// (no VGPR staging, no ds_write).  The LDS image is lane-linear per wave instruction, so the bank-conflict swizzle is
// applied to the per-lane SOURCE address and undone by the same XOR on the fragment read.  Every wave owns a
// 128x64 sub-tile = 4x2 MFMA 32x32x16 blocks (128 accumulator VGPRs).  Two geometries:
//   * 256x256x64, 8 waves (2x4), two LDS stages (128 KiB): the next K-tile streams in while this one is multiplied;
//     one barrier per K-tile, one workgroup per CU, two waves per SIMD.                       (large problems)
//   * 256x128x64, 4 waves (2x2), one LDS stage (48 KiB): 2-3 workgroups per CU hide each other's load latency.
// M/N tails are handled by clamping source rows (garbage only reaches output rows/cols that are never stored).
// SWAP: MFMA issued as D'[n][m] (lane owns one output row, 4 consecutive columns -> vector stores); !SWAP: D[m][n]
// (lane owns one output column -> a wave's fp32 atomics hit 2 x 128 contiguous bytes): used for split-K wgrad.
constexpr int FBM = 256;

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef __attribute__((address_space(3))) u32x2_t* lds_u32x2_ptr;
typedef __attribute__((address_space(3))) u32x4_t* lds_u32x4_ptr;

// Byte offsets (from the tile's base row/column `rel0`, default row0) of this lane's 16-byte pieces of the operand
// image that starts at row/column row0.
template <bool TRANS, int ROWS, int NI, int NW>
__device__ __forceinline__ void fast_offsets(const OperandView& v, int R, int row0, int lane, int wave, unsigned (&off)[NI],
                                             int rel0 = -1) {
  rel0 = rel0 < 0 ? row0 : rel0;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int q = j * NW + wave;  // 1 KiB chunk index inside the tile image
    if (!TRANS) {
      const int row = q * 8 + (lane >> 3), phys = lane & 7;
      const int c16 = phys ^ ((row >> 1) & 7);
      int gr = row0 + row;
      gr = gr < R ? gr : R - 1;
      off[j] = (unsigned)(((long)(gr - rel0) * v.ld + c16 * 8) * 2);
    } else {
      constexpr int CPR = ROWS / 8;  // 16-byte pieces per k-row
      constexpr int KPC = 64 / CPR;  // k-rows per 1 KiB chunk
      const int krow = q * KPC + lane / CPR, phys = lane % CPR;
      const int c16 = phys ^ ((krow & 3) << 2);
      int col = row0 + c16 * 8;
      col = col + 8 <= R ? col : R - 8;
      off[j] = (unsigned)(((long)krow * v.ld + (col - rel0)) * 2);
    }
  }
}

template <bool TRANS, int RB /*row bytes of a transposed tile*/>
__device__ __forceinline__ bf16x8_t fast_frag(const char* lds, int sub, int ks, int lane) {
  if (!TRANS) {
    const int row = sub + (lane & 31);
    const int c16 = ks * 2 + (lane >> 5);
    return *(const bf16x8_t*)(lds + row * 128 + ((c16 ^ ((row >> 1) & 7)) << 4));
  } else {
    const int G = lane >> 4, i = lane & 15;
    const int krow = ks * 16 + (G >> 1) * 8 + (i >> 2);
    const int col = sub + (G & 1) * 16 + (i & 3) * 4;
    const int addr = krow * RB + ((((col >> 3)) ^ ((krow & 3) << 2)) << 4) + (col & 7) * 2;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lds + addr));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lds + addr + 4 * RB));
    const s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, v);
  }
}

// 16 bytes per lane straight into LDS at (wave-uniform) dst + lane*16.  Kept in a non-template __device__ function:
// hipcc 7.2 silently drops the host-side kernel handle when this builtin appears in a template-dependent expression.
__device__ __forceinline__ void glds16(const __amdgpu_buffer_rsrc_t rs, char* dst, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_ptr)dst, 16, voff, 0, 0, 0);
}

// Issue this wave's share of one K-tile: NIA + NIB direct-to-LDS 1 KiB pieces (A image first, B image behind it).
#define OASR_STAGE_TILE(KT, DST)                                                                                          \
  do {                                                                                                                    \
    const __amdgpu_buffer_rsrc_t ra_ = make_rsrc(baseA + (KT) * stepA);                                                   \
    const __amdgpu_buffer_rsrc_t rb_ = make_rsrc(baseB + (KT) * stepB);                                                   \
    char* dst_ = (DST);                                                                                                   \
    _Pragma("unroll") for (int j_ = 0; j_ < NIA; ++j_)                                                                    \
        glds16(ra_, dst_ + (j_ * NW + wave) * 1024, offA[j_]);                                               \
    _Pragma("unroll") for (int j_ = 0; j_ < NIB; ++j_)                                                                    \
        glds16(rb_, dst_ + A_BYTES + (j_ * NW + wave) * 1024, offB[j_]);                                     \
  } while (0)

// The common bf16-output epilogues (everything but fp32 / positional-embedding / odd-stride outputs), organised so that
// every runtime option is one wave-uniform branch per 32-row block and all global traffic is 16 bytes per lane:
//   pass A  acc * alpha (+ bias) -> bf16 (the autocast rounding point of the Linear) -> wave-private LDS tile
//           ([32 rows][128 B], 16-byte chunks XOR-swizzled by row);
//   pass B  each lane takes 8 consecutive columns of one row back out of LDS and applies what follows the rounding
//           point in GemmArgs order (GELU, x dGELU(u), + residual; u and the residual are read with the same
//           coalesced 16-byte accesses as the stores and are prefetched before pass A), accumulates the fused
//           bias-gradient column sums, and stores out_pre / out.
// Bit-identical to epilogue_math(): every later op starts from the bf16-rounded pre-activation.
__host__ __device__ __forceinline__ bool fast_rows_ok(const GemmArgs& p) {
  const unsigned long al = (unsigned long)p.out | (unsigned long)p.out_pre | (unsigned long)p.resid | (unsigned long)p.dgelu_u;
  return !p.out_f32 && (!p.pos || (!p.resid && !p.dgelu_u)) && (p.N % 8) == 0 && (p.ldc % 8) == 0 && (al & 15) == 0 &&
         (!p.resid || (p.ldr % 8) == 0) && (!p.dgelu_u || (p.ldu % 8) == 0) && !(p.resid && p.dgelu_u);
}

// 16-byte global accesses of the epilogue, optionally non-temporal (streaming: the output / side-input bytes are touched once
// and should not displace the operand panels the main loops re-read from L2)
__device__ __forceinline__ void store16(bf16_t* ptr, const u32x4_t& v, bool nt) {
  if (nt) __builtin_nontemporal_store(v, (u32x4_t*)ptr);
  else *(u32x4_t*)ptr = v;
}
__device__ __forceinline__ u32x4_t load16(const bf16_t* ptr, bool nt) {
  return nt ? __builtin_nontemporal_load((const u32x4_t*)ptr) : *(const u32x4_t*)ptr;
}
// The runtime options of the epilogue, as a bit set.  The body below is written once over these flags; for the combinations the
// training step actually launches (EPI_MODES) it is instantiated with the flags as compile-time constants, so that the ~12 wave-uniform
// branches per 8-column chunk, the per-row bounds predicates and the dead arithmetic disappear from the instruction stream -- the
// epilogue is instruction-bound (profiles/r03_gemm_tile_stamps.txt).  Anything else runs the generic instantiation (MODE < 0).
enum : unsigned {
  EPI_BIAS = 1, EPI_RESID = 2, EPI_U = 4, EPI_UDERIV = 8, EPI_PRE = 16, EPI_OUT = 32, EPI_GELU = 64, EPI_DERIV = 128, EPI_POS = 256,
  EPI_SCALE = 512, EPI_NT_ST = 1024, EPI_NT_LD = 2048, EPI_PARTIAL = 4096
};
__device__ __forceinline__ unsigned epi_flags_of(const GemmArgs& p, bool partial) {
  return (p.bias ? EPI_BIAS : 0) | (p.resid ? EPI_RESID : 0) | (p.dgelu_u ? EPI_U : 0) | (p.dgelu_deriv ? EPI_UDERIV : 0) |
         (p.out_pre ? EPI_PRE : 0) | (p.out ? EPI_OUT : 0) | (p.act != 0 ? EPI_GELU : 0) | (p.act == 2 ? EPI_DERIV : 0) |
         (p.pos ? EPI_POS : 0) | (p.alpha != 1.0f ? EPI_SCALE : 0) | ((p.epi_flags & 1) ? EPI_NT_ST : 0) |
         ((p.epi_flags & 2) ? EPI_NT_LD : 0) | (partial ? EPI_PARTIAL : 0);
}
constexpr unsigned EPI_MODES[] = {
    EPI_BIAS | EPI_OUT,                                   // Linear with bias (q|k|v, cross q / k|v)
    EPI_BIAS | EPI_OUT | EPI_RESID,                       // attention / MLP output projection + residual
    EPI_BIAS | EPI_OUT | EPI_PRE | EPI_GELU | EPI_DERIV,  // mlp.0 in training: GELU out, GELU' saved
    EPI_BIAS | EPI_OUT | EPI_GELU,                        // mlp.0 in inference
    EPI_OUT,                                              // dgrad
    EPI_OUT | EPI_RESID,                                  // dgrad accumulating into an existing gradient
    EPI_OUT | EPI_U | EPI_UDERIV,                         // dgrad through GELU (saved derivative)
};
constexpr int EPI_NMODES = (int)(sizeof(EPI_MODES) / sizeof(EPI_MODES[0]));

// stg: this wave's staging tile, 8-row groups of 1 KiB placed GS bytes apart; bias_lds: 64 floats of wave-private LDS.
template <bool CSUM, int GS, bool PF, int MODE>
__device__ __forceinline__ void fast_epilogue_rows_impl(const GemmArgs& p, f32x16_t (&acc)[4][2], char* stg, float* bias_lds,
                                                        int mrow0, int ncol0, int lane, unsigned rt_flags) {
  const unsigned F = MODE >= 0 ? EPI_MODES[MODE >= 0 ? MODE : 0] : rt_flags;  // compile-time constant when MODE >= 0
  const int h = lane >> 5, row = lane & 31;
  const bool has_bias = (F & EPI_BIAS) != 0, has_side = (F & (EPI_RESID | EPI_U)) != 0, has_u = (F & EPI_U) != 0;
  const bool has_pre = (F & EPI_PRE) != 0, has_out = (F & EPI_OUT) != 0, gelu = (F & EPI_GELU) != 0, has_pos = (F & EPI_POS) != 0;
  const bool save_deriv = (F & EPI_DERIV) != 0, u_is_deriv = (F & EPI_UDERIV) != 0;
  const bool nt_st = (F & EPI_NT_ST) != 0, nt_ld = (F & EPI_NT_LD) != 0;
  const bool partial = (F & EPI_PARTIAL) != 0;  // false: this wave's 128 x 64 block lies inside the matrix, no bounds predicates
  const bf16_t* side = has_u ? p.dgelu_u : p.resid;  // at most one of the two (fast_rows_ok)
  const long lds_ = has_u ? p.ldu : p.ldr;
  const int ch = lane & 7, nn = ncol0 + ch * 8;
  const bool n_ok = !partial || nn < p.N;
  if (has_bias) bias_lds[lane] = (!partial || ncol0 + lane < p.N) ? p.bias[ncol0 + lane] : 0.f;
  float cs[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) cs[e] = 0.f;
  u32x4_t sd[4];
  if (has_side && PF) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int mm = mrow0 + i * 8 + (lane >> 3);
      if (n_ok && (!partial || mm < p.M)) sd[i] = load16(side + (long)mm * lds_ + nn, nt_ld);
    }
  }
  __builtin_amdgcn_wave_barrier();
  // The epilogue is instruction-bound, not memory-bound (in-kernel cycle stamps, profiles/r03_gemm_tile_stamps.txt: 9.0 k cycles per
  // tile with bias only, 17 k with a side input -- with every global load and store REMOVED), so the per-value work is kept minimal:
  // the bias vectors of this wave's 64 columns are fetched once (not per 32-row block), alpha == 1 costs nothing, and values that
  // are already bf16 (unpacked from the staged pre-activation, no GELU / positional term applied) are not rounded again.
  const bool scale = (F & EPI_SCALE) != 0;
  const bool exact = !(gelu || has_pos);  // pass B's x[] are bf16 values exactly as unpacked
  // (hoisted only where 32 more registers are free: not beside side-input or GELU' temporaries, not in the generic instantiation)
  constexpr bool HOIST = MODE >= 0 && (EPI_MODES[MODE >= 0 ? MODE : 0] & (EPI_RESID | EPI_U | EPI_DERIV)) == 0;
  f32x4_t bv[2][4];
  if (HOIST) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q) bv[nt][q] = has_bias ? *(const f32x4_t*)(bias_lds + nt * 32 + 8 * q + 4 * h) : f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  // staging addresses: the swizzle XOR only touches bits 4-6, so every chunk address is (lane base) ^ (chunk << 4)
  unsigned wbase = (unsigned)(size_t)stg + (row >> 3) * GS + (row & 7) * 128 + h * 8 + ((row & 7) << 4);
  unsigned rbase = (unsigned)(size_t)stg + (lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) << 4);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    // (opaque re-definition per block: otherwise hipcc keeps all derived addresses of all four blocks live and spills)
    asm volatile("" : "+v"(wbase), "+v"(rbase));
    // pass A
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v[4];
        const f32x4_t b4 = HOIST ? bv[nt][q] : has_bias ? *(const f32x4_t*)(bias_lds + nt * 32 + 8 * q + 4 * h) : f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (scale) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = acc[mt][nt][q * 4 + i] * p.alpha + b4[i];
        } else if (has_bias) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = acc[mt][nt][q * 4 + i] + b4[i];
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = acc[mt][nt][q * 4 + i];
        }
        u32x2_t pk;
        pk[0] = pack_bf2(v[0], v[1]);
        pk[1] = pack_bf2(v[2], v[3]);
        *(lds_u32x2_ptr)(size_t)(wbase ^ ((nt * 4 + q) << 4)) = pk;
      }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_sched_barrier(0);  // keep the blocks apart: hipcc otherwise hoists work across them and spills
    // (PF == false, register-capped kernels: this block's own side inputs, right after its accumulators died)
    // side inputs of the next 32-row block (its accumulators' predecessors are dead now)
    u32x4_t sn[4];
    if (has_side && (mt < 3 || !PF)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int mm = mrow0 + (mt + (PF ? 1 : 0)) * 32 + i * 8 + (lane >> 3);
        if (n_ok && (!partial || mm < p.M)) sn[i] = load16(side + (long)mm * lds_ + nn, nt_ld);
      }
    }
    // pass B
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r2 = i * 8 + (lane >> 3);
      const int mm = mrow0 + mt * 32 + r2;
      const bool ok = n_ok && (!partial || mm < p.M);
      const u32x4_t pre = *(lds_u32x4_ptr)(size_t)(rbase + i * GS);  // rows r2 = i*8 + (lane >> 3): (r2 & 7) == lane >> 3
      if (has_pre && !save_deriv && ok) store16(p.out_pre + (long)mm * p.ldc + nn, pre, nt_st);
      u32x4_t fin = pre;
      if (gelu || has_side || has_pos) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          x[2 * e] = bf_lo(pre[e]);
          x[2 * e + 1] = bf_hi(pre[e]);
        }
        if (save_deriv) {  // GELU and GELU' share the erf polynomial and the exponential
          float dv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) gelu_and_deriv(x[e], x[e], dv[e]);
          if (has_pre && ok) {
            u32x4_t dpk;
#pragma unroll
            for (int e = 0; e < 4; ++e) dpk[e] = pack_bf2(dv[2 * e], dv[2 * e + 1]);
            store16(p.out_pre + (long)mm * p.ldc + nn, dpk, nt_st);
          }
        } else if (gelu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = gelu_f(x[e]);
        }
        if (has_pos && ok) {  // positional embedding row (m mod period), fp32 (conv2 + sinusoids)
          const float* pp = p.pos + (long)(mm % p.pos_period) * p.N + nn;
          const f32x4_t p0 = *(const f32x4_t*)pp, p1 = *(const f32x4_t*)(pp + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            x[e] = bf_round(x[e]) + p0[e];
            x[4 + e] = bf_round(x[4 + e]) + p1[e];
          }
        }
        if (has_side) {
          const u32x4_t sv = PF ? sd[i] : sn[i];
          if (has_u && u_is_deriv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              x[2 * e] = (exact ? x[2 * e] : bf_round(x[2 * e])) * bf_lo(sv[e]);
              x[2 * e + 1] = (exact ? x[2 * e + 1] : bf_round(x[2 * e + 1])) * bf_hi(sv[e]);
            }
          } else if (has_u) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              x[2 * e] = (exact ? x[2 * e] : bf_round(x[2 * e])) * dgelu_f(bf_lo(sv[e]));
              x[2 * e + 1] = (exact ? x[2 * e + 1] : bf_round(x[2 * e + 1])) * dgelu_f(bf_hi(sv[e]));
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              x[2 * e] = (exact ? x[2 * e] : bf_round(x[2 * e])) + bf_lo(sv[e]);
              x[2 * e + 1] = (exact ? x[2 * e + 1] : bf_round(x[2 * e + 1])) + bf_hi(sv[e]);
            }
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) fin[e] = pack_bf2(x[2 * e], x[2 * e + 1]);
      }
      if (CSUM && ok) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          cs[2 * e] += bf_lo(fin[e]);
          cs[2 * e + 1] += bf_hi(fin[e]);
        }
      }
      if (has_out && ok) store16(p.out + (long)mm * p.ldc + nn, fin, nt_st);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (PF) {
#pragma unroll
      for (int i = 0; i < 4; ++i) sd[i] = sn[i];
    }
  }
  if (CSUM) {  // lanes sharing (lane & 7) hold partial sums of the same 8 columns
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = cs[e];
      t += __shfl_xor(t, 8);
      t += __shfl_xor(t, 16);
      t += __shfl_xor(t, 32);
      cs[e] = t;
    }
    if (lane < 8 && n_ok) {
      if (p.colsum_scratch) {  // one partial row per 128-row wave block, written exactly once: no atomics
        float* dst = p.colsum_scratch + (long)(mrow0 >> 7) * p.N + nn;
        *(f32x4_t*)dst = f32x4_t{cs[0], cs[1], cs[2], cs[3]};
        *(f32x4_t*)(dst + 4) = f32x4_t{cs[4], cs[5], cs[6], cs[7]};
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) unsafeAtomicAdd(p.colsum + nn + e, cs[e]);
      }
    }
  }
}

template <bool CSUM, int GS, bool PF, int MODE>
__device__ __forceinline__ bool fast_epilogue_rows_try(const GemmArgs& p, f32x16_t (&acc)[4][2], char* stg, float* bias_lds, int mrow0,
                                                       int ncol0, int lane, unsigned flags) {
  if constexpr (MODE >= EPI_NMODES) {
    return false;
  } else {
    if (flags == EPI_MODES[MODE]) {
      fast_epilogue_rows_impl<CSUM, GS, PF, MODE>(p, acc, stg, bias_lds, mrow0, ncol0, lane, flags);
      return true;
    }
    return fast_epilogue_rows_try<CSUM, GS, PF, MODE + 1>(p, acc, stg, bias_lds, mrow0, ncol0, lane, flags);
  }
}
template <bool CSUM, int GS = 1024, bool PF = true>
__device__ __forceinline__ void fast_epilogue_rows(const GemmArgs& p, f32x16_t (&acc)[4][2], char* stg, float* bias_lds,
                                                   int mrow0, int ncol0, int lane) {
  const bool partial = mrow0 + 128 > p.M || ncol0 + 64 > p.N;  // wave-uniform
  const unsigned flags = epi_flags_of(p, partial);
  if (!fast_epilogue_rows_try<CSUM, GS, PF, 0>(p, acc, stg, bias_lds, mrow0, ncol0, lane, flags))
    fast_epilogue_rows_impl<CSUM, GS, PF, -1>(p, acc, stg, bias_lds, mrow0, ncol0, lane, flags);
}

// Epilogue shared by the direct-to-LDS kernels: each wave owns a 128 x 64 block of the output tile as acc[4][2]
// 32x32 MFMA blocks (rows m0 + wm*128 + mt*32, columns n0 + wn*64 + nt*32).  All waves of the workgroup must be past
// their last main-loop LDS read (the staging tiles alias the operand images).
// stg: this wave's 4 KiB staging tile, bias_lds: this wave's 64 floats.
template <bool SWAP, bool CSUM, bool PF = true>
__device__ __forceinline__ void fast_epilogue(const GemmArgs& p, f32x16_t (&acc)[4][2], char* stg, float* bias_lds, int m0, int n0,
                                              int wm, int wn, int lane) {
  if (SWAP) {  // bf16 outputs (the host routes anything fast_rows_ok() rejects to the general kernel)
    fast_epilogue_rows<CSUM, 1024, PF>(p, acc, stg, bias_lds, m0 + wm * 128, n0 + wn * 64, lane);
    return;
  }
  const int h = lane >> 5;
  {
    // D[m][n]: lane owns column n = .. + (lane & 31), rows (r & 3) + 8*(r >> 2) + 4*h.  fp32 atomic accumulate only.
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int n = n0 + wn * 64 + nt * 32 + (lane & 31);
      if (n >= p.N) continue;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * 128 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (m < p.M) unsafeAtomicAdd(p.out_f32 + (long)m * p.ldc32 + n, p.alpha * acc[mt][nt][r]);
        }
      }
    }
  }
}

}  // namespace

// (external linkage: hipcc 7.2 drops the host-side handle of this instantiation set when it has internal linkage)
template <bool TA, bool TB, int FBN, int NWN, int NSTAGE, bool SWAP, bool CSUM>
// (3 workgroups per CU only for the split-K / atomic variant: its epilogue needs no registers beyond the accumulators)
__global__ __launch_bounds__(128 * NWN, (NWN == 2 && !SWAP) ? 3 : 2) void oasr_gemm_fast_kernel(GemmArgs p) {
  constexpr int NW = 2 * NWN;                     // waves per workgroup
  constexpr int A_BYTES = FBM * 64 * 2, B_BYTES = FBN * 64 * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int NIA = (A_BYTES / 1024) / NW, NIB = (B_BYTES / 1024) / NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  const int tiles_m = (p.M + FBM - 1) / FBM, tiles_n = (p.N + FBN - 1) / FBN;
  // Rasterisation (speed only; any mapping is correct).  Block b runs on XCD b % 8, each XCD has a private 4 MiB L2.
  //  * split-K (SWAP == false, wgrad): the K-range index is tied to the XCD -- XCD x owns splits [x*S/8, (x+1)*S/8) for
  //    every output tile, walking all tiles of one split before the next.  The token slab a split streams is then
  //    fetched from HBM by exactly one L2 and shared by all concurrently running tiles (instead of once per XCD).
  //  * otherwise: XCD-contiguous chunks of the tile list, grouped GM rows of tiles x all column tiles with GM chosen so
  //    one group is resident on the XCD at once: the group's A panels stream through L2 once while every column tile
  //    consumes them, weights (small) are re-read from L2/MALL.
  int tm, tn, ksplit;
  {
    const int ntile = tiles_m * tiles_n;
    if (!SWAP && (p.split_k & 7) == 0 && gridDim.y == 1) {
      const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;  // j-th block dispatched to this XCD
      const int s8 = p.split_k >> 3;
      const int t = j % ntile;
      ksplit = xcd * s8 + j / ntile;
      tm = t / tiles_n;
      tn = t - tm * tiles_n;
    } else {
      const int bid = xcd_remap(blockIdx.x, gridDim.x);
      constexpr int RESIDENT = 32 * (NWN == 2 ? (SWAP ? 2 : 3) : 1);  // workgroups resident per XCD
      int gm = p.raster_gm > 0 ? p.raster_gm : RESIDENT / tiles_n;
      gm = gm < 1 ? 1 : (gm > 16 ? 16 : gm);
      const int per_group = gm * tiles_n;
      const int group = bid / per_group, in_group = bid - group * per_group;
      const int first_m = group * gm;
      const int gsz = min(gm, tiles_m - first_m);
      tm = first_m + in_group % gsz;
      tn = in_group / gsz;
      ksplit = blockIdx.y;
    }
  }
  const int m0 = tm * FBM, n0 = tn * FBN;

  const int kt_total = p.K / BK;
  const int per = (kt_total + p.split_k - 1) / p.split_k;
  const int kt0 = ksplit * per;
  const int kt1 = min(kt_total, kt0 + per);
  if (kt0 >= kt1) return;

  unsigned offA[NIA], offB[NIB];
  fast_offsets<TA, FBM, NIA, NW>(p.A, p.M, m0, lane, wave, offA);
  fast_offsets<TB, FBN, NIB, NW>(p.B, p.N, n0, lane, wave, offB);
  const bf16_t* baseA = TA ? p.A.ptr + m0 : p.A.ptr + (long)m0 * p.A.ld;
  const bf16_t* baseB = TB ? p.B.ptr + n0 : p.B.ptr + (long)n0 * p.B.ld;
  const long stepA = TA ? (long)BK * p.A.ld : BK, stepB = TB ? (long)BK * p.B.ld : BK;

  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (NSTAGE == 2) {
    OASR_STAGE_TILE(kt0, smem);
    __syncthreads();
  }
  for (int kt = kt0; kt < kt1; ++kt) {
    const char* cur;
    if (NSTAGE == 2) {
      cur = smem + ((kt - kt0) & 1) * STAGE;
      if (kt + 1 < kt1)
        OASR_STAGE_TILE(kt + 1, smem + (((kt - kt0) & 1) ^ 1) * STAGE);
    } else {
      cur = smem;
      OASR_STAGE_TILE(kt, smem);
      __syncthreads();  // hipcc drains vmcnt(0) for the LDS-DMA before the barrier
    }
    const char* sA = cur;
    const char* sB = cur + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8_t af[4], bfr[2];
#pragma unroll
      for (int t = 0; t < 4; ++t) af[t] = fast_frag<TA, FBM * 2>(sA, wm * 128 + t * 32, ks, lane);
#pragma unroll
      for (int t = 0; t < 2; ++t) bfr[t] = fast_frag<TB, FBN * 2>(sB, wn * 64 + t * 32, ks, lane);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          if (SWAP)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[nt], af[mt], acc[mt][nt], 0, 0, 0);
          else
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt], bfr[nt], acc[mt][nt], 0, 0, 0);
        }
    }
    __syncthreads();  // stage fully consumed (and, with two stages, the prefetched tile has landed)
  }

  fast_epilogue<SWAP, CSUM, true>(p, acc, smem + wave * 8192, (float*)(smem + wave * 8192 + 4096), m0, n0, wm, wn, lane);
}


// ---- 256 x 256 x 64 "ping-pong" kernel ---------------------------------------------------------------------------
// 8 waves (2 along M x 4 along N, 128 x 64 outputs each), one workgroup per CU, 128 KiB of LDS = two K-tile buffers of
// four 16 KiB half-tile images (A rows 0-127 / 128-255, B columns 0-127 / 128-255; each image is laid out exactly like
// a 128-row operand tile of the kernel above, so the fragment readers are shared).
//
// A K-tile is consumed in four phases, one 64 x 32 output quadrant (8 MFMAs) each:
//     phase  LDS -> VGPR          MFMA        direct-to-LDS staging issued (2 x 1 KiB pieces per wave)
//       0    B0 (4), A0 (8)       A0 x B0     A rows   0-127 of tile t+1 -> other buffer
//       1    B1 (4)               A0 x B1     A rows 128-255 of tile t+1 -> other buffer
//       2    A1 (8)               A1 x B1     B cols   0-127 of tile t+2 -> THIS buffer (its B images are dead)
//       3    -                    A1 x B0     B cols 128-255 of tile t+2 -> THIS buffer;  s_waitcnt vmcnt(4)
// Every phase is  { ds_reads; staging; s_barrier; MFMAs; s_barrier }.  The waves of the upper M half run one barrier
// behind the lower half, so on every SIMD one wave is in its MFMA section while the other issues LDS reads and DMA:
// the matrix pipe never waits for a barrier-drained staging step (the 1-workgroup-per-CU failure mode of the
// kernel above).  The DMA queue is never drained inside the loop: the single counted wait per K-tile (phase 3) leaves
// the two newest half-tiles (tile t+2's B) in flight across the barriers.
// Hazards (DMA writes are ordered with LDS reads only by the issuing wave's vmcnt wait followed by a barrier):
//   RAW  tile t+1 is complete after phase 3's wait of tile t + the barriers up to the first read in phase 0 of t+1
//        (two barrier events later even for the trailing wave group);
//   WAR  an image is re-staged >= 2 phases after its last ds_read, or in the next phase when the reading phase
//        retired its reads (lgkmcnt(0)) before its first barrier (B1 in phase 1 -> B restaged in phase 2).
#define OASR_PP_BARRIER() asm volatile("s_barrier" ::: "memory")

template <bool TA, bool TB, bool SWAP, bool CSUM, int VAR>
__global__ __launch_bounds__(512, 2) void oasr_gemm_pp_kernel(GemmArgs p) {
  // VAR bit 0: the direct-to-LDS pieces are issued between the MFMAs (else in the fragment-read section);
  //     bit 1: every MFMA section is pinned between its two barriers.  Without the pin hipcc sinks most of a section's MFMAs
  //            below the closing s_barrier (the asm-volatile barrier orders memory, not register-only instructions, and the
  //            conditional DMA issue splits the basic block the sched_barrier fences act in): the ISA of round 1's kernel has
  //            1 + 7 MFMAs around the first barrier pair instead of 8 inside it.
  constexpr bool DMA_IN_MMA = (VAR & 1) != 0, PIN = (VAR & 2) != 0;
  constexpr int HALF = 128 * 64 * 2, BUF = 4 * HALF;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int tiles_m = (p.M + 255) / 256, tiles_n = (p.N + 255) / 256;
  const int kt_total = p.K / BK;
  const int per = (kt_total + p.split_k - 1) / p.split_k;
  const long stepA = TA ? (long)BK * p.A.ld : BK, stepB = TB ? (long)BK * p.B.ld : BK;
  // PERSISTENT launches (p.vgrid > gridDim.x, one workgroup per CU): workgroup w walks the virtual blocks w, w + gridDim.x, ...
  // of the grid a plain launch would have used -- gridDim.x is a multiple of 8, so the XCD a virtual block lands on, and with it
  // the rasterisation below, is exactly that of the plain launch.
  const int vgrid = p.vgrid > 0 ? p.vgrid : (int)gridDim.x;
  int m0, n0, nt;
  unsigned off[4][2];  // staging offsets: image i (0,1 = A halves; 2,3 = B halves), 2 pieces per wave
  const bf16_t *baseA, *baseB;
#define OASR_PP_TILE(V)                                                                                      \
  do {                                                                                                       \
    int tm_, tn_, ksplit_;                                                                                   \
    const int ntile_ = tiles_m * tiles_n;                                                                    \
    if (!SWAP && (p.split_k & 7) == 0 && gridDim.y == 1) { /* split-K range tied to the XCD (see the kernel above) */ \
      const int xcd_ = (V) & 7, j_ = (V) >> 3;                                                               \
      const int s8_ = p.split_k >> 3;                                                                        \
      const int t_ = j_ % ntile_;                                                                            \
      ksplit_ = xcd_ * s8_ + j_ / ntile_;                                                                    \
      tm_ = t_ / tiles_n;                                                                                    \
      tn_ = t_ - tm_ * tiles_n;                                                                              \
    } else {                                                                                                 \
      const int bid_ = xcd_remap((V), vgrid);                                                                \
      /* groups of 8 tile rows, walked rows-first: the 32 tiles an XCD runs at once are 8 row panels x 4 column panels */ \
      /* (12 operand panels through its L2 instead of 2 x 16 = 18: -6 % on the N = 4096 layers, scripts/gemm_ab.py) */ \
      int gm_ = p.raster_gm > 0 ? p.raster_gm : 8;                                                           \
      gm_ = gm_ < 1 ? 1 : (gm_ > 16 ? 16 : gm_);                                                             \
      const int per_group_ = gm_ * tiles_n;                                                                  \
      const int group_ = bid_ / per_group_, in_group_ = bid_ - group_ * per_group_;                          \
      const int first_m_ = group_ * gm_;                                                                     \
      const int gsz_ = min(gm_, tiles_m - first_m_);                                                         \
      tm_ = first_m_ + in_group_ % gsz_;                                                                     \
      tn_ = in_group_ / gsz_;                                                                                \
      ksplit_ = blockIdx.y;                                                                                  \
    }                                                                                                        \
    m0 = tm_ * 256;                                                                                          \
    n0 = tn_ * 256;                                                                                          \
    const int kt0_ = ksplit_ * per;                                                                          \
    nt = min(kt_total, kt0_ + per) - kt0_; /* K-tiles of this output tile */                                 \
    fast_offsets<TA, 128, 2, 8>(p.A, p.M, m0, lane, wave, off[0], m0);                                       \
    fast_offsets<TA, 128, 2, 8>(p.A, p.M, m0 + 128, lane, wave, off[1], m0);                                 \
    fast_offsets<TB, 128, 2, 8>(p.B, p.N, n0, lane, wave, off[2], n0);                                       \
    fast_offsets<TB, 128, 2, 8>(p.B, p.N, n0 + 128, lane, wave, off[3], n0);                                 \
    baseA = (TA ? p.A.ptr + m0 : p.A.ptr + (long)m0 * p.A.ld) + (long)kt0_ * stepA;                         \
    baseB = (TB ? p.B.ptr + n0 : p.B.ptr + (long)n0 * p.B.ld) + (long)kt0_ * stepB;                         \
  } while (0)
  int v = blockIdx.x;
  OASR_PP_TILE(v);
  if (nt <= 0) return;  // (uneven split-K; the host never launches those persistent)
  // Phase stagger of the first wave of workgroups (one per CU): with equal-length tiles every CU reaches its epilogue at
  // the same moment and the 256 x 128 KiB of output must drain to HBM in one burst while the matrix cores idle; delaying
  // the CUs of each XCD by k/phases of a tile period spreads the stores under the other CUs' main loops.
  if (p.stagger > 0 && p.stagger_phases > 1 && blockIdx.x < 256 && blockIdx.y == 0) {
    const int ph = (blockIdx.x >> 3) % p.stagger_phases;
    for (int i = 0; i < ph * p.stagger; ++i) __builtin_amdgcn_s_sleep(127);
  }

#define OASR_PP_STAGE(IMG, T, BUFP)                                                                          \
  do {                                                                                                       \
    const __amdgpu_buffer_rsrc_t rs_ = make_rsrc((IMG) < 2 ? baseA + (T) * stepA : baseB + (T) * stepB);     \
    char* d_ = (BUFP) + (IMG) * HALF + wave * 1024;                                                          \
    glds16(rs_, d_, off[IMG][0]);                                                                            \
    glds16(rs_, d_ + 8192, off[IMG][1]);                                                                     \
  } while (0)

  // prologue of an output tile: all of K-tile 0 into buffer P0, the B images of K-tile 1 into the other buffer
#define OASR_PP_PROLOGUE(P0, P1)      \
  do {                                \
    OASR_PP_STAGE(0, 0, P0);          \
    OASR_PP_STAGE(1, 0, P0);          \
    OASR_PP_STAGE(2, 0, P0);          \
    OASR_PP_STAGE(3, 0, P0);          \
    if (nt > 1) {                     \
      OASR_PP_STAGE(2, 1, P1);        \
      OASR_PP_STAGE(3, 1, P1);        \
    }                                 \
  } while (0)
  f32x16_t acc[4][2];
  int parity = 0;     // buffer that holds K-tile 0 of the current output tile
  bool first = true;  // first output tile of this workgroup (its prologue was not issued from an epilogue)
  OASR_PP_PROLOGUE(smem, smem + BUF);
for (;;) {  // output tiles of this workgroup (one iteration unless the launch is persistent)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  if (first && nt > 1) {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {  // later tiles: the prologue pieces are older than the previous epilogue's stores, so the count is drained
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  OASR_PP_BARRIER();
  if (wm == 1 && !(VAR & 8)) OASR_PP_BARRIER();  // the upper half trails by one barrier from here on (VAR bit 3: lockstep experiment)

  const int aoff = wm * HALF, boff = (2 + (wn >> 1)) * HALF, bsub = (wn & 1) * 64;
  bf16x8_t fa[2][4], fb0[4], fb1[4];
#define OASR_PP_MFMA1(MT0, NT, FB, KS, T)                                                                     \
  do {                                                                                                       \
    if (SWAP)                                                                                                \
      acc[MT0 + T][NT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FB[KS], fa[T][KS], acc[MT0 + T][NT], 0, 0, 0); \
    else                                                                                                     \
      acc[MT0 + T][NT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[T][KS], FB[KS], acc[MT0 + T][NT], 0, 0, 0); \
  } while (0)
  // 8 MFMAs at priority 1.  DMA: the phase's two direct-to-LDS pieces are issued from INSIDE the section (after the 2nd and
  // the 4th MFMA) when OASR_PP_DMA_IN_MMA: a piece costs ~60 cycles of issue among bare MFMAs, which the 256 cycles of
  // matrix-pipe time cover, against 100-185 cycles in a read section already carrying 8-12 ds_read_b128
  // (MI355X_MICROARCH.md, "LDS-DMA piece issue cost") -- there it made the read section longer than the partner's MFMAs.
#define OASR_PP_MMA(MT0, NT, FB, DO_STAGE, IMG, T, BUFP)                                                     \
  do {                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    __builtin_amdgcn_s_setprio(1);                                                                           \
    if (PIN) asm volatile("" : "+v"(acc[MT0][NT]), "+v"(acc[MT0 + 1][NT]));                                  \
    OASR_PP_MFMA1(MT0, NT, FB, 0, 0);                                                                        \
    OASR_PP_MFMA1(MT0, NT, FB, 0, 1);                                                                        \
    if (DMA_IN_MMA && (DO_STAGE)) {                                                                          \
      __builtin_amdgcn_sched_barrier(0);                                                                     \
      const __amdgpu_buffer_rsrc_t rs_ = make_rsrc((IMG) < 2 ? baseA + (T) * stepA : baseB + (T) * stepB);   \
      glds16(rs_, (BUFP) + (IMG) * HALF + wave * 1024, off[IMG][0]);                                         \
      __builtin_amdgcn_sched_barrier(0);                                                                     \
    }                                                                                                        \
    OASR_PP_MFMA1(MT0, NT, FB, 1, 0);                                                                        \
    OASR_PP_MFMA1(MT0, NT, FB, 1, 1);                                                                        \
    if (DMA_IN_MMA && (DO_STAGE)) {                                                                          \
      __builtin_amdgcn_sched_barrier(0);                                                                     \
      const __amdgpu_buffer_rsrc_t rs_ = make_rsrc((IMG) < 2 ? baseA + (T) * stepA : baseB + (T) * stepB);   \
      glds16(rs_, (BUFP) + (IMG) * HALF + wave * 1024 + 8192, off[IMG][1]);                                  \
      __builtin_amdgcn_sched_barrier(0);                                                                     \
    }                                                                                                        \
    OASR_PP_MFMA1(MT0, NT, FB, 2, 0);                                                                        \
    OASR_PP_MFMA1(MT0, NT, FB, 2, 1);                                                                        \
    OASR_PP_MFMA1(MT0, NT, FB, 3, 0);                                                                        \
    OASR_PP_MFMA1(MT0, NT, FB, 3, 1);                                                                        \
    if (PIN) asm volatile("" : "+v"(acc[MT0][NT]), "+v"(acc[MT0 + 1][NT]));                                  \
    __builtin_amdgcn_s_setprio(0);                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
  } while (0)

  if (VAR & 4) {
    // Two sections of 16 MFMAs per K-tile (4 barriers instead of 8): S0 = rows 0-63 of this wave's 128 x 64 block against
    // both column halves, S1 = rows 64-127.  Staging (hazards as in the header, with "phase" = section):
    //   S0 read : A image 0 of tile t+1 -> oth (its last reader, the leading group, is past its own S1 MFMAs);
    //             lgkmcnt(0) BEFORE the barrier, so the B images of `cur` are dead once both groups passed it
    //   S0 MFMA : A image 1 of tile t+1 -> oth, between the MFMAs (the trailing group finished reading it one barrier ago)
    //   S1 read : B images of tile t+2 -> cur; vmcnt(4): everything of tile t+1 has landed, only those 4 pieces are in flight
    for (int t = 0; t < nt; ++t) {
      char* cur = smem + ((t & 1) ^ parity) * BUF;
      char* oth = smem + ((t & 1) ^ parity ^ 1) * BUF;
      const bool next1 = t + 1 < nt, next2 = t + 2 < nt;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        fb0[ks] = fast_frag<TB, 256>(cur + boff, bsub, ks, lane);
        fb1[ks] = fast_frag<TB, 256>(cur + boff, bsub + 32, ks, lane);
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i][ks] = fast_frag<TA, 256>(cur + aoff, i * 32, ks, lane);
      if (next1) OASR_PP_STAGE(0, t + 1, oth);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      OASR_PP_BARRIER();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      asm volatile("" : "+v"(acc[0][0]), "+v"(acc[1][0]), "+v"(acc[0][1]), "+v"(acc[1][1]));
      OASR_PP_MFMA1(0, 0, fb0, 0, 0);
      OASR_PP_MFMA1(0, 0, fb0, 0, 1);
      OASR_PP_MFMA1(0, 1, fb1, 0, 0);
      OASR_PP_MFMA1(0, 1, fb1, 0, 1);
      if (next1) {
        __builtin_amdgcn_sched_barrier(0);
        const __amdgpu_buffer_rsrc_t rs_ = make_rsrc(baseA + (t + 1) * stepA);
        glds16(rs_, oth + 1 * HALF + wave * 1024, off[1][0]);
        __builtin_amdgcn_sched_barrier(0);
      }
      OASR_PP_MFMA1(0, 0, fb0, 1, 0);
      OASR_PP_MFMA1(0, 0, fb0, 1, 1);
      OASR_PP_MFMA1(0, 1, fb1, 1, 0);
      OASR_PP_MFMA1(0, 1, fb1, 1, 1);
      if (next1) {
        __builtin_amdgcn_sched_barrier(0);
        const __amdgpu_buffer_rsrc_t rs_ = make_rsrc(baseA + (t + 1) * stepA);
        glds16(rs_, oth + 1 * HALF + wave * 1024 + 8192, off[1][1]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int ks = 2; ks < 4; ++ks) {
        OASR_PP_MFMA1(0, 0, fb0, ks, 0);
        OASR_PP_MFMA1(0, 0, fb0, ks, 1);
        OASR_PP_MFMA1(0, 1, fb1, ks, 0);
        OASR_PP_MFMA1(0, 1, fb1, ks, 1);
      }
      asm volatile("" : "+v"(acc[0][0]), "+v"(acc[1][0]), "+v"(acc[0][1]), "+v"(acc[1][1]));
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      OASR_PP_BARRIER();
      // ---- S1
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i][ks] = fast_frag<TA, 256>(cur + aoff, 64 + i * 32, ks, lane);
      if (next2) {
        OASR_PP_STAGE(2, t + 2, cur);
        if (DMA_IN_MMA) {  // B image 1 of tile t+2 goes out between this section's MFMAs: only B image 0's 2 pieces are newer
          asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
          OASR_PP_STAGE(3, t + 2, cur);
          asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      OASR_PP_BARRIER();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      asm volatile("" : "+v"(acc[2][0]), "+v"(acc[3][0]), "+v"(acc[2][1]), "+v"(acc[3][1]));
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        OASR_PP_MFMA1(2, 0, fb0, ks, 0);
        OASR_PP_MFMA1(2, 0, fb0, ks, 1);
        OASR_PP_MFMA1(2, 1, fb1, ks, 0);
        OASR_PP_MFMA1(2, 1, fb1, ks, 1);
        if (DMA_IN_MMA && next2 && ks < 2) {
          __builtin_amdgcn_sched_barrier(0);
          const __amdgpu_buffer_rsrc_t rs_ = make_rsrc(baseB + (t + 2) * stepB);
          glds16(rs_, cur + 3 * HALF + wave * 1024 + ks * 8192, off[3][ks]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      asm volatile("" : "+v"(acc[2][0]), "+v"(acc[3][0]), "+v"(acc[2][1]), "+v"(acc[3][1]));
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      OASR_PP_BARRIER();
    }
  } else
  for (int t = 0; t < nt; ++t) {
    char* cur = smem + ((t & 1) ^ parity) * BUF;
    char* oth = smem + ((t & 1) ^ parity ^ 1) * BUF;
    const bool next1 = t + 1 < nt, next2 = t + 2 < nt;
    // ---- phase 0
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fb0[ks] = fast_frag<TB, 256>(cur + boff, bsub, ks, lane);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i][ks] = fast_frag<TA, 256>(cur + aoff, i * 32, ks, lane);
    if (!DMA_IN_MMA && next1) OASR_PP_STAGE(0, t + 1, oth);
    OASR_PP_BARRIER();
    OASR_PP_MMA(0, 0, fb0, next1, 0, t + 1, oth);
    OASR_PP_BARRIER();
    // ---- phase 1
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fb1[ks] = fast_frag<TB, 256>(cur + boff, bsub + 32, ks, lane);
    if (!DMA_IN_MMA && next1) OASR_PP_STAGE(1, t + 1, oth);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // B images of `cur` are dead before this phase's barrier
    OASR_PP_BARRIER();
    OASR_PP_MMA(0, 1, fb1, next1, 1, t + 1, oth);
    OASR_PP_BARRIER();
    // ---- phase 2
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i][ks] = fast_frag<TA, 256>(cur + aoff, 64 + i * 32, ks, lane);
    if (!DMA_IN_MMA && next2) OASR_PP_STAGE(2, t + 2, cur);
    OASR_PP_BARRIER();
    OASR_PP_MMA(2, 1, fb1, next2, 2, t + 2, cur);
    OASR_PP_BARRIER();
    // ---- phase 3: everything of tile t+1 (this wave's pieces) must have landed before the barrier that ends the tile
    if (DMA_IN_MMA) {
      // outstanding, oldest first: [B0 B1](t+1) [A0 A1](t+1) B0(t+2) -> all but the last two pieces
      if (next2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (next2) {
      OASR_PP_STAGE(3, t + 2, cur);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // everything but tile t+2's B has landed (this wave's pieces)
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    OASR_PP_BARRIER();
    OASR_PP_MMA(2, 0, fb0, next2, 3, t + 2, cur);
    OASR_PP_BARRIER();
  }
  if (wm == 0 && !(VAR & 8)) OASR_PP_BARRIER();  // re-join: the trailing half has finished its LDS reads after this
  // Persistent launch: the NEXT output tile's prologue is issued before this tile's epilogue, into the buffer the epilogue does
  // not stage through (K-tile 0 -> buffer parity ^ 1, K-tile 1's B images -> the upper half of buffer `parity`; the epilogue's
  // wave-private staging tiles live in the lower half of buffer `parity`, the bias rows behind both buffers).  The operand fetch
  // latency, the workgroup relaunch and the drain of this tile's stores then overlap instead of adding up per tile.
  const int em0 = m0, en0 = n0;
  char* const stg = smem + parity * BUF + wave * 4096;
  const int vn = v + (int)gridDim.x;
  const bool has_next = vn < vgrid;
  if (has_next) {
    OASR_PP_TILE(vn);
    parity ^= 1;
    OASR_PP_PROLOGUE(smem + parity * BUF, smem + (parity ^ 1) * BUF);
  }
  fast_epilogue<SWAP, CSUM>(p, acc, stg, (float*)(smem + 2 * BUF + wave * 256), em0, en0, wm, wn, lane);
  if (!has_next) break;
  v = vn;
  asm volatile("" : "+s"(v));  // re-derive the tile's offsets here instead of keeping 8 VGPRs live across the epilogue
  OASR_PP_TILE(v);
  first = false;
}
#undef OASR_PP_STAGE
#undef OASR_PP_PROLOGUE
#undef OASR_PP_TILE
#undef OASR_PP_MMA
#undef OASR_PP_MFMA1
}


namespace {

// ---- skinny GEMM for the decode path: out[M <= 64][N] = x[M][K] . W[N][K]^T, NT layout --------------------------------
// A decoder step multiplies a handful of token rows with every weight matrix; the 256-wide tiles above would run 3-24
// workgroups that each walk K serially (measured 25-32 us per GEMM, 40x the time the weight bytes need).  Here one
// workgroup owns 32 output columns, its 4 waves split K four ways and stream the weight rows straight from L2/HBM
// into MFMA operands (16 bytes per lane, no LDS staging: every weight byte is used exactly once), x comes through
// L1/L2.  Partial accumulators meet in LDS; the epilogue is the general kernel's.
template <int MT>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmArgs p) {
  __shared__ float red[4][MT][16][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  const int n0 = blockIdx.x * 32;
  int col = n0 + (lane & 31);
  col = col < p.N ? col : p.N - 1;
  const bf16_t* wp = p.B.ptr + (long)col * p.B.ld + h * 8;
  const bf16_t* xp[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int row = mt * 32 + (lane & 31);
    row = row < p.M ? row : p.M - 1;
    xp[mt] = p.A.ptr + (long)row * p.A.ld + h * 8;
  }
  f32x16_t acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
  const int kq = p.K / 4;  // K % 64 == 0 (host): every wave gets a multiple of 16
  const int k_begin = wave * kq, k_end = k_begin + kq;
  int k = k_begin;
  for (; k + 64 <= k_end; k += 64) {  // 4 k-steps per trip: 4 (+ 4 MT) independent 16-byte loads in flight per lane
    u32x4_t wq[4], xq[MT][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      wq[j] = *(const u32x4_t*)(wp + k + 16 * j);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) xq[mt][j] = *(const u32x4_t*)(xp[mt] + k + 16 * j);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)  // D'[n][m]: lane owns output row m
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wq[j]), __builtin_bit_cast(bf16x8_t, xq[mt][j]),
                                                          acc[mt], 0, 0, 0);
  }
  for (; k < k_end; k += 16) {
    const bf16x8_t wf = __builtin_bit_cast(bf16x8_t, *(const u32x4_t*)(wp + k));
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const bf16x8_t xf = __builtin_bit_cast(bf16x8_t, *(const u32x4_t*)(xp[mt] + k));
      acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, acc[mt], 0, 0, 0);
    }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][mt][r][lane] = acc[mt][r];
  __syncthreads();
  // wave w finishes register group q = w: columns n0 + 8w + 4h .. +3 of output row (lane & 31)
  const int n = n0 + 8 * wave + 4 * h;
  if (n < p.N) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = mt * 32 + (lane & 31);
      if (m < p.M) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          v[i] = red[0][mt][wave * 4 + i][lane] + red[1][mt][wave * 4 + i][lane] + red[2][mt][wave * 4 + i][lane] +
                 red[3][mt][wave * 4 + i][lane];
        epilogue_store(p, m, n, p.pos ? (m % p.pos_period) : 0, v);
      }
    }
  }
}


template <bool TA, bool TB, int FBN, int NWN, int NSTAGE, bool SWAP, bool CSUM = false>
int launch_fast_cfg(const GemmArgs& a, hipStream_t stream) {
  static LdsAttrOnce attr;
  const int lds = NSTAGE * (FBM * 64 * 2 + FBN * 64 * 2);
  { const int rc_ = ensure_dynamic_lds(attr, (const void*)oasr_gemm_fast_kernel<TA, TB, FBN, NWN, NSTAGE, SWAP, CSUM>, lds); if (rc_) return rc_; }
  const int tiles = cdiv(a.M, FBM) * cdiv(a.N, FBN);
  dim3 grid(tiles, a.split_k);
  if (!SWAP && (a.split_k & 7) == 0) grid = dim3(tiles * a.split_k, 1);  // split index tied to the XCD (see kernel)
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
    auto tf = [](bool b) { return b ? "true" : "false"; };
    static const std::string name = std::string("oasr_gemm_fast_kernel<") + tf(TA) + ", " + tf(TB) + ", " + std::to_string(FBN) + ", " +
                                    std::to_string(NWN) + ", " + std::to_string(NSTAGE) + ", " + tf(SWAP) + ", " + tf(CSUM) + ">";
    g_prof.push({(TA ? 2 : 0) + (TB ? 1 : 0), 2.0 * (double)a.M * (double)a.N * (double)a.K, name.c_str(), a.M, a.N, a.K,
                           (a.bias ? 1 : 0) | (a.resid ? 2 : 0) | (a.act ? 4 : 0) | (a.dgelu_u ? 8 : 0) | (a.colsum ? 16 : 0) | (a.atomic ? 32 : 0)});
    OASR_CHECK_HIP(hipEventRecord(e0, stream));
  }
  hipLaunchKernelGGL((oasr_gemm_fast_kernel<TA, TB, FBN, NWN, NSTAGE, SWAP, CSUM>), grid, dim3(128 * NWN), lds, stream, a);
  OASR_LAUNCH_CHECK();
  if (e1) OASR_CHECK_HIP(hipEventRecord(e1, stream));
  return OASR_OK;
}

template <bool TA, bool TB, bool SWAP, bool CSUM, int DMA>
int launch_pp_variant(const GemmArgs& a, hipStream_t stream) {
  static LdsAttrOnce attr;
  const int lds = 2 * 4 * 128 * 64 * 2 + 8 * 256;  // two K-tile buffers + the epilogue's per-wave bias rows
  { const int rc_ = ensure_dynamic_lds(attr, (const void*)oasr_gemm_pp_kernel<TA, TB, SWAP, CSUM, DMA>, lds); if (rc_) return rc_; }
  const int tiles = cdiv(a.M, 256) * cdiv(a.N, 256);
  dim3 grid(tiles, a.split_k);
  if (!SWAP && (a.split_k & 7) == 0) grid = dim3(tiles * a.split_k, 1);
  // Persistent launch (one workgroup per CU walking the same virtual grid, next tile's prologue issued ahead of the epilogue):
  // opt-in (OASR_PP_PERSISTENT=1 or oasr_gemm_set_variant() bits 4-5).  Measured level with plain launches on every layer shape
  // (profiles/r02_gemm_tile_cost_model.txt): the chip is at its power budget in these kernels (effective clock 1.53 GHz of 2.4,
  // profiles/r02_gemm_effective_clock.txt), so cycles saved between tiles come back as a lower clock, and plain launches keep the
  // hardware's dynamic tile dispatch (robust when RCCL kernels hold CUs).
  static const int n_cu = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
    n &= ~7;  // (the virtual-block -> XCD correspondence needs a multiple of 8)
    return n >= 8 ? n : 8;
  }();
  static const int env_persist = [] {
    const char* e = oasr_experiment_env("OASR_PP_PERSISTENT");
    return e ? atoi(e) : -1;
  }();
  GemmArgs pa = a;
  pa.vgrid = (int)grid.x;
  // (experiment: OASR_PP_PERSIST_GELU=1 launches only the GELU-epilogue shapes persistent -- mlp.0 forward and the dgrad through GELU',
  // whose 12.7 k-cycle epilogues are the largest fixed cost per tile, profiles/r03_gemm_tile_stamps.txt)
  static const int env_persist_gelu = [] {
    const char* e = oasr_experiment_env("OASR_PP_PERSIST_GELU");
    return e ? atoi(e) : 0;
  }();
  int want = g_pp_persistent >= 0 ? g_pp_persistent : env_persist;
  if (want < 0 && env_persist_gelu && (a.act != 0 || a.dgelu_u)) want = 1;
  const bool can_persist = grid.y == 1 && (int)grid.x > n_cu && ((a.K / BK) % a.split_k) == 0;
  if (can_persist && want == 1) grid = dim3(n_cu, 1);
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
    auto tf = [](bool b) { return b ? "true" : "false"; };
    static const std::string name = std::string("oasr_gemm_pp_kernel<") + tf(TA) + ", " + tf(TB) + ", " + tf(SWAP) + ", " + tf(CSUM) +
                                    ", " + std::to_string(DMA) + ">";
    g_prof.push({(TA ? 2 : 0) + (TB ? 1 : 0), 2.0 * (double)a.M * (double)a.N * (double)a.K, name.c_str(), a.M, a.N, a.K,
                           (a.bias ? 1 : 0) | (a.resid ? 2 : 0) | (a.act ? 4 : 0) | (a.dgelu_u ? 8 : 0) | (a.colsum ? 16 : 0) | (a.atomic ? 32 : 0)});
    OASR_CHECK_HIP(hipEventRecord(e0, stream));
  }
  hipLaunchKernelGGL((oasr_gemm_pp_kernel<TA, TB, SWAP, CSUM, DMA>), grid, dim3(512), lds, stream, pa);
  OASR_LAUNCH_CHECK();
  if (e1) OASR_CHECK_HIP(hipEventRecord(e1, stream));
  return OASR_OK;
}
template <bool TA, bool TB, bool SWAP, bool CSUM = false>
int launch_pp_cfg(const GemmArgs& a, hipStream_t stream) {
  // Where the direct-to-LDS pieces are issued: between the MFMAs (+3..5 % on the NT forward shapes at M = 192000,
  // profiles/r02_gemm_variants_ab.txt) or in the fragment-read section (level or better when an operand is read
  // through ds_read_b64_tr_b16: those read sections are longer and hide the issue).  g_pp_dma_in_mma: -1 = this rule,
  // 0 / 1 = forced (scripts/gemm_stagger_ab.py variant).
  // Default (profiles/r02_gemm_variants_ab.txt, r02_step_gemm_variants_same_box.txt): two pinned 16-MFMA sections per K-tile with
  // the DMA pieces spread over read and MFMA sections (7): +8 % on the NT forward shapes at K = 1024, +16..19 % on the NN dgrads
  // whose B fragments come through ds_read_b64_tr_b16.  The fused-colsum dgrad keeps four pinned 8-MFMA sections (2), which won
  // the whole-step A/B for it.  (NT at K = 4096 is 2 % faster on 2; not worth a second instantiation per layout.)
  static const int env_var = [] {
    const char* e = oasr_experiment_env("OASR_PP_VARIANT");  // experiments: force one variant for a whole process
    return e ? atoi(e) : -1;
  }();
  const int forced = g_pp_dma_in_mma >= 0 ? g_pp_dma_in_mma : env_var;
  const int var = forced >= 0 ? forced : (CSUM ? 2 : 7);
  switch (var & 15) {
    case 1: return launch_pp_variant<TA, TB, SWAP, CSUM, 1>(a, stream);
    case 2: return launch_pp_variant<TA, TB, SWAP, CSUM, 2>(a, stream);
    case 3: return launch_pp_variant<TA, TB, SWAP, CSUM, 3>(a, stream);
    case 6: return launch_pp_variant<TA, TB, SWAP, CSUM, 6>(a, stream);
    case 7: return launch_pp_variant<TA, TB, SWAP, CSUM, 7>(a, stream);
    case 15: return launch_pp_variant<TA, TB, SWAP, CSUM, 15>(a, stream);
    default: return launch_pp_variant<TA, TB, SWAP, CSUM, 0>(a, stream);
  }
}

// Geometry choice for bf16-output GEMMs, from interleaved A/B runs on the OLMoASR-medium shapes (scripts/gemm_ab.py,
// encoder M = 96000 and decoder M = 28672 tokens): the ping-pong kernel is faster on every layer shape (+4..20 % on
// the encoder, +4..43 % on the decoder where 256 x 128 tiles quantise badly) except the dGELU-epilogue dgrad (-3 %).
// Below ~half a wave of 256 x 256 tiles the 256 x 128 geometry keeps more CUs busy.
bool prefer_pingpong(const GemmArgs& a) {
  return a.K >= 2 * BK && (long)cdiv(a.M, 256) * cdiv(a.N, 256) > 128;
}

int launch_skinny(const GemmArgs& a, hipStream_t stream) {
  const dim3 grid(cdiv(a.N, 32));
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
    g_prof.push({0, 2.0 * (double)a.M * (double)a.N * (double)a.K, a.M <= 32 ? "gemm_skinny_kernel<1>" : "gemm_skinny_kernel<2>"});
    OASR_CHECK_HIP(hipEventRecord(e0, stream));
  }
  if (a.M <= 32)
    hipLaunchKernelGGL(gemm_skinny_kernel<1>, grid, dim3(256), 0, stream, a);
  else
    hipLaunchKernelGGL(gemm_skinny_kernel<2>, grid, dim3(256), 0, stream, a);
  OASR_LAUNCH_CHECK();
  if (e1) OASR_CHECK_HIP(hipEventRecord(e1, stream));
  return OASR_OK;
}

template <bool TA, bool TB>
int launch_fast_t(const GemmArgs& a, hipStream_t stream) {
  const bool atomic_only = a.atomic && a.out_f32 && !a.out && !a.out_pre;
  // Geometry (measured: scripts/gemm_ab.py, scripts/wgrad_sweep.py): bf16 outputs -> the ping-pong kernel unless the
  // problem has under half a wave of 256x256 tiles; split-K / atomic outputs (wgrad) -> 256x128 with 3 workgroups per
  // CU, whose co-resident workgroups hide the atomic epilogues.  OASR_GEMM_GEOM=1|2|3 overrides (experiments).
  static const int env_geom = [] {
    const char* e = oasr_experiment_env("OASR_GEMM_GEOM");
    return e ? atoi(e) : 0;
  }();
  const int geom = g_fast_geometry ? g_fast_geometry : env_geom;
  const bool big = geom == 2;  // (the 256x256 / 2-stage geometry lost to 256x128 on every measured shape incl. small split-K outputs: scripts/wgrad_sweep.py)
  if (geom == 3 || (geom == 0 && !atomic_only && prefer_pingpong(a)) ||
      (geom == 0 && atomic_only && a.atomic_on_pp && (a.M % 256) == 0 && (a.N % 256) == 0)) {  // 256x256 ping-pong kernel
    if (atomic_only) return launch_pp_cfg<TA, TB, false>(a, stream);
    if (a.colsum && !TA && TB) {
      const int rc = launch_pp_cfg<false, true, true, true>(a, stream);
      if (rc || !a.colsum_scratch) return rc;  // (without scratch the kernel used atomics)
      return launch_colsum_accum(a.colsum_scratch, a.N, 2L * cdiv(a.M, 256), a.N, a.colsum, stream);
    }
    const int rc = launch_pp_cfg<TA, TB, true>(a, stream);
    return (rc || !a.colsum) ? rc : launch_colsum_accum(a.out, a.ldc, a.M, a.N, a.colsum, stream);
  }
  if (atomic_only) {
    if (big) return launch_fast_cfg<TA, TB, 256, 4, 2, false>(a, stream);
    return launch_fast_cfg<TA, TB, 128, 2, 1, false>(a, stream);
  }
  if (big) {
    const int rc = launch_fast_cfg<TA, TB, 256, 4, 2, true>(a, stream);
    return (rc || !a.colsum) ? rc : launch_colsum_accum(a.out, a.ldc, a.M, a.N, a.colsum, stream);
  }
  if (a.colsum) {
    if (!TA && TB) {  // dgrad + fused bias gradient
      const int rc = launch_fast_cfg<false, true, 128, 2, 1, true, true>(a, stream);
      if (rc || !a.colsum_scratch) return rc;
      return launch_colsum_accum(a.colsum_scratch, a.N, 2L * cdiv(a.M, 256), a.N, a.colsum, stream);
    }
    const int rc = launch_fast_cfg<TA, TB, 128, 2, 1, true>(a, stream);
    return rc ? rc : launch_colsum_accum(a.out, a.ldc, a.M, a.N, a.colsum, stream);
  }
  return launch_fast_cfg<TA, TB, 128, 2, 1, true>(a, stream);
}

template <bool TA, bool TB>
int launch_t(const GemmArgs& a, hipStream_t stream) {
  static LdsAttrOnce attr;
  const int lds = 4 * TILE_BYTES;
  { const int rc_ = ensure_dynamic_lds(attr, (const void*)gemm_kernel<TA, TB>, lds); if (rc_) return rc_; }
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
    static const std::string name = std::string("gemm_kernel<") + (TA ? "true" : "false") + ", " + (TB ? "true" : "false") + ">";
    g_prof.push({(TA ? 2 : 0) + (TB ? 1 : 0), 2.0 * (double)a.M * nn * kk, name.c_str()});
    OASR_CHECK_HIP(hipEventRecord(e0, stream));
  }
  hipLaunchKernelGGL((gemm_kernel<TA, TB>), grid, dim3(256), lds, stream, a);
  OASR_LAUNCH_CHECK();
  if (e1) OASR_CHECK_HIP(hipEventRecord(e1, stream));
  return OASR_OK;
}

}  // namespace

// v < 0: defaults.  bits 0-3: kernel variant (8 = the per-layout default); bits 4-5: 0 = default launch rule, 1 = never
// persistent, 2 = persistent whenever possible.  (24 = default kernels, plain launches; 40 = default kernels, persistent)
void gemm_set_variant(int v) {
  g_pp_dma_in_mma = (v < 0 || (v & 15) == 8) ? -1 : (v & 15);
  g_pp_persistent = v < 0 ? -1 : (((v >> 4) & 3) == 0 ? -1 : ((v >> 4) & 3) - 1);
  g_epi_flags = v < 0 ? -1 : ((v >> 6) & 3);  // bit 6: non-temporal output stores, bit 7: non-temporal side-input loads
}
void gemm_set_stagger(int sleeps, int phases) {  // sleeps < 0: stagger off everywhere (A/B baseline)
  g_stagger = sleeps;
  g_stagger_phases = phases < 2 ? 2 : phases;
}

int launch_gemm(const GemmArgs& a, hipStream_t stream) {
  OASR_REQUIRE(a.A.ptr && a.B.ptr, "gemm: null operand");
  if (g_epi_flags >= 0 && a.epi_flags != g_epi_flags + 256) {
    GemmArgs b = a;
    b.epi_flags = g_epi_flags + 256;  // (+256: marks the override as applied)
    return launch_gemm(b, stream);
  }
  if (g_stagger > 0 && a.stagger == 0) {
    GemmArgs b = a;
    b.stagger = g_stagger;
    b.stagger_phases = g_stagger_phases;
    return launch_gemm(b, stream);
  }
  // measured (scripts/gemm_stagger_ab.py, M = 192000): +14 % on N = 1024, K = 1024 (tiles of ~8 us whose epilogues collide),
  // 0..-3 % on every longer tile -> only the short ones are staggered: 8 phases x one s_sleep(127)
  if (g_stagger == 0 && a.stagger == 0 && a.N <= 1024 && a.K <= 1024 && a.M >= 16384 && a.out && !a.out_f32) {
    GemmArgs b = a;
    b.stagger = 1;
    b.stagger_phases = 8;
    return launch_gemm(b, stream);
  }
  OASR_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm: bad shape %d %d %d", a.M, a.N, a.K);
  OASR_REQUIRE((a.N % 4) == 0, "gemm: N (%d) must be a multiple of 4", a.N);
  OASR_REQUIRE((a.A.ld % 8) == 0 && (a.B.ld % 8) == 0, "gemm: operand leading dims must be multiples of 8 (16-byte loads)");
  OASR_REQUIRE(a.ta || (a.K % 8) == 0 || a.A.rpb, "gemm: K must be a multiple of 8 for k-contiguous A");
  OASR_REQUIRE(!a.ta || (a.M % 8) == 0 || true, "gemm");
  OASR_REQUIRE(a.split_k >= 1, "gemm: split_k");
  OASR_REQUIRE(a.split_k == 1 || (a.atomic && a.out_f32 && !a.out && !a.out_pre), "gemm: split_k > 1 needs atomic fp32 output only");
  OASR_REQUIRE(a.out || a.out_f32 || a.out_pre, "gemm: no output");
  OASR_REQUIRE(!a.colsum || (a.out && (a.N % 8) == 0 && a.split_k == 1), "gemm: colsum needs a bf16 `out`, N % 8 == 0, split_k == 1");
  static const int env_gm = [] {
    const char* e = oasr_experiment_env("OASR_GEMM_GM");
    return e ? atoi(e) : 0;
  }();
  if (a.raster_gm == 0 && env_gm > 0) {
    GemmArgs b = a;
    b.raster_gm = env_gm;
    return launch_gemm(b, stream);
  }
  // decode-sized problems: a few token rows against a whole weight matrix
  if (a.M <= 64 && !a.ta && !a.tb && !a.A.rpb && !a.B.rpb && (a.K % BK) == 0 && a.split_k == 1 && !a.atomic && !a.colsum &&
      !g_force_general && g_fast_geometry == 0)
    return launch_skinny(a, stream);
  const bool atomic_only = a.atomic && a.out_f32 && !a.out && !a.out_pre;
  const bool fast = !a.A.rpb && !a.B.rpb && (a.K % BK) == 0 && (!a.ta || (a.M % 8) == 0) && (!a.tb || (a.N % 8) == 0) &&
                    a.M >= 8 && a.N >= 8 && !g_force_general && (atomic_only || fast_rows_ok(a));
  if (fast) {
    if (!a.ta && !a.tb) return launch_fast_t<false, false>(a, stream);
    if (!a.ta && a.tb) return launch_fast_t<false, true>(a, stream);
    if (a.ta && !a.tb) return launch_fast_t<true, false>(a, stream);
    return launch_fast_t<true, true>(a, stream);
  }
  int rc;
  if (!a.ta && !a.tb) rc = launch_t<false, false>(a, stream);
  else if (!a.ta && a.tb) rc = launch_t<false, true>(a, stream);
  else if (a.ta && !a.tb) rc = launch_t<true, false>(a, stream);
  else rc = launch_t<true, true>(a, stream);
  if (rc == OASR_OK && a.colsum) rc = launch_colsum_accum(a.out, a.ldc, a.M, a.N, a.colsum, stream);  // unfused fallback
  return rc;
}

void gemm_profile_enable(int on) {
  g_prof.on = on != 0;
  if (on) g_prof.recs.clear();
}
int gemm_profile_lane(int lane) {
  const int old = g_prof.lane;
  if (lane >= 0) g_prof.lane = lane;  // (negative: query only)
  return old;
}

// Sums elapsed ms / algorithmic flops / launch count per operand layout (index = 2*ta + tb) and, as text
// "symbol\tlaunches\tms\tflops\n", per kernel symbol (so bench.py can be checked against rocprofv3's per-kernel
// averages).  Synchronises.
int gemm_profile_collect(double ms[4], double flops[4], long count[4], char* by_symbol, int cap) {
  for (int i = 0; i < 4; ++i) {
    ms[i] = 0;
    flops[i] = 0;
    count[i] = 0;
  }
  struct Agg {
    long n = 0;
    double ms = 0, flops = 0;
  };
  std::map<std::string, Agg> agg;
  for (size_t i = 0; i < g_prof.recs.size(); ++i) {
    OASR_CHECK_HIP(hipEventSynchronize(g_prof.events[2 * i + 1]));
    float t = 0.f;
    OASR_CHECK_HIP(hipEventElapsedTime(&t, g_prof.events[2 * i], g_prof.events[2 * i + 1]));
    const int k = g_prof.recs[i].kind;
    if (g_prof.recs[i].lane != 1) {  // per-layout totals: main-stream launches only (side-stream spans are queueing times)
      ms[k] += t;
      flops[k] += g_prof.recs[i].flops;
      count[k] += 1;
    }
    static const bool by_shape = oasr_experiment_env("OASR_PROF_SHAPES") != nullptr;  // experiments: one line per (symbol, shape, epilogue)
    std::string key = g_prof.recs[i].name;
    if (g_prof.recs[i].lane) key += g_prof.recs[i].lane == 1 ? " [side]" : " [shared]";
    if (by_shape) {
      char sfx[96];
      snprintf(sfx, sizeof(sfx), " M=%d N=%d K=%d epi=%d", g_prof.recs[i].M, g_prof.recs[i].N, g_prof.recs[i].K, g_prof.recs[i].epi);
      key += sfx;
    }
    Agg& a = agg[key];
    a.n += 1;
    a.ms += t;
    a.flops += g_prof.recs[i].flops;
  }
  if (by_symbol && cap > 0) {
    std::string out;
    for (auto& kv : agg) {
      char line[512];
      snprintf(line, sizeof(line), "%s\t%ld\t%.6f\t%.6e\n", kv.first.c_str(), kv.second.n, kv.second.ms, kv.second.flops);
      out += line;
    }
    snprintf(by_symbol, cap, "%s", out.c_str());
  }
  g_prof.recs.clear();
  return OASR_OK;
}

void gemm_force_general(int on) {
  g_force_general = (on == 1);
  g_fast_geometry = on >= 2 ? on - 1 : 0;  // 2 -> force 256x128, 3 -> force 256x256 (2-stage), 4 -> force 256x256 ping-pong
}

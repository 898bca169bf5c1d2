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
  };
  std::vector<Rec> recs;
};
GemmProfile g_prof;
bool g_force_general = false;
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

template <bool TRANS, int ROWS, int NI, int NW>
__device__ __forceinline__ void fast_offsets(const OperandView& v, int R, int row0, int lane, int wave, unsigned (&off)[NI]) {
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int q = j * NW + wave;  // 1 KiB chunk index inside the tile image
    if (!TRANS) {
      const int row = q * 8 + (lane >> 3), phys = lane & 7;
      const int c16 = phys ^ ((row >> 1) & 7);
      int gr = row0 + row;
      gr = gr < R ? gr : R - 1;
      off[j] = (unsigned)(((long)(gr - row0) * v.ld + c16 * 8) * 2);
    } else {
      constexpr int CPR = ROWS / 8;  // 16-byte pieces per k-row
      constexpr int KPC = 64 / CPR;  // k-rows per 1 KiB chunk
      const int krow = q * KPC + lane / CPR, phys = lane % CPR;
      const int c16 = phys ^ ((krow & 3) << 2);
      int col = row0 + c16 * 8;
      col = col + 8 <= R ? col : R - 8;
      off[j] = (unsigned)(((long)krow * v.ld + (col - row0)) * 2);
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

}  // namespace

// (external linkage: hipcc 7.2 drops the host-side handle of this instantiation set when it has internal linkage)
template <bool TA, bool TB, int FBN, int NWN, int NSTAGE, bool SWAP, bool CSUM>
__global__ __launch_bounds__(128 * NWN, NWN == 2 ? 3 : 2) void oasr_gemm_fast_kernel(GemmArgs p) {
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
      constexpr int RESIDENT = 32 * (NWN == 2 ? 3 : 1);  // workgroups resident per XCD
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

  const int h = lane >> 5;
  if (SWAP) {
    // bf16 outputs leave through a per-wave 8 KiB LDS staging tile ([64 rows][128 B], 16-byte chunks XOR-swizzled
    // by row) so that global stores are 16 bytes per lane and 128 contiguous bytes per output row, instead of 8-byte
    // pieces scattered over 32 rows.  All LDS is free here: the main loop ended on a barrier.
    char* stg = smem + wave * 8192;  // [0,4K): pre-activation rows, [4K,8K): final rows; 32 rows x 128 B each
    const bool vec_ok = (p.N % 8) == 0 && (p.ldc % 8) == 0;
    float ctot = 0.f;  // CSUM: this lane's column (n0 + wn*64 + lane) summed over the wave's 128 rows
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int m = m0 + wm * 128 + mt * 32 + (lane & 31);
      const int mc = m < p.M ? m : p.M - 1;  // clamp: loads stay in range, stores are masked below
      const int pos_row = p.pos ? (mc % p.pos_period) : 0;
      const int row = lane & 31;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          int n = n0 + wn * 64 + nt * 32 + 8 * q + 4 * h;
          const bool n_ok = n < p.N;
          n = n_ok ? n : 0;
          float v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = acc[mt][nt][q * 4 + i];
          u32x2_t pre, fin;
          epilogue_math(p, mc, n, pos_row, v, pre, fin);
          if (p.out_f32 && m < p.M && n_ok) epilogue_f32(p, m, n, v);
          if (!vec_ok) {
            if (m < p.M && n_ok) {
              if (p.out_pre) *(u32x2_t*)(p.out_pre + (long)m * p.ldc + n) = pre;
              if (p.out) *(u32x2_t*)(p.out + (long)m * p.ldc + n) = fin;
            }
          } else {
            const int a = row * 128 + (((nt * 4 + q) ^ (row & 7)) << 4) + h * 8;
            if (p.out_pre) *(u32x2_t*)(stg + a) = pre;
            if (p.out) *(u32x2_t*)(stg + 4096 + a) = fin;
          }
        }
      }
      if (vec_ok) {
        __builtin_amdgcn_wave_barrier();  // wave-private staging: LDS ops of one wave execute in order
        if (CSUM) {  // fused bias gradient: column sums of the staged (bf16) output rows, valid rows only
          const int mrow0 = m0 + wm * 128 + mt * 32;
#pragma unroll
          for (int r2 = 0; r2 < 32; ++r2) {
            const bf16_t e = *(const bf16_t*)(stg + 4096 + r2 * 128 + (((lane >> 3) ^ (r2 & 7)) << 4) + (lane & 7) * 2);
            if (mrow0 + r2 < p.M) ctot += bf2f(e);
          }
        }
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
          bf16_t* dst = pass == 0 ? p.out_pre : p.out;
          if (!dst) continue;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r2 = i * 8 + (lane >> 3), ch = lane & 7;
            const u32x4_t val = *(const u32x4_t*)(stg + pass * 4096 + r2 * 128 + ((ch ^ (r2 & 7)) << 4));
            const int mm = m0 + wm * 128 + mt * 32 + r2;
            const int nn = n0 + wn * 64 + ch * 8;
            if (mm < p.M && nn < p.N) *(u32x4_t*)(dst + (long)mm * p.ldc + nn) = val;
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (CSUM) {
      const int n = n0 + wn * 64 + lane;
      if (n < p.N) unsafeAtomicAdd(p.colsum + n, ctot);
    }
  } else {
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

namespace {

template <bool TA, bool TB, int FBN, int NWN, int NSTAGE, bool SWAP, bool CSUM = false>
int launch_fast_cfg(const GemmArgs& a, hipStream_t stream) {
  static bool attr = false;
  const int lds = NSTAGE * (FBM * 64 * 2 + FBN * 64 * 2);
  if (!attr) {
    OASR_CHECK_HIP(hipFuncSetAttribute((const void*)oasr_gemm_fast_kernel<TA, TB, FBN, NWN, NSTAGE, SWAP, CSUM>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr = true;
  }
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
    g_prof.recs.push_back({(TA ? 2 : 0) + (TB ? 1 : 0), 2.0 * (double)a.M * (double)a.N * (double)a.K, name.c_str()});
    OASR_CHECK_HIP(hipEventRecord(e0, stream));
  }
  hipLaunchKernelGGL((oasr_gemm_fast_kernel<TA, TB, FBN, NWN, NSTAGE, SWAP, CSUM>), grid, dim3(128 * NWN), lds, stream, a);
  OASR_LAUNCH_CHECK();
  if (e1) OASR_CHECK_HIP(hipEventRecord(e1, stream));
  return OASR_OK;
}

template <bool TA, bool TB>
int launch_fast_t(const GemmArgs& a, hipStream_t stream) {
  const bool atomic_only = a.atomic && a.out_f32 && !a.out && !a.out_pre;
  const long big_tiles = (long)cdiv(a.M, FBM) * cdiv(a.N, 256) * a.split_k;
  // Geometry: measured on the OLMoASR-medium shapes (scripts/gemm_bench.py) the 256x128 / 3-workgroups-per-CU
  // geometry is at least as fast as the 1-workgroup-per-CU 256x256 one everywhere except very small split-K outputs;
  // co-resident workgroups overlap one's VALU-heavy epilogue with another's MFMA main loop.
  // OASR_GEMM_GEOM=1|2 overrides (experiments).
  static const int env_geom = [] {
    const char* e = getenv("OASR_GEMM_GEOM");
    return e ? atoi(e) : 0;
  }();
  const int geom = g_fast_geometry ? g_fast_geometry : env_geom;
  const bool big = geom == 2 || (geom == 0 && atomic_only && (long)cdiv(a.M, FBM) * cdiv(a.N, 128) <= 32 && big_tiles >= 128);
  if (atomic_only) {
    if (big) return launch_fast_cfg<TA, TB, 256, 4, 2, false>(a, stream);
    return launch_fast_cfg<TA, TB, 128, 2, 1, false>(a, stream);
  }
  if (big) {
    const int rc = launch_fast_cfg<TA, TB, 256, 4, 2, true>(a, stream);
    return (rc || !a.colsum) ? rc : launch_colsum_accum(a.out, a.ldc, a.M, a.N, a.colsum, stream);
  }
  if (a.colsum) {
    if (!TA && TB) return launch_fast_cfg<false, true, 128, 2, 1, true, true>(a, stream);  // dgrad + fused bias gradient
    const int rc = launch_fast_cfg<TA, TB, 128, 2, 1, true>(a, stream);
    return rc ? rc : launch_colsum_accum(a.out, a.ldc, a.M, a.N, a.colsum, stream);
  }
  return launch_fast_cfg<TA, TB, 128, 2, 1, true>(a, stream);
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
    static const std::string name = std::string("gemm_kernel<") + (TA ? "true" : "false") + ", " + (TB ? "true" : "false") + ">";
    g_prof.recs.push_back({(TA ? 2 : 0) + (TB ? 1 : 0), 2.0 * (double)a.M * nn * kk, name.c_str()});
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
  OASR_REQUIRE(!a.colsum || (a.out && (a.N % 8) == 0 && a.split_k == 1), "gemm: colsum needs a bf16 `out`, N % 8 == 0, split_k == 1");
  static const int env_gm = [] {
    const char* e = getenv("OASR_GEMM_GM");
    return e ? atoi(e) : 0;
  }();
  if (a.raster_gm == 0 && env_gm > 0) {
    GemmArgs b = a;
    b.raster_gm = env_gm;
    return launch_gemm(b, stream);
  }
  const bool fast = !a.A.rpb && !a.B.rpb && (a.K % BK) == 0 && (!a.ta || (a.M % 8) == 0) && (!a.tb || (a.N % 8) == 0) &&
                    a.M >= 8 && a.N >= 8 && !g_force_general;
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
    ms[k] += t;
    flops[k] += g_prof.recs[i].flops;
    count[k] += 1;
    Agg& a = agg[g_prof.recs[i].name];
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
  g_fast_geometry = on >= 2 ? on - 1 : 0;  // 2 -> force 256x128, 3 -> force 256x256
}

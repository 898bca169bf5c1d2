// Flash attention (head_dim 64) forward + backward for gfx950, bf16 in/out, fp32 softmax and accumulation.
//
// Replaces F.scaled_dot_product_attention as called by the reference MultiHeadAttention
// (olmoasr/model.py:317-340): encoder self-attention (no mask), decoder self-attention (causal + key padding
// mask, model.py:740-741, given here as kv_len[b] -- the [B,448,448] additive mask of
// train_timestamps.py:314-315 is column-only) and cross-attention (no mask, Tq=448, Tk=1500), SURVEY.md K6-K8, K14.
//
// Register layout trick: every score tile is computed TRANSPOSED, S^T[k][q] = K.Q^T, with
// v_mfma_f32_32x32x16_bf16.  In the 32x32 accumulator layout lane l owns column q = l & 31 and 16 rows (keys), so
// the online softmax (row max / exp / row sum / rescale) is lane-local except for one exchange between lanes
// l and l^32.  The exponentiated tile, packed to bf16, already IS the B operand of O^T[d][q] = V^T.P^T, and V^T is
// produced from the row-major V tile in LDS by ds_read_b64_tr_b16 -- P never touches LDS.  The same idea drives
// the backward: dK/dV kernel computes S[q][k] with the key lane-local, dQ kernel computes S^T with the query
// lane-local; P and dS feed the second-stage MFMAs straight from registers.
//
// forward : grid (ceil(Tq/128), B*H), 4 waves x 32 queries, KV tiles of 64 keys through a 3-stage LDS-DMA ring.
// backward: bwd_dq   grid (ceil(Tq/128), B*H)  -- also produces delta = rowsum(dO * O)
//           bwd_dkdv grid (ceil(Tk/128), B*H)  -- loops over 64-query tiles of Q / dO / lse / delta.
#include "decode_shared.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float SCALE = 0.125f;  // 1/sqrt(64)
constexpr float NEG = -1.0e30f;
constexpr int TILE = 64 * 64 * 2;  // one 64x64 bf16 tile

typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;

// Bank-conflict swizzle of a [64 rows][128 B] tile: chunk c16 of row r lives at chunk c16 ^ f((r >> 1) & 7), f(x) = x ^ ((x & 1) << 2).
//  * ds_read_b128 (frag_rows) is served in 16-lane groups whose rows cover every value of (r >> 1) & 7 exactly twice (once per
//    row parity = bank half): any bijection of that value keeps the 16 x 4 banks distinct;
//  * ds_read_b64_tr_b16 (frag_cols) is served in 32-lane groups = 4 consecutive rows x 4 consecutive chunks: rows r and r + 2 share
//    the bank half, so their chunk sets must be disjoint, i.e. f must differ in bit 2 between (r >> 1) and (r >> 1) + 1.  The plain
//    XOR with (r >> 1) & 7 did not (measured: 20-25 % of the LDS cycles of the three kernels were conflicts).
__device__ __forceinline__ int tile_addr(int row, int c16) {
  const int x = (row >> 1) & 7;
  return row * 128 + ((c16 ^ x ^ ((x & 1) << 2)) << 4);
}

// cooperative 64x64 tile: each of 256 threads moves 2 x 16 bytes.  Bounds-checked buffer loads: the descriptor covers
// rows [0, rmax) of this (batch, head) slice, so rows past the end read as zeros (their scores are masked, and zero V
// rows keep 0 * V finite); the per-thread offsets are loop invariants and the tile advance is a scalar offset, so the
// loop carries no vector address arithmetic.
struct TileSrc {
  __amdgpu_buffer_rsrc_t rs;
  unsigned voff[2];
  unsigned row_bytes;
};
__device__ __forceinline__ TileSrc tile_src(const bf16_t* base, long ld, int rmax, int tid) {
  TileSrc t;
  // last valid byte: row (rmax-1), 64 columns
  t.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(base), 0, (int)(((long)(rmax - 1) * ld + 64) * 2), 0x00020000);
  t.row_bytes = (unsigned)(ld * 2);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = tid + 256 * i;
    t.voff[i] = (unsigned)((id >> 3) * ld * 2 + (id & 7) * 16);
  }
  return t;
}
__device__ __forceinline__ void tile_issue(const TileSrc& t, int row0, u32x4_t (&r)[2]) {
  const unsigned soff = (unsigned)row0 * t.row_bytes;
#pragma unroll
  // the tile offset goes into the VGPR offset: on gfx9-class raw buffers the range check does not cover soffset
  for (int i = 0; i < 2; ++i) r[i] = __builtin_amdgcn_raw_buffer_load_b128(t.rs, t.voff[i] + soff, 0, 0);
}
__device__ __forceinline__ void tile_commit(char* lds, int tid, const u32x4_t (&r)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int id = tid + 256 * i;
    *(u32x4_t*)(lds + tile_addr(id >> 3, id & 7)) = r[i];
  }
}
// MFMA operand: lane holds tile row (sub + (l & 31)), 8 contiguous columns ds*16 + (l >> 5)*8 ..
__device__ __forceinline__ bf16x8_t frag_rows(const char* lds, int sub, int ds, int lane) {
  return *(const bf16x8_t*)(lds + tile_addr(sub + (lane & 31), ds * 2 + (lane >> 5)));
}
// MFMA operand from the TRANSPOSE of the tile: lane holds column (csub + (l & 31)) and the 8 rows
// rbase + 8*(j >> 2) + 4*(l >> 5) + (j & 3), j = 0..7 -- exactly the rows a 32x32 accumulator's registers
// 8u..8u+7 cover, so a packed accumulator half is the matching other operand.
__device__ __forceinline__ bf16x8_t frag_cols(const char* lds, int rbase, int csub, int lane) {
  const int G = lane >> 4, i = lane & 15;
  const int row = rbase + 4 * (G >> 1) + (i >> 2);
  const int col = csub + (G & 1) * 16 + (i & 3) * 4;
  const int a0 = tile_addr(row, col >> 3) + (col & 7) * 2;
  const int a1 = tile_addr(row + 8, col >> 3) + (col & 7) * 2;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lds + a0));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lds + a1));
  const s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ bf16x8_t pack_half(const f32x16_t& p, int u) {
  u32x4_t o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = pack_bf2(p[8 * u + 2 * i], p[8 * u + 2 * i + 1]);
  return __builtin_bit_cast(bf16x8_t, o);
}
__device__ __forceinline__ bf16x8_t ld_frag_global(const bf16_t* p) { return __builtin_bit_cast(bf16x8_t, *(const u32x4_t*)p); }
__device__ __forceinline__ int acc_row(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int t = __shfl_xor(v, o, 64);
    v = t > v ? t : v;
  }
  return v;
}

// ---- row-coalesced output stores --------------------------------------------------------------------------------------
// The kernels hold their results transposed, T[dt][r] = out[row = lane & 31][col = dt*32 + acc_row(r, lane >> 5)]: a lane owns
// one output row but only 4 consecutive columns per register group, so direct stores are 8-byte pieces scattered over
// 32 rows per instruction (measured: the forward kernel spends 14 % of its time issuing them).  Instead each wave
// transposes through a private LDS tile and every lane stores 16 contiguous bytes, 8 lanes covering one 128-byte row
// segment.  `stg`: wave-private, 4 KiB; all waves must be done with the operand tiles.
// `wave_sums` (optional, 64 floats of LDS owned by this wave): receives the column sums of the bf16 values stored for
// the valid rows (the caller combines the waves and writes ONE partial row per workgroup -- fp32 atomics from every wave
// onto the same d addresses were measured to cost more than the kernel itself).
__device__ __forceinline__ void store_rows_bf16(char* stg, const f32x16_t (&T)[2], float scale, bf16_t* gbase, long ld, int rows_valid,
                                                int lane, float* wave_sums = nullptr) {
  const int row = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      u32x2_t o;
      o[0] = pack_bf2(T[dt][4 * g4] * scale, T[dt][4 * g4 + 1] * scale);
      o[1] = pack_bf2(T[dt][4 * g4 + 2] * scale, T[dt][4 * g4 + 3] * scale);
      *(u32x2_t*)(stg + row * 128 + (((dt * 4 + g4) ^ (row & 7)) << 4) + hh * 8) = o;
    }
  __builtin_amdgcn_wave_barrier();
  float cs[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) cs[e] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r2 = i * 8 + (lane >> 3), ch = lane & 7;
    const u32x4_t v = *(const u32x4_t*)(stg + r2 * 128 + ((ch ^ (r2 & 7)) << 4));
    if (r2 < rows_valid) {
      *(u32x4_t*)(gbase + (long)r2 * ld + ch * 8) = v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        cs[2 * e] += bf_lo(v[e]);
        cs[2 * e + 1] += bf_hi(v[e]);
      }
    }
  }
  if (wave_sums) {  // (wave-uniform) lanes sharing (lane & 7) hold partial sums of the same 8 columns
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = cs[e];
      t += __shfl_xor(t, 8, 64);
      t += __shfl_xor(t, 16, 64);
      t += __shfl_xor(t, 32, 64);
      cs[e] = t;
    }
    if (lane < 8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) wave_sums[lane * 8 + e] = cs[e];  // zeros when this wave has no valid row
    }
  }
  __builtin_amdgcn_wave_barrier();
}

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// Chunked token rows (AttnArgs.q_rows / k_rows, kernels.h): first row of the 64-position chunk that holds position t of sample b.
// Wave-uniform when t is (scalar load); entries past the last chunk hold a row far outside every tensor.
__device__ __forceinline__ int chunk_row(const int32_t* rows, int b, int t) { return rows[b * OASR_ROWTAB + (t >> 6)] + (t & 63); }

// Softmax arithmetic on pairs: gfx950's v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 process two fp32 values per lane per
// issue slot.  With head_dim 64 these kernels are VALU-issue bound (16 MFMAs against ~145 scalar VALU per 64-key tile and
// wave in the forward), so halving the fma / add / mul counts is what moves them; the exponential stays one v_exp_f32 each.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t pk_fma(f32x2_t a, f32x2_t b, f32x2_t c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2_t pk_exp2(f32x2_t a) {
  f32x2_t r;
  r[0] = __builtin_amdgcn_exp2f(a[0]);
  r[1] = __builtin_amdgcn_exp2f(a[1]);
  return r;
}
// Lazy rescale (log2 domain): the running reference maximum is only raised -- and O / l only rescaled -- when some row of
// the wave grows past it by more than this; until then P = exp2(s - m_ref) <= 2^8, harmless in fp32 / bf16.
constexpr float RESCALE_THR = 8.0f;

// ---- LDS-DMA ring helpers (used by the forward below and by the ping-pong backward kernels further down, where the scheme is described) ----
#define ATTN_FENCE() __builtin_amdgcn_sched_barrier(0)
constexpr int PNS = 4;          // ring stages
constexpr int PSTG = 2 * TILE;  // two 64-row tiles per stage

// The DMA is issued from inline assembly on purpose: hipcc orders every later ds_read behind a compiler-visible LDS-DMA with
// s_waitcnt vmcnt(0) (it cannot tell the ring stages apart), which would drain the tiles in flight at every step.  The loops
// count their own pieces (s_waitcnt vmcnt(n) + s_barrier before a stage is read); in-order return makes any compiler-placed
// vmcnt for its own loads merely conservative.  (m0 is reserved: the compiler-generated code of these kernels has no other user.)
__device__ __forceinline__ void glds16_asm(const u32x4_t rs, unsigned lds_addr, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs) : "memory");
}
struct PipeSrc1 {
  u32x4_t rs;  // raw buffer descriptor over rows [0, rmax) of this (batch, head) slice, 64 columns
  unsigned voff;
  unsigned tile_bytes;
};
// Piece p (1 KiB = 8 rows) of a 64-row tile lands at LDS bytes [p * 1024, +1024) of the tile; lane l writes chunk c' = l & 7 of row
// 8p + (l >> 3), which under tile_addr's swizzle holds source chunk c' ^ g(row).  Wave w of 8 moves piece w of each tile.
__device__ __forceinline__ PipeSrc1 pipe_src1(const bf16_t* base, long ld, int rmax, int wave, int lane) {
  PipeSrc1 t;
  const unsigned long addr = (unsigned long)base;
  t.rs[0] = (unsigned)addr;
  t.rs[1] = (unsigned)(addr >> 32) & 0xffffu;  // stride 0: raw buffer
  t.rs[2] = (unsigned)(((long)(rmax - 1) * ld + 64) * 2);
  t.rs[3] = 0x00020000u;
  t.tile_bytes = (unsigned)(64 * ld * 2);
  const int row = wave * 8 + (lane >> 3);
  const int x = (row >> 1) & 7;
  const int c = (lane & 7) ^ x ^ ((x & 1) << 2);
  t.voff = (unsigned)(row * ld * 2 + c * 16);  // the tile offset is added here too: the range check does not cover soffset
  return t;
}
#define ATTN_BARRIER()                          \
  do {                                          \
    ATTN_FENCE();                               \
    asm volatile("s_barrier" ::: "memory");     \
    ATTN_FENCE();                               \
  } while (0)


// ------------------------------------------------------------------------------------------------------------
// ROWS: the query side (and, when a.k_rows is set, the key side) lives in chunked token rows (kernels.h) -- the decoder of a
// span-limited training step.  ROWS == false is the plain strided layout and compiles to exactly the code it always was.
template <bool CAUSAL, bool ROWS>
__global__ __launch_bounds__(256, 3) void attn_fwd_kernel(AttnArgs a) {
  // K/V tiles move global -> LDS by DMA, two tiles ahead through a 3-stage ring (3 x 16 KiB per workgroup, three workgroups per CU): no staging
  // registers, no commit stores, and a tile has a whole iteration to land before anybody waits for it.  The ONE barrier of an iteration sits
  // between the softmax and the O^T MFMAs: it releases tile t+1, whose first four K fragments are then read under the O^T MFMAs of tile t, so the
  // next iteration's S^T MFMAs start without an LDS round trip.  (In-kernel cycle stamps of the register-staged form it replaces,
  // profiles/r04_attention_forward_stamps.txt: of a wave's 3400 cycles per tile 600 went into waiting for the next tile's global loads and
  // writing them to LDS, 870 into the K fragment round trip + S^T issue.  Results are bit-identical to that form: scripts/attn_fwd_crc.py.)
  __shared__ __attribute__((aligned(1024))) char smem[3 * 2 * TILE];  // stage s: K at 2s*TILE, V at (2s+1)*TILE
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
  // 1-D grid, XCD-aware: all query blocks of one (batch, head) run on the same XCD so K/V are filled into that XCD's
  // L2 once and re-used by the other query blocks (round-robin dispatch would fetch them through all 8 L2s).
  const int nqb = (a.Tq + 127) >> 7;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = bid / nqb;
  const int b = bh / a.H, h = bh - b * a.H;
  const int q0 = (bid - bh * nqb) * 128;
  const int myq = q0 + wave * 32 + (lane & 31);
  const int myq_c = myq < a.Tq ? myq : a.Tq - 1;
  const bool krows = ROWS && a.k_rows != nullptr;  // (block-uniform)
  // span-limited FORWARD (opt-in of the span step): query positions >= q_span[b] are not computed at all -- no output row written
  const int q_lim = (ROWS && a.q_span) ? min(a.q_span[b], a.Tq) : a.Tq;
  if (ROWS && q0 >= q_lim) return;  // (block-uniform, before any barrier)
  const bool w_act = !ROWS || q0 + wave * 32 < q_lim;  // (wave-uniform) an inactive wave computes on whatever its rows hold and stores nothing

  const bf16_t* qp = ROWS ? a.q + (long)chunk_row(a.q_rows, b, myq_c) * a.ldq + h * 64 : a.q + (long)b * a.bsq + (long)myq_c * a.ldq + h * 64;
  bf16x8_t qf[4];
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) qf[ds] = ld_frag_global(qp + ds * 16 + hh * 8);

  int kv_len = a.kv_len ? a.kv_len[b] : a.Tk;
  kv_len = kv_len < a.Tk ? kv_len : a.Tk;
  int kv_end = kv_len;
  if (CAUSAL) kv_end = kv_end < q0 + 128 ? kv_end : q0 + 128;
  const int ntiles = kv_end > 0 ? (kv_end + 63) >> 6 : 1;  // >= 1: a fully masked tile yields l = 0 -> O = 0

  // first row of key tile t (64 keys = one chunk)
  auto krow0 = [&](int t) { return krows ? __builtin_amdgcn_readfirstlane(a.k_rows[b * OASR_ROWTAB + t]) : t * 64; };
  const PipeSrc1 ksrc = krows ? pipe_src1(a.k + h * 64, a.ldk, a.B * a.Tk, wave, lane) : pipe_src1(a.k + (long)b * a.bsk + h * 64, a.ldk, a.Tk, wave, lane);
  const PipeSrc1 vsrc = krows ? pipe_src1(a.v + h * 64, a.ldv, a.B * a.Tk, wave, lane) : pipe_src1(a.v + (long)b * a.bsv + h * 64, a.ldv, a.Tk, wave, lane);
  const unsigned smem_a = (unsigned)(size_t)smem;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);  // (the DMA's LDS base travels in m0: a scalar)
  const unsigned krb = (unsigned)(a.ldk * 2), vrb = (unsigned)(a.ldv * 2);
  // this wave's four 1 KiB pieces of key tile t2 (rows 8w.. and 32+8w.. of K and of V); tiles past the last one read as zeros (offset out of range)
  auto dma_tile = [&](int t2, int stage) {
    unsigned ko = 0x7fffff00u, vo = 0x7fffff00u;
    if (t2 < ntiles) {
      const unsigned r0 = (unsigned)krow0(t2);
      ko = ksrc.voff + r0 * krb;
      vo = vsrc.voff + r0 * vrb;
    }
    const unsigned dst = smem_a + stage * 2 * TILE + wave_u * 1024;
    glds16_asm(ksrc.rs, dst, ko);
    glds16_asm(ksrc.rs, dst + 4096, ko + 32 * krb);
    glds16_asm(vsrc.rs, dst + TILE, vo);
    glds16_asm(vsrc.rs, dst + TILE + 4096, vo + 32 * vrb);
  };

  f32x16_t oT[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oT[i][r] = 0.f;
  float m_run = NEG;
  f32x2_t l_run2 = {0.f, 0.f};

  // Ring protocol (every wave issues 4 pieces per tile; vmcnt counts them in order):
  //   prologue      : tiles 0, 1 in flight; wait for tile 0 (vmcnt(4)) + barrier; K fragments (first 32 keys) of tile 0
  //   iteration t   : S^T MFMAs on tile t | softmax | vmcnt(0) + barrier: tile t+1 has landed for everybody and everybody is past its reads of
  //                   tile t-1 | DMA of tile t+2 into tile t-1's stage | K fragments of tile t+1 | O^T MFMAs on tile t
  asm volatile("" ::"v"(qf[0]), "v"(qf[1]), "v"(qf[2]), "v"(qf[3]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the Q fragments; from here on the loop counts its own DMA pieces)
  dma_tile(0, 0);
  dma_tile(1, 1);
  int stg = 0;
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  ATTN_BARRIER();
  bf16x8_t kf0[4];
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) kf0[ds] = frag_rows(smem, 0, ds, lane);
  for (int t = 0; t < ntiles; ++t) {
    const char* kb = smem + stg * 2 * TILE;
    const char* vb = kb + TILE;
    stg = stg == 2 ? 0 : stg + 1;
    f32x16_t sT[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sT[kt][r] = 0.f;
#pragma unroll
      for (int ds = 0; ds < 4; ++ds) sT[kt] = MFMA(kt == 0 ? kf0[ds] : frag_rows(kb, 32, ds, lane), qf[ds], sT[kt]);
    }
    // Only boundary tiles need the per-element mask (wave-uniform test): last partial tile of kv_len, and for the
    // causal case the tiles that cross this wave's diagonal.  Scores stay raw; the softmax scale rides in the fma.
    constexpr float C = SCALE * LOG2E;
    const int q_lo = q0 + wave * 32;  // smallest query row of this wave
    const bool full = (t * 64 + 64 <= kv_len) && (!CAUSAL || (t * 64 + 63 <= q_lo));
    float m_tile = NEG;
    if (full) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) m_tile = fmaxf(m_tile, sT[kt][r]);
    } else {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = t * 64 + kt * 32 + acc_row(r, hh);
          const bool ok = (key < kv_len) && (!CAUSAL || key <= myq);
          const float sv = ok ? sT[kt][r] : NEG;
          sT[kt][r] = sv;
          m_tile = fmaxf(m_tile, sv);
        }
    }
    m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32, 64));
    const float m_cand = m_tile * C;  // NEG * C stays hugely negative
    if (__any(m_cand > m_run + RESCALE_THR)) {  // wave-uniform; rare after the first tiles
      const float m_new = fmaxf(m_run, m_cand);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      const f32x2_t a2 = {alpha, alpha};
      l_run2 *= a2;
      m_run = m_new;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          f32x2_t o2 = {oT[dt][r], oT[dt][r + 1]};
          o2 *= a2;
          oT[dt][r] = o2[0];
          oT[dt][r + 1] = o2[1];
        }
    }
    {
      const float nm = -m_run;
      float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const float p0 = __builtin_amdgcn_exp2f(fmaf(sT[kt][r], C, nm));
          const float p1 = __builtin_amdgcn_exp2f(fmaf(sT[kt][r + 1], C, nm));
          sT[kt][r] = p0;
          sT[kt][r + 1] = p1;
          ps0 += p0;
          ps1 += p1;
        }
      l_run2[0] += ps0;
      l_run2[1] += ps1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of tile t+1 have landed
    ATTN_BARRIER();                                    // ... and everybody's; everybody is past the O^T reads of tile t-1: its stage is free
    dma_tile(t + 2, stg == 2 ? 0 : stg + 1);           // (stg already names tile t+1's stage)
    if (t + 1 < ntiles) {
#pragma unroll
      for (int ds = 0; ds < 4; ++ds) kf0[ds] = frag_rows(smem + stg * 2 * TILE, 0, ds, lane);
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const bf16x8_t pf = pack_half(sT[kt], u);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) oT[dt] = MFMA(frag_cols(vb, kt * 32 + 16 * u, dt * 32, lane), pf, oT[dt]);
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tail pieces (zeros) must not land on the staging tiles below
  __syncthreads();

  const float l_run = l_run2[0] + l_run2[1];
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  {
    // (the loop ended on a barrier: the K/V stages are free; 4 KiB of staging per wave)
    char* stg = smem + wave * 4096;
    const int rows_valid = w_act ? a.Tq - (q0 + wave * 32) : 0;  // may be <= 0 or > 32
    // (a wave's 32 rows lie inside one chunk; past Tq the table entry is a sentinel and rows_valid <= 0 keeps it unused)
    const long row0 = ROWS ? (long)chunk_row(a.q_rows, b, q0 + wave * 32) * a.ldo + h * 64 : (long)b * a.bso + (long)(q0 + wave * 32) * a.ldo + h * 64;
    store_rows_bf16(stg, oT, inv, a.o + row0, a.ldo, rows_valid, lane);
    // rounding residual of O for the backward's delta term (bf16 O alone loses it when mean(V) dominates V's variation;
    // O + residual is fp32-grade at 4 bytes per element instead of 2 + 4)
    if (a.o_lo) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = oT[dt][r] * inv;
          oT[dt][r] = v - bf_round(v);
        }
      store_rows_bf16(stg, oT, 1.0f, a.o_lo + row0, a.ldo, rows_valid, lane);
    }
    if (myq < a.Tq && hh == 0 && a.lse && w_act) a.lse[((long)b * a.H + h) * a.Tq + myq] = (m_run + __builtin_amdgcn_logf(l_tot)) * LN2;
  }
}

// ------------------------------------------------------------------------------------------------------------
// dQ[q][:] = scale * sum_k dS[q][k] K[k][:],  dS = P o (dP - delta),  P = exp(scale*S - lse),  dP = dO V^T
template <bool CAUSAL, bool ROWS>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
  // 1-D grid, XCD-aware: all query blocks of one (batch, head) run on the same XCD so K/V are filled into that XCD's
  // L2 once and re-used by the other query blocks (round-robin dispatch would fetch them through all 8 L2s).
  const int nqb = (a.Tq + 127) >> 7;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = bid / nqb;
  const int b = bh / a.H, h = bh - b * a.H;
  const int q0 = (bid - bh * nqb) * 128;
  const int myq = q0 + wave * 32 + (lane & 31);
  const bool krows = ROWS && a.k_rows != nullptr;  // (block-uniform)
  // span-limited backward: query positions >= q_span[b] hold no gradient -- their d_o / o rows are not read, dq is not written
  const int q_lim = (ROWS && a.q_span) ? min(a.q_span[b], a.Tq) : a.Tq;
  if (ROWS && q0 >= q_lim) {  // (block-uniform) nothing to do, but the partial bias-gradient row of this block must read as zeros
    if (a.dq_colsum && tid < 64) a.colsum_scratch[((long)(b * nqb + (bid - bh * nqb)) * a.H + h) * 64 + tid] = 0.f;
    return;
  }
  const bool w_act = !ROWS || q0 + wave * 32 < q_lim;        // (wave-uniform; spans are multiples of 64)
  const int myq_c = !w_act ? q0 + (lane & 31) : (myq < a.Tq ? myq : a.Tq - 1);  // an inactive wave re-reads rows of the block's first chunk

  const long qrow = ROWS ? (long)chunk_row(a.q_rows, b, myq_c) : 0;
  const bf16_t* qp = ROWS ? a.q + qrow * a.ldq + h * 64 : a.q + (long)b * a.bsq + (long)myq_c * a.ldq + h * 64;
  const long orow_off = ROWS ? qrow * a.ldo + h * 64 : (long)b * a.bso + (long)myq_c * a.ldo + h * 64;
  const bf16_t* dop = a.d_o + orow_off;
  const bf16_t* op = a.o + orow_off;
  const bf16_t* olop = a.o_lo ? a.o_lo + orow_off : nullptr;
  bf16x8_t qf[4], dof[4];
  float dpart = 0.f;
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) {
    qf[ds] = ld_frag_global(qp + ds * 16 + hh * 8);
    u32x4_t d4 = *(const u32x4_t*)(dop + ds * 16 + hh * 8);
    if (ROWS && !w_act) d4 = u32x4_t{0u, 0u, 0u, 0u};
    dof[ds] = __builtin_bit_cast(bf16x8_t, d4);
    const u32x4_t o4 = *(const u32x4_t*)(op + ds * 16 + hh * 8);
    if (olop) {  // O = bf16 O + bf16 rounding residual (an fp32-grade O in 4 bytes per element, like the forward wrote it)
      const u32x4_t r4 = *(const u32x4_t*)(olop + ds * 16 + hh * 8);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        dpart += bf_lo(d4[i]) * (bf_lo(o4[i]) + bf_lo(r4[i])) + bf_hi(d4[i]) * (bf_hi(o4[i]) + bf_hi(r4[i]));
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) dpart += bf_lo(d4[i]) * bf_lo(o4[i]) + bf_hi(d4[i]) * bf_hi(o4[i]);
    }
  }
  const float delta = dpart + __shfl_xor(dpart, 32, 64);
  const long stat_idx = ((long)b * a.H + h) * a.Tq + myq_c;
  if (hh == 0 && myq < a.Tq && w_act) a.delta[stat_idx] = delta;
  const float lse2 = a.lse[stat_idx] * LOG2E;

  int kv_len = a.kv_len ? a.kv_len[b] : a.Tk;
  kv_len = kv_len < a.Tk ? kv_len : a.Tk;
  int kv_end = kv_len;
  if (CAUSAL) kv_end = kv_end < q0 + 128 ? kv_end : q0 + 128;
  int ntiles = kv_end > 0 ? (kv_end + 63) >> 6 : 1;  // >= 1: a fully masked tile yields l = 0 -> O = 0
  if (a.qtile_flags) {  // which 64-query tiles of d_o hold anything: recorded for the dK/dV kernel; a block of zeros has dQ = 0
    __shared__ int nzs[4];
    bool nz = false;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
      const u32x4_t d4 = __builtin_bit_cast(u32x4_t, dof[ds]);
      nz = nz || ((d4[0] | d4[1] | d4[2] | d4[3]) & 0x7fff7fffu) != 0u;
    }
    const bool wnz = __any(nz);
    if (lane == 0) nzs[wave] = wnz ? 1 : 0;
    __syncthreads();
    const int t0nz = nzs[0] | nzs[1], t1nz = nzs[2] | nzs[3];
    if (tid == 0) {
      const int ntq = (a.Tq + 63) >> 6, tq = q0 >> 6;
      int32_t* f = a.qtile_flags + ((long)b * a.H + h) * ntq;
      if (tq < ntq) f[tq] = t0nz;
      if (tq + 1 < ntq) f[tq + 1] = t1nz;
    }
    if (!(t0nz | t1nz)) ntiles = 0;
  }
  const TileSrc ksrc = krows ? tile_src(a.k + h * 64, a.ldk, a.B * a.Tk, tid) : tile_src(a.k + (long)b * a.bsk + h * 64, a.ldk, a.Tk, tid);
  const TileSrc vsrc = krows ? tile_src(a.v + h * 64, a.ldv, a.B * a.Tk, tid) : tile_src(a.v + (long)b * a.bsv + h * 64, a.ldv, a.Tk, tid);
  auto krow0 = [&](int t) { return krows ? __builtin_amdgcn_readfirstlane(a.k_rows[b * OASR_ROWTAB + t]) : t * 64; };

  f32x16_t dqT[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dqT[i][r] = 0.f;

  // Unconditional prologue (ntiles >= 1 by construction): a guarded one leaves "Q/dO fragment loads may be pending" in
  // the compiler's wait-count state at the loop header, and hipcc then re-waits vmcnt(3..0) in front of the first MFMAs
  // of EVERY iteration -- i.e. for the K/V prefetch it has just issued -- serialising the load latency into the loop.
  u32x4_t rk[2], rv[2];
  tile_issue(ksrc, krow0(0), rk);
  tile_issue(vsrc, krow0(0), rv);
  tile_commit(smem, tid, rk);
  tile_commit(smem + TILE, tid, rv);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const char* kb = smem + (t & 1) * 2 * TILE;
    const char* vb = kb + TILE;
    const bool more = t + 1 < ntiles;
    if (more) {
      const int r0 = krow0(t + 1);
      tile_issue(ksrc, r0, rk);
      tile_issue(vsrc, r0, rv);
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      f32x16_t sT, dpT;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        sT[r] = 0.f;
        dpT[r] = 0.f;
      }
#pragma unroll
      for (int ds = 0; ds < 4; ++ds) {
        sT = MFMA(frag_rows(kb, kt * 32, ds, lane), qf[ds], sT);
        dpT = MFMA(frag_rows(vb, kt * 32, ds, lane), dof[ds], dpT);
      }
      const bool full = (t * 64 + kt * 32 + 32 <= kv_len) && (!CAUSAL || (t * 64 + kt * 32 + 31 <= q0 + wave * 32));
      if (full) {
        const f32x2_t c2 = {SCALE * LOG2E, SCALE * LOG2E}, nl2 = {-lse2, -lse2}, nd2 = {-delta, -delta};
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2_t s2 = {sT[r], sT[r + 1]}, dp2 = {dpT[r], dpT[r + 1]};
          const f32x2_t ds2 = pk_exp2(pk_fma(s2, c2, nl2)) * (dp2 + nd2);
          sT[r] = ds2[0];
          sT[r + 1] = ds2[1];
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = t * 64 + kt * 32 + acc_row(r, hh);
          const bool ok = (key < kv_len) && (!CAUSAL || key <= myq);
          const float p = ok ? __builtin_amdgcn_exp2f(fmaf(sT[r], SCALE * LOG2E, -lse2)) : 0.f;
          sT[r] = p * (dpT[r] - delta);
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const bf16x8_t dsf = pack_half(sT, u);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) dqT[dt] = MFMA(frag_cols(kb, kt * 32 + 16 * u, dt * 32, lane), dsf, dqT[dt]);
      }
    }
    if (more) {
      char* nb = smem + ((t + 1) & 1) * 2 * TILE;
      tile_commit(nb, tid, rk);
      tile_commit(nb + TILE, tid, rv);
    }
    __syncthreads();
  }
  // (the loop ended on a barrier: the K/V stages are free for the per-wave staging tiles)
  float* wsum = (float*)(smem + 16384);  // [4 waves][64], behind the staging tiles
  {
    const long drow0 = ROWS ? (long)chunk_row(a.q_rows, b, q0 + wave * 32) * a.ldq + h * 64 : (long)b * a.bsq + (long)(q0 + wave * 32) * a.ldq + h * 64;
    store_rows_bf16(smem + wave * 4096, dqT, SCALE, a.dq + drow0, a.ldq, w_act ? a.Tq - (q0 + wave * 32) : 0, lane,
                    a.dq_colsum ? wsum + wave * 64 : nullptr);
  }
  if (a.dq_colsum) {  // one partial row per workgroup: colsum_scratch[(b, query block)][h*64 + c], reduced by the launcher
    __syncthreads();
    if (tid < 64)
      a.colsum_scratch[((long)(b * nqb + (bid - bh * nqb)) * a.H + h) * 64 + tid] =
          (wsum[tid] + wsum[64 + tid]) + (wsum[128 + tid] + wsum[192 + tid]);
  }
}

// ------------------------------------------------------------------------------------------------------------
// Ping-pong kernels for the UNMASKED case (no causal mask, no per-sequence key length: encoder self-attention and
// cross-attention, 98 % of the attention time of a training step).
//
// What the measurements on the kernels above say (scripts/attn_kernel_times.py under rocprofv3, in-kernel s_memtime stamps;
// profiles/r03_attention_pingpong.txt):
//  * they issue the way they are written -- fragment reads -> wait -> S/dP MFMAs -> softmax VALU -> second-stage MFMAs -- and with
//    two waves per SIMD the matrix pipe idles through every LDS round trip and VALU block (MFMA-busy 0.33);
//  * interleaving everything in ONE wave's stream (reads of step i+2, MFMAs of step i+1, VALU of step i) does not fix it: an MFMA
//    whose operand comes from a ds_read issued even a full step earlier in the same wave still stalls (S/dP MFMAs alone 151 us,
//    with dependent fragment reads 277 us, with the same reads and nothing depending on them 168 us);
//  * v_pk_mul_f32 / v_pk_add_f32 beside MFMAs cost ~16 cycles each where two scalar ops cost ~0 (the compiler SLP-packs
//    adjacent fp32 multiplies: this file is built with -fno-slp-vectorize).
// So: 8 waves, two per SIMD, alternating roles between workgroup barriers -- one COMPUTEs (12 MFMAs with the softmax VALU in
// their shadow, every operand already in registers, nothing inside the segment depends on anything else inside it) while its
// partner LOADs (fragment reads for its next segment ending in lgkmcnt(0), DMA issue).  Waves 4-7 run one barrier behind
// waves 0-3.  K/V tiles move global -> LDS by DMA (buffer_load ... lds: no staging registers, no commit stores) two tiles ahead
// through a 4-stage ring; rows past the end read as zeros (buffer range check), which also makes the tail steps (the loop runs
// in groups of four tiles) contribute nothing.  No masks, no branches in the loop.
//   LOAD(j)   : operand fragments of step j+1, transposed fragments of step j-1; on even j the DMA of tile j/2+2 and the wait for
//               this wave's pieces of tile j/2+1
//   COMPUTE(j): first-stage MFMAs of step j+1, softmax VALU of step j, second-stage MFMAs of step j-1
// ROWS: the query side lives in chunked token rows and is limited to q_span (kernels.h): the cross-attention of a span-limited step
template <bool ROWS>
__global__ __launch_bounds__(512, 1) void attn_bwd_dq_pp_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(1024))) char smem[PNS * PSTG];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), hh = lane >> 5;
  const int grp = wave >> 2;
  const unsigned smem_a = (unsigned)(size_t)smem;
  const int nqb = (a.Tq + 255) >> 8;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = bid / nqb;
  const int b = bh / a.H, h = bh - b * a.H;
  const int q0 = (bid - bh * nqb) * 256;
  const int myq = q0 + wave * 32 + (lane & 31);
  const int q_lim = (ROWS && a.q_span) ? min(a.q_span[b], a.Tq) : a.Tq;
  if (ROWS && q0 >= q_lim) {  // (block-uniform) no gradient in this block: only its two partial bias-gradient rows must read as zeros
    if (a.dq_colsum && tid < 128) {
      const int nqb128 = (a.Tq + 127) >> 7, blk128 = (bid - bh * nqb) * 2 + (tid >> 6);
      if (blk128 < nqb128) a.colsum_scratch[((long)(b * nqb128 + blk128) * a.H + h) * 64 + (tid & 63)] = 0.f;
    }
    return;
  }
  const bool w_act = !ROWS || q0 + wave * 32 < q_lim;  // (wave-uniform)
  const int myq_c = !w_act ? q0 + (lane & 31) : (myq < a.Tq ? myq : a.Tq - 1);  // an inactive wave re-reads rows of the first chunk

  const long qrow = ROWS ? (long)chunk_row(a.q_rows, b, myq_c) : 0;
  const bf16_t* qp = ROWS ? a.q + qrow * a.ldq + h * 64 : a.q + (long)b * a.bsq + (long)myq_c * a.ldq + h * 64;
  const long orow_off = ROWS ? qrow * a.ldo + h * 64 : (long)b * a.bso + (long)myq_c * a.ldo + h * 64;
  const bf16_t* dop = a.d_o + orow_off;
  const bf16_t* op = a.o + orow_off;
  const bf16_t* olop = a.o_lo ? a.o_lo + orow_off : nullptr;
  bf16x8_t qf[4], dof[4];
  float dpart = 0.f;
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) {
    qf[ds] = ld_frag_global(qp + ds * 16 + hh * 8);
    u32x4_t d4 = *(const u32x4_t*)(dop + ds * 16 + hh * 8);
    if (ROWS && !w_act) d4 = u32x4_t{0u, 0u, 0u, 0u};
    dof[ds] = __builtin_bit_cast(bf16x8_t, d4);
    const u32x4_t o4 = *(const u32x4_t*)(op + ds * 16 + hh * 8);
    if (olop) {
      const u32x4_t r4 = *(const u32x4_t*)(olop + ds * 16 + hh * 8);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        dpart += bf_lo(d4[i]) * (bf_lo(o4[i]) + bf_lo(r4[i])) + bf_hi(d4[i]) * (bf_hi(o4[i]) + bf_hi(r4[i]));
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) dpart += bf_lo(d4[i]) * bf_lo(o4[i]) + bf_hi(d4[i]) * bf_hi(o4[i]);
    }
  }
  const float delta = dpart + __shfl_xor(dpart, 32, 64);
  const long stat_idx = ((long)b * a.H + h) * a.Tq + myq_c;
  if (hh == 0 && myq < a.Tq && w_act) a.delta[stat_idx] = delta;
  const float lse2 = a.lse[stat_idx] * LOG2E;
  asm volatile("" ::"v"(lse2), "v"(delta), "v"(qf[3]), "v"(dof[3]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the loop counts its own DMA pieces
  bool blk_nz = true;
  if (a.qtile_flags) {  // which 64-query tiles of d_o hold anything: recorded for the dK/dV kernel; a block of zeros has dQ = 0
    int* nzs = (int*)smem;  // (the ring is not in use yet; a barrier below releases these bytes before the first DMA piece)
    bool nz = false;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
      const u32x4_t d4 = __builtin_bit_cast(u32x4_t, dof[ds]);
      nz = nz || ((d4[0] | d4[1] | d4[2] | d4[3]) & 0x7fff7fffu) != 0u;
    }
    const bool wnz = __any(nz);
    if (lane == 0) nzs[wave] = wnz ? 1 : 0;
    __syncthreads();
    if (tid < 4) {
      const int ntq = (a.Tq + 63) >> 6, tq = (q0 >> 6) + tid;
      if (tq < ntq) a.qtile_flags[((long)b * a.H + h) * ntq + tq] = nzs[2 * tid] | nzs[2 * tid + 1];
    }
    blk_nz = (nzs[0] | nzs[1] | nzs[2] | nzs[3] | nzs[4] | nzs[5] | nzs[6] | nzs[7]) != 0;
    __syncthreads();
  }

  const int ntiles = (a.Tk + 63) >> 6;
  const int ngroups = (ntiles + PNS - 1) / PNS;
  const PipeSrc1 ksrc = pipe_src1(a.k + (long)b * a.bsk + h * 64, a.ldk, a.Tk, wave, lane);
  const PipeSrc1 vsrc = pipe_src1(a.v + (long)b * a.bsv + h * 64, a.ldv, a.Tk, wave, lane);
  auto dma_tile = [&](int tile, int stage) {
    const unsigned dst = smem_a + stage * PSTG + wave * 1024;
    glds16_asm(ksrc.rs, dst, ksrc.voff + (unsigned)tile * ksrc.tile_bytes);
    glds16_asm(vsrc.rs, dst + TILE, vsrc.voff + (unsigned)tile * vsrc.tile_bytes);
  };

  int ar[4], ac[2][2];
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) ar[ds] = tile_addr(lane & 31, ds * 2 + hh);
  {
    const int G = lane >> 4, i = lane & 15;
    const int row = 4 * (G >> 1) + (i >> 2);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const int col = dt * 32 + (G & 1) * 16 + (i & 3) * 4;
      ac[dt][0] = tile_addr(row, col >> 3) + (col & 7) * 2;
      ac[dt][1] = tile_addr(row + 8, col >> 3) + (col & 7) * 2;
    }
  }
  auto rd_rows = [&](int off, int ds) { return *(const bf16x8_t*)(smem + ar[ds] + off); };
  auto rd_tr = [&](int off, int dt, int half) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(smem + ac[dt][half] + off));
  };

  bf16x8_t Ak[4], Av[4];
  s16x4_t kc[2][2][2];  // [u][dt][half]
  f32x16_t dqT[2], S[2], dP[2], ND;
  u32x4_t w[2][2];  // [step parity][u]: bf16 dS^T, the B operand of the dQ MFMAs one segment later
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      dqT[i][r] = 0.f;
      S[i][r] = 0.f;
    }
#pragma unroll
  for (int r = 0; r < 16; ++r) ND[r] = -delta;  // dP accumulators start at -delta (a lane's 16 rows all belong to its query)
  dP[0] = ND;
  dP[1] = ND;
  const float c1 = SCALE * LOG2E, nl1 = -lse2;

  // LOAD(j), j = 2t + kt, stage st = t & 3: K/V fragments of step j+1, transposed-K fragments of step j-1
  auto load_seg = [&](int t, int st, int kt) {
    if (kt == 0) dma_tile(t + 2, (st + 2) & 3);
    const int a_st = kt ? ((st + 1) & 3) : st;                    // step j+1 lives in the next tile when kt == 1
    const int c_st = kt ? st : ((st + 3) & 3);                    // step j-1 lives in the previous tile when kt == 0
    const int a_off = a_st * PSTG + (kt ^ 1) * 32 * 128;
    const int c_off = c_st * PSTG + (kt ^ 1) * 32 * 128;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
      Ak[ds] = rd_rows(a_off, ds);
      Av[ds] = rd_rows(a_off + TILE, ds);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        kc[u][dt][0] = rd_tr(c_off + u * 16 * 128, dt, 0);
        kc[u][dt][1] = rd_tr(c_off + u * 16 * 128, dt, 1);
      }
    if (kt == 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");  // this wave's pieces of tile t+1 have landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  // COMPUTE(j): S/dP MFMAs of step j+1, softmax VALU of step j, dQ MFMAs of step j-1 -- nothing here depends on anything else here
  auto compute_seg = [&](int kt) {
    const int n = kt ^ 1;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
      if (ds == 0) {
        f32x16_t z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        S[n] = MFMA(Ak[0], qf[0], z);
        dP[n] = MFMA(Av[0], dof[0], ND);
      } else {
        S[n] = MFMA(Ak[ds], qf[ds], S[n]);
        dP[n] = MFMA(Av[ds], dof[ds], dP[n]);
      }
      if (ds < 2) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const s16x8_t kv8 = __builtin_shufflevector(kc[ds][dt][0], kc[ds][dt][1], 0, 1, 2, 3, 4, 5, 6, 7);
          dqT[dt] = MFMA(__builtin_bit_cast(bf16x8_t, kv8), __builtin_bit_cast(bf16x8_t, w[n][ds]), dqT[dt]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float p0 = __builtin_amdgcn_exp2f(fmaf(S[kt][2 * j], c1, nl1));
      const float p1 = __builtin_amdgcn_exp2f(fmaf(S[kt][2 * j + 1], c1, nl1));
      w[kt][j >> 2][j & 3] = pack_bf2(p0 * dP[kt][2 * j], p1 * dP[kt][2 * j + 1]);
    }
    // schedule request for this barrier-to-barrier region: 12 MFMA slots, the 56 VALU spread over them
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
    }
  };

  if (blk_nz) {  // (workgroup-uniform)
  // prologue: tiles 0, 1 in flight; S/dP of step 0; LOAD(0) with nothing for the dQ MFMAs of "step -1" to add
  dma_tile(0, 0);
  dma_tile(1, 1);
  asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  ATTN_BARRIER();
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) {
    Ak[ds] = rd_rows(0, ds);
    Av[ds] = rd_rows(TILE, ds);
  }
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) {
    S[0] = MFMA(Ak[ds], qf[ds], S[0]);
    dP[0] = MFMA(Av[ds], dof[ds], dP[0]);
  }
  ATTN_FENCE();
  load_seg(0, 0, 0);  // (its transposed-K reads hit stage 3, which nothing has written: overwritten with zeros below)
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    w[1][u] = u32x4_t{0u, 0u, 0u, 0u};
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      kc[u][dt][0] = s16x4_t{0, 0, 0, 0};
      kc[u][dt][1] = s16x4_t{0, 0, 0, 0};
    }
  }
  ATTN_BARRIER();
  if (grp == 1) ATTN_BARRIER();  // the upper half trails by one barrier from here on

  for (int g = 0; g < ngroups; ++g) {
#pragma unroll
    for (int st = 0; st < PNS; ++st) {
      const int t = g * PNS + st;
      if (w_act) compute_seg(0);  // (a wave whose 32 queries lie past the span has nothing to compute: it keeps its DMA / barrier duties only)
      ATTN_BARRIER();
      load_seg(t, st, 1);
      ATTN_BARRIER();
      if (w_act) compute_seg(1);
      ATTN_BARRIER();
      load_seg(t + 1, (st + 1) & 3, 0);
      ATTN_BARRIER();
    }
  }
  // the dQ MFMAs of the last step (its transposed-K fragments came with the last LOAD)
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const s16x8_t kv8 = __builtin_shufflevector(kc[u][dt][0], kc[u][dt][1], 0, 1, 2, 3, 4, 5, 6, 7);
      dqT[dt] = MFMA(__builtin_bit_cast(bf16x8_t, kv8), __builtin_bit_cast(bf16x8_t, w[1][u]), dqT[dt]);
    }
  if (grp == 0) ATTN_BARRIER();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tail DMA pieces (zeros) must not land on the staging tiles below
  }
  __syncthreads();
  float* wsum = (float*)(smem + 32768);  // [8 waves][64], behind the staging tiles
  {
    const long drow0 = ROWS ? (long)chunk_row(a.q_rows, b, q0 + wave * 32) * a.ldq + h * 64 : (long)b * a.bsq + (long)(q0 + wave * 32) * a.ldq + h * 64;
    store_rows_bf16(smem + wave * 4096, dqT, SCALE, a.dq + drow0, a.ldq, w_act ? a.Tq - (q0 + wave * 32) : 0, lane,
                    a.dq_colsum ? wsum + wave * 64 : nullptr);
  }
  if (a.dq_colsum) {  // two partial rows per workgroup, in the 128-query row numbering of the colsum scratch
    __syncthreads();
    if (tid < 128) {
      const int half = tid >> 6, c = tid & 63;
      const int nqb128 = (a.Tq + 127) >> 7;
      const int blk128 = (bid - bh * nqb) * 2 + half;
      if (blk128 < nqb128)
        a.colsum_scratch[((long)(b * nqb128 + blk128) * a.H + h) * 64 + c] =
            (wsum[half * 256 + c] + wsum[half * 256 + 64 + c]) + (wsum[half * 256 + 128 + c] + wsum[half * 256 + 192 + c]);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// dK[k][:] = scale * sum_q dS[q][k] Q[q][:],  dV[k][:] = sum_q P[q][k] dO[q][:]
// index of the last 64-query tile whose d_o the dQ kernel found non-zero (-1: none); every wave computes it for itself
__device__ __forceinline__ int last_nonzero_qtile(const int32_t* flags, int n, int lane) {
  int last = -1;
  for (int i = lane; i < n; i += 64)
    if (flags[i] != 0) last = i;
  return __builtin_amdgcn_readfirstlane(wave_max_i(last));
}

template <bool CAUSAL, bool ROWS>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_kernel(AttnArgs a) {
  // stage s: Q tile, dO tile, then lse[64] | delta[64] floats
  constexpr int STAGE = 2 * TILE + 512;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
  const int nkb = (a.Tk + 127) >> 7;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = bid / nkb;
  const int b = bh / a.H, h = bh - b * a.H;
  const int k0 = (bid - bh * nkb) * 128;
  const int mykey = k0 + wave * 32 + (lane & 31);
  const bool krows = ROWS && a.k_rows != nullptr;  // (block-uniform) chunked key rows: decoder self-attention
  // span-limited backward: query positions >= q_span[b] hold no gradient (their d_o rows are never read); with chunked key rows the
  // same positions are keys no supervised query sees -- their dk / dv rows lie outside the active rows and are not written
  const int q_lim = (ROWS && a.q_span) ? min(a.q_span[b], a.Tq) : a.Tq;
  const long dv_scratch_off0 = (long)a.B * ((a.Tq + 127) >> 7) * a.H * 64;
  if (krows && a.q_span && k0 >= q_lim) {
    if (a.dv_colsum && tid < 64) a.colsum_scratch[dv_scratch_off0 + ((long)(b * nkb + (bid - bh * nkb)) * a.H + h) * 64 + tid] = 0.f;
    return;
  }
  const bool w_store = !(krows && a.q_span) || k0 + wave * 32 < q_lim;  // (wave-uniform)
  const int mykey_c = !w_store ? k0 + (lane & 31) : (mykey < a.Tk ? mykey : a.Tk - 1);

  int kv_len = a.kv_len ? a.kv_len[b] : a.Tk;
  kv_len = kv_len < a.Tk ? kv_len : a.Tk;

  const bf16_t* kp = krows ? a.k + (long)chunk_row(a.k_rows, b, mykey_c) * a.ldk + h * 64 : a.k + (long)b * a.bsk + (long)mykey_c * a.ldk + h * 64;
  const bf16_t* vp = krows ? a.v + (long)chunk_row(a.k_rows, b, mykey_c) * a.ldv + h * 64 : a.v + (long)b * a.bsv + (long)mykey_c * a.ldv + h * 64;
  bf16x8_t kf[4], vf[4];
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) {
    kf[ds] = ld_frag_global(kp + ds * 16 + hh * 8);
    vf[ds] = ld_frag_global(vp + ds * 16 + hh * 8);
  }
  f32x16_t dkT[2], dvT[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      dkT[i][r] = 0.f;
      dvT[i][r] = 0.f;
    }

  const TileSrc qsrc = ROWS ? tile_src(a.q + h * 64, a.ldq, a.B * a.Tq, tid) : tile_src(a.q + (long)b * a.bsq + h * 64, a.ldq, a.Tq, tid);
  const TileSrc dosrc = ROWS ? tile_src(a.d_o + h * 64, a.ldo, a.B * a.Tq, tid) : tile_src(a.d_o + (long)b * a.bso + h * 64, a.ldo, a.Tq, tid);
  const float* lse_b = a.lse + ((long)b * a.H + h) * a.Tq;
  const float* delta_b = a.delta + ((long)b * a.H + h) * a.Tq;

  int nqt = (a.Tq + 63) >> 6;
  if (a.qtile_flags) nqt = last_nonzero_qtile(a.qtile_flags + ((long)b * a.H + h) * nqt, nqt, lane) + 1;  // d_o is zero past it
  if (ROWS && (q_lim >> 6) < nqt) nqt = q_lim >> 6;
  const int t_begin = (CAUSAL && k0 < kv_len) ? (k0 >> 6) : 0;
  const bool any = k0 < kv_len && t_begin < nqt;  // otherwise every P (or every d_o row) is zero: fall through and write zeros

  u32x4_t rq[2], rd[2];
  float rstat = 0.f;
  auto issue = [&](int t) {
    const int qr0 = ROWS ? __builtin_amdgcn_readfirstlane(a.q_rows[b * OASR_ROWTAB + t]) : t * 64;
    tile_issue(qsrc, qr0, rq);
    tile_issue(dosrc, qr0, rd);
    if (tid < 128) {
      int qi = t * 64 + (tid & 63);
      qi = qi < a.Tq ? qi : a.Tq - 1;
      rstat = tid < 64 ? lse_b[qi] * LOG2E : delta_b[qi];
    }
  };
  auto commit = [&](char* st) {
    tile_commit(st, tid, rq);
    tile_commit(st + TILE, tid, rd);
    if (tid < 128) ((float*)(st + 2 * TILE))[tid] = rstat;
  };
  // dv partial rows live behind the dq ones: [B * ceil(Tq/128)][H*64] then [B * ceil(Tk/128)][H*64]
  const long dv_scratch_off = dv_scratch_off0;
  if (!any) {  // every key of this block is padding: dK = dV = 0 (uniform early exit, before any load is issued)
    if (a.dv_colsum && tid < 64) a.colsum_scratch[dv_scratch_off + ((long)(b * nkb + (bid - bh * nkb)) * a.H + h) * 64 + tid] = 0.f;
    if (mykey < a.Tk && w_store) {
      bf16_t* dkp0 = krows ? a.dk + (long)chunk_row(a.k_rows, b, mykey) * a.ldk + h * 64 : a.dk + (long)b * a.bsk + (long)mykey * a.ldk + h * 64;
      bf16_t* dvp0 = krows ? a.dv + (long)chunk_row(a.k_rows, b, mykey) * a.ldv + h * 64 : a.dv + (long)b * a.bsv + (long)mykey * a.ldv + h * 64;
      const u32x4_t z = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        *(u32x4_t*)(dkp0 + hh * 32 + i * 8) = z;
        *(u32x4_t*)(dvp0 + hh * 32 + i * 8) = z;
      }
    }
    return;
  }
  // unconditional prologue: see attn_fwd_kernel (a guarded one makes hipcc re-wait the prefetch at the loop top)
  issue(t_begin);
  commit(smem);
  __syncthreads();
  for (int t = t_begin; t < nqt; ++t) {
    const int cur = (t - t_begin) & 1;
    const char* st = smem + cur * STAGE;
    const char* qb = st;
    const char* dob = st + TILE;
    const float* stat = (const float*)(st + 2 * TILE);
    const bool more = t + 1 < nqt;
    if (more) issue(t + 1);
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      f32x16_t s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = 0.f;
        dp[r] = 0.f;
      }
#pragma unroll
      for (int ds = 0; ds < 4; ++ds) {
        s = MFMA(frag_rows(qb, qt * 32, ds, lane), kf[ds], s);
        dp = MFMA(frag_rows(dob, qt * 32, ds, lane), vf[ds], dp);
      }
      const int key_hi = k0 + wave * 32 + 31;  // largest key of this wave
      const bool full = (t * 64 + qt * 32 + 32 <= a.Tq) && (key_hi < kv_len) && (!CAUSAL || key_hi <= t * 64 + qt * 32);
      if (full) {  // wave-uniform: no masks, pairs of queries per packed instruction
        const f32x2_t c2 = {SCALE * LOG2E, SCALE * LOG2E};
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int ql = qt * 32 + 8 * g4 + 4 * hh;
          const f32x4_t l4 = *(const f32x4_t*)(stat + ql);
          const f32x4_t d4 = *(const f32x4_t*)(stat + 64 + ql);
#pragma unroll
          for (int i = 0; i < 4; i += 2) {
            const int r = 4 * g4 + i;
            const f32x2_t s2 = {s[r], s[r + 1]}, dp2 = {dp[r], dp[r + 1]}, nl2 = {-l4[i], -l4[i + 1]}, nd2 = {-d4[i], -d4[i + 1]};
            const f32x2_t p2 = pk_exp2(pk_fma(s2, c2, nl2));
            const f32x2_t ds2 = p2 * (dp2 + nd2);
            s[r] = p2[0];
            s[r + 1] = p2[1];
            dp[r] = ds2[0];
            dp[r + 1] = ds2[1];
          }
        }
      } else {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int ql = qt * 32 + 8 * g4 + 4 * hh;
          const f32x4_t l4 = *(const f32x4_t*)(stat + ql);
          const f32x4_t d4 = *(const f32x4_t*)(stat + 64 + ql);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 4 * g4 + i;
            const int qg = t * 64 + ql + i;
            const bool ok = (qg < a.Tq) && (mykey < kv_len) && (!CAUSAL || mykey <= qg);
            const float p = ok ? __builtin_amdgcn_exp2f(fmaf(s[r], SCALE * LOG2E, -l4[i])) : 0.f;
            s[r] = p;
            dp[r] = p * (dp[r] - d4[i]);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const bf16x8_t pf = pack_half(s, u);
        const bf16x8_t dsf = pack_half(dp, u);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          dvT[dt] = MFMA(frag_cols(dob, qt * 32 + 16 * u, dt * 32, lane), pf, dvT[dt]);
          dkT[dt] = MFMA(frag_cols(qb, qt * 32 + 16 * u, dt * 32, lane), dsf, dkT[dt]);
        }
      }
    }
    if (more) commit(smem + (cur ^ 1) * STAGE);
    __syncthreads();
  }
  {
    char* stg = smem + wave * 4096;  // the loop ended on a barrier: both stages are free
    const int rows_valid = w_store ? a.Tk - (k0 + wave * 32) : 0;
    const long krow0 = krows ? (long)chunk_row(a.k_rows, b, k0 + wave * 32) : 0;
    store_rows_bf16(stg, dkT, SCALE, krows ? a.dk + krow0 * a.ldk + h * 64 : a.dk + (long)b * a.bsk + (long)(k0 + wave * 32) * a.ldk + h * 64,
                    a.ldk, rows_valid, lane);
    float* wsum = (float*)(smem + 16384);
    store_rows_bf16(stg, dvT, 1.0f, krows ? a.dv + krow0 * a.ldv + h * 64 : a.dv + (long)b * a.bsv + (long)(k0 + wave * 32) * a.ldv + h * 64,
                    a.ldv, rows_valid, lane, a.dv_colsum ? wsum + wave * 64 : nullptr);
    if (a.dv_colsum) {
      __syncthreads();
      if (tid < 64)
        a.colsum_scratch[dv_scratch_off + ((long)(b * nkb + (bid - bh * nkb)) * a.H + h) * 64 + tid] =
            (wsum[tid] + wsum[64 + tid]) + (wsum[128 + tid] + wsum[192 + tid]);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Ping-pong dK/dV kernel for the unmasked case: 8 waves x 32 keys, Q / dO tiles (64 queries) + their lse / delta rows stream
// through the ring.  The key operands are held as K' = -K/8 and V' = -V (exact in bf16) and the first-stage accumulators START at
// lse[q] / delta[q] (read from LDS straight into the accumulator registers by the LOAD segment), so that
//   S_acc = lse - scale * S      -> P = exp2(-log2e * S_acc)          (one multiply + one exponential per score)
//   dP_acc = delta - dP          -> -dS = P * dP_acc                  (one multiply; the sign rides in dK's store scale)
// with no per-row statistics in registers.
constexpr int KSTAT = 2048;            // LDS: [4 stages x (lse[64] | delta[64])] then the 4 tile stages
constexpr int KLDS = KSTAT + PNS * PSTG;

__device__ __forceinline__ void glds4_asm(const u32x4_t rs, unsigned lds_addr, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs) : "memory");
}

// ROWS: the query side (q, d_o) lives in chunked token rows and is limited to q_span (kernels.h)
template <bool ROWS>
__global__ __launch_bounds__(512, 1) void attn_bwd_dkdv_pp_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), hh = lane >> 5;
  const int grp = wave >> 2;
  const unsigned smem_a = (unsigned)(size_t)smem;
  const int nkb = (a.Tk + 255) >> 8;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = bid / nkb;
  const int b = bh / a.H, h = bh - b * a.H;
  const int k0 = (bid - bh * nkb) * 256;
  const int mykey = k0 + wave * 32 + (lane & 31);
  const int mykey_c = mykey < a.Tk ? mykey : a.Tk - 1;

  const bf16_t* kp = a.k + (long)b * a.bsk + (long)mykey_c * a.ldk + h * 64;
  const bf16_t* vp = a.v + (long)b * a.bsv + (long)mykey_c * a.ldv + h * 64;
  bf16x8_t kf[4], vf[4];
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) {
    const u32x4_t k4 = *(const u32x4_t*)(kp + ds * 16 + hh * 8);
    const u32x4_t v4 = *(const u32x4_t*)(vp + ds * 16 + hh * 8);
    u32x4_t kn, vn;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      kn[i] = pack_bf2(bf_lo(k4[i]) * -0.125f, bf_hi(k4[i]) * -0.125f);  // exact: a power of two
      vn[i] = v4[i] ^ 0x80008000u;
    }
    kf[ds] = __builtin_bit_cast(bf16x8_t, kn);
    vf[ds] = __builtin_bit_cast(bf16x8_t, vn);
  }
  asm volatile("" ::"v"(kf[3]), "v"(vf[3]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the loop counts its own DMA pieces

  int ntiles = (a.Tq + 63) >> 6;
  if (a.qtile_flags) ntiles = last_nonzero_qtile(a.qtile_flags + ((long)b * a.H + h) * ntiles, ntiles, lane) + 1;  // d_o is zero past it
  if (ROWS && a.q_span) ntiles = min(ntiles, min(a.q_span[b], a.Tq) >> 6);  // d_o rows past the span are not even written
  const int ngroups = (ntiles + PNS - 1) / PNS;
  const PipeSrc1 qsrc = ROWS ? pipe_src1(a.q + h * 64, a.ldq, a.B * a.Tq, wave, lane) : pipe_src1(a.q + (long)b * a.bsq + h * 64, a.ldq, a.Tq, wave, lane);
  const PipeSrc1 dosrc = ROWS ? pipe_src1(a.d_o + h * 64, a.ldo, a.B * a.Tq, wave, lane) : pipe_src1(a.d_o + (long)b * a.bso + h * 64, a.ldo, a.Tq, wave, lane);
  // ROWS: the first token row of every 64-query tile, tiles at or past ntiles mapped far outside the tensors (the DMA then writes
  // zeros, exactly what the plain layout gets from its range check for the tail steps of the last group of four)
  int qrow[8];
  if (ROWS) {
#pragma unroll
    for (int i = 0; i < 8; ++i) qrow[i] = i < ntiles ? __builtin_amdgcn_readfirstlane(a.q_rows[b * OASR_ROWTAB + i]) : 0x3fffffff;
  }
  auto qrow_of = [&](int tile) {  // (scalar select chain: `tile` is wave-uniform, no register indexing)
    int r = 0x3fffffff;
#pragma unroll
    for (int i = 0; i < 8; ++i) r = tile == i ? qrow[i] : r;
    return r;
  };
  // lse / delta rows of a tile: 2 x 256 bytes; even waves bring lse, odd waves delta (four identical copies each: one DMA per
  // wave keeps every wave's vmcnt arithmetic the same)
  u32x4_t srs;
  {
    const unsigned long addr = (unsigned long)(((wave & 1) ? a.delta : a.lse) + ((long)b * a.H + h) * a.Tq);
    srs[0] = (unsigned)addr;
    srs[1] = (unsigned)(addr >> 32) & 0xffffu;
    srs[2] = (unsigned)(a.Tq * 4);
    srs[3] = 0x00020000u;
  }
  auto dma_tile = [&](int tile, int stage) {
    const unsigned dst = smem_a + KSTAT + stage * PSTG + wave * 1024;
    if (ROWS) {
      // a sentinel row times the row pitch wraps in 32 bits: clamp the byte offset to "past the end" instead
      const int r0 = qrow_of(tile);
      const unsigned qo = r0 == 0x3fffffff ? 0x7fffff00u : (unsigned)r0 * (unsigned)(a.ldq * 2);
      const unsigned oo = r0 == 0x3fffffff ? 0x7fffff00u : (unsigned)r0 * (unsigned)(a.ldo * 2);
      glds16_asm(qsrc.rs, dst, r0 == 0x3fffffff ? qo : qsrc.voff + qo);
      glds16_asm(dosrc.rs, dst + TILE, r0 == 0x3fffffff ? oo : dosrc.voff + oo);
      glds4_asm(srs, smem_a + stage * 512 + (wave & 1) * 256, r0 == 0x3fffffff ? 0x7fffff00u : (unsigned)(lane * 4 + tile * 256));
    } else {
      glds16_asm(qsrc.rs, dst, qsrc.voff + (unsigned)tile * qsrc.tile_bytes);
      glds16_asm(dosrc.rs, dst + TILE, dosrc.voff + (unsigned)tile * dosrc.tile_bytes);
      glds4_asm(srs, smem_a + stage * 512 + (wave & 1) * 256, (unsigned)(lane * 4 + tile * 256));
    }
  };

  int ar[4], ac[2][2];
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) ar[ds] = KSTAT + tile_addr(lane & 31, ds * 2 + hh);
  {
    const int G = lane >> 4, i = lane & 15;
    const int row = 4 * (G >> 1) + (i >> 2);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const int col = dt * 32 + (G & 1) * 16 + (i & 3) * 4;
      ac[dt][0] = KSTAT + tile_addr(row, col >> 3) + (col & 7) * 2;
      ac[dt][1] = KSTAT + tile_addr(row + 8, col >> 3) + (col & 7) * 2;
    }
  }
  const int astat = hh * 16;  // byte offset of this lane's first row group in a 32-query statistics row
  auto rd_rows = [&](int off, int ds) { return *(const bf16x8_t*)(smem + ar[ds] + off); };
  auto rd_tr = [&](int off, int dt, int half) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(smem + ac[dt][half] + off));
  };

  bf16x8_t Aq[4], Ado[4];
  s16x4_t cq[2][2][2], cdo[2][2][2];  // [u][dt][half]
  f32x16_t dkT[2], dvT[2], S[2], dP[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      dkT[i][r] = 0.f;
      dvT[i][r] = 0.f;
    }

  // LOAD(j), j = 2t + kt, stage st = t & 3: Q / dO fragments + accumulator seeds of step j+1, transposed fragments of step j.
  // (The second stage runs in the SAME segment as its softmax here -- one pipeline stage less than the dQ kernel: a third stage
  // would keep a second set of bf16 P / dS tiles alive, and 96 accumulator + key registers already leave no room for it.)
  auto load_seg = [&](int t, int st, int kt) {
    if (kt == 0) dma_tile(t + 3, (st + 3) & 3);
    const int a_st = kt ? ((st + 1) & 3) : st;
    const int a_off = a_st * PSTG + (kt ^ 1) * 32 * 128;
    const int c_off = st * PSTG + kt * 32 * 128;
    const int s_off = a_st * 512 + (kt ^ 1) * 128 + astat;
    const int n = kt ^ 1;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
      Aq[ds] = rd_rows(a_off, ds);
      Ado[ds] = rd_rows(a_off + TILE, ds);
    }
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const f32x4_t l4 = *(const f32x4_t*)(smem + s_off + g4 * 32);
      const f32x4_t d4 = *(const f32x4_t*)(smem + s_off + 256 + g4 * 32);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        S[n][4 * g4 + i] = l4[i];
        dP[n][4 * g4 + i] = d4[i];
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          cq[u][dt][hf] = rd_tr(c_off + u * 16 * 128, dt, hf);
          cdo[u][dt][hf] = rd_tr(c_off + TILE + u * 16 * 128, dt, hf);
        }
    if (kt == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // this wave's pieces of tile t+1 have landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  // COMPUTE(j): S / dP MFMAs of step j+1 (onto their seeds) with the softmax VALU of step j in their shadow, then the dV / dK
  // MFMAs of step j (the u = 0 half of P / dS is packed first, so its four MFMAs cover the rest of the VALU)
  auto compute_seg = [&](int kt) {
    const int n = kt ^ 1;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
      S[n] = MFMA(Aq[ds], kf[ds], S[n]);
      dP[n] = MFMA(Ado[ds], vf[ds], dP[n]);
    }
    u32x4_t pf[2], dsf[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float p0 = __builtin_amdgcn_exp2f(S[kt][2 * j] * -LOG2E);
      const float p1 = __builtin_amdgcn_exp2f(S[kt][2 * j + 1] * -LOG2E);
      pf[j >> 2][j & 3] = pack_bf2(p0, p1);
      dsf[j >> 2][j & 3] = pack_bf2(p0 * dP[kt][2 * j], p1 * dP[kt][2 * j + 1]);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const s16x8_t c1 = __builtin_shufflevector(cdo[u][dt][0], cdo[u][dt][1], 0, 1, 2, 3, 4, 5, 6, 7);
        const s16x8_t c2 = __builtin_shufflevector(cq[u][dt][0], cq[u][dt][1], 0, 1, 2, 3, 4, 5, 6, 7);
        dvT[dt] = MFMA(__builtin_bit_cast(bf16x8_t, c1), __builtin_bit_cast(bf16x8_t, pf[u]), dvT[dt]);
        dkT[dt] = MFMA(__builtin_bit_cast(bf16x8_t, c2), __builtin_bit_cast(bf16x8_t, dsf[u]), dkT[dt]);
      }
    // 64 VALU: 6 beside each of the 8 first-stage MFMAs (all of u = 0 and half of u = 1), the rest beside the u = 0 second-stage MFMAs
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x402, 6, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x402, 4, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
  };

  if (ntiles > 0) {  // (workgroup-uniform)
  // prologue: tiles 0..2 in flight; seeds + S / dP of step 0; LOAD(0)
  dma_tile(0, 0);
  dma_tile(1, 1);
  dma_tile(2, 2);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  ATTN_BARRIER();
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) {
    Aq[ds] = rd_rows(0, ds);
    Ado[ds] = rd_rows(TILE, ds);
  }
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    const f32x4_t l4 = *(const f32x4_t*)(smem + astat + g4 * 32);
    const f32x4_t d4 = *(const f32x4_t*)(smem + astat + 256 + g4 * 32);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      S[0][4 * g4 + i] = l4[i];
      dP[0][4 * g4 + i] = d4[i];
    }
  }
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) {
    S[0] = MFMA(Aq[ds], kf[ds], S[0]);
    dP[0] = MFMA(Ado[ds], vf[ds], dP[0]);
  }
  ATTN_FENCE();
  load_seg(0, 0, 0);
  ATTN_BARRIER();
  if (grp == 1) ATTN_BARRIER();  // the upper half trails by one barrier from here on

  for (int g = 0; g < ngroups; ++g) {
#pragma unroll
    for (int st = 0; st < PNS; ++st) {
      const int t = g * PNS + st;
      compute_seg(0);
      ATTN_BARRIER();
      load_seg(t, st, 1);
      ATTN_BARRIER();
      compute_seg(1);
      ATTN_BARRIER();
      load_seg(t + 1, (st + 1) & 3, 0);
      ATTN_BARRIER();
    }
  }
  if (grp == 0) ATTN_BARRIER();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tail DMA pieces (zeros) must not land on the staging tiles below
  }
  __syncthreads();
  {
    char* stg = smem + wave * 4096;
    const int rows_valid = a.Tk - (k0 + wave * 32);
    store_rows_bf16(stg, dkT, -SCALE, a.dk + (long)b * a.bsk + (long)(k0 + wave * 32) * a.ldk + h * 64, a.ldk, rows_valid, lane);
    float* wsum = (float*)(smem + 32768);
    store_rows_bf16(stg, dvT, 1.0f, a.dv + (long)b * a.bsv + (long)(k0 + wave * 32) * a.ldv + h * 64, a.ldv, rows_valid, lane,
                    a.dv_colsum ? wsum + wave * 64 : nullptr);
    if (a.dv_colsum) {  // two partial rows per workgroup, in the 128-key row numbering of the colsum scratch
      __syncthreads();
      if (tid < 128) {
        const int half = tid >> 6, c = tid & 63;
        const int nkb128 = (a.Tk + 127) >> 7;
        const int blk128 = (bid - bh * nkb) * 2 + half;
        const long dv_scratch_off = (long)a.B * ((a.Tq + 127) >> 7) * a.H * 64;
        if (blk128 < nkb128)
          a.colsum_scratch[dv_scratch_off + ((long)(b * nkb128 + blk128) * a.H + h) * 64 + c] =
              (wsum[half * 256 + c] + wsum[half * 256 + 64 + c]) + (wsum[half * 256 + 128 + c] + wsum[half * 256 + 192 + c]);
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------------
// One query row per (batch, head): the KV-cached decode step (Tq == 1).  The 128-row flash tile above spends a whole
// workgroup's prologue/epilogue on one row (measured 16 us per call, 27 % of a decode step); here a workgroup streams
// the K rows once (8 lanes x 16 bytes per key, 32 keys per pass), keeps the scores in LDS, and streams V once.
// Numerics as the tiled kernel: fp32 scores and normaliser, P rounded to bf16 before it multiplies V.
// Keys are taken in segments of <= dec::SEG_KEYS with their own maximum, merged in order (decode_shared.h): the arithmetic the
// one-launch step engine (decode_xcd.hip) spreads over workgroups -- the two are bit-identical; one segment (every self-attention)
// is the plain two-pass softmax.
__global__ __launch_bounds__(256) void attn_decode_kernel(AttnArgs a) {
  __shared__ float sc[dec::SEG_KEYS];
  __shared__ float red[32][64];
  __shared__ float lsum[32];
  __shared__ float wmax[4];
  const int tid = threadIdx.x, l8 = tid & 7, grp = tid >> 3, wave = tid >> 6, lane = tid & 63;
  const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
  int Tk = a.kv_len ? a.kv_len[b] : a.Tk;
  Tk = Tk < a.Tk ? Tk : a.Tk;
  float qv[8];
  dec::load_q8(*(const u32x4_t*)(a.q + (long)b * a.bsq + h * 64 + l8 * 8), qv);
  const bf16_t* kp = a.k + (long)b * a.bsk + h * 64 + l8 * 8;
  const bf16_t* vp = a.v + (long)b * a.bsv + h * 64 + l8 * 8;
  const int ns = dec::n_segments(a.Tk);  // (the segmentation follows the cache extent, not kv_len: a row's result does not depend on its neighbours)
  float m_s[dec::MAX_SEG] = {NEG, NEG}, l_s[dec::MAX_SEG] = {0.f, 0.f}, o_s[dec::MAX_SEG] = {0.f, 0.f};
#pragma unroll
  for (int sg = 0; sg < dec::MAX_SEG; ++sg) {
    if (sg >= ns) break;
    const int k0 = sg * dec::SEG_KEYS;
    int n = Tk - k0;  // keys of this segment
    n = n < 0 ? 0 : (n > dec::SEG_KEYS ? dec::SEG_KEYS : n);
    float mx = NEG;
    for (int t0 = grp; t0 < n; t0 += 32 * 8) {  // 8 independent 16-byte loads in flight per lane
      u32x4_t k4[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = t0 + 32 * u;
        k4[u] = *(const u32x4_t*)(kp + (long)(k0 + (t < n ? t : n - 1)) * a.ldk);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = t0 + 32 * u;
        const float s2 = dec::score8(qv, k4[u]);
        if (t < n) {
          if (l8 == 0) sc[t] = s2;
          mx = fmaxf(mx, s2);
        }
      }
    }
    mx = wave_max(mx);
    if (lane == 0) wmax[wave] = mx;
    __syncthreads();
    const float m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    float l = 0.f, o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = 0.f;
    for (int t0 = grp; t0 < n; t0 += 32 * 8) {
      u32x4_t v4[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = t0 + 32 * u;
        v4[u] = *(const u32x4_t*)(vp + (long)(k0 + (t < n ? t : n - 1)) * a.ldv);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = t0 + 32 * u;
        dec::accum_pv(t < n ? __builtin_amdgcn_exp2f(sc[t] - m) : 0.f, v4[u], l, o);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[grp][l8 * 8 + j] = o[j];
    if (l8 == 0) lsum[grp] = l;
    __syncthreads();
    if (tid < 64) {
      dec::reduce_groups(red, lsum, tid, o_s[sg], l_s[sg]);
      m_s[sg] = m;
    }
    __syncthreads();  // sc / red / lsum / wmax are reused by the next segment
  }
  if (tid < 64) {
    float m, lt;
    const float val = dec::merge_segments(m_s, l_s, o_s, ns, m, lt);
    const float nb = __shfl_xor(val, 1, 64);
    if ((tid & 1) == 0) *(uint32_t*)(a.o + (long)b * a.bso + h * 64 + tid) = pack_bf2(val, nb);
    if (tid == 0 && a.lse) a.lse[(long)b * a.H + h] = lt > 0.f ? (m + __builtin_amdgcn_logf(lt)) * LN2 : NEG;
  }
}

int g_attn_pingpong = 1;  // tests / A-B (oasr_attention_set_pingpong): 0 routes the unmasked case through the general kernels

int check_args(const AttnArgs& a, bool bwd) {
  OASR_REQUIRE(a.q && a.k && a.v && a.o, "attention: null pointer");
  OASR_REQUIRE(a.B > 0 && a.H > 0 && a.Tq > 0 && a.Tk > 0, "attention: bad shape");
  OASR_REQUIRE((a.ldq % 8) == 0 && (a.ldk % 8) == 0 && (a.ldv % 8) == 0 && (a.ldo % 8) == 0, "attention: strides must be multiples of 8");
  OASR_REQUIRE(!a.causal || a.Tq == a.Tk, "attention: causal needs Tq == Tk");
  if (bwd) OASR_REQUIRE(a.d_o && a.lse && a.delta && a.dq && a.dk && a.dv, "attention_bwd: null pointer");
  if (bwd) OASR_REQUIRE((!a.dq_colsum && !a.dv_colsum) || a.colsum_scratch, "attention_bwd: fused bias gradients need colsum_scratch");
  OASR_REQUIRE(!a.k_rows || a.q_rows, "attention: k_rows needs q_rows");
  OASR_REQUIRE(!a.q_span || a.q_rows, "attention: q_span needs q_rows");
  OASR_REQUIRE(!a.q_rows || ((a.Tq % 64) == 0 && a.Tq <= 64 * OASR_ROWTAB), "attention: chunked query rows need Tq %% 64 == 0 and Tq <= %d", 64 * OASR_ROWTAB);
  OASR_REQUIRE(!a.k_rows || ((a.Tk % 64) == 0 && a.Tk <= 64 * OASR_ROWTAB), "attention: chunked key rows need Tk %% 64 == 0 and Tk <= %d", 64 * OASR_ROWTAB);
  return OASR_OK;
}

}  // namespace

void attention_set_pingpong(int on) { g_attn_pingpong = on; }

int launch_attention_fwd(const AttnArgs& a, hipStream_t s) {
  int rc = check_args(a, false);
  if (rc) return rc;
  if (a.Tq == 1 && a.Tk <= 1536 && !a.o_lo && !a.q_rows) {  // decode step (causality is implied: every cached key is visible)
    hipLaunchKernelGGL(attn_decode_kernel, dim3(a.B * a.H), dim3(256), 0, s, a);
    OASR_LAUNCH_CHECK();
    return OASR_OK;
  }
  dim3 grid(cdiv(a.Tq, 128) * a.B * a.H);
  if (a.q_rows) {
    if (a.causal)
      hipLaunchKernelGGL((attn_fwd_kernel<true, true>), grid, dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL((attn_fwd_kernel<false, true>), grid, dim3(256), 0, s, a);
  } else if (a.causal) {
    hipLaunchKernelGGL((attn_fwd_kernel<true, false>), grid, dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL((attn_fwd_kernel<false, false>), grid, dim3(256), 0, s, a);
  }
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}

int launch_attention_bwd(const AttnArgs& a, hipStream_t s) {
  int rc = check_args(a, true);
  if (rc) return rc;
  dim3 gq(cdiv(a.Tq, 128) * a.B * a.H), gk(cdiv(a.Tk, 128) * a.B * a.H);
  const bool rows = a.q_rows != nullptr;
  if (a.causal) {
    if (rows) {
      hipLaunchKernelGGL((attn_bwd_dq_kernel<true, true>), gq, dim3(256), 0, s, a);
      hipLaunchKernelGGL((attn_bwd_dkdv_kernel<true, true>), gk, dim3(256), 0, s, a);
    } else {
      hipLaunchKernelGGL((attn_bwd_dq_kernel<true, false>), gq, dim3(256), 0, s, a);
      hipLaunchKernelGGL((attn_bwd_dkdv_kernel<true, false>), gk, dim3(256), 0, s, a);
    }
  } else {
    // the ping-pong kernels take the unmasked case; with chunked rows only the cross-attention form (plain key side, <= 8 query tiles)
    const bool pp = !a.kv_len && g_attn_pingpong && (!rows || (!a.k_rows && a.Tq <= 512));
    // The chunked-row case is the cross-attention backward of a span step: ~2 active 64-query tiles per sample.  Kernel choice per side measured with an
    // experiment build (bit 0 / bit 1 = dQ / dK-dV on the ping-pong kernels; (profiles/r04_cross_attention_kernel_choice.txt, whole step, same box): both 1363.5 /
    // 1367.1 ms, dQ ping-pong + dK/dV general 1358.8, dQ general + dK/dV ping-pong 1377.2, both general 1371.6.  The 0.3-0.6 % of "1" are
    // NOT taken: with the same kernels on both sides the span step reproduces the plain step's arithmetic row by row (gradients equal to
    // 1e-7, fp32 summation order only; tests/test_gpu_span.py), with a different dK/dV kernel only to bf16 rounding (2.6e-4).
    if (pp) {
      if (rows) hipLaunchKernelGGL(attn_bwd_dq_pp_kernel<true>, dim3(cdiv(a.Tq, 256) * a.B * a.H), dim3(512), 0, s, a);
      else hipLaunchKernelGGL(attn_bwd_dq_pp_kernel<false>, dim3(cdiv(a.Tq, 256) * a.B * a.H), dim3(512), 0, s, a);
      static LdsAttrOnce attr_rows, attr_plain;
      rc = rows ? ensure_dynamic_lds(attr_rows, (const void*)attn_bwd_dkdv_pp_kernel<true>, KLDS)
                : ensure_dynamic_lds(attr_plain, (const void*)attn_bwd_dkdv_pp_kernel<false>, KLDS);
      if (rc) return rc;
      if (rows) hipLaunchKernelGGL(attn_bwd_dkdv_pp_kernel<true>, dim3(cdiv(a.Tk, 256) * a.B * a.H), dim3(512), KLDS, s, a);
      else hipLaunchKernelGGL(attn_bwd_dkdv_pp_kernel<false>, dim3(cdiv(a.Tk, 256) * a.B * a.H), dim3(512), KLDS, s, a);
    } else if (rows) {
      hipLaunchKernelGGL((attn_bwd_dq_kernel<false, true>), gq, dim3(256), 0, s, a);
      hipLaunchKernelGGL((attn_bwd_dkdv_kernel<false, true>), gk, dim3(256), 0, s, a);
    } else {
      hipLaunchKernelGGL((attn_bwd_dq_kernel<false, false>), gq, dim3(256), 0, s, a);
      hipLaunchKernelGGL((attn_bwd_dkdv_kernel<false, false>), gk, dim3(256), 0, s, a);
    }
  }
  OASR_LAUNCH_CHECK();
  // fused query / value bias gradients: reduce the per-workgroup partial rows (a few MB) into the fp32 gradients
  const int d = a.H * 64;
  const long rq = (long)a.B * cdiv(a.Tq, 128), rk = (long)a.B * cdiv(a.Tk, 128);
  if (a.dq_colsum) {
    int rc2 = launch_colsum_accum(a.colsum_scratch, d, rq, d, a.dq_colsum, s);
    if (rc2) return rc2;
  }
  if (a.dv_colsum) {
    int rc2 = launch_colsum_accum(a.colsum_scratch + rq * d, d, rk, d, a.dv_colsum, s);
    if (rc2) return rc2;
  }
  return OASR_OK;
}

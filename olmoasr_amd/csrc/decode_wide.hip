// Chip-wide one-launch KV-cached decoder step for ONE sequence on gfx950 -- TextDecoder.forward for one new token, olmoasr/model.py:786-817
// with the kv_cache hooks of :925-964 (inference twin: olmoasr/inf_model.py:150-196, 320-362).
//
// Why a second one-launch engine: decode_xcd.hip keeps its team on ONE XCD (32 CUs) so that its barrier is an L2-local counter, and a team of
// 32 CUs is issue-bound at ~1.2 TB/s (profiles/r05_decode_xcd_stamps_v5_issue_bound.txt) -- about what one XCD's fabric port delivers anyway.
// A token at medium moves 960 MB, so that engine cannot go below ~0.8 ms of streaming however cheap its barriers are.  Spreading the SAME
// engine over the chip does not help (profiles/r05_decode_xcd_probe_v6.txt: 1.87 ms): its 32-row MFMA tiles give an N = d projection only 32
// workgroups' worth of work.  What the measurements ask for is the opposite trade: ALL 256 CUs, a few weight rows each, so that the
// streaming time all but disappears (115 KB per workgroup and layer) and a token costs (phases) x (one exchange across the chip).
//   * Work split: a projection's N output rows are dealt out in contiguous runs of R = 2 ceil(N / (2 nwg)) rows per workgroup (4 of mlp.2's
//     1024 at medium, 16 of mlp.0's 4096).  A (row, 512-element K span) pair is a UNIT = one 16-byte load per lane; a workgroup's units are dealt
//     round-robin to its 8 waves (1-4 units per wave and phase at medium).  One row against one token is a dot product: plain fp32 FMAs on
//     the unpacked bf16 pairs, no MFMA (at M = 1 an MFMA tile is 31/32 padding) -- 24 VALU instructions per unit.
//   * The weights of the NEXT projection phase are requested into registers (<= 8 x 16 bytes per lane) as soon as the current phase's
//     products are done: they are in flight while the workgroup publishes its rows and waits for everybody else's.
//   * Exchange: no counter.  Every workgroup owns one 32-bit flag = the number of the last phase it has completed (epoch-based, never
//     reset); a consumer polls the 1 KB flag array with ONE 16-byte agent-scope load per lane and goes on when every flag has reached the
//     phase before its own.  Rows travel through agent-scope (sc1) stores and loads only, so nothing depends on which XCD a workgroup is on.
//     A workgroup without work in a phase does not poll at all.
//   * Rows that EVERY workgroup consumes in the very next phase (block output, attention / cross output rows, cross query) skip flag + row: values and
//     the phase's epoch in ONE 16-byte store per replica, polled directly by the consumer ("packets", below).
//   * Attention: self-attention = one workgroup per head (<= 448 cached keys: 7 per 8-lane group, K and V rows requested together);
//     cross-attention = one workgroup per (head, quarter of the 1500 keys) whose K / V rows -- static data -- are requested BEFORE the poll;
//     the four partials (m, l, o[64]) of a head are merged by the consumer phase's operand stage (decode_shared.h's segment merge).
// Arithmetic: fp32 accumulation, bf16 rounding at the same points as the multi-launch step (decode_shared.h: LayerNorm output, Linear output,
// GELU input, residual sum, P before it multiplies V); the K reduction order differs (512-element spans, summed in order), so results agree
// with the other engines to fp32 rounding of the accumulation, not to the last bit (tests/test_gpu_decode_step.py states the tolerance).
#include "decode_shared.h"

namespace {

#ifndef DW_WT
#define DW_WT 576
#endif
constexpr int WT = DW_WT;  // threads per workgroup
constexpr int WW = WT / 64;
constexpr int WC = WW - 1;  // compute waves 1 .. WW-1; wave 0 is the helper: poll, operand row, epilogue, stores, flag -- it requests no weights, so nothing slow sits in front of its polls
constexpr int WNG = WT / 8;  // 8-lane groups (attention: one key per group and step)
constexpr int WMAXU = (64 + WC - 1) / WC;  // units per compute wave and phase (64 units per workgroup at most)
#ifndef DW_NS_LOG
#define DW_NS_LOG 3
#endif
constexpr int WNS_LOG = DW_NS_LOG;
constexpr int WNS = 1 << WNS_LOG;  // cross-attention key segments per head (one workgroup each)
constexpr int WSK = (448 + WNG - 1) / WNG;  // self-attention keys per 8-lane group: S_max <= 448
constexpr int WCK = (1536 / WNS + WNG - 1) / WNG;  // cross-attention keys per group and segment: Te <= 1536
constexpr int WMAXD = 1280;
constexpr int WLNC = 3;    // 16-byte chunks per lane of a LayerNorm row: d <= 1536
constexpr int WMAXL = 32;
constexpr int WMAXWG = 256;
constexpr unsigned WSPIN = 1u << 20;
constexpr int WREP = 8;      // flag replicas: 256 workgroups polling the same eight cache lines serialise on one memory channel; a replica per XCC, 4 KB apart
constexpr int WFS = 1024;    // dwords between replicas
// Rows that every workgroup consumes in the very next phase travel as PACKETS: the producing workgroup's (<= 6) bf16 values and the phase's epoch
// number in ONE 16-byte store per replica; a consumer polls the packets themselves.  No acknowledgement wait before a flag, no separate
// flag-then-row round trip.  Vectors: the block output x (mlp.2 -> the next attn_ln, or the final LayerNorm), x2 (attention output projection ->
// cross_attn_ln), x3 (cross output projection -> mlp_ln), q (cross query -> cross-attention).
constexpr int WPKV = 5, WV_X = 0, WV_X2 = 1, WV_X3 = 2, WV_Q = 3, WV_O = 4;  // (O: the self-attention output, 11 packets per head, slot h 11 + j)
constexpr int WQKS = 3 * 256;  // slots per replica of the q | k | v row of the new position: <= 3 packets per workgroup (R <= 18), slot wg 3 + j
constexpr int WPKS = 256;    // packet slots per replica: one per workgroup
constexpr int WPKR = 6;      // values per packet at most: R <= 6 for an N = d projection (the kernel is instantiated per R: 2, 4, 6)

#define WWAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

struct WLayer {  // element offsets of one decoder layer (the order of DecodeXcdArgs::layer_offsets)
  long ln1g, ln1b, wqkv, bqkv_aux, wo, bo, lncg, lncb, wcq, bcq, wco, bco, ln2g, ln2b, w1, b1, w2, b2;
};
struct WArgs {
  const bf16_t* wflat;
  const float* params;
  const float* aux;
  bf16_t* cache;
  long cache_lstride;
  bf16_t *x, *x2, *x3, *q, *o, *hg;
  float* part;      // [H * WNS][66]: m, l, o[64]
  unsigned* ctrl;   // [1] error flag  [3] XCC ids seen  [4] epoch base of this engine
  u32x4_t* pk;      // [WPKV][WREP][WPKS] packets: {bf16 x 6, epoch}
  u32x4_t* pkqkv;   // [WREP][WQKS] packets of the new position's q | k | v row
  unsigned* flagv;  // [WREP][WFS]: replica r (polled by the workgroups on XCC r) of the per-workgroup flags = last completed phase (epoch-based)
  int d, H, Te, S_max, L, pos, nwg, flags, swg;
  unsigned long long* stamps;
  const bf16_t* w_logits;  // the launch's last phase: logits = tok_emb . LayerNorm(x) (null logits_out: left to the caller)
  const float *lnf_g, *lnf_b;
  float* logits_out;
  int V;
  WLayer l0;
  long lstride, astride;
};
// ---- agent-scope accesses --------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32x4_t ld16_agent(const void* p) {
  const uint64_t a = __hip_atomic_load((const uint64_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint64_t b = __hip_atomic_load((const uint64_t*)((const char*)p + 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  u32x4_t r;
  r[0] = (unsigned)a, r[1] = (unsigned)(a >> 32), r[2] = (unsigned)b, r[3] = (unsigned)(b >> 32);
  return r;
}
__device__ __forceinline__ u32x4_t ld16_agent_off(const void* sbase, unsigned off) { return ld16_agent((const char*)sbase + (size_t)off); }
__device__ __forceinline__ unsigned ld4_agent(const void* p) { return __hip_atomic_load((const unsigned*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ldf_agent(const float* p) { return __uint_as_float(ld4_agent(p)); }
__device__ __forceinline__ void st4_agent(void* p, unsigned v) { __hip_atomic_store((unsigned*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// 16-byte agent-scope accesses as ONE instruction each (the HIP atomics stop at 8 bytes; a packet must not tear): issued from inline assembly, so the
// compiler does not count them -- every reader below waits with an explicit s_waitcnt before it looks at the data
__device__ __forceinline__ void st16_agent(u32x4_t* p, const u32x4_t& v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void ld16_agent_issue(u32x4_t& r, const u32x4_t* p) { asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r) : "v"(p) : "memory"); }
__device__ __forceinline__ float pk_value(const u32x4_t& w, int i) { return (i & 1) ? bf_hi(w[i >> 1]) : bf_lo(w[i >> 1]); }

// lane L's packet of a vector whose value i sits in lane i (bf16-rounded, 0 beyond the end): values 6 L .. 6 L + 5 + the epoch
__device__ __forceinline__ u32x4_t gather_packet(float vb, int lane, unsigned epoch) {
  float g[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) g[i] = __shfl(vb, (6 * lane + i) & 63, 64);
  u32x4_t w;
#pragma unroll
  for (int i = 0; i < 3; ++i) w[i] = pack_bf2(g[2 * i], g[2 * i + 1]);
  w[3] = epoch;
  return w;
}
// ... stored into every replica's slot (lanes < n, each its own packet): WREP store instructions
__device__ __forceinline__ void store_packets(u32x4_t* slot0, size_t rep_stride, const u32x4_t& w, bool active) {
#pragma unroll
  for (int r = 0; r < WREP; ++r)
    if (active) st16_agent(slot0 + r * rep_stride, w);
}

// ---- wave reductions on the DPP path (register to register: ~10 instructions; __shfl_xor is six dependent ds_bpermute round trips, ~700 cycles
// per sum where a phase has ~4000 to spend).  All 64 lanes must be active.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float rdl(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ float wsum(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1, 0, 3, 2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2, 3, 0, 1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror: every lane holds its 16-lane row's sum
  return (rdl(v, 0) + rdl(v, 16)) + (rdl(v, 32) + rdl(v, 48));
}
__device__ __forceinline__ float wmaxf(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
  return fmaxf(fmaxf(rdl(v, 0), rdl(v, 16)), fmaxf(rdl(v, 32), rdl(v, 48)));
}
// wave 0: every workgroup has completed phase `target` (or the wait gave up and poisoned the launch)
__device__ __forceinline__ void wide_wait(const WArgs& a, const unsigned* myflags, unsigned target, int lane) {
  unsigned spins = 0;
  for (;;) {
    bool ok = true;
    if (lane * 4 < a.nwg) {  // (nwg is a multiple of 4, the flag array 16-byte aligned)
      const u32x4_t f = ld16_agent(myflags + lane * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) ok = ok && (int)(f[i] - target) >= 0;
    }
    if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
    if (!(a.flags & 4)) __builtin_amdgcn_s_sleep(1);
    if (++spins > WSPIN || ((spins & 63) == 0 && ld4_agent(a.ctrl + 1) != 0)) {
      if (lane == 0) __hip_atomic_fetch_or(a.ctrl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // somebody never arrived: poison, do not hang
      break;
    }
  }
}

// wave 0: the packets of row vector `vec` with epoch `expect`, slot lane + 64 c in pkt[c] (slots >= nslots: not waited for)
constexpr int WPKL = WPKS / 64;  // packets per lane
__device__ __forceinline__ void pk_poll(const WArgs& a, int vec, int rep, unsigned expect, int nslots, int lane, u32x4_t (&pkt)[WPKL]) {
  const u32x4_t* src = a.pk + (size_t)(vec * WREP + rep) * WPKS;
  bool need[WPKL];
#pragma unroll
  for (int c = 0; c < WPKL; ++c) {
    need[c] = lane + 64 * c < nslots;
    pkt[c] = u32x4_t{0u, 0u, 0u, 0u};
  }
  unsigned spins = 0;
  for (;;) {
    // (every lane loads every time -- slots it does not wait for: slot 0 --: an assembly load under a branch would leave the register allocator free to
    // copy its not-yet-landed destination at the join)
#pragma unroll
    for (int c = 0; c < WPKL; ++c) ld16_agent_issue(pkt[c], src + (lane + 64 * c < nslots ? lane + 64 * c : 0));
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(pkt[0]), "+v"(pkt[1]), "+v"(pkt[2]), "+v"(pkt[3])::"memory");
    bool missing = false;
#pragma unroll
    for (int c = 0; c < WPKL; ++c)
      if (need[c]) {
        if (pkt[c][3] == expect) need[c] = false;
        else missing = true;
      }
    if (__builtin_amdgcn_ballot_w64(missing) == 0) break;
    if (!(a.flags & 4)) __builtin_amdgcn_s_sleep(1);
    if (++spins > WSPIN || ((spins & 63) == 0 && ld4_agent(a.ctrl + 1) != 0)) {
      if (lane == 0) __hip_atomic_fetch_or(a.ctrl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // a packet never came: poison, do not hang
      break;
    }
  }
}
static_assert(WPKL == 4, "pk_poll pins four packets per lane");
// wave 0: LayerNorm of a row vector held as packets (slot lane + 64 c = rows slot RD .. slot RD + RD - 1 of d) -> bf16 operand row in LDS.  Values centred once,
// two passes, the expression of dec::ln_apply8; gamma / beta from LDS (handed over by the compute waves).  RD = rows per workgroup of an N = d projection.
template <int RD>
__device__ __forceinline__ void ln_packets(const u32x4_t (&pkt)[WPKL], int d, int lane, const float* lng, const float* lnb, unsigned short* xs) {
  float v[WPKL][RD], sum = 0.f;
#pragma unroll
  for (int c = 0; c < WPKL; ++c)
#pragma unroll
    for (int i = 0; i < RD; ++i) {
      v[c][i] = (lane + 64 * c) * RD + i < d ? pk_value(pkt[c], i) : 0.f;
      sum += v[c][i];
    }
  const float mean = wsum(sum) / (float)d;
  float sq = 0.f;
#pragma unroll
  for (int c = 0; c < WPKL; ++c)
#pragma unroll
    for (int i = 0; i < RD; ++i)
      if ((lane + 64 * c) * RD + i < d) {
        v[c][i] -= mean;
        sq += v[c][i] * v[c][i];
      }
  const float rstd = rsqrtf(wsum(sq) / (float)d + 1e-5f);
#pragma unroll
  for (int c = 0; c < WPKL; ++c)
#pragma unroll
    for (int i = 0; i < RD; i += 2) {
      const int k = (lane + 64 * c) * RD + i;
      if (k < d) {  // (RD and d are even: pairs stay together; k is even: 8-byte reads)
        const oasr_f32x2_t g = *(const oasr_f32x2_t*)&lng[k], b = *(const oasr_f32x2_t*)&lnb[k];
        *(uint32_t*)(xs + k) = pack_bf2(v[c][i] * rstd * g[0] + b[0], v[c][i + 1] * rstd * g[1] + b[1]);
      }
    }
}
// wave 0: a PLAIN bf16 row (the embedding launch's, layer 0) fetched in packet layout
template <int RD>
__device__ __forceinline__ void pk_from_plain(const bf16_t* x, int d, int lane, u32x4_t (&pkt)[WPKL]) {
#pragma unroll
  for (int c = 0; c < WPKL; ++c) {
    pkt[c] = u32x4_t{0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < RD / 2; ++j) {
      const int k = (lane + 64 * c) * RD + 2 * j;
      if (k < d) pkt[c][j] = ld4_agent(x + k);
    }
  }
}

struct WGemv {  // one projection phase
  const bf16_t* w;  // [N][K]
  int K, N, R;
  const bf16_t* xin;
  const float *ln_g, *ln_b;
  bool merge;
  const float* bias;
  bool gelu;
  const bf16_t* resid;
  bf16_t* out;
};
template <int PH>
struct WPh {  // what projection phase PH is, at compile time
  static constexpr bool LN = PH == 0 || PH == 3 || PH == 6, MERGE = PH == 5, GELU = PH == 6, RESID = PH == 2 || PH == 5 || PH == 7;
  static constexpr int NEXT = PH == 7 ? 0 : PH + 1;  // the phase after it
  // rows in / out as packets: the vector id, or -1 (in: the operand row; PH 0 takes packets from the layer before, a plain row in layer 0)
  static constexpr int PKIN = PH == 0 ? WV_X : PH == 3 ? WV_X2 : PH == 6 ? WV_X3 : -1;
  static constexpr int PKOUT = PH == 2 ? WV_X2 : PH == 3 ? WV_Q : PH == 5 ? WV_X3 : PH == 7 ? WV_X : -1;
  // parity of the projection's index in the layer's six (0, 2, 3, 5, 6, 7): consecutive projections use different LDS hand-over buffers (the
  // compute waves run ahead into the next projection -- through an attention phase without a workgroup barrier -- while the helper wave still
  // reads the current one's in its epilogue)
  static constexpr int PAR = (PH == 2 || PH == 5 || PH == 7) ? 1 : 0;
};
// units per compute wave and phase at most, at 256 workgroups (decode_wide_supports checks the shape against it): keeps the unrolled unit loops -- and the
// kernel under the 64 KB of instruction cache two CUs share -- as short as the instantiation allows
template <int RD, int PH>
constexpr int wmu() {  // ceil(units of the workgroup at most / compute waves); units: RD = 2: 6 / 2 / 8 / 8 (q|k|v, N = d, mlp.0, mlp.2), 4: 24 / 8 / 32 / 32, 6 (d <= 1280): 48 / 18 / 60 / 60
  constexpr int u = RD == 2 ? (PH == 0 ? 6 : (PH == 6 || PH == 7) ? 8 : 2) : RD == 4 ? (PH == 0 ? 24 : (PH == 6 || PH == 7) ? 32 : 8) : (PH == 0 ? 48 : (PH == 6 || PH == 7) ? 60 : 18);
  return (u + WC - 1) / WC;
}
template <int PH>
__device__ __forceinline__ WGemv gemv_of(const WArgs& a, int layer) {
  const long s = (long)layer * a.lstride;
  WGemv p;
  p.merge = WPh<PH>::MERGE, p.gelu = WPh<PH>::GELU, p.ln_g = p.ln_b = nullptr, p.resid = nullptr, p.K = a.d, p.N = a.d, p.xin = nullptr;
  if constexpr (PH == 0) {  // attn_ln -> q | k | v of position pos, straight into the cache row
    p.w = a.wflat + (a.l0.wqkv + s), p.xin = a.x, p.N = 3 * a.d, p.ln_g = a.params + (a.l0.ln1g + s), p.ln_b = a.params + (a.l0.ln1b + s);
    p.bias = a.aux + (a.l0.bqkv_aux + (long)layer * a.astride);
    p.out = a.cache + (long)layer * a.cache_lstride + (long)a.pos * 3 * a.d;
  } else if constexpr (PH == 2) {  // self-attention output projection + residual
    p.w = a.wflat + (a.l0.wo + s), p.xin = a.o, p.bias = a.params + (a.l0.bo + s), p.resid = a.x, p.out = a.x2;
  } else if constexpr (PH == 3) {  // cross_attn_ln -> cross query
    p.w = a.wflat + (a.l0.wcq + s), p.xin = a.x2, p.ln_g = a.params + (a.l0.lncg + s), p.ln_b = a.params + (a.l0.lncb + s), p.bias = a.params + (a.l0.bcq + s);
    p.out = a.q;
  } else if constexpr (PH == 5) {  // cross-attention output projection + residual (operand = merged segment partials)
    p.w = a.wflat + (a.l0.wco + s), p.bias = a.params + (a.l0.bco + s), p.resid = a.x2, p.out = a.x3;
  } else if constexpr (PH == 6) {  // mlp_ln -> mlp.0 + GELU
    p.w = a.wflat + (a.l0.w1 + s), p.xin = a.x3, p.N = 4 * a.d, p.ln_g = a.params + (a.l0.ln2g + s), p.ln_b = a.params + (a.l0.ln2b + s), p.bias = a.params + (a.l0.b1 + s);
    p.out = a.hg;
  } else {  // 7: mlp.2 + residual
    p.w = a.wflat + (a.l0.w2 + s), p.xin = a.hg, p.K = 4 * a.d, p.bias = a.params + (a.l0.b2 + s), p.resid = a.x3, p.out = a.x;
  }
  p.R = 2 * ((p.N + 2 * a.nwg - 1) / (2 * a.nwg));
  return p;
}

// this wave's weight chunks of a projection phase -> registers (unit u = wave + 8 i: row u / J of the workgroup's run, K span u % J)
// Units (row r of the workgroup's run, 512-element K span j), numbered u = r J + j, are dealt to the compute waves in contiguous runs of
// upw = ceil(U / WC): one division per wave and phase, then (r, j) by stepping.
struct WUnits {
  int U, upw, u0, r0, j0, J, KC;
};
__host__ __device__ __forceinline__ WUnits units_of(const WGemv& p, int wg, int wave) {
  WUnits q;
  q.KC = p.K >> 3, q.J = (q.KC + 63) >> 6;
  int rows = p.N - wg * p.R;
  rows = rows < 0 ? 0 : (rows > p.R ? p.R : rows);
  q.U = rows * q.J;
  q.upw = (q.U + WC - 1) / WC;  // (WC is a compile-time constant)
  q.u0 = (wave - 1) * q.upw;
  q.r0 = q.u0 / q.J, q.j0 = q.u0 - q.r0 * q.J;
  return q;
}
template <int MU>
__device__ __forceinline__ void request_units(const WGemv& p, int wg, int wave, int lane, u32x4_t (&wreg)[WMAXU]) {
  if (wave == 0) return;
  const WUnits q = units_of(p, wg, wave);
  const bf16_t* row = p.w + (long)(wg * p.R + q.r0) * p.K;
  int j = q.j0;
  const int cnt = q.U - q.u0 < q.upw ? q.U - q.u0 : q.upw;  // this wave's units
#pragma unroll
  for (int i = 0; i < MU; ++i) {
    if (i >= cnt) break;  // (one branch out instead of a skipped body per remaining slot: a wave's instructions cost ~10 cycles each here)
    const int c = j * 64 + lane;
    if (c < q.KC) wreg[i] = __builtin_nontemporal_load((const u32x4_t*)(row + (long)c * 8));  // (else: never read)
    if (++j == q.J) j = 0, row += p.K;
  }
}
// What the compute waves fetch for the helper wave along with a phase's weights: the LayerNorm parameters of the operand row and the bias of the
// workgroup's rows.  They go through LDS at the top of the phase.  The helper wave itself has NO load in flight when it starts to poll: loads
// return in order, so a parameter row that misses to HBM in front of the poll would hold the poll's answer back by its latency.
struct WStage {
  f32x4_t g, b;  // elements 4 i .. 4 i + 3 of gamma / beta, i = tid - 64
  float bias;    // row i of the workgroup's run
};
// wave 0, lane i < rows: the (bf16) values this workgroup produced for rows i of x / x2 / x3 -- the NEXT projection onto the same rows adds them as its
// residual, so they never travel
struct WRes {
  float x, x2, x3;
};
// a FRESH (undefined) value: registers that are assigned under a condition inside the layer loop would otherwise carry their previous contents
// around the loop as far as the register allocator can tell (measured: the two attention phases' K / V rows were live at the same time)
template <typename V>
__device__ __forceinline__ void fresh(V& v) {
  asm volatile("" : "=v"(v));
}
// the registers of a phase's weights hold nothing any more (request_units assigns them under conditions: without this the OLD values stay live
// across the attention phases as far as the register allocator can tell)
__device__ __forceinline__ void kill_units(u32x4_t (&wreg)[WMAXU]) {
#pragma unroll
  for (int i = 0; i < WMAXU; ++i) asm volatile("" : "=v"(wreg[i]));
}
// (v_dot2c_f32_bf16 would do two products per instruction; through __builtin_amdgcn_fdot2_f32_bf16 it returned wrong sums here -- max |d| 8.6 on
// the logits, profiles/r06_decode_wide.txt -- so the products are plain fp32 FMAs on the unpacked halves)
__device__ __forceinline__ float dot8(const u32x4_t& w, const u32x4_t& x) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += bf_lo(w[i]) * bf_lo(x[i]) + bf_hi(w[i]) * bf_hi(x[i]);
  return s;
}
__device__ __forceinline__ float rdlane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ float merge_w(const float (&m_s)[WNS], const float (&l_s)[WNS], const float (&o_s)[WNS]) {
  float m = m_s[0];
#pragma unroll
  for (int s = 1; s < WNS; ++s) m = fmaxf(m, m_s[s]);
  float L = 0.f, O = 0.f;
#pragma unroll
  for (int s = 0; s < WNS; ++s) {
    const float w = __builtin_amdgcn_exp2f(m_s[s] - m);
    L += l_s[s] * w;
    O += o_s[s] * w;
  }
  return L > 0.f ? O / L : 0.f;
}

// one query row against this thread's NK keys (key u of the thread = key grp + WNG u of the workgroup's range, n keys in it; K / V rows already
// in registers): scores, maximum over the workgroup, P (bf16) . V, and the sum over the WNG groups in two short stages (a wave's eight groups
// first, then the waves) instead of one WNG-step walk.  Threads 0-63 (dimension tid) return the segment's maximum, normaliser and unnormalised
// output value.
struct WAttLds {
  float ared[WNG][64];
  float ared2[WW][64];
  float lsum2[WW];
  float wmax[WW];
};
template <int NK>
__device__ __forceinline__ void attend(const float (&qv)[8], const u32x4_t (&k4)[NK], const u32x4_t (&v4)[NK], int n, int tid, WAttLds& L, float& m_out,
                                       float& lt_out, float& acc_out) {
  const int lane = tid & 63, l8 = tid & 7, grp = tid >> 3;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float s2[NK], mx = dec::NEG;
#pragma unroll
  for (int u = 0; u < NK; ++u) {
    s2[u] = dec::NEG;
    if (WNG * u < n) {
      const float sc = dec::score8(qv, k4[u]);
      if (grp + WNG * u < n) s2[u] = sc, mx = fmaxf(mx, sc);
    }
  }
  mx = wmaxf(mx);
  if (lane == 0) L.wmax[wave] = mx;
  __syncthreads();
  float m = L.wmax[0];
#pragma unroll
  for (int w = 1; w < WW; ++w) m = fmaxf(m, L.wmax[w]);
  float l = 0.f, o[8];
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) o[jj] = 0.f;
#pragma unroll
  for (int u = 0; u < NK; ++u)
    if (WNG * u < n) dec::accum_pv(grp + WNG * u < n ? __builtin_amdgcn_exp2f(s2[u] - m) : 0.f, v4[u], l, o);
  *(f32x4_t*)&L.ared[grp][l8 * 8] = f32x4_t{o[0], o[1], o[2], o[3]};
  *(f32x4_t*)&L.ared[grp][l8 * 8 + 4] = f32x4_t{o[4], o[5], o[6], o[7]};
  const float lw = wsum(l) * 0.125f;  // the wave's eight groups (every group's l sits in its eight lanes)
  if (lane == 0) L.lsum2[wave] = lw;
  __syncthreads();
  {  // stage 1: this wave's eight groups, dimension `lane`
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += L.ared[wave * 8 + k][lane];
    L.ared2[wave][lane] = acc;
  }
  __syncthreads();
  m_out = m, lt_out = 0.f, acc_out = 0.f;
  if (tid < 64) {
#pragma unroll
    for (int w = 0; w < WW; ++w) acc_out += L.ared2[w][tid], lt_out += L.lsum2[w];
  }
}

#define WSTAMP(K)                                                                                                \
  do {                                                                                                           \
    if (STAMPS && wg == a.swg && layer == 1 && tid == 0) a.stamps[ph * 8 + (K)] = __builtin_amdgcn_s_memtime(); \
  } while (0)

#define WPHASE_PROLOGUE                                                                                                       \
  /* thread and workgroup ids re-materialised per phase: otherwise every address that depends on them only is hoisted out of the \
     layer loop and spilled -- 130 VGPRs' worth */                                                                                \
  int tid = threadIdx.x, wg = blockIdx.x;                                                                                         \
  asm volatile("" : "+v"(tid), "+s"(wg));                                                                                       \
  const int lane = tid & 63, l8 = tid & 7, grp = tid >> 3;                                                                        \
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                                                                      \
  const unsigned target = base + gp; /* every workgroup has completed the phase before this one */                                \
  (void)l8, (void)grp, (void)lane, (void)wave, (void)target

struct WLds {
  __attribute__((aligned(16))) unsigned short xs[4 * WMAXD];  // the operand row (bf16)
  __attribute__((aligned(32))) float rowsum[2][64 * WC];  // [phase parity][row of the workgroup's run][compute wave]
  __attribute__((aligned(16))) float lng[WMAXD], lnb[WMAXD];  // LayerNorm parameters of the current phase's operand row
  float biasv[2][64];            // [phase parity] bias of the workgroup's rows
  WAttLds att;
};
// this workgroup's rows of the phase are stored (wave 0 has waited for the acknowledgements): publish -- lanes 0-7 of wave 0, one store
// instruction, eight replicas
#define WPUBLISH() \
  if (tid < WREP) st4_agent(a.flagv + tid * WFS + wg, base + gp + 1)

// ---------------- self-attention: one workgroup per head ----------------
template <bool STAMPS>
__device__ __forceinline__ void self_phase(const WArgs& a, int layer, unsigned gp, unsigned base, const unsigned* myflags, int rep, WLds& lds) {
  constexpr int ph = 1;
  WPHASE_PROLOGUE;
  WAttLds& att = lds.att;
  const bf16_t* self = a.cache + (long)layer * a.cache_lstride;
  if (wg < a.H) {
    const int h = wg, n = a.pos + 1;
    // K / V rows of the positions before this one were written by earlier launches: requested before the poll (plain loads); row `pos` itself
    // (written a phase ago by other workgroups) after it, with the query, through agent-scope loads
    const bf16_t* kb = self + a.d + h * 64;  // (uniform bases + one 32-bit offset per key: K and V rows share it)
    const bf16_t* vb = kb + a.d;
    u32x4_t k4[WSK], v4[WSK];
#pragma unroll
    for (int u = 0; u < WSK; ++u) {
      fresh(k4[u]), fresh(v4[u]);
      const int t = grp + WNG * u;
      if (WNG * u < a.pos && t < a.pos) {
        const size_t off = (size_t)(unsigned)((t * 3 * a.d + l8 * 8) * 2);
        k4[u] = __builtin_nontemporal_load((const u32x4_t*)((const char*)kb + off));
        v4[u] = __builtin_nontemporal_load((const u32x4_t*)((const char*)vb + off));
      }
    }
    WSTAMP(0);
    if (wave == 0) {
      // the new position's q | k | v values of this head arrive as packets (the projection a phase ago): lane (t, wi, j) polls packet j of the wi-th
      // workgroup that holds rows of range t (0 q, 1 k, 2 v: rows t d + h 64 .. + 63 of the 3 d), and spreads its six values out in LDS
      const int R0 = 2 * ((3 * a.d + 2 * a.nwg - 1) / (2 * a.nwg)), NP0 = (R0 + 5) / 6;
      const int t = lane >> 4, li = lane & 15;
      const int wi = li / NP0, j = li - wi * NP0;
      const int lo = t * a.d + h * 64;               // first row of the range
      const int wgp = lo / R0 + wi;                  // producing workgroup
      const bool mine = t < 3 && wi * NP0 + j < 16 && wgp <= (lo + 63) / R0 && wgp * R0 + 6 * j < 3 * a.d;
      const u32x4_t* src = a.pkqkv + (size_t)rep * WQKS;
      u32x4_t pq = {0u, 0u, 0u, 0u};
      bool need = mine;
      unsigned spins = 0;
      for (;;) {
        ld16_agent_issue(pq, src + (mine ? wgp * 3 + j : 0));  // (unconditional: see pk_poll)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(pq)::"memory");
        if (need && pq[3] == target) need = false;
        if (__builtin_amdgcn_ballot_w64(need) == 0) break;
        if (!(a.flags & 4)) __builtin_amdgcn_s_sleep(1);
        if (++spins > WSPIN || ((spins & 63) == 0 && ld4_agent(a.ctrl + 1) != 0)) {
          if (lane == 0) __hip_atomic_fetch_or(a.ctrl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
      if (mine) {
#pragma unroll
        for (int e = 0; e < 6; ++e) {
          const int r = wgp * R0 + 6 * j + e - lo;  // (a packet's values past the workgroup's run belong to nobody: 6 j + e < R0)
          if (6 * j + e < R0 && r >= 0 && r < 64) lds.xs[t * 64 + r] = (unsigned short)(pq[e >> 1] >> ((e & 1) * 16));
        }
      }
    }
    __syncthreads();
    WSTAMP(1);
    const u32x4_t q4 = *(const u32x4_t*)(lds.xs + l8 * 8);
#pragma unroll
    for (int u = 0; u < WSK; ++u) {
      const int t = grp + WNG * u;
      if (WNG * u <= a.pos && a.pos < WNG * (u + 1) && t >= a.pos) {  // the step that holds row pos: its lanes at and past pos take that row (past: masked, finite)
        k4[u] = *(const u32x4_t*)(lds.xs + 64 + l8 * 8), v4[u] = *(const u32x4_t*)(lds.xs + 128 + l8 * 8);
      }
    }
    if (STAMPS) {
      WWAIT_VM0();
      WSTAMP(3);
    }
    float qv[8];
    dec::load_q8(q4, qv);
    float m, lt, acc;
    attend<WSK>(qv, k4, v4, n, tid, att, m, lt, acc);
    WSTAMP(6);
    if (tid < 64) {  // the head's 64 output values as 11 packets of six (lane L: values 6 L .. 6 L + 5): no acknowledgement wait
      const float val = lt > 0.f ? acc / lt : 0.f;
      const u32x4_t w = gather_packet(bf_round(val), lane, base + gp + 1);
      store_packets(a.pk + (size_t)(WV_O * WREP) * WPKS + h * 11 + lane, WPKS, w, lane < 11);
    }
    WSTAMP(2);
  }
  WPUBLISH();
  WSTAMP(7);
}

// ---------------- cross-attention: one workgroup per (head, key segment) ----------------
template <bool STAMPS, int RD>
__device__ __forceinline__ void cross_phase(const WArgs& a, int layer, unsigned gp, unsigned base, const unsigned* myflags, int rep, WLds& lds) {
  constexpr int ph = 4;
  WPHASE_PROLOGUE;
  WAttLds& att = lds.att;
  const bf16_t* cross = a.cache + (long)layer * a.cache_lstride + (long)3 * a.S_max * a.d;
  if (wg < a.H * WNS) {
    const int h = wg >> WNS_LOG, sg = wg & (WNS - 1);
    const int SL = (a.Te + WNS - 1) / WNS;
    const int t0 = sg * SL;
    int n = a.Te - t0;
    n = n > SL ? SL : (n < 0 ? 0 : n);
    // the segment's K / V rows do not depend on the token: requested before the poll
    const bf16_t* kb = cross + (long)t0 * 2 * a.d + h * 64;
    const bf16_t* vb = kb + a.d;
    u32x4_t k4[WCK], v4[WCK];
#pragma unroll
    for (int u = 0; u < WCK; ++u) {
      fresh(k4[u]), fresh(v4[u]);
      const int t = grp + WNG * u;
      if (WNG * u < n) {
        const size_t off = (size_t)(unsigned)(((t < n ? t : n - 1) * 2 * a.d + l8 * 8) * 2);
        k4[u] = __builtin_nontemporal_load((const u32x4_t*)((const char*)kb + off));
        v4[u] = __builtin_nontemporal_load((const u32x4_t*)((const char*)vb + off));
      }
    }
    WSTAMP(0);
    if (wave == 0) {  // the head's 64 query values arrive as packets of R_d values (the cross query projection a phase ago): poll them, spread them out in LDS
      constexpr int R_d = RD;
      const int s0 = (h * 64) / R_d, s1 = (h * 64 + 63) / R_d;  // slots that hold rows h 64 .. h 64 + 63
      const u32x4_t* src = a.pk + (size_t)(WV_Q * WREP + rep) * WPKS;
      u32x4_t pq = {0u, 0u, 0u, 0u};
      bool need = s0 + lane <= s1;
      unsigned spins = 0;
      for (;;) {
        ld16_agent_issue(pq, src + (s0 + lane <= s1 ? s0 + lane : s0));  // (unconditional: see pk_poll)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(pq)::"memory");
        if (need && pq[3] == target) need = false;
        if (__builtin_amdgcn_ballot_w64(need) == 0) break;
        if (!(a.flags & 4)) __builtin_amdgcn_s_sleep(1);
        if (++spins > WSPIN || ((spins & 63) == 0 && ld4_agent(a.ctrl + 1) != 0)) {
          if (lane == 0) __hip_atomic_fetch_or(a.ctrl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
      if (s0 + lane <= s1) {
#pragma unroll
        for (int i = 0; i < RD; ++i) {
          const int r = (s0 + lane) * R_d + i - h * 64;
          if (r >= 0 && r < 64) lds.xs[r] = (unsigned short)(pq[i >> 1] >> ((i & 1) * 16));
        }
      }
    }
    __syncthreads();
    WSTAMP(1);
    float qv[8];
    dec::load_q8(*(const u32x4_t*)(lds.xs + l8 * 8), qv);
    float m, lt, acc;
    attend<WCK>(qv, k4, v4, n, tid, att, m, lt, acc);
    if (tid < 64) {
      float* dst = a.part + (long)wg * 66;
      st4_agent(dst + 2 + lane, __float_as_uint(acc));
      if (lane == 0) st4_agent(dst, __float_as_uint(m)), st4_agent(dst + 1, __float_as_uint(lt));
      WWAIT_VM0();
    }
    WSTAMP(2);
  }
  WPUBLISH();
  WSTAMP(7);
}

// ---------------- projection PH: rows wg R .. wg R + R - 1 ----------------
template <bool STAMPS, int RD, int PH>
__device__ __forceinline__ void gemv_phase(const WArgs& a, int layer, unsigned gp, unsigned base, const unsigned* myflags, int rep, WLds& lds,
                                           u32x4_t (&wreg)[WMAXU], const WStage& st, WRes& res) {
  constexpr int ph = PH;
  WPHASE_PROLOGUE;
  unsigned short* xs = lds.xs;
  // (two buffers, WPh::PAR: the other waves zero the next projection's while wave 0 may still be reading this one's in its epilogue)
  float* rowsum = lds.rowsum[WPh<PH>::PAR];
  if (tid < 64 * WC) rowsum[tid] = 0.f;  // (before the workgroup barriers below)
  const WGemv p = gemv_of<PH>(a, layer);
  const int row0 = wg * p.R;
  if (row0 < p.N) {
    const int KC = p.K >> 3, J = (KC + 63) >> 6;
    int rows = p.N - row0;
    rows = rows > p.R ? p.R : rows;
    const int U = rows * J;
    WUnits q = {};  // (before the poll: off the path from the flags to the stores; the helper wave has no units)
  if (wave > 0) q = units_of(p, wg, wave);
    // the compute waves hand over what they fetched for the helper wave (request_phase)
    float* biasv = lds.biasv[WPh<PH>::PAR];
    if (tid >= 64) {
      const int i = tid - 64;
      if (WPh<PH>::LN && 4 * i < p.K) *(f32x4_t*)&lds.lng[4 * i] = st.g, *(f32x4_t*)&lds.lnb[4 * i] = st.b;
      if (i < rows) biasv[i] = st.bias;
    }
    WSTAMP(0);
    constexpr int PKIN = WPh<PH>::PKIN, PKOUT = WPh<PH>::PKOUT;
    const bool pk_in = PKIN >= 0 && !(PH == 0 && layer == 0);  // (layer 0's attn_ln reads the embedding launch's plain row)
    u32x4_t pkt[WPKL];
    if (wave == 0) {
      if (pk_in) pk_poll(a, PKIN, rep, target, (a.d + RD - 1) / RD, lane, pkt);
      else if (PH == 0) pk_from_plain<RD>(a.x, a.d, lane, pkt);  // (layer 0; preceded by no phase of this launch)
      else if (PH == 2) pk_poll(a, WV_O, rep, target, a.H * 11, lane, pkt);
      else wide_wait(a, myflags, target, lane);
    }
    __syncthreads();
    WSTAMP(1);
    float resid_v = 0.f;  // (the residual row is two phases old; needed by the epilogue only, so requested behind the operand row's loads)
    unsigned resid_z = 0u;
    // ---- operand row -> LDS ----
    if constexpr (WPh<PH>::MERGE) {  // merged cross-attention output: wave w merges heads w, w + 8, ...; lane = dimension; (m, l) of segment s come in through lane s
      // (two heads per pass: both heads' partials are requested before either is merged -- one round trip for H <= 2 WW)
#pragma unroll 1
      for (int h0 = wave; h0 < a.H; h0 += 2 * WW) {
        const int sl = lane < WNS ? lane : WNS - 1;
        float mv[2], lv[2], o_s[2][WNS];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int h = h0 + WW * k < a.H ? h0 + WW * k : h0;
          const float* src = a.part + (long)h * (WNS * 66);
          mv[k] = ldf_agent(src + sl * 66), lv[k] = ldf_agent(src + sl * 66 + 1);
#pragma unroll
          for (int sg = 0; sg < WNS; ++sg) o_s[k][sg] = ldf_agent(src + sg * 66 + 2 + lane);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int h = h0 + WW * k;
          if (h < a.H) {
            float m_s[WNS], l_s[WNS];
#pragma unroll
            for (int sg = 0; sg < WNS; ++sg) m_s[sg] = rdlane(mv[k], sg), l_s[sg] = rdlane(lv[k], sg);
            const float val = merge_w(m_s, l_s, o_s[k]), nb = dec::xor_lane<1>(val);
            if ((lane & 1) == 0) *(uint32_t*)(xs + h * 64 + lane) = pack_bf2(val, nb);
          }
        }
      }
    } else if constexpr (WPh<PH>::LN) {  // LayerNorm folded into the operand (K = d): wave 0
      if (wave == 0) ln_packets<RD>(pkt, a.d, lane, lds.lng, lds.lnb, xs);
    } else if constexpr (PH == 2) {  // the self-attention output arrived as packets (polled above): slot h 11 + j = values h 64 + 6 j .. + 5
      if (wave == 0) {
#pragma unroll
        for (int c = 0; c < WPKL; ++c) {
          const int sl = lane + 64 * c;
          if (sl < a.H * 11) {
            const int h = sl / 11, j = sl - h * 11;
#pragma unroll
            for (int e = 0; e < 3; ++e)
              if (6 * j + 2 * e < 64) *(uint32_t*)(xs + h * 64 + 6 * j + 2 * e) = pkt[c][e];
          }
        }
      }
    } else {  // plain row: K / 8 chunks over the 512 threads
      constexpr int NC = (4 * WMAXD / 8 + WT - 1) / WT;
      u32x4_t raw[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        fresh(raw[c]);
        if (tid + WT * c < KC) raw[c] = ld16_agent(p.xin + (tid + WT * c) * 8);
      }
#pragma unroll
      for (int c = 0; c < NC; ++c)
        if (tid + WT * c < KC) *(u32x4_t*)(xs + (tid + WT * c) * 8) = raw[c];
    }
    if (PH == 2 && layer == 0 && wave == 0 && lane < rows) resid_z = ld4_agent(p.resid + ((row0 + lane) & ~1));  // (the embedding launch's row; later: WRes)
    __syncthreads();
    WSTAMP(2);
    // ---- this wave's units ----
    if (wave > 0) {
      if (STAMPS) {
        WWAIT_VM0();
        if (wg == a.swg && layer == 1 && tid == 64) a.stamps[ph * 8 + 5] = __builtin_amdgcn_s_memtime();
      }
      const WUnits q = units_of(p, wg, wave);
      // lane-wise partial products of consecutive spans of the SAME row are added up before the wave reduction: one reduction per (wave, row),
      // its result in rowsum[row][wave] (zeroed at the top of the phase); the epilogue adds a row's WC slots in wave order
      const int cnt = q.U - q.u0 < q.upw ? q.U - q.u0 : q.upw;
      float accl = 0.f;
      int j = q.j0, r = q.r0;
#pragma unroll
      for (int i = 0; i < wmu<RD, PH>(); ++i) {
        if (i >= cnt) break;
        const int c = j * 64 + lane;
        if (c < q.KC) accl += dot8(wreg[i], *(const u32x4_t*)(xs + c * 8));
        ++j;
        if (j == q.J || i == cnt - 1) {
          const float sum = wsum(accl);
          if (lane == 0) rowsum[r * WC + (wave - 1)] = sum;
          accl = 0.f, j = 0, ++r;
        }
      }
      kill_units(wreg);
      if (STAMPS && wg == a.swg && layer == 1 && tid == 64) a.stamps[ph * 8 + 6] = __builtin_amdgcn_s_memtime();
    }
    __syncthreads();
    WSTAMP(3);
    if (wave == 0) {
      float y = 0.f;
      if (lane < rows) {
        float acc = 0.f;
#pragma unroll
      for (int w = 0; w < WC; ++w) acc += rowsum[lane * WC + w];
        if constexpr (PH == 2) resid_v = layer == 0 ? (((row0 + lane) & 1) ? bf_hi(resid_z) : bf_lo(resid_z)) : res.x;
        if constexpr (PH == 5) resid_v = res.x2;
        if constexpr (PH == 7) resid_v = res.x3;
        y = dec::epi_value(acc, biasv[lane], WPh<PH>::GELU, WPh<PH>::RESID, resid_v);
      }
      if constexpr (PKOUT >= 0) {
        // the workgroup's <= 6 values + the phase's epoch in ONE 16-byte store per replica (lanes 0-7): nothing to wait for before the flag
        const float yb = bf_round(y);
        if constexpr (PH == 2) res.x2 = yb;
        if constexpr (PH == 5) res.x3 = yb;
        if constexpr (PH == 7) res.x = yb;
        u32x4_t w;
#pragma unroll
        for (int i = 0; i < 3; ++i) w[i] = i < RD / 2 ? pack_bf2(rdl(yb, 2 * i), rdl(yb, 2 * i + 1)) : 0u;  // (lanes >= rows hold 0)
        w[3] = base + gp + 1;
        if (lane < WREP) st16_agent(a.pk + (size_t)(PKOUT * WREP + lane) * WPKS + wg, w);
      } else if constexpr (PH == 0) {
        // q | k | v of the new position: into the cache row for the steps to come (nobody reads it in THIS launch: no acknowledgement wait), and as
        // packets of six for the self-attention heads of this one
        const float nb = dec::xor_lane<1>(y);
        if (lane < rows && (lane & 1) == 0) st4_agent(p.out + row0 + lane, pack_bf2(y, nb));
        const u32x4_t w = gather_packet(bf_round(y), lane, base + gp + 1);
        store_packets(a.pkqkv + wg * 3 + lane, WQKS, w, lane * 6 < rows);
      } else {
        const float nb = dec::xor_lane<1>(y);
        if (lane < rows && (lane & 1) == 0) st4_agent(p.out + row0 + lane, pack_bf2(y, nb));
        WWAIT_VM0();
      }
    }
    WSTAMP(4);
  }
  WPUBLISH();
  WSTAMP(7);
}
// ---------------- the last phase: final LayerNorm + tied logits projection (TextDecoder.forward, olmoasr/model.py:812-817) ----------------
// V rows of d bf16 weights (106 MB at medium) dealt out in runs of ceil(V / nwg) rows per workgroup, row k of a run to compute wave k % 8; a wave
// streams its rows in groups of WLG (the first group requested before the poll: the weights are static), one wave reduction per row, the row's
// logit parked in lane (row ordinal) and stored once at the end.  bf16-rounded like the autocast Linear's output (decode_shared.h::epi_logit).
constexpr int WLG = 4;   // rows per group
constexpr int WLJ = 3;   // 512-element spans per row: d <= 1536
template <bool STAMPS, int RD>
__device__ __forceinline__ void logits_phase(const WArgs& a, unsigned gp, unsigned base, const unsigned* myflags, int rep, WLds& lds) {
  WPHASE_PROLOGUE;
  unsigned short* xs = lds.xs;
  const int KC = a.d >> 3, J = (KC + 63) >> 6;
  const int RW = (a.V + a.nwg - 1) / a.nwg;
  const int row0 = wg * RW;
  int rows = a.V - row0;
  rows = rows > RW ? RW : (rows < 0 ? 0 : rows);
  const int cw = wave - 1;
  const int mine = wave > 0 && rows > cw ? (rows - cw + WC - 1) / WC : 0;  // rows of this wave: cw, cw + WC, ...
  u32x4_t wq[WLG][WLJ];
  auto request = [&](int k0) {  // rows k0 .. k0 + WLG - 1 of this wave
#pragma unroll
    for (int g = 0; g < WLG; ++g) {
#pragma unroll
      for (int j = 0; j < WLJ; ++j) {
        fresh(wq[g][j]);
        const int c = j * 64 + lane;
        if (k0 + g < mine && j < J && c < KC)
          wq[g][j] = __builtin_nontemporal_load((const u32x4_t*)(a.w_logits + (long)(row0 + cw + WC * (k0 + g)) * a.d + (long)c * 8));
      }
    }
  };
  request(0);
  // the final LayerNorm's parameters: fetched by the compute waves, handed over through LDS (the helper wave polls with no load in flight)
  if (tid >= 64) {
    const int i = tid - 64;
    if (4 * i < a.d) {
      const f32x4_t g4 = *(const f32x4_t*)(a.lnf_g + 4 * i), b4 = *(const f32x4_t*)(a.lnf_b + 4 * i);
      *(f32x4_t*)&lds.lng[4 * i] = g4, *(f32x4_t*)&lds.lnb[4 * i] = b4;
    }
  }
  u32x4_t pkt[WPKL];
  if (wave == 0) pk_poll(a, WV_X, rep, target, (a.d + RD - 1) / RD, lane, pkt);  // the last block's output arrives as packets
  __syncthreads();  // (... and the parameters are in LDS)
  if (wave == 0) ln_packets<RD>(pkt, a.d, lane, lds.lng, lds.lnb, xs);
  __syncthreads();
  if (mine > 0) {
    u32x4_t xq[WLJ];  // this lane's chunks of the operand row
#pragma unroll
    for (int j = 0; j < WLJ; ++j) {
      fresh(xq[j]);
      if (j < J && j * 64 + lane < KC) xq[j] = *(const u32x4_t*)(xs + (j * 64 + lane) * 8);
    }
    float res = 0.f;
#pragma unroll 1
    for (int k0 = 0; k0 < mine; k0 += WLG) {
      float part[WLG];
#pragma unroll
      for (int g = 0; g < WLG; ++g) {
        part[g] = 0.f;
#pragma unroll
        for (int j = 0; j < WLJ; ++j)
          if (k0 + g < mine && j < J && j * 64 + lane < KC) part[g] += dot8(wq[g][j], xq[j]);
      }
      if (k0 + WLG < mine) request(k0 + WLG);  // (the products above are done with the registers)
#pragma unroll
      for (int g = 0; g < WLG; ++g)
        if (k0 + g < mine) {
          const float y = dec::epi_logit(wsum(part[g]), 0.f);
          if (lane == k0 + g) res = y;
        }
    }
    if (lane < mine) a.logits_out[row0 + cw + WC * lane] = res;
  }
}

// the weights of projection PH into this wave's registers
template <int RD, int PH>
__device__ __forceinline__ void request_phase(const WArgs& a, int layer, u32x4_t (&wreg)[WMAXU], WStage& st) {
  int tid = threadIdx.x, wg = blockIdx.x;
  asm volatile("" : "+v"(tid), "+s"(wg));
  kill_units(wreg);
  fresh(st.g), fresh(st.b), fresh(st.bias);
  if (layer >= a.L) return;
  const WGemv p = gemv_of<PH>(a, layer);
  request_units<wmu<RD, PH>()>(p, wg, __builtin_amdgcn_readfirstlane(tid >> 6), tid & 63, wreg);
  const int i = tid - 64;
  if (i >= 0) {
    if (WPh<PH>::LN && 4 * i < p.K) st.g = *(const f32x4_t*)(p.ln_g + 4 * i), st.b = *(const f32x4_t*)(p.ln_b + 4 * i);
    if (i < p.R && wg * p.R + i < p.N) st.bias = p.bias[wg * p.R + i];
  }
}

// STAMPS: the measurement instantiation (scripts/decode_xcd_probe.py; its stores cost waits of their own).  RD: rows per workgroup of an N = d projection
// = values per packet (2: d <= 512, 4: d <= 1024, 6: d <= 1536 at 256 workgroups)
template <bool STAMPS, int RD>
__global__ __launch_bounds__(WT) void decode_wide_kernel(WArgs a) {
  __shared__ WLds lds;
  const unsigned base = a.ctrl[4];
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const int rep = (int)(xcc & (WREP - 1));
  const unsigned* myflags = a.flagv + (size_t)rep * WFS;
  if (threadIdx.x == 0) __hip_atomic_fetch_or(a.ctrl + 3, 1u << (xcc & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  u32x4_t wreg[WMAXU];
  WStage st;
  WRes res = {0.f, 0.f, 0.f};
  request_phase<RD, 0>(a, 0, wreg, st);
  unsigned gp = 0;  // phases before the current one
  // A phase's weights are requested at the end of the phase before it: in flight through the exchange.  (Not across an attention phase: with
  // the attention's K / V rows the registers of weights would not fit, and the compiler would park them in scratch = wait for them on the spot.)
#pragma unroll 1
  for (int layer = 0; layer < a.L; ++layer) {
    gemv_phase<STAMPS, RD, 0>(a, layer, gp++, base, myflags, rep, lds, wreg, st, res);
    self_phase<STAMPS>(a, layer, gp++, base, myflags, rep, lds);
    request_phase<RD, 2>(a, layer, wreg, st);
    gemv_phase<STAMPS, RD, 2>(a, layer, gp++, base, myflags, rep, lds, wreg, st, res);
    request_phase<RD, 3>(a, layer, wreg, st);
    gemv_phase<STAMPS, RD, 3>(a, layer, gp++, base, myflags, rep, lds, wreg, st, res);
    cross_phase<STAMPS, RD>(a, layer, gp++, base, myflags, rep, lds);
    request_phase<RD, 5>(a, layer, wreg, st);
    gemv_phase<STAMPS, RD, 5>(a, layer, gp++, base, myflags, rep, lds, wreg, st, res);
    request_phase<RD, 6>(a, layer, wreg, st);
    gemv_phase<STAMPS, RD, 6>(a, layer, gp++, base, myflags, rep, lds, wreg, st, res);
    request_phase<RD, 7>(a, layer, wreg, st);
    gemv_phase<STAMPS, RD, 7>(a, layer, gp++, base, myflags, rep, lds, wreg, st, res);
    request_phase<RD, 0>(a, layer + 1, wreg, st);
  }
  if (a.logits_out) logits_phase<STAMPS, RD>(a, gp, base, myflags, rep, lds);
  if (blockIdx.x == 0 && threadIdx.x < 64) {  // everybody has read the epoch base long ago; publish the next launch's once every workgroup is through
    wide_wait(a, myflags, base + gp, threadIdx.x);
    if (threadIdx.x == 0) a.ctrl[4] = base + gp;
  }
}

}  // namespace

// ---- host side ----------------------------------------------------------------------------------------------------------------------
size_t decode_wide_part_floats(int H) { return (size_t)H * WNS * 66; }

bool decode_wide_supports(int d, int H, int Te, int S_max, int L, int M, int nwg) {
  if (!(M == 1 && L >= 1 && L <= WMAXL && d % 64 == 0 && d == H * 64 && d <= WMAXD && H <= 32 && S_max >= 1 && S_max <= WNG * WSK && Te >= 1 &&
        (Te + WNS - 1) / WNS <= WNG * WCK && nwg >= 64 && nwg <= WMAXWG && nwg % 4 == 0 && H * WNS <= nwg))
    return false;
  if (2 * ((d + 2 * nwg - 1) / (2 * nwg)) > WPKR) return false;  // an N = d projection's rows per workgroup fill one packet
  {
    const int R0 = 2 * ((3 * d + 2 * nwg - 1) / (2 * nwg)), NP0 = (R0 + 5) / 6;  // the q | k | v row: <= 3 packets per workgroup, <= 16 packets per head and range
    if (NP0 > 3 || ((63 + R0 - 1) / R0 + 1) * NP0 > 16 || H * 11 > WPKS) return false;
  }
  // units per wave: rows R x spans J dealt to 8 waves, within what the instantiation for this R_d unrolls (wmu)
  {
    const int rd = 2 * ((d + 2 * nwg - 1) / (2 * nwg));
    const int mu0 = rd == 2 ? wmu<2, 0>() : rd == 4 ? wmu<4, 0>() : wmu<6, 0>(), mu2 = rd == 2 ? wmu<2, 2>() : rd == 4 ? wmu<4, 2>() : wmu<6, 2>();
    const int mu6 = rd == 2 ? wmu<2, 6>() : rd == 4 ? wmu<4, 6>() : wmu<6, 6>();
    auto upw = [&](int N, int K) { return (2 * ((N + 2 * nwg - 1) / (2 * nwg)) * ((K / 8 + 63) / 64) + WC - 1) / WC; };
    if (upw(3 * d, d) > mu0 || upw(d, d) > mu2 || upw(4 * d, d) > mu6 || upw(d, 4 * d) > mu6) return false;
  }
  const int Ns[3] = {3 * d, d, 4 * d}, Ks[3] = {d, 4 * d, d};
  for (int i = 0; i < 3; ++i) {
    const int R = 2 * ((Ns[i] + 2 * nwg - 1) / (2 * nwg)), J = (Ks[i] / 8 + 63) / 64;
    if (R * J > 64 || R > 64) return false;
  }
  return true;
}

int launch_decode_wide(const DecodeXcdArgs& h, hipStream_t s) {
  OASR_REQUIRE(decode_wide_supports(h.d, h.H, h.Te, h.S_max, h.L, h.M, h.team), "decode_wide: unsupported shape");
  OASR_REQUIRE(h.pos >= 0 && h.pos < h.S_max && h.layer_offsets, "decode_wide: bad launch shape");
  WArgs a;
  a.wflat = h.wflat, a.params = h.params, a.aux = h.aux, a.cache = h.cache, a.cache_lstride = h.cache_lstride;
  a.x = h.x, a.x2 = h.x2, a.x3 = h.x3, a.q = h.q, a.o = h.o, a.hg = h.hg, a.part = h.part;
  a.ctrl = h.ctrl, a.flagv = h.ctrl + 1024;  // the flag replicas start 4 KB into the cache's control tail
  a.pk = (u32x4_t*)(h.ctrl + 16384);         // ... the packets 64 KB in (5 vectors x 8 replicas x 256 x 16 bytes = 160 KB)
  a.pkqkv = a.pk + (size_t)WPKV * WREP * WPKS;  // ... and the q | k | v row's behind them (8 replicas x 768 x 16 bytes = 96 KB)
  a.d = h.d, a.H = h.H, a.Te = h.Te, a.S_max = h.S_max, a.L = h.L, a.pos = h.pos, a.nwg = h.team, a.flags = h.flags;
  a.stamps = (unsigned long long*)h.stamps;
  a.w_logits = h.w_logits, a.lnf_g = h.lnf_g, a.lnf_b = h.lnf_b, a.logits_out = h.logits_out, a.V = h.V;
  OASR_REQUIRE(!h.logits_out || (h.w_logits && h.lnf_g && h.lnf_b && h.V > 0 && (h.V + h.team - 1) / h.team <= 64 * WC), "decode_wide: bad logits phase");
  a.swg = (h.flags >> 8) & 0xff;  // (the engine passes OASR_XCD_FLAGS bits 9-16 here: the workgroup whose stamps are taken)
  {
    long* dst = &a.l0.ln1g;
    for (int i = 0; i < 18; ++i) dst[i] = (long)h.layer_offsets[i];
    a.lstride = h.lstride, a.astride = h.astride;
  }
  const int rd = 2 * ((h.d + 2 * h.team - 1) / (2 * h.team));
#define DW_LAUNCH(ST, R) hipLaunchKernelGGL((decode_wide_kernel<ST, R>), dim3(h.team), dim3(WT), 0, s, a)
  if (a.stamps) {  // (measurement: medium / small only)
    OASR_REQUIRE(rd == 4, "decode_wide: the stamped instantiation is built for 4 rows per workgroup");
    DW_LAUNCH(true, 4);
  } else if (rd == 2) DW_LAUNCH(false, 2);
  else if (rd == 4) DW_LAUNCH(false, 4);
  else DW_LAUNCH(false, 6);
#undef DW_LAUNCH
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}

// Test hooks (include/oasr_testing.h): the host-side view of the work split -- whether a shape runs on this engine, and which (row, K span) units compute
// wave `wave` (1 .. 8) of workgroup `wg` is dealt for an [N x K] projection: out[2 i] = row (absolute), out[2 i + 1] = span; returns the count, or -1
// past the instantiation's unrolled bound (wmu).  tests/test_native_abi.py checks that the units of all waves tile the matrix exactly once.
extern "C" int oasr_wide_supports_debug(int d, int H, int Te, int S_max, int L, int M, int nwg) { return decode_wide_supports(d, H, Te, S_max, L, M, nwg) ? 1 : 0; }
extern "C" int oasr_wide_plan_debug(int d, int nwg, int N, int K, int wg, int wave, int* out, int max_units) {
  WGemv p{};
  p.N = N, p.K = K, p.R = 2 * ((N + 2 * nwg - 1) / (2 * nwg));
  if (wave < 1 || wave > WC || wg < 0 || wg >= nwg) return -1;
  const WUnits q = units_of(p, wg, wave);
  const int cnt = q.U - q.u0 < q.upw ? q.U - q.u0 : q.upw;
  const int rd = 2 * ((d + 2 * nwg - 1) / (2 * nwg));
  const int ph = N == 3 * d ? 0 : (N == 4 * d ? 6 : (K == 4 * d ? 7 : 2));
  const int mu = rd == 2 ? (ph == 0 ? wmu<2, 0>() : ph == 2 ? wmu<2, 2>() : wmu<2, 6>()) : rd == 4 ? (ph == 0 ? wmu<4, 0>() : ph == 2 ? wmu<4, 2>() : wmu<4, 6>())
                                                                                                   : (ph == 0 ? wmu<6, 0>() : ph == 2 ? wmu<6, 2>() : wmu<6, 6>());
  if (cnt > mu) return -1;
  int n = 0, j = q.j0, r = q.r0;
  for (int i = 0; i < cnt && n < max_units; ++i, ++n) {
    out[2 * n] = wg * p.R + r, out[2 * n + 1] = j;
    if (++j == q.J) j = 0, ++r;
  }
  return n;
}

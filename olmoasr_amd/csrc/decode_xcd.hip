// One-launch KV-cached decoder step for a handful of sequences (B <= 4) on ONE XCD, gfx950 (round 5; since round 6 the default for one sequence is
// the chip-wide engine of decode_wide.hip, and this team engine runs where it is forced -- oasr_decode_set_ln_fold(2..4) -- or where that one does not apply)
// -- TextDecoder.forward for one new token per sequence, olmoasr/model.py:786-817 with the kv_cache hooks of :925-964
// (inference twin: olmoasr/inf_model.py:150-196, 320-362).
//
// Why: the multi-launch step (engine.hip::oasr_decode_step_impl) is ~8 dependent launches per layer, each a 4.7 us dispatch floor plus a
// chain of dependent memory round trips (weights that must first miss to HBM, a reduction, a store): 2.4 ms per token at medium = 0.05
// of the HBM roof (profiles/r02_decode_step.txt, r03_decode_xcd.txt).  What those measurements asked for is built here:
//   * ONE persistent launch; the `team` workgroups (one per CU, by default the 32 CUs of ONE XCD: blockIdx % 8 == 0) walk the 8 phases
//     of every layer and meet at a team barrier (relaxed agent-scope counter, 1.2 us inside an XCD against 4.7 us per launch and 12 us for
//     a device-wide release/acquire barrier);
//   * everything that does NOT depend on the token -- the weight tiles and the cross-attention K/V a workgroup will consume -- is a STATIC
//     per-wave stream that runs AHEAD of the phases through a wave-private LDS ring (buffer_load ... lds issued from inline assembly, one
//     counted vmcnt per block, never drained): while a workgroup waits at a barrier its next XR blocks are already landing, so a phase
//     costs one round trip on the activations instead of five;
//   * a fifth "helper" wave per workgroup owns everything that is NOT streaming: the barrier, the activation rows (LayerNorm folded into the
//     operand exactly like decode_proj.hip), the K-split reduction + epilogue and the stores -- the four streaming waves issue no other
//     vector-memory instruction, so their vmcnt counts ring blocks only.
// Data crosses workgroups ONLY through agent-scope (sc1) atomic loads / stores, so results do not depend on where the team's workgroups
// land; the XCD placement is a speed matter (recorded in ctrl[3]).  Arithmetic is decode_shared.h's: bit-identical to the multi-launch step.
#include "decode_shared.h"

namespace {

constexpr int XR = 6;        // ring slots per streaming wave
constexpr int SLOT = 4096;   // bytes per slot: 4 DMA instructions of 64 lanes x 16 B
constexpr int XMAXM = 4;     // sequences this engine takes
constexpr int XMAXL = 32;    // decoder layers
constexpr int NSEG = 7;      // streamed segments per layer (the self-attention phase streams nothing)
constexpr unsigned SPIN_LIMIT = 1u << 21;
constexpr int XMAXPAIR = 40;  // (head, key segment) pairs of one sequence: H * ns <= 40 (d <= 1280)
constexpr int XHG = 8;        // heads merged per batch of partial loads

struct XLayer {  // element offsets of one decoder layer: w* into the bf16 shadow, the rest into the fp32 parameter arena / aux region
  long ln1g, ln1b, wqkv, bqkv_aux, wo, bo, lncg, lncb, wcq, bcq, wco, bco, ln2g, ln2b, w1, b1, w2, b2;
};
struct XArgs {
  const bf16_t* wflat;
  const float* params;
  const float* aux;
  bf16_t* cache;       // per layer: self q|k|v [M, S_max, 3d] then cross k|v [M, Te, 2d]
  long cache_lstride;  // elements
  bf16_t *x, *x2, *x3, *q, *o, *hg;  // activation rows [M][d] ([M][4d] for hg), exchanged through agent-scope accesses
  float* part;         // cross-attention partials [M * H * ns][66]: m, l, o[64]
  unsigned* ctrl;      // [0] team barrier counter  [1] error flag  [2] epoch base  [3] XCC ids seen (bit mask)
  int d, H, Te, S_max, L, M, pos, team, stride;
  int flags;             // experiments: bit 0 = plain (not agent-scope) activation stores, bit 2 = no sleep in the barrier poll, bit 3 = checked
                         // instantiation, bit 4 = consume ring blocks without waiting for them, bit 5 = issue no DMA (4, 5: timing only, garbage results)
  unsigned long long* stamps;  // optional [8 phases][8] s_memtime stamps of workgroup 0 in layer 1 (null: off)
  XLayer l0;             // decoder layer 0; layer l = l0 + l * (lstride | astride): the blocks are laid out back to back (host-checked)
  long lstride, astride;
};
__host__ __device__ __forceinline__ XLayer layer_of(const XArgs& a, int l) {
  XLayer y = a.l0;
  const long s = (long)l * a.lstride;
  y.ln1g += s, y.ln1b += s, y.wqkv += s, y.wo += s, y.bo += s, y.lncg += s, y.lncb += s, y.wcq += s, y.bcq += s, y.wco += s, y.bco += s;
  y.ln2g += s, y.ln2b += s, y.w1 += s, y.b1 += s, y.w2 += s, y.b2 += s;
  y.bqkv_aux += (long)l * a.astride;
  return y;
}

// ---- the static block sequence of one workgroup (identical for its four streaming waves) ---------------------------------------------
struct XGeom {
  int d, H, Te, M, team, wg, L, ns;
  int nst_d, nst_4d;  // k16 steps per wave and tile for K = d / 4d
  int nkb_d, nkb_4d;  // ring blocks per wave and tile
  int cnt_qkv, cnt_d, cnt_4d, cnt_it;  // tiles of this workgroup in a phase of 3d/32, d/32, 4d/32 tiles; cross-attention items
};
__host__ __device__ __forceinline__ int cnt_of(int n, int wg, int team) {  // tiles t = wg, wg + team, ... < n (a short loop instead of a division)
  int c = 0;
  for (int t = wg; t < n; t += team) ++c;
  return c;
}
__host__ __device__ __forceinline__ XGeom make_geom(int d, int H, int Te, int M, int L, int team, int wg) {
  XGeom g;
  g.d = d, g.H = H, g.Te = Te, g.M = M, g.team = team, g.wg = wg, g.L = L;
  g.ns = Te > dec::SEG_KEYS ? 2 : 1;
  g.nst_d = d >> 6, g.nst_4d = d >> 4;
  g.nkb_d = (g.nst_d + 3) >> 2, g.nkb_4d = (g.nst_4d + 3) >> 2;
  g.cnt_qkv = cnt_of((3 * d) >> 5, wg, team);
  g.cnt_d = cnt_of(d >> 5, wg, team);
  g.cnt_it = cnt_of(M * H * g.ns, wg, team);
  g.cnt_4d = cnt_of((4 * d) >> 5, wg, team);
  return g;
}
// tiles (items for segment 3) of this workgroup in streamed segment `seg` (selects, not a table: the cursors live in SGPRs)
__host__ __device__ __forceinline__ int seg_cnt(const XGeom& g, int seg) { return seg == 0 ? g.cnt_qkv : seg == 3 ? g.cnt_it : seg == 5 ? g.cnt_4d : g.cnt_d; }
struct XItem {
  int b, h, sg, n, kvb;  // sequence, head, key segment, keys in it, ring blocks per K (or V) stream
};
// (no integer division anywhere near the cursors: hipcc expands a 32-bit division through VALU float reciprocals, whose result -- and
// everything computed from it, i.e. the whole block bookkeeping -- then lives in VGPRs under exec-mask branches instead of SGPRs)
__host__ __device__ __forceinline__ XItem item_of(const XGeom& g, int idx) {
  const int id = g.wg + g.team * idx;
  const int hn = g.H * g.ns;
  XItem it;
  it.b = (id >= hn ? 1 : 0) + (id >= 2 * hn ? 1 : 0) + (id >= 3 * hn ? 1 : 0);  // M <= 4 sequences
  const int rem = id - it.b * hn;
  it.h = g.ns == 2 ? rem >> 1 : rem;  // ns <= MAX_SEG = 2
  it.sg = g.ns == 2 ? rem & 1 : 0;
  int n = g.Te - it.sg * dec::SEG_KEYS;
  it.n = n > dec::SEG_KEYS ? dec::SEG_KEYS : n;
  it.kvb = (((it.n + 31) >> 5) + 3) >> 2;
  return it;
}
struct XCur {
  int layer, seg, idx, sub;
};
#if defined(__HIP_DEVICE_COMPILE__)
#define XU(x) __builtin_amdgcn_readfirstlane(x)  // pin a wave-uniform value to an SGPR (hipcc's uniformity analysis gives up on the cursors)
#else
#define XU(x) (x)
#endif
__host__ __device__ __forceinline__ int nsub_of(const XGeom& g, int seg, int idx) {
  if (seg == 3) return 2 * item_of(g, idx).kvb;
  return seg == 6 ? g.nkb_4d : g.nkb_d;
}
// the cursor's sub already points one past the unit's last full block: roll over into the next tile / item / segment if the unit ends there
__host__ __device__ __forceinline__ bool cur_normalise(const XGeom& g, XCur& c);
__host__ __device__ __forceinline__ int nsub_of(const XGeom& g, int seg, int idx);
__host__ __device__ __forceinline__ bool cur_advance_from(const XGeom& g, XCur& c) {
  if (c.sub >= nsub_of(g, c.seg, c.idx)) c.sub = 0, ++c.idx;
  const bool more = cur_normalise(g, c);
  c.layer = XU(c.layer), c.seg = XU(c.seg), c.idx = XU(c.idx), c.sub = XU(c.sub);
  return more;
}
// skip empty segments; returns false past the last layer
__host__ __device__ __forceinline__ bool cur_normalise(const XGeom& g, XCur& c) {
  while (c.layer < g.L && c.idx >= seg_cnt(g, c.seg)) {
    c.idx = 0;
    if (++c.seg == NSEG) c.seg = 0, ++c.layer;
  }
  return c.layer < g.L;
}
__host__ __device__ __forceinline__ bool cur_advance(const XGeom& g, XCur& c) {
  if (++c.sub >= nsub_of(g, c.seg, c.idx)) c.sub = 0, ++c.idx;
  const bool more = cur_normalise(g, c);
  c.layer = XU(c.layer), c.seg = XU(c.seg), c.idx = XU(c.idx), c.sub = XU(c.sub);
  return more;
}

// ---- agent-scope data accesses (coherent across workgroups wherever they run) ---------------------------------------------------------
__device__ __forceinline__ uint64_t ld8_agent(const void* p) {
  return __hip_atomic_load((const uint64_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u32x4_t ld16_agent(const void* p) {
  const uint64_t a = ld8_agent(p), b = ld8_agent((const char*)p + 8);
  u32x4_t r;
  r[0] = (unsigned)a, r[1] = (unsigned)(a >> 32), r[2] = (unsigned)b, r[3] = (unsigned)(b >> 32);
  return r;
}
__device__ __forceinline__ float ldf_agent(const float* p) {
  return __uint_as_float(__hip_atomic_load((const unsigned*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st4_agent(void* p, unsigned v) { __hip_atomic_store((unsigned*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// (experiment switch: plain stores -- they stay in this XCD's L2, which is all a team on ONE XCD needs; not valid for a spread team)
__device__ __forceinline__ void st4_x(void* p, unsigned v, int flags) {
  if (flags & 1) *(volatile unsigned*)p = v;
  else st4_agent(p, v);
}

// ---- LDS-DMA of one ring block (4 x 1 KiB), issued from inline assembly (hipcc would otherwise order every later ds_read behind it with
// s_waitcnt vmcnt(0) and drain the ring) ----------------------------------------------------------------------------------------------
__device__ __forceinline__ void dma16(const u32x4_t rs_in, unsigned lds_addr, unsigned voff, unsigned soff) {
  u32x4_t rs;  // (re-pinned: a descriptor that travelled through a loop-carried struct comes back as VGPRs)
  rs[0] = __builtin_amdgcn_readfirstlane(rs_in[0]), rs[1] = __builtin_amdgcn_readfirstlane(rs_in[1]);
  rs[2] = 0x7fffffffu, rs[3] = 0x00020000u;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(__builtin_amdgcn_readfirstlane(lds_addr)), "v"(voff), "s"(rs),
               "s"(__builtin_amdgcn_readfirstlane(soff))
               : "memory");
}
// (descriptor already in SGPRs)
__device__ __forceinline__ void dma16_s(const u32x4_t rs_in, unsigned lds_addr, unsigned voff, unsigned soff) {
  u32x4_t rs;
  rs[0] = __builtin_amdgcn_readfirstlane(rs_in[0]), rs[1] = __builtin_amdgcn_readfirstlane(rs_in[1]);
  rs[2] = 0x7fffffffu, rs[3] = 0x00020000u;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(__builtin_amdgcn_readfirstlane(lds_addr)), "v"(voff), "s"(rs),
               "s"(__builtin_amdgcn_readfirstlane(soff))
               : "memory");
}
__device__ __forceinline__ u32x4_t rsrc_of(const void* base) {
  const unsigned long addr = (unsigned long)base;
  u32x4_t rs;  // (readfirstlane: the descriptor must sit in SGPRs; the inline-asm "s" constraint does not move it there by itself)
  rs[0] = __builtin_amdgcn_readfirstlane((unsigned)addr);
  rs[1] = __builtin_amdgcn_readfirstlane((unsigned)(addr >> 32) & 0xffffu);  // stride 0: raw buffer
  rs[2] = 0x7fffffffu;
  rs[3] = 0x00020000u;
  return rs;
}
#define XWAIT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define XBAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// ---- the producer: a table of UNITS per wave ------------------------------------------------------------------------------------------
// A unit = a run of ring blocks that share one descriptor and one set of per-lane offsets: all tiles of one projection segment that belong
// to this workgroup (consecutive tiles lie a constant distance apart), the full blocks of one item's K (or V) stream, or the one clamped
// block that ends a stream.  The units of decoder layer 0 are listed ONCE per launch by walking the cursor (the CPU-tested sequence) and
// kept in a wave-private LDS table; layer l adds l x (layer stride) to the soffset.  Issuing a block is then: 4 DMA instructions, three
// scalar updates, and -- at a unit's end -- one 16-byte LDS read.  (Versions 1-5 advanced the cursor per block: ~2 k cycles of single-wave
// instruction issue at every unit boundary, profiles/r05_decode_xcd_stamps_v5.txt.)
constexpr int XTABN = 32;  // units per layer and wave (6 projection segments + 4 per cross-attention item, <= 5 items)
struct XStream {  // per streaming wave (all fields wave-uniform, SGPR-resident)
  bool more;      // the producer has blocks left
  int issued, consumed;
  int islot, cslot;  // issued % XR, consumed % XR, kept incrementally
  int u, nunits, layer;  // current unit, units per layer, layer of the block to issue next
  unsigned soff, jump;   // soffset of the next block; step from a tile's last block to this workgroup's next tile (weights)
  int left, kind;        // blocks left in the unit; lane-offset set: 0 weights K = d, 1 weights K = 4d, 2 cross K/V, 3 cross K/V last (clamped) block
  int tleft, tnkb;       // blocks left in the tile being issued (incl. the next one), blocks per tile
};
// per-lane byte offsets that do not depend on the block (computed once per wave): the DMA source offsets of a weight block for K = d / K = 4d,
// of a K/V block, of the clamped last K/V block of the partial key segment, and the LDS fragment offsets of the 4 k16 steps of a weight block
struct XLane {
  unsigned w_d[4], w_4d[4], kv[4], kvc[4], frag[4];
};
__device__ __forceinline__ XLane make_lane(int d, int Te, int wave, int lane) {
  XLane v;
  const int r8 = lane >> 3, l8 = lane & 7, row = lane & 31, kc = lane >> 5;
  const int n_p = Te - (Te > dec::SEG_KEYS ? dec::SEG_KEYS : 0);  // keys of the last (possibly partial) segment
  const int last = n_p - 1 - 128 * ((n_p - 1) >> 7);              // last valid key of its last block, relative to that block
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c16 = l8 ^ ((q * 4 + (r8 >> 1)) & 7);
    v.w_d[q] = (unsigned)(((q * 8 + r8) * d + c16 * 8) * 2);
    v.w_4d[q] = (unsigned)(((q * 8 + r8) * 4 * d + c16 * 8) * 2);
    const int t = 32 * q + 8 * wave + r8;
    v.kv[q] = (unsigned)((t * 2 * d + l8 * 8) * 2);
    v.kvc[q] = (unsigned)(((t < last ? t : last) * 2 * d + l8 * 8) * 2);  // keys past the segment's end re-read its last key (never used)
    v.frag[q] = (unsigned)(row * 128 + (((2 * q + kc) ^ ((row >> 1) & 7)) << 4));
  }
  return v;
}

// List the units of decoder layer 0 for this wave into its LDS table (4 words each: soffset, blocks, kind | blocks per tile << 8, tile jump);
// returns their number.  Runs once per launch; walks the same cursor functions the CPU test pins.
struct XBuild {  // what the unit walk needs, BY VALUE (a reference to the kernel arguments would pull them out of SGPRs into scratch memory)
  int d, H, Te, M, L, team, wg;
  long w[6];  // layer 0 weight offsets by streamed segment: qkv, attn.out, cross q, (3 unused), cross out, mlp.0, mlp.2 -> indices 0,1,2,3,4,5
};
__device__ __attribute__((noinline)) int build_units(XBuild p, int wave, int lane, unsigned* tab) {
  const XGeom g = make_geom(p.d, p.H, p.Te, p.M, p.L, p.team, p.wg);
  XCur c{0, 0, 0, 0};
  bool more = cur_normalise(g, c);
  int n = 0;
  while (more && c.layer == 0 && n < XTABN) {
    unsigned soff, jump = 0;
    int nblk, kind, tnkb = 0x400000;
    if (c.seg == 3) {
      const XItem it = item_of(g, c.idx);
      const bool is_v = c.sub >= it.kvb;
      const int j = is_v ? c.sub - it.kvb : c.sub;
      soff = (unsigned)((((long)it.b * p.Te + (long)it.sg * dec::SEG_KEYS + 128 * j) * 2 * p.d + it.h * 64 + (is_v ? p.d : 0)) * 2);
      const int nfull = it.n >> 7;
      if (nfull > j) nblk = nfull - j, kind = 2;
      else nblk = 1, kind = 3;
      c.sub += nblk;
      more = cur_advance_from(g, c);
    } else {
      const long woff = p.w[c.seg < 3 ? c.seg : c.seg - 1];
      const int K = c.seg == 6 ? 4 * p.d : p.d, nkb = c.seg == 6 ? g.nkb_4d : g.nkb_d;
      const int tile = g.wg + g.team * c.idx;
      soff = (unsigned)((unsigned long)((woff + ((long)tile * 32) * K + (long)wave * (K >> 2)) * 2));  // relative to the arena's bf16 shadow
      nblk = (seg_cnt(g, c.seg) - c.idx) * nkb, kind = c.seg == 6 ? 1 : 0, tnkb = nkb;
      jump = (unsigned)((long)g.team * 32 * K * 2 - (long)(nkb - 1) * 128);
      c.idx = seg_cnt(g, c.seg), c.sub = 0;
      more = cur_normalise(g, c);
    }
    if (lane == 0) {
      tab[4 * n] = soff, tab[4 * n + 1] = (unsigned)nblk, tab[4 * n + 2] = (unsigned)kind | ((unsigned)tnkb << 8), tab[4 * n + 3] = jump;
    }
    ++n;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (read back by this wave only)
  return n;
}
// load unit st.u of layer st.layer into the producer state
__device__ __forceinline__ void unit_load(const XArgs& a, XStream& st, const unsigned* tab) {
  const u32x4_t e = *(const u32x4_t*)(tab + 4 * st.u);
  const int kind = XU(e[2] & 0xffu);
  const long stride = kind >= 2 ? a.cache_lstride : a.lstride;
  st.soff = XU(e[0] + (unsigned)((long)st.layer * stride * 2));
  st.left = XU(e[1]);
  st.kind = kind;
  st.tnkb = XU(e[2] >> 8);
  st.tleft = st.tnkb;
  st.jump = XU(e[3]);
}
// after a block has been issued: next block of the unit, next tile, or next unit / layer
__device__ __forceinline__ void unit_advance(const XArgs& a, XStream& st, const unsigned* tab) {
  st.left = XU(st.left - 1);
  if (st.left == 0) {
    st.u = XU(st.u + 1);
    if (st.u == st.nunits) st.u = 0, st.layer = XU(st.layer + 1);
    if (st.layer == a.L) st.more = false;
    else unit_load(a, st, tab);
  } else if (st.tleft == 1) {
    st.soff = XU(st.soff + st.jump), st.tleft = st.tnkb;  // the tile's last block: on to this workgroup's next tile of the segment
  } else {
    st.soff = XU(st.soff + (st.kind >= 2 ? (unsigned)(128 * 2 * a.d * 2) : 128u)), st.tleft = XU(st.tleft - 1);
  }
}
// the four DMA instructions of one block
__device__ __forceinline__ void issue_block(const XStream& st, const XLane& lv, const u32x4_t rs_w, const u32x4_t rs_kv, unsigned dst) {
  const int kind = XU(st.kind);
  const unsigned soff = XU(st.soff);
  if (kind == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) dma16_s(rs_w, dst + q * 1024, lv.w_d[q], soff);
  } else if (kind == 1) {
#pragma unroll
    for (int q = 0; q < 4; ++q) dma16_s(rs_w, dst + q * 1024, lv.w_4d[q], soff);
  } else if (kind == 2) {
#pragma unroll
    for (int q = 0; q < 4; ++q) dma16_s(rs_kv, dst + q * 1024, lv.kv[q], soff);
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) dma16_s(rs_kv, dst + q * 1024, lv.kvc[q], soff);
  }
}
struct XProd {  // launch constants of the producer
  u32x4_t rs_w, rs_kv;  // descriptors: the bf16 shadow of the parameter arena; layer 0's cross K/V
  const unsigned* tab;  // this wave's unit table (LDS)
  unsigned ring_lds;    // LDS byte address of this wave's ring
};
// issue the next block into ring slot islot and advance
__device__ __forceinline__ void ring_issue(const XArgs& a, XStream& st, const XLane& lv, const XProd& pr) {
  if (!st.more) return;
  issue_block(st, lv, pr.rs_w, pr.rs_kv, pr.ring_lds + (unsigned)st.islot * SLOT);
  st.issued = XU(st.issued + 1);
  st.islot = XU(st.islot + 1 == XR ? 0 : st.islot + 1);
  unit_advance(a, st, pr.tab);
}
// block `consumed` has landed: at most (issued - consumed - 1) younger blocks may still be in flight (loads return in order)
__device__ __forceinline__ void ring_wait(const XStream& st, int = 0) {
  const int younger = XU(st.issued - st.consumed - 1);
  if (younger >= 5) XWAIT_VM(20);
  else if (younger == 4) XWAIT_VM(16);
  else if (younger == 3) XWAIT_VM(12);
  else if (younger == 2) XWAIT_VM(8);
  else if (younger == 1) XWAIT_VM(4);
  else XWAIT_VM(0);
}

// ---- team barrier (helper wave) ----------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void team_arrive(unsigned* ctrl) {
  XWAIT_VM(0);  // this workgroup's stores of the phase have been acknowledged
  if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(ctrl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void team_wait(unsigned* ctrl, unsigned target, int flags = 0) {
  if ((threadIdx.x & 63) == 0) {  // one lane polls; the wave reconverges behind it
    unsigned spins = 0;
    while ((int)(__hip_atomic_load(ctrl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
      if (!(flags & 4)) __builtin_amdgcn_s_sleep(1);
      if (++spins > SPIN_LIMIT || ((spins & 63) == 0 && __hip_atomic_load(ctrl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
        __hip_atomic_fetch_or(ctrl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // a team member never arrived: poison, do not hang
        break;
      }
    }
  }
}

// dynamic LDS block, byte offsets (compile-time: d <= XMAXD fixes the operand row stride)
constexpr int XMAXD = 1280;
constexpr unsigned XL_RING = 0;
constexpr unsigned XL_XS = 4 * XMAXD * 2 + 16;  // row stride of the activation operand: K_max = 4d, + 16 so the rows sit on different banks
constexpr unsigned XL_XBUF = XL_RING + 4 * XR * SLOT;
constexpr unsigned XL_RED = XL_XBUF + XMAXM * XL_XS;
constexpr unsigned XL_SC = XL_RED + 4 * 4 * 16 * XMAXM * 2 * 4;  // NRED reduction buffers
constexpr unsigned XL_ARED = XL_SC + dec::SEG_KEYS * 4;  // (sc | ared contiguous: the cross-attention merge stages the partials there)
constexpr unsigned XL_LSUM = XL_ARED + 32 * 64 * 4;
constexpr unsigned XL_WMAX = XL_LSUM + 32 * 4;
constexpr unsigned XL_AOUT = XL_WMAX + 4 * 4;
constexpr unsigned XL_TAB = (XL_AOUT + 66 * 4 + 8 + 15) & ~15u;  // 4 waves x XTABN units x 16 bytes
constexpr unsigned XL_TOTAL = XL_TAB + 4 * XTABN * 16;
static_assert(XL_TOTAL <= 160 * 1024, "one-launch decode step: LDS budget");

struct XGemv {  // one projection phase
  int seg;
  const bf16_t* xin;    // activation rows [M][K] (global)
  int K, N;
  const float *ln_g, *ln_b;  // LayerNorm folded into the operand, or null
  bool merge_attn;      // the operand is the merged cross-attention output (a.part), not xin
  const float* bias;
  bool gelu;
  const bf16_t* resid;  // [M][d] or null
  bf16_t* out;
  long ldc;
};

__device__ __forceinline__ float rdlane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// helper wave: activation rows of a projection phase -> LDS operand rows (bf16).  Every global load of a row is issued before the first
// one is used: ONE round trip per row, not one per chunk.
constexpr int XCH = 4;  // 16-byte chunks per lane and batch of a plain operand row (register budget: the kernel sits at 254 of 256 VGPRs)
__device__ __forceinline__ void helper_operand(const XArgs& a, const XGeom& g, const XGemv& ph, char* smem, int lane) {
  if (ph.merge_attn) {  // rows = merge of the cross-attention segments' partials (decode_shared.h), dimension `lane` of every head
    // ONE round trip per row: lane j < H * ns fetches (m, l) of pair j, every lane its dimension of every pair's o; the (m, l) of a pair
    // reach all lanes by a wave shuffle
    const int npair = a.H * g.ns;  // <= 40 (decode_xcd_supports)
    for (int b = 0; b < a.M; ++b) {
      const float* src = a.part + (long)b * npair * 66;
      const int jl = lane < npair ? lane : npair - 1;
      const float m_l = ldf_agent(src + jl * 66), l_l = ldf_agent(src + jl * 66 + 1);
      for (int h0 = 0; h0 < a.H; h0 += XHG) {  // XHG heads per batch of loads (register budget)
        float o[XHG * dec::MAX_SEG];
#pragma unroll
        for (int j = 0; j < XHG * dec::MAX_SEG; ++j) {
          const int pj = h0 * g.ns + j;
          o[j] = ldf_agent(src + (pj < npair ? pj : npair - 1) * 66 + 2 + lane);  // (unconditional: one batch of loads)
        }
#pragma unroll
        for (int hh = 0; hh < XHG; ++hh) {
          const int h = h0 + hh;
          if (h < a.H) {
            float m_s[dec::MAX_SEG] = {dec::NEG, dec::NEG}, l_s[dec::MAX_SEG] = {0.f, 0.f}, o_s[dec::MAX_SEG] = {0.f, 0.f};
            // (m, l) of a pair sit in lane `pair` of m_l / l_l: v_readlane (a scalar broadcast), not a ds_bpermute round trip
            if (g.ns == 2) {
              m_s[0] = rdlane(m_l, 2 * h), m_s[1] = rdlane(m_l, 2 * h + 1);
              l_s[0] = rdlane(l_l, 2 * h), l_s[1] = rdlane(l_l, 2 * h + 1);
              o_s[0] = o[2 * hh], o_s[1] = o[2 * hh + 1];
            } else {
              m_s[0] = rdlane(m_l, h), l_s[0] = rdlane(l_l, h), o_s[0] = o[hh];
            }
            float m, lt;
            const float val = dec::merge_segments(m_s, l_s, o_s, g.ns, m, lt);
            const float nb = dec::xor_lane<1>(val);
            if ((lane & 1) == 0) *(uint32_t*)(smem + XL_XBUF + b * XL_XS + (h * 64 + lane) * 2) = pack_bf2(val, nb);
          }
        }
      }
    }
    return;
  }
  const int nchunk = ph.K >> 3;
  for (int b = 0; b < a.M; ++b) {
    if (ph.ln_g) {  // K == d <= 2048: the row fits MAXC chunks per lane
      u32x4_t raw[dec::MAXC];
#pragma unroll
      for (int c = 0; c < dec::MAXC; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nchunk) raw[c] = ld16_agent(ph.xin + (long)b * ph.K + ch * 8);
      }
      float mean, rstd;
      dec::row_stats(raw, lane, nchunk, ph.K, mean, rstd);
#pragma unroll
      for (int c = 0; c < dec::MAXC; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nchunk) *(u32x4_t*)(smem + XL_XBUF + b * XL_XS + ch * 16) = dec::ln_apply8(raw[c], mean, rstd, ph.ln_g, ph.ln_b, ch * 8);
      }
    } else {
      for (int c0 = 0; c0 * 64 < nchunk; c0 += XCH) {  // (K = 4096: two batches)
        u32x4_t raw[XCH];
#pragma unroll
        for (int c = 0; c < XCH; ++c)
          if (lane + 64 * (c0 + c) < nchunk) raw[c] = ld16_agent(ph.xin + (long)b * ph.K + (lane + 64 * (c0 + c)) * 8);
#pragma unroll
        for (int c = 0; c < XCH; ++c)
          if (lane + 64 * (c0 + c) < nchunk) *(u32x4_t*)(smem + XL_XBUF + b * XL_XS + (lane + 64 * (c0 + c)) * 16) = raw[c];
      }
    }
  }
}
// helper wave: query rows of an attention phase -> LDS
__device__ __forceinline__ void helper_query_rows(const XArgs& a, const bf16_t* src, long row_stride, char* smem, int lane) {
  const int nchunk = a.d >> 3;
  for (int b = 0; b < a.M; ++b) {
    u32x4_t raw[dec::MAXC];
#pragma unroll
    for (int c = 0; c < dec::MAXC; ++c)
      if (lane + 64 * c < nchunk) raw[c] = ld16_agent(src + (long)b * row_stride + (lane + 64 * c) * 8);
#pragma unroll
    for (int c = 0; c < dec::MAXC; ++c)
      if (lane + 64 * c < nchunk) *(u32x4_t*)(smem + XL_XBUF + b * XL_XS + (lane + 64 * c) * 16) = raw[c];
  }
}

// helper wave: K-split reduction + epilogue + store of one 32-column tile (lane: row m = lane >> 4, columns 2 (lane & 15), + 1).
// The residual pair and the two biases do not depend on the tile's accumulators: they are requested BEFORE the workgroup barrier that
// releases the accumulators (helper_epilogue_inputs), so the epilogue itself is LDS reads + one store.
struct XEpiIn {
  unsigned rz;
  float b0, b1;
};
__device__ __forceinline__ XEpiIn helper_epilogue_inputs(const XArgs& a, const XGemv& ph, int n0, int lane) {
  const int m = lane >> 4, cp = lane & 15;
  XEpiIn in{0u, 0.f, 0.f};
  if (m >= a.M) return in;
  if (ph.resid) in.rz = __hip_atomic_load((const unsigned*)(ph.resid + (long)m * a.d + n0 + 2 * cp), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (ph.bias) in.b0 = ph.bias[n0 + 2 * cp], in.b1 = ph.bias[n0 + 2 * cp + 1];
  return in;
}
__device__ __forceinline__ void helper_epilogue(const XArgs& a, const XGemv& ph, const XEpiIn& in, const char* smem, int buf, int n0,
                                                int lane) {
  const int m = lane >> 4, cp = lane & 15;
  if (m >= a.M) return;
  const float* red = (const float*)(smem + XL_RED) + buf * (4 * 16 * XMAXM * 2);
  float y[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int c = 2 * cp + e;
    const int r = (c >> 3) * 4 + (c & 3), h = (c >> 2) & 1;
    const int idx = (r * XMAXM + m) * 2 + h;
    const float acc = red[0 * 16 * XMAXM * 2 + idx] + red[1 * 16 * XMAXM * 2 + idx] + red[2 * 16 * XMAXM * 2 + idx] + red[3 * 16 * XMAXM * 2 + idx];
    y[e] = dec::epi_value(acc, e ? in.b1 : in.b0, ph.gelu, ph.resid != nullptr, e ? bf_hi(in.rz) : bf_lo(in.rz));
  }
  st4_x(ph.out + (long)m * ph.ldc + n0 + 2 * cp, pack_bf2(y[0], y[1]), a.flags);
}

// ---- fast run: `run` consecutive ring blocks during which nothing special happens on either side -- the consumer's blocks are full, the
// producer stays inside its current unit (st.left full blocks with one descriptor / lane-offset set) and the ring is in its steady state
// (XR blocks in flight: the block consumed and the block issued share a slot).  Per block: one counted wait, the consumer's body on the
// slot, four DMA instructions, a handful of scalar updates -- no cursor arithmetic (the general path costs ~1.7 k cycles of
// instruction issue per block, profiles/r05_decode_xcd_stamps_v3_instruction_bound.txt).
__device__ __forceinline__ int fast_run_len(const XStream& st, int consumer_full_left) {
  return (consumer_full_left > 0 && st.more && st.issued - st.consumed == XR) ? consumer_full_left : 0;
}
template <typename F>
__device__ __forceinline__ void fast_run(const XArgs& a, XStream& st, const XLane& lv, const XProd& pr, char* smem, int wave, int run, F&& body) {
  int i = 0;
  for (; i < run && st.more; ++i) {
    XWAIT_VM(20);  // 4 * (XR - 1): the oldest block in flight has landed
    body(smem + XL_RING + (wave * XR + st.cslot) * SLOT, i);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the body's reads of the slot have retired: it may be re-staged
    issue_block(st, lv, pr.rs_w, pr.rs_kv, pr.ring_lds + (unsigned)st.cslot * SLOT);  // (steady state: islot == cslot)
    st.cslot = XU(st.cslot + 1 == XR ? 0 : st.cslot + 1);
    unit_advance(a, st, pr.tab);
  }
  st.islot = st.cslot;
  st.issued = XU(st.issued + i), st.consumed = XU(st.consumed + i);
  // the producer ran dry inside the run (the step's last blocks): the rest of the run is consumed without issuing
  for (; i < run; ++i) {
    ring_wait(st, a.flags);
    body(smem + XL_RING + (wave * XR + st.cslot) * SLOT, i);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    st.consumed = XU(st.consumed + 1);
    st.cslot = XU(st.cslot + 1 == XR ? 0 : st.cslot + 1);
  }
}

// streaming waves: the tiles of one projection phase, back to back.  A tile = this wave's K quarter of 32 weight rows against the operand
// rows, through the ring; its 16 accumulators go to one of NRED reduction buffers.  The helper wave is met once per GROUP of NRED tiles
// (every phase of every model up to d = 1024 is one group), not once per tile.
constexpr int NRED = 4;
template <bool DBG>
__device__ __forceinline__ void stream_phase(const XArgs& a, const XGeom& g, XStream& st, XCur& cc, const XLane& lv, const XProd& pr, const XGemv& ph,
                                             char* smem, int wave, int lane, int ntile) {
  const int nst = ph.seg == 6 ? g.nst_4d : g.nst_d, nkb = ph.seg == 6 ? g.nkb_4d : g.nkb_d;
  const bool allfull = (nst & 3) == 0;
  const int row = lane & 31, kc = lane >> 5;
  const int kq0 = wave * (ph.K >> 2);
  const char* xrow = smem + XL_XBUF + (row & (XMAXM - 1)) * XL_XS + (kq0 + kc * 8) * 2;
  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  int tile = 0, kb = 0;
  auto tile_done = [&]() {  // accumulators -> reduction buffer (tile & (NRED - 1)); next tile
    if (row < XMAXM) {
      float* red = (float*)(smem + XL_RED) + (tile & (NRED - 1)) * (4 * 16 * XMAXM * 2) + wave * (16 * XMAXM * 2);
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(r * XMAXM + row) * 2 + kc] = acc[r];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    kb = 0;
    ++tile;
  };
  while (tile < ntile) {
    int gend = (tile & ~(NRED - 1)) + NRED;  // end of this group of tiles
    gend = gend < ntile ? gend : ntile;
    const int cfull = allfull ? (gend - tile) * nkb - kb : (nst >> 2) - kb;  // full blocks ahead of the consumer before anything special
    const int run = DBG ? 0 : fast_run_len(st, cfull);
    if (run > 0) {
      fast_run(a, st, lv, pr, smem, wave, run, [&](const char* slot, int) {
        bf16x8_t wf[4], xf[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          wf[q] = *(const bf16x8_t*)(slot + lv.frag[q]);
          xf[q] = *(const bf16x8_t*)(xrow + kb * 128 + q * 32);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[q], xf[q], acc, 0, 0, 0);
        if (++kb == nkb) tile_done();
      });
    } else {
      if (DBG) {
        if (cc.seg != ph.seg || cc.idx != tile || cc.sub != kb) __hip_atomic_fetch_or(a.ctrl + 1, 0x100u | (unsigned)ph.seg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cur_advance(g, cc);
      }
      ring_wait(st, a.flags);
      const char* slot = smem + XL_RING + (wave * XR + st.cslot) * SLOT;
      int steps = nst - kb * 4;
      steps = steps > 4 ? 4 : steps;
      for (int q = 0; q < steps; ++q) {
        const bf16x8_t wf = *(const bf16x8_t*)(slot + row * 128 + (((2 * q + kc) ^ ((row >> 1) & 7)) << 4));
        const bf16x8_t xf = *(const bf16x8_t*)(xrow + (kb * 4 + q) * 32);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, acc, 0, 0, 0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slot's reads have retired before it is re-staged
      st.consumed = XU(st.consumed + 1);
      st.cslot = XU(st.cslot + 1 == XR ? 0 : st.cslot + 1);
      ring_issue(a, st, lv, pr);
      if (++kb == nkb) tile_done();
    }
    if (kb == 0 && (tile == ntile || (tile & (NRED - 1)) == 0)) {  // a group is complete: hand its accumulators to the helper wave
      XBAR();
      if (tile < ntile) XBAR();  // ... and wait until it has read them before the next group overwrites the buffers
    }
  }
}

__device__ __forceinline__ XGemv gemv_of(const XArgs& a, int layer, int ph) {
  XLayer ly = layer_of(a, layer);
  bf16_t* self = a.cache + (long)layer * a.cache_lstride;
  XGemv p;
  p.merge_attn = false, p.gelu = false, p.ln_g = p.ln_b = nullptr, p.resid = nullptr, p.K = a.d, p.N = a.d, p.ldc = a.d;
  switch (ph) {
    case 0:  // attn_ln -> q | k | v of position pos, straight into the cache row
      p.seg = 0, p.xin = a.x, p.N = 3 * a.d, p.ln_g = a.params + ly.ln1g, p.ln_b = a.params + ly.ln1b, p.bias = a.aux + ly.bqkv_aux;
      p.out = self + (long)a.pos * 3 * a.d, p.ldc = (long)a.S_max * 3 * a.d;
      break;
    case 2:  // self-attention output projection + residual
      p.seg = 1, p.xin = a.o, p.bias = a.params + ly.bo, p.resid = a.x, p.out = a.x2;
      break;
    case 3:  // cross_attn_ln -> cross query
      p.seg = 2, p.xin = a.x2, p.ln_g = a.params + ly.lncg, p.ln_b = a.params + ly.lncb, p.bias = a.params + ly.bcq, p.out = a.q;
      break;
    case 5:  // cross-attention output projection + residual (operand = merged segment partials)
      p.seg = 4, p.xin = nullptr, p.merge_attn = true, p.bias = a.params + ly.bco, p.resid = a.x2, p.out = a.x3;
      break;
    case 6:  // mlp_ln -> mlp.0 + GELU
      p.seg = 5, p.xin = a.x3, p.N = 4 * a.d, p.ln_g = a.params + ly.ln2g, p.ln_b = a.params + ly.ln2b, p.bias = a.params + ly.b1, p.gelu = true;
      p.out = a.hg, p.ldc = 4 * a.d;
      break;
    default:  // 7: mlp.2 + residual
      p.seg = 6, p.xin = a.hg, p.K = 4 * a.d, p.bias = a.params + ly.b2, p.resid = a.x3, p.out = a.x;
      break;
  }
  return p;
}
// measurement: s_memtime of workgroup 0's helper wave (slots 0-4: phase start, barrier passed, operand / query rows in LDS, last tile stored,
// arrived) and of its streaming wave 0 (5, 6: released, last tile handed over) in decoder layer 1
#define XSTAMP(K)                                                                                          \
  do {                                                                                                     \
    if (a.stamps && wg == 0 && layer == 1 && lane == 0) a.stamps[ph * 8 + (K)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
template <bool DBG>  // DBG: the consumer checks every block against a mirror of the producer's cursor (tests; costs scalar work per block)
__global__ __launch_bounds__(320) void decode_xcd_kernel(XArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  if (a.stride == 8 && (blockIdx.x & 7) != 0) return;
  const int wg = a.stride == 8 ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool helper = wave == 4;
  XGeom g = make_geom(a.d, a.H, a.Te, a.M, a.L, a.team, wg);
  g.cnt_qkv = __builtin_amdgcn_readfirstlane(g.cnt_qkv), g.cnt_d = __builtin_amdgcn_readfirstlane(g.cnt_d);
  g.cnt_4d = __builtin_amdgcn_readfirstlane(g.cnt_4d), g.cnt_it = __builtin_amdgcn_readfirstlane(g.cnt_it);
  const unsigned base = a.ctrl[2];  // epoch: the counter is never reset, every launch adds (phases x team) to it
  const unsigned ring_lds = (unsigned)(size_t)(smem + XL_RING) + (unsigned)wave * XR * SLOT;
  unsigned gphase = 0;  // phases completed by the whole team before the current one
  const XLane lv = make_lane(a.d, a.Te, wave, lane);

  XStream st;
  XCur cc;  // consumer's mirror of the block sequence (checked instantiation: a desynchronised producer is reported, not silently consumed)
  st.issued = st.consumed = 0, st.islot = st.cslot = 0;
  st.u = 0, st.nunits = 0, st.layer = 0, st.soff = 0, st.jump = 0, st.left = 0, st.kind = 0, st.tleft = st.tnkb = 0x400000;
  cc = XCur{0, 0, 0, 0};
  XProd pr;
  pr.tab = (const unsigned*)(smem + XL_TAB) + (wave & 3) * (XTABN * 4);
  pr.ring_lds = ring_lds;
  pr.rs_w = rsrc_of(a.wflat);
  pr.rs_kv = rsrc_of(a.cache + (long)3 * a.M * a.S_max * a.d);
  if (!helper) {
    {
      XBuild bp;
      bp.d = a.d, bp.H = a.H, bp.Te = a.Te, bp.M = a.M, bp.L = a.L, bp.team = a.team, bp.wg = wg;
      bp.w[0] = a.l0.wqkv, bp.w[1] = a.l0.wo, bp.w[2] = a.l0.wcq, bp.w[3] = a.l0.wco, bp.w[4] = a.l0.w1, bp.w[5] = a.l0.w2;
      st.nunits = XU(build_units(bp, wave, lane, (unsigned*)(smem + XL_TAB) + wave * (XTABN * 4)));
    }
    st.more = st.nunits > 0 && st.nunits < XTABN;  // (a full table means the walk was cut short: never the case for a supported shape)
    if (st.nunits >= XTABN && lane == 0) __hip_atomic_fetch_or(a.ctrl + 1, 0x200u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (st.more) unit_load(a, st, pr.tab);
    cur_normalise(g, cc);
    for (int i = 0; i < XR; ++i) ring_issue(a, st, lv, pr);
  } else {
    st.more = false;
    if (lane == 0) {
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      __hip_atomic_fetch_or(a.ctrl + 3, 1u << (xcc & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int i = lane; i < (int)(XMAXM * XL_XS / 4); i += 64) ((unsigned*)(smem + XL_XBUF))[i] = 0u;  // operand rows >= M read as zeros
  }

#pragma unroll 1
  for (int layer = 0; layer < a.L; ++layer) {
#pragma unroll 1
    for (int ph = 0; ph < 8; ++ph, ++gphase) {
      const unsigned target = base + gphase * (unsigned)a.team;
      if (ph == 1 || ph == 4) {
        // ---------------- attention: items (b, h[, key segment]) of this workgroup ----------------
        const bool cross = ph == 4;
        const int nitem = cross ? g.cnt_it : cnt_of(a.M * a.H, wg, a.team);
        const bf16_t* self = a.cache + (long)layer * a.cache_lstride;
        if (helper) {
          XSTAMP(0);
          team_wait(a.ctrl, target, a.flags);
          XSTAMP(1);
          // query rows -> LDS (self: the q third of cache row pos; cross: the cross query)
          if (cross) helper_query_rows(a, a.q, a.d, smem, lane);
          else helper_query_rows(a, self + (long)a.pos * 3 * a.d, (long)a.S_max * 3 * a.d, smem, lane);
          XSTAMP(2);
          XBAR();  // A
          for (int j = 0; j < nitem; ++j) {
            XBAR();  // 1
            XBAR();  // 2
            XBAR();  // 3: the item's result is in XL_AOUT
            const float* ao = (const float*)(smem + XL_AOUT);
            if (cross) {
              const XItem it = item_of(g, j);
              float* dst = a.part + ((long)(it.b * a.H + it.h) * g.ns + it.sg) * 66;
              st4_x(dst + 2 + lane, __float_as_uint(ao[2 + lane]), a.flags);
              if (lane < 2) st4_x(dst + lane, __float_as_uint(ao[lane]), a.flags);
            } else {
              const int id = wg + a.team * j, b = (id >= a.H ? 1 : 0) + (id >= 2 * a.H ? 1 : 0) + (id >= 3 * a.H ? 1 : 0), h = id - b * a.H;
              const float val = ao[2 + lane], nb = dec::xor_lane<1>(val);
              if ((lane & 1) == 0) st4_x(a.o + (long)b * a.d + h * 64 + lane, pack_bf2(val, nb), a.flags);
            }
          }
          XSTAMP(3);
          team_arrive(a.ctrl);
          XSTAMP(4);
        } else {
          XBAR();  // A
          float* sc = (float*)(smem + XL_SC);
          float(*ared)[64] = (float(*)[64])(smem + XL_ARED);
          float* lsum = (float*)(smem + XL_LSUM);
          float* wmax = (float*)(smem + XL_WMAX);
          float* ao = (float*)(smem + XL_AOUT);
          const int l8 = tid & 7, grp = tid >> 3;
          for (int j = 0; j < nitem; ++j) {
            int b, h, n;
            XItem it;
            if (cross) {
              it = item_of(g, j);
              b = it.b, h = it.h, n = it.n;
            } else {
              const int id = wg + a.team * j;
              b = (id >= a.H ? 1 : 0) + (id >= 2 * a.H ? 1 : 0) + (id >= 3 * a.H ? 1 : 0), h = id - b * a.H, n = a.pos + 1;
            }
            float qv[8];
            dec::load_q8(*(const u32x4_t*)(smem + XL_XBUF + b * XL_XS + (h * 64 + l8 * 8) * 2), qv);
            float mx = dec::NEG, l = 0.f, o[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) o[jj] = 0.f;
            if (cross) {
              // K blocks, then V blocks, of this wave's keys (t == 8 wave + (lane >> 3) mod 32) through the ring
              for (int kb = 0; kb < it.kvb;) {
                const int run = DBG ? 0 : fast_run_len(st, (n >> 7) - kb);  // blocks whose 128 keys all lie inside the segment
                if (run > 0) {
                  fast_run(a, st, lv, pr, smem, wave, run, [&](const char* slot, int i) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                      const int t = 32 * ((kb + i) * 4 + q) + grp;
                      const float s2 = dec::score8(qv, *(const u32x4_t*)(slot + q * 1024 + lane * 16));
                      if (l8 == 0) sc[t] = s2;
                      mx = fmaxf(mx, s2);
                    }
                  });
                  kb += run;
                  continue;
                }
                if (DBG) {
                  if (cc.seg != 3 || cc.idx != j || cc.sub != kb) __hip_atomic_fetch_or(a.ctrl + 1, 0x103u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  cur_advance(g, cc);
                }
                ring_wait(st, a.flags);
                const char* slot = smem + XL_RING + (wave * XR + st.cslot) * SLOT;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  const int t = 32 * (kb * 4 + q) + grp;
                  const float s2 = dec::score8(qv, *(const u32x4_t*)(slot + q * 1024 + lane * 16));
                  if (t < n) {
                    if (l8 == 0) sc[t] = s2;
                    mx = fmaxf(mx, s2);
                  }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                st.consumed = XU(st.consumed + 1);
                st.cslot = XU(st.cslot + 1 == XR ? 0 : st.cslot + 1);
                ring_issue(a, st, lv, pr);
                ++kb;
              }
            } else {
              // self-attention: <= S_max cached keys, q | k | v rows of this sequence (agent-scope: row pos was written in this launch)
              const bf16_t* kp = self + (long)b * a.S_max * 3 * a.d + a.d + h * 64 + l8 * 8;
              for (int t0 = grp; t0 < n; t0 += 32 * 8) {
                u32x4_t k4[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                  const int t = t0 + 32 * u;
                  k4[u] = ld16_agent(kp + (long)(t < n ? t : n - 1) * 3 * a.d);
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
            }
            mx = wave_max(mx);
            if (lane == 0) wmax[wave] = mx;
            XBAR();  // 1
            const float m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
            if (cross) {
              for (int kb = 0; kb < it.kvb;) {
                const int run = DBG ? 0 : fast_run_len(st, (n >> 7) - kb);
                if (run > 0) {
                  fast_run(a, st, lv, pr, smem, wave, run, [&](const char* slot, int i) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                      const int t = 32 * ((kb + i) * 4 + q) + grp;
                      dec::accum_pv(__builtin_amdgcn_exp2f(sc[t] - m), *(const u32x4_t*)(slot + q * 1024 + lane * 16), l, o);
                    }
                  });
                  kb += run;
                  continue;
                }
                if (DBG) {
                  if (cc.seg != 3 || cc.idx != j || cc.sub != it.kvb + kb) __hip_atomic_fetch_or(a.ctrl + 1, 0x113u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  cur_advance(g, cc);
                }
                ring_wait(st, a.flags);
                const char* slot = smem + XL_RING + (wave * XR + st.cslot) * SLOT;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  const int t = 32 * (kb * 4 + q) + grp;
                  dec::accum_pv(t < n ? __builtin_amdgcn_exp2f(sc[t] - m) : 0.f, *(const u32x4_t*)(slot + q * 1024 + lane * 16), l, o);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                st.consumed = XU(st.consumed + 1);
                st.cslot = XU(st.cslot + 1 == XR ? 0 : st.cslot + 1);
                ring_issue(a, st, lv, pr);
                ++kb;
              }
            } else {
              const bf16_t* vp = self + (long)b * a.S_max * 3 * a.d + 2 * a.d + h * 64 + l8 * 8;
              for (int t0 = grp; t0 < n; t0 += 32 * 8) {
                u32x4_t v4[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                  const int t = t0 + 32 * u;
                  v4[u] = ld16_agent(vp + (long)(t < n ? t : n - 1) * 3 * a.d);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                  const int t = t0 + 32 * u;
                  dec::accum_pv(t < n ? __builtin_amdgcn_exp2f(sc[t] - m) : 0.f, v4[u], l, o);
                }
              }
            }
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) ared[grp][l8 * 8 + jj] = o[jj];
            if (l8 == 0) lsum[grp] = l;
            XBAR();  // 2
            if (tid < 64) {
              float acc, lt;
              dec::reduce_groups(ared, lsum, tid, acc, lt);
              if (cross) {  // the segment's partial, merged by the consumer phase
                ao[2 + tid] = acc;
                if (tid == 0) ao[0] = m, ao[1] = lt;
              } else {
                float m_s[dec::MAX_SEG] = {m, dec::NEG}, l_s[dec::MAX_SEG] = {lt, 0.f}, o_s[dec::MAX_SEG] = {acc, 0.f};
                float mo, lo;
                ao[2 + tid] = dec::merge_segments(m_s, l_s, o_s, 1, mo, lo);
              }
            }
            XBAR();  // 3
          }
        }
      } else {
        // ---------------- projection: tiles t = wg, wg + team, ... ----------------
        const XGemv p = gemv_of(a, layer, ph);
        const int ntile = seg_cnt(g, p.seg);
        if (helper) {
          XSTAMP(0);
          team_wait(a.ctrl, target, a.flags);
          XSTAMP(1);
          if (ntile > 0) helper_operand(a, g, p, smem, lane);
          XSTAMP(2);
          XBAR();  // A: operand rows ready
          for (int i0 = 0; i0 < ntile; i0 += NRED) {  // groups of NRED tiles (stream_phase)
            const int i1 = i0 + NRED < ntile ? i0 + NRED : ntile;
            XEpiIn in[NRED];
#pragma unroll
            for (int k = 0; k < NRED; ++k)
              if (i0 + k < i1) in[k] = helper_epilogue_inputs(a, p, (wg + a.team * (i0 + k)) * 32, lane);
            XBAR();  // B: the group's accumulators are in the reduction buffers
#pragma unroll
            for (int k = 0; k < NRED; ++k)
              if (i0 + k < i1) helper_epilogue(a, p, in[k], smem, k, (wg + a.team * (i0 + k)) * 32, lane);
            if (i1 < ntile) XBAR();  // buffers free again
          }
          XSTAMP(3);
          team_arrive(a.ctrl);
          XSTAMP(4);
        } else {
          XBAR();  // A
          if (wave == 0) XSTAMP(5);
          stream_phase<DBG>(a, g, st, cc, lv, pr, p, smem, wave, lane, ntile);
          if (wave == 0) XSTAMP(6);
        }
      }
    }
  }
  if (helper) {
    if (wg == 0) {  // everybody has read the epoch base long ago; publish the next launch's once the whole team is through
      team_wait(a.ctrl, base + gphase * (unsigned)a.team);
      if (lane == 0) a.ctrl[2] = base + gphase * (unsigned)a.team;
    }
  }
}

}  // namespace

// ---- host side ----------------------------------------------------------------------------------------------------------------------
size_t decode_xcd_part_floats(int M, int H, int Te) { return (size_t)M * H * dec::n_segments(Te) * 66; }

bool decode_xcd_supports(int d, int H, int Te, int S_max, int L, int M) {
  // (d % 256: a wave's K quarter is whole 64-element ring blocks)
  return M >= 1 && M <= XMAXM && L >= 1 && L <= XMAXL && d % 256 == 0 && d == H * 64 && d <= XMAXD && S_max <= dec::SEG_KEYS &&
         Te <= dec::MAX_SEG * dec::SEG_KEYS && Te >= 1 && H * dec::n_segments(Te) <= XMAXPAIR;
}

// The ring DMA addresses everything through raw buffer descriptors with 32-bit byte offsets from the start of the bf16 weight shadow and
// from layer 0's cross K/V rows (rsrc_of / dma16: num_records 0x7fffffff): an offset at or past 2 GiB would read zeros without any error.
// True when the furthest byte any block of this launch can address stays below that (weights: OLMoASR-large ends at ~1.7 GB because the
// decoder sits at the start of the arena); the engine takes the multi-launch step otherwise.
bool decode_xcd_offsets_ok(const int64_t* layer0, long lstride, long cache_lstride, int d, int Te, int L, int M) {
  const long kLimit = (1L << 31) - (1L << 20);  // 1 MiB of slack for the per-lane part of an address
  const int wsel[6] = {2, 4, 8, 10, 14, 16};  // XLayer: wqkv, wo, wcq, wco, w1, w2
  const long wsize[6] = {3L * d * d, (long)d * d, (long)d * d, (long)d * d, 4L * d * d, 4L * d * d};
  // layer l = layer 0 + l * lstride; the arena keeps the decoder blocks in gradient-completion order (last block first), so the stride
  // is NEGATIVE in the product's layout: both ends of the walk are checked
  const long span = (long)(L - 1) * lstride;
  for (int k = 0; k < 6; ++k) {
    const long first = layer0[wsel[k]] + (span < 0 ? span : 0), last = layer0[wsel[k]] + (span > 0 ? span : 0) + wsize[k];
    if (first < 0 || last * 2 >= kLimit) return false;
  }
  const long kv_end = (long)(L - 1) * cache_lstride + (long)M * Te * 2 * d;  // from layer 0's cross K/V rows
  return cache_lstride >= 0 && kv_end * 2 < kLimit;
}

extern "C" int oasr_xcd_offsets_ok_debug(const int64_t* layer0, long long lstride, long long cache_lstride, int d, int Te, int L, int M) {
  return decode_xcd_offsets_ok(layer0, (long)lstride, (long)cache_lstride, d, Te, L, M) ? 1 : 0;
}

// Debug / CPU test: the block sequence (seg, idx, sub) of workgroup `wg` as the kernel's cursors generate it
extern "C" int oasr_xcd_plan_debug(int d, int H, int Te, int M, int L, int team, int wg, int* out, int max_blocks) {
  const XGeom g = make_geom(d, H, Te, M, L, team, wg);
  XCur c{0, 0, 0, 0};
  int n = 0;
  bool more = cur_normalise(g, c);
  while (more) {
    if (n < max_blocks) out[4 * n] = c.layer, out[4 * n + 1] = c.seg, out[4 * n + 2] = c.idx, out[4 * n + 3] = c.sub;
    ++n;
    more = cur_advance(g, c);
  }
  return n;
}

int launch_decode_xcd(const DecodeXcdArgs& h, hipStream_t s) {
  OASR_REQUIRE(decode_xcd_supports(h.d, h.H, h.Te, h.S_max, h.L, h.M), "decode_xcd: unsupported shape (d=%d H=%d Te=%d S=%d L=%d M=%d)", h.d, h.H,
               h.Te, h.S_max, h.L, h.M);
  OASR_REQUIRE(h.team >= 1 && h.team <= 256 && (h.stride == 1 || h.stride == 8) && h.pos >= 0 && h.pos < h.S_max, "decode_xcd: bad launch shape");
  OASR_REQUIRE(decode_xcd_offsets_ok(h.layer_offsets, h.lstride, h.cache_lstride, h.d, h.Te, h.L, h.M),
               "decode_xcd: a weight / cross K/V byte offset reaches 2 GiB (32-bit buffer offsets); use the multi-launch step");
  XArgs a;
  a.wflat = h.wflat, a.params = h.params, a.aux = h.aux, a.cache = h.cache, a.cache_lstride = h.cache_lstride;
  a.x = h.x, a.x2 = h.x2, a.x3 = h.x3, a.q = h.q, a.o = h.o, a.hg = h.hg, a.part = h.part, a.ctrl = h.ctrl;
  a.d = h.d, a.H = h.H, a.Te = h.Te, a.S_max = h.S_max, a.L = h.L, a.M = h.M, a.pos = h.pos, a.team = h.team, a.stride = h.stride;
  a.flags = h.flags, a.stamps = (unsigned long long*)h.stamps;
  {
    const int64_t* o = h.layer_offsets;  // layer 0
    a.l0 = XLayer{o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[8], o[9], o[10], o[11], o[12], o[13], o[14], o[15], o[16], o[17]};
    a.lstride = h.lstride, a.astride = h.astride;
  }
  const unsigned lds = XL_TOTAL;
  static LdsAttrOnce attr;
  static LdsAttrOnce attr_dbg;
  if (h.flags & 8) {  // checked instantiation (tests)
    { const int rc_ = ensure_dynamic_lds(attr_dbg, (const void*)decode_xcd_kernel<true>, (int)lds); if (rc_) return rc_; }
    hipLaunchKernelGGL(decode_xcd_kernel<true>, dim3(h.team * h.stride), dim3(320), lds, s, a);
  } else {
    { const int rc_ = ensure_dynamic_lds(attr, (const void*)decode_xcd_kernel<false>, (int)lds); if (rc_) return rc_; }
    hipLaunchKernelGGL(decode_xcd_kernel<false>, dim3(h.team * h.stride), dim3(320), lds, s, a);
  }
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}

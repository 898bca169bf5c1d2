// Host-side engine: parameter layout, workspace planning and the explicit forward / backward schedule of the
// OLMoASR training micro-step.  No autograd tape: every saved activation has a planned slot in the caller's
// workspace and the backward is written out by hand, layer by layer, in the order gradients become final
// (so per-segment events can release RCCL buckets while the rest of the backward is still running).
//
// Reference schedule being replaced: OLMoASR.forward (olmoasr/model.py:856-887) -> AudioEncoder.forward (:571-623)
// -> TextDecoder.forward (:688-775) -> F.cross_entropy(ignore_index=51864)/accum (train_timestamps.py:1444-1450)
// -> scaler.scale(loss).backward() (:1454) -> unscale_/clip_grad_norm_/AdamW (:1509-1512).
#include <string>
#include <vector>

#include "../../include/oasr.h"
#include <math.h>

#include "kernels.h"

namespace {

constexpr long PAD_ID = 51864;

struct Tensor {
  std::string name;
  int64_t off, numel;
  int ndim;
  int64_t shape[4];
};

struct AttnP {
  int64_t qw, kw, vw, ow, qb, vb, ob;  // arena offsets (elements); qw,kw,vw are contiguous -> fused [3d,d]
  int64_t fused_bias;                  // offset (floats) into the aux fp32 region: [qb | 0 | vb]
};
struct BlockP {
  int64_t attn_ln_w, attn_ln_b, cln_w, cln_b, mlp_ln_w, mlp_ln_b, w1, b1, w2, b2;
  AttnP attn, cattn;
  bool cross;
};
struct Segment {
  int64_t off, numel;
};

}  // namespace

// Runner::side_mode of the span step.  7 = the decoder backward's R-row weight gradients, the cross-attention key|value gradients AND the
// forward's key|value projections on the lowest-priority side streams: -1.4..-1.55 % of the step against no side streams, -0.7 % against mode 5
// (same-box A/Bs, profiles/r05_side_streams.txt; round 6 re-measured: profiles/r06_side_streams.txt).  The 48 forward projections share the
// dominant forward kernel's symbol; as filler their begin-to-end spans are queueing times, so the GEMM launch statistics carry the lane a
// launch ran on (gemm_profile_lane, set by Runner::OnStream) and bench.py's `roofline` / scripts/rocprof_summary.py price main-stream launches only.
constexpr int SIDE_STREAMS_DEFAULT = 7;

struct oasr_ctx {
  oasr_dims dims;
  int d, H, L_enc, L_dec, Te, T1, S_max, V, Vp;  // V = n_vocab+1 rows (train model), Vp = padded to 128
  std::vector<Tensor> tensors;
  std::vector<Segment> segments;
  std::vector<BlockP> enc, dec;
  int64_t dec_ln_w, dec_ln_b, dec_pos, enc_lnp_w, enc_lnp_b, conv1_w, conv1_b, conv2_w, conv2_b, tok_emb;
  int64_t numel;
  // bound memory
  float *params = nullptr, *grads = nullptr, *m = nullptr, *v = nullptr;
  const float* enc_pos = nullptr;
  char* shadow = nullptr;
  // shadow layout (bytes)
  size_t sh_flat, sh_w1p, sh_w2p, sh_aux, sh_total;
  int64_t aux_floats;
  int f32 = 0;  // compute_dtype: 0 = bf16 production kernels, 1 = fp32 validation kernels (fp32ref.hip)
  std::vector<int64_t> xcd_offsets;  // decoder layer 0's 18 tensor offsets in decode_xcd.hip::XLayer order (empty: irregular layout, engine off)
  int64_t xcd_lstride = 0, xcd_astride = 0;  // layer l = layer 0 + l * stride
  // Side stream of the span step's decoder backward (Runner::wgrad_side): created on first use, lowest priority, so its weight-gradient
  // workgroups fill the compute units the main stream's launches leave idle
  struct Side {
    hipStream_t stream = nullptr;  // the R-row weight gradients of the decoder backward
    hipStream_t big = nullptr;     // the encoder-sized GEMMs of the cross-attention key|value side (forward projection, its two gradients)
    hipEvent_t fork[4] = {nullptr, nullptr, nullptr, nullptr}, join[4] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> kv_ready;  // [L_dec]: layer i's key|value projection has been written (forward)
    unsigned nf = 0, nj = 0;
  };
  mutable Side side;
  // set by oasr_decode_check when the one-launch decoder step (decode_xcd.hip) reported a poisoned team barrier: its 32 workgroups must be
  // resident at once, which a shared / CU-masked device does not guarantee.  From then on this context decodes on the multi-launch engine.
  bool xcd_disabled = false;
  int n_cu = 0;  // compute units of the device (queried by the first decoder step)
  ~oasr_ctx() {
    for (hipEvent_t e : side.fork)
      if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : side.join)
      if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : side.kv_ready)
      if (e) (void)hipEventDestroy(e);
    if (side.stream) (void)hipStreamDestroy(side.stream);
    if (side.big) (void)hipStreamDestroy(side.big);
  }
  // compute copy of the weight at arena offset `off`: the bf16 shadow, or -- fp32 validation -- the master weights themselves
  template <typename T>
  const T* Wt(int64_t off) const;
  template <typename T>
  const T* w1p() const { return (const T*)(shadow + sh_w1p); }  // packed conv1 kernel [d][256]
  template <typename T>
  const T* w2p() const { return (const T*)(shadow + sh_w2p); }  // packed conv2 kernel [d][3d]
  const float* P(int64_t off) const { return params + off; }
  float* G(int64_t off) const { return grads + off; }
  const float* aux(int64_t off) const { return (const float*)(shadow + sh_aux) + off; }
};

template <>
inline const bf16_t* oasr_ctx::Wt<bf16_t>(int64_t off) const { return (const bf16_t*)(shadow + sh_flat) + off; }
template <>
inline const float* oasr_ctx::Wt<float>(int64_t off) const { return params + off; }

namespace {

struct Builder {
  oasr_ctx* c;
  int64_t cur = 0;
  int64_t add(const std::string& name, std::initializer_list<int64_t> shape) {
    Tensor t;
    t.name = name;
    t.off = cur;
    t.ndim = (int)shape.size();
    t.numel = 1;
    int i = 0;
    for (auto s : shape) {
      t.shape[i++] = s;
      t.numel *= s;
    }
    for (; i < 4; ++i) t.shape[i] = 1;
    cur += t.numel;
    c->tensors.push_back(t);
    return t.off;
  }
  void attn(const std::string& p, AttnP& a, int d) {
    a.qw = add(p + ".query.weight", {d, d});
    a.kw = add(p + ".key.weight", {d, d});
    a.vw = add(p + ".value.weight", {d, d});
    a.ow = add(p + ".out.weight", {d, d});
    a.qb = add(p + ".query.bias", {d});
    a.vb = add(p + ".value.bias", {d});
    a.ob = add(p + ".out.bias", {d});
    a.fused_bias = c->aux_floats;
    c->aux_floats += 3 * d;
  }
  void block(const std::string& p, BlockP& b, int d, bool cross) {
    const int64_t start = cur;
    b.cross = cross;
    b.w2 = add(p + ".mlp.2.weight", {d, 4 * d});
    b.b2 = add(p + ".mlp.2.bias", {d});
    b.w1 = add(p + ".mlp.0.weight", {4 * d, d});
    b.b1 = add(p + ".mlp.0.bias", {4 * d});
    b.mlp_ln_w = add(p + ".mlp_ln.weight", {d});
    b.mlp_ln_b = add(p + ".mlp_ln.bias", {d});
    if (cross) {
      attn(p + ".cross_attn", b.cattn, d);
      b.cln_w = add(p + ".cross_attn_ln.weight", {d});
      b.cln_b = add(p + ".cross_attn_ln.bias", {d});
    }
    attn(p + ".attn", b.attn, d);
    b.attn_ln_w = add(p + ".attn_ln.weight", {d});
    b.attn_ln_b = add(p + ".attn_ln.bias", {d});
    c->segments.push_back({start, cur - start});
  }
};

// ---- workspace bump allocator (dry-run when base == nullptr) --------------------------------------------------
struct Arena {
  char* base;
  size_t cur = 0, cap;
  Arena(void* b, size_t c) : base((char*)b), cap(c) {}
  void* raw(size_t bytes) {
    cur = (cur + 255) & ~(size_t)255;
    void* p = base ? base + cur : (void*)(uintptr_t)(cur + 256);  // non-null fake in dry-run
    cur += bytes;
    return p;
  }
  template <typename T>
  T* act(size_t n) { return (T*)raw((n + 32) * sizeof(T)); }  // +32 elements: conv windows / 16-byte tails may over-read
  float* f32(size_t n) { return (float*)raw(n * 4); }
};

#define RC(x)            \
  do {                   \
    int _rc = (x);       \
    if (_rc) return _rc; \
  } while (0)

// Everything below is written once for both compute dtypes: T = bf16_t (production) or float (validation).
template <typename T>
struct Engine {
  typedef OperandViewT<T> View;
  typedef GemmArgsT<T> Gemm;
  typedef AttnArgsT<T> Attn;

struct AttnSave {
  T *ln, *qkv, *o;  // self: qkv [M,3d];  cross: qkv = q [M,d]
  T* kv;            // cross only: [B*Te, 2d]
  float *mean, *rstd, *lse;
  T* o_lo;  // training only: bf16 rounding residual of the attention output (o + o_lo = fp32-grade O for the backward's delta)
};
struct BlockSave {
  T* x_in;  // residual stream entering the block (owned by the previous stage)
  AttnSave sa, ca;
  T *x_mid, *x_mid2, *ln2, *u, *hg, *x_out;
  float *mean2, *rstd2;
};

struct Plan {
  // encoder
  T *mel_tm, *u1, *h1, *u2, *x0, *xa;
  float *mean_p, *rstd_p;
  std::vector<BlockSave> enc, dec;
  // decoder
  T *dx0, *lnf, *logits;
  float *mean_f, *rstd_f, *row_loss;
  int32_t* n_valid;
  // backward temporaries
  T *ga, *gb, *gc, *gln, *gqkv, *go, *gu, *gxa, *gkv, *gq, *gA2;
  float *delta, *tmp_w1p, *tmp_w2p, *cs_scratch, *gemm_cs_scratch;
  int32_t* qtile_flags;  // attention backward of the decoder: which 64-position tiles of d_o are non-zero
  // supervised-span step (oasr_train_fwd_bwd_span): chunk-row table of the decoder's token rows, spans, targets in row order
  int32_t *rows, *span_dev;
  int64_t* targets_phys;
};

static void plan_attn(Arena& A, AttnSave& s, long M, long Mkv, int d, int B, int H, long Tq, bool cross, bool train) {
  s.ln = A.template act<T>(M * d);
  s.qkv = A.template act<T>(M * (cross ? d : 3 * d));
  s.kv = cross ? A.template act<T>(Mkv * 2 * d) : nullptr;
  s.o = A.template act<T>(M * d);
  s.mean = A.f32(M);
  s.rstd = A.f32(M);
  s.lse = A.f32((long)B * H * Tq);
  s.o_lo = train ? A.template act<T>(M * d) : nullptr;
}

// In inference mode the per-layer buffers are shared between layers (allocated once); in training each layer
// gets its own slots because the backward needs them.
static void make_plan(const oasr_ctx* c, Arena& A, Plan& p, int B, int S, bool train) {
  const int d = c->d;
  const long Me = (long)B * c->Te, M1 = (long)B * c->T1, Md = (long)B * S;
  p.mel_tm = A.template act<T>(M1 * c->dims.n_mels + 2 * 256) + 256;  // zeroed guard rows on both sides (conv1 weight gradient windows)
  p.u1 = A.template act<T>(M1 * d);
  p.h1 = A.template act<T>(M1 * d + d) + d;  // one zeroed time row in front: the conv2 weight gradient reads h1 as overlapping windows from h1 - d
  p.u2 = A.template act<T>(Me * d);
  p.x0 = A.template act<T>(Me * d);
  auto plan_block = [&](BlockSave& s, long M, long Tq, bool cross) {
    plan_attn(A, s.sa, M, 0, d, B, c->H, Tq, false, train);
    s.x_mid = A.template act<T>(M * d);
    if (cross) {
      plan_attn(A, s.ca, M, Me, d, B, c->H, Tq, true, train);
      s.x_mid2 = A.template act<T>(M * d);
    } else {
      s.x_mid2 = nullptr;
    }
    s.ln2 = A.template act<T>(M * d);
    s.u = A.template act<T>(M * 4 * d);
    s.hg = A.template act<T>(M * 4 * d);
    s.mean2 = A.f32(M);
    s.rstd2 = A.f32(M);
    s.x_out = A.template act<T>(M * d);
  };
  p.enc.resize(c->L_enc);
  p.dec.resize(c->L_dec);
  if (train) {
    for (auto& s : p.enc) plan_block(s, Me, c->Te, false);
  } else {
    BlockSave s0, s1;
    plan_block(s0, Me, c->Te, false);
    s1 = s0;
    s1.x_out = A.template act<T>(Me * d);  // ping-pong the residual stream
    for (int i = 0; i < c->L_enc; ++i) p.enc[i] = (i & 1) ? s1 : s0;
  }
  p.xa = A.template act<T>(Me * d);
  p.mean_p = A.f32(Me);
  p.rstd_p = A.f32(Me);
  p.dx0 = A.template act<T>(Md * d);
  if (train) {
    for (auto& s : p.dec) plan_block(s, Md, S, true);
  } else {
    BlockSave s0, s1;
    plan_block(s0, Md, S, true);
    s1 = s0;
    s1.x_out = A.template act<T>(Md * d);
    for (int i = 0; i < c->L_dec; ++i) p.dec[i] = (i & 1) ? s1 : s0;
  }
  p.lnf = A.template act<T>(Md * d);
  p.mean_f = A.f32(Md);
  p.rstd_f = A.f32(Md);
  p.logits = A.template act<T>(Md * c->Vp);
  p.row_loss = A.f32(Md);
  p.n_valid = (int32_t*)A.raw(256);
  p.rows = p.span_dev = nullptr;
  p.targets_phys = nullptr;
  if (train) {
    p.rows = (int32_t*)A.raw((size_t)B * OASR_ROWTAB * 4);
    p.span_dev = (int32_t*)A.raw((size_t)B * 4);
    p.targets_phys = (int64_t*)A.raw((size_t)Md * 8);
    const long Mmax = Me > Md ? Me : Md;
    p.ga = A.template act<T>(Mmax * d);
    p.gb = A.template act<T>(Mmax * d);
    p.gc = A.template act<T>(Mmax * d);
    p.gln = A.template act<T>(Mmax * d);
    p.gqkv = A.template act<T>(Mmax * 3 * d);
    p.go = A.template act<T>(Mmax * d);
    p.gu = A.template act<T>(M1 * d > Mmax * 4 * d ? M1 * d : Mmax * 4 * d);  // also holds dpre1 [B*3000, d]
    p.gxa = A.template act<T>(Me * d);
    p.gkv = A.template act<T>(Me * 2 * d);
    p.gq = A.template act<T>(Md * d);
    p.gA2 = A.template act<T>(Me * 3 * d);
    p.delta = A.f32((long)B * c->H * c->Te);
    p.cs_scratch = A.f32(attn_colsum_scratch_floats(B, c->H, c->Te, c->Te));  // (the largest of the three attention shapes)
    p.qtile_flags = (int32_t*)A.f32((size_t)B * c->H * ((S + 63) / 64) + 16);
    p.gemm_cs_scratch = A.f32((size_t)2 * cdiv(Mmax, 256) * 4 * d + 64);
    p.tmp_w1p = A.f32((long)d * 256);
    p.tmp_w2p = A.f32((long)d * 3 * d);
    // the residual stream entering each block (block_fwd records the same pointers): a plan re-made for a backward-only call
    // (oasr_train_bwd) must be complete without having run the forward
    for (int i = 0; i < c->L_enc; ++i) p.enc[i].x_in = i ? p.enc[i - 1].x_out : p.x0;
    for (int i = 0; i < c->L_dec; ++i) p.dec[i].x_in = i ? p.dec[i - 1].x_out : p.dx0;
  }
}

struct Runner {
  const oasr_ctx* c;
  hipStream_t st;
  int B, S;
  const int32_t* text_len;
  bool train = false;  // the training forward saves GELU'(u) in place of u (GemmArgs.act == 2)
  float* cs_scratch = nullptr;  // partial rows of fused bias-gradient column sums (GemmArgs.colsum_scratch)
  // Supervised-span step: the decoder's token rows are CHUNKED (64 positions per chunk, kernels.h: AttnArgs.q_rows) with every
  // chunk that can carry gradient first, so the whole decoder backward runs on the first `dec_rows_bwd` rows as plain matrices.
  const int32_t* dec_rows = nullptr;  // chunk-row table [B][OASR_ROWTAB] (device) or null = plain [B, S] rows
  const int32_t* dec_span = nullptr;  // [B] spans rounded up to 64 (device); backward only
  long dec_rows_bwd = 0;              // active decoder rows (0 = all B*S)
  // opt-in (OASR_SPAN_FORWARD_ACTIVE): the decoder's FORWARD covers the active rows only as well.  The rows left out are the padded
  // positions whose logits the reference computes and nothing ever reads (no supervised query attends to them, the loss ignores them).
  long dec_rows_fwd = 0;
  const float* mel_clip_max = nullptr;  // [B] or null: `mel` is oasr_log_mel_raw's output, finalized in the time-major transpose

  int linear(const T* x, long M, int K, const T* W, int N, const float* bias, int act, const T* resid, T* out,
             T* out_pre) {
    Gemm g = gemm_defaults_t<T>();
    g.A = plain_view(x, K);
    g.B = plain_view(W, K);
    g.M = (int)M;
    g.N = N;
    g.K = K;
    g.bias = bias;
    g.act = act;
    g.resid = resid;
    g.ldr = N;
    g.out = out;
    g.out_pre = out_pre;
    g.ldc = N;
    return launch_gemm(g, st);
  }
  // dx[M,K] = dy[M,N] . W[N,K]  (* gelu'(u))  (+ resid)
  int dgrad(const T* dy, long M, int N, const T* W, int K, const T* dgelu_u, const T* resid, T* dx,
            float* colsum = nullptr, bool u_is_deriv = false) {
    Gemm g = gemm_defaults_t<T>();
    g.A = plain_view(dy, N);
    g.B = plain_view(W, K);
    g.tb = 1;
    g.M = (int)M;
    g.N = K;
    g.K = N;
    g.dgelu_u = dgelu_u;
    g.dgelu_deriv = u_is_deriv ? 1 : 0;
    g.ldu = K;
    g.resid = resid;
    g.ldr = K;
    g.out = dx;
    g.ldc = K;
    g.colsum = colsum;
    g.colsum_scratch = colsum ? cs_scratch : nullptr;
    return launch_gemm(g, st);
  }
  // dW[N,K] += dy[M,N]^T . x[M,K]   (fp32 atomics, split over the token dimension)
  int wgrad(const T* dy, long ldy, long M, int N, const View& x, int K, float* dW, long ldw) {
    Gemm g = gemm_defaults_t<T>();
    g.A = plain_view(dy, ldy);
    g.ta = 1;
    g.B = x;
    g.tb = 1;
    g.M = N;
    g.N = K;
    g.K = (int)M;
    g.out_f32 = dW;
    g.ldc32 = ldw;
    g.atomic = 1;
    const long tiles = (long)cdiv(N, 256) * cdiv(K, 128);
    const long kt = cdiv(M, 64);
    // Split-K choice (scripts/wgrad_sweep.py): 768 workgroups are resident at once (3 per CU); what matters is how the
    // tiles x split grid quantises onto them (1.33 waves is the worst case), the (16 + split) K-tiles' worth of atomic epilogue
    // every workgroup adds, and that multiples of 8 let every XCD own whole K-ranges (gemm.hip).
    long split = 1;
    double best = 1e30;
    static const int cand[] = {1, 2, 4, 8, 16, 24, 32};
    for (int s_ : cand) {
      if (s_ > 1 && (kt / s_ < 8 || tiles >= 768)) break;
      const double w = (double)tiles * s_, per = (double)kt / s_ + 16.0 + s_;  // atomics get slower the more splits collide
      const double waves = w <= 768.0 ? 0.7 + 0.3 * w / 768.0 : ceil(w / 768.0);
      const double score = per * waves;
      if (score < 0.97 * best) {
        best = score;
        split = s_;
      }
    }
    g.split_k = (int)split;
    // With its MFMA sections pinned the 256x256 ping-pong loop beats the 256x128 kernel on the square and the 4:1 weight shapes
    // (scripts/wgrad_sweep.py, profiles/r02_wgrad_sweep.txt; TF/s pp vs 256x128): 192k tokens [1024x1024] 1071 vs 972 (split 16),
    // [4096x1024] 1157 vs 1117 (8), [1024x4096] 1158 vs 1064 (4); 57k tokens [4096x1024] 1032 vs 956 (4), [1024x4096] 1131 vs 996
    // (4); [3072x1024] and the 57k-token square stay on the 256x128 kernel.  One workgroup per CU: splits give 256-512 workgroups.
    if (M >= 40000 && x.rpb == 0 && (N % 256) == 0 && (K % 256) == 0) {
      const long t256 = (long)(N / 256) * (K / 256);
      const bool long_tokens = M >= 150000;
      int pp_split = 0;
      if (t256 == 16 && long_tokens) pp_split = 16;
      else if (t256 == 64) pp_split = (long_tokens && N > K) ? 8 : 4;
      // round 4 (profiles/r04_wgrad_sweep.txt): the cross-attention key|value gradient [2048 x 1024] over the 192k encoder tokens, never swept
      // before: ping-pong split 8 = 0.687 ms (1173 TFLOP/s) vs 0.754 ms (1068) for the best 256x128 split; [3072 x 1024] stays (1.055 vs 1.081)
      else if (t256 == 32 && long_tokens) pp_split = 8;
      if (pp_split) {
        g.atomic_on_pp = 1;
        g.split_k = pp_split;
      }
    }
    return launch_gemm(g, st);
  }
  // ---- side stream (decoder backward of a span step) ----------------------------------------------------------------------------
  // The decoder-side GEMMs of a span step run over R ~ 18.7k rows: 292 tiles of 256 x 256 for an N = 1024 output = 1.14 rounds over the 256
  // CUs, the second round 14 % full.  A weight gradient and the data gradient launched after it are independent (both read dy), so the
  // weight gradients go to a second, lowest-priority stream whose workgroups take the CUs the main stream's tails leave idle; the main
  // stream waits for them (join_side) before the LayerNorm backward that ends each section of block_bwd -- the next kernel that may
  // overwrite something a weight gradient reads -- so the per-block events (DDP buckets) still mean "this block's gradients are complete".
  // The encoder-sized GEMMs of the cross-attention key|value side (the projection of xa in the forward -- it depends on the encoder output
  // only, so all L_dec of them are issued when the decoder starts; its weight gradient and d(xa) in the backward) run on a second side
  // stream the same way: short workgroups by the thousand, the filler for every tail of the decoder's own launches.
  struct OnStream {  // launches of this scope go to `to`
    hipStream_t& ref;
    hipStream_t keep;
    int lane;  // (bench.py's per-launch GEMM statistics keep side-stream spans -- queueing times -- apart from main-stream kernel times)
    OnStream(hipStream_t& r, hipStream_t to) : ref(r), keep(r), lane(gemm_profile_lane(to != r ? 1 : -1)) { ref = to; }
    ~OnStream() {
      ref = keep;
      gemm_profile_lane(lane);
    }
  };
  int side_mode = 0;  // bit 0: R-row weight gradients, bit 1: forward key|value projections, bit 2: backward key|value gradients
  bool side_pending = false, big_pending = false;
  int side_begin(int mode) {
    oasr_ctx::Side& sd = c->side;
    if (!sd.stream) {
      // built into a local and published only when every call has succeeded: a failure half way must not leave a non-null stream
      // beside null events for the next step to trip over (whatever was created is destroyed again)
      oasr_ctx::Side nw;
      nw.kv_ready.assign((size_t)c->L_dec, nullptr);
      auto build = [&]() -> int {
        int least = 0, greatest = 0;
        OASR_CHECK_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        OASR_CHECK_HIP(hipStreamCreateWithPriority(&nw.stream, hipStreamNonBlocking, least));
        OASR_CHECK_HIP(hipStreamCreateWithPriority(&nw.big, hipStreamNonBlocking, least));
        for (hipEvent_t& e : nw.fork) OASR_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (hipEvent_t& e : nw.join) OASR_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (hipEvent_t& e : nw.kv_ready) OASR_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        return OASR_OK;
      };
      if (const int rc = build()) {
        for (hipEvent_t e : nw.fork) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : nw.join) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : nw.kv_ready) if (e) (void)hipEventDestroy(e);
        if (nw.stream) (void)hipStreamDestroy(nw.stream);
        if (nw.big) (void)hipStreamDestroy(nw.big);
        return rc;
      }
      sd = nw;
    }
    side_mode = mode;
    return OASR_OK;
  }
  // everything launched on `st` so far happens before whatever is launched on `to` next
  int fork_to(hipStream_t to) {
    oasr_ctx::Side& sd = c->side;
    hipEvent_t e = sd.fork[sd.nf++ & 3];
    OASR_CHECK_HIP(hipEventRecord(e, st));
    OASR_CHECK_HIP(hipStreamWaitEvent(to, e, 0));
    return OASR_OK;
  }
  int join_from(hipStream_t from) {
    oasr_ctx::Side& sd = c->side;
    hipEvent_t e = sd.join[sd.nj++ & 3];
    OASR_CHECK_HIP(hipEventRecord(e, from));
    OASR_CHECK_HIP(hipStreamWaitEvent(st, e, 0));
    return OASR_OK;
  }
  int wgrad_side(const T* dy, long ldy, long M, int N, const View& x, int K, float* dW, long ldw) {
    if (!(side_mode & 1)) return wgrad(dy, ldy, M, N, x, K, dW, ldw);
    RC(fork_to(c->side.stream));
    OnStream on(st, c->side.stream);
    side_pending = true;
    return wgrad(dy, ldy, M, N, x, K, dW, ldw);
  }
  int join_side() {
    if (!side_pending) return OASR_OK;
    side_pending = false;
    return join_from(c->side.stream);
  }
  int join_big() {
    if (!big_pending) return OASR_OK;
    big_pending = false;
    return join_from(c->side.big);
  }

  int attn_args(Attn& a, const AttnSave& s, bool cross, long Tq, long Tk, bool causal) {
    const int d = c->d;
    memset(&a, 0, sizeof(a));
    if (!cross) {
      a.q = s.qkv;
      a.k = s.qkv + d;
      a.v = s.qkv + 2 * d;
      a.ldq = a.ldk = a.ldv = 3 * d;
      a.bsq = a.bsk = a.bsv = Tq * 3 * d;
    } else {
      a.q = s.qkv;
      a.ldq = d;
      a.bsq = Tq * d;
      a.k = s.kv;
      a.v = s.kv + d;
      a.ldk = a.ldv = 2 * d;
      a.bsk = a.bsv = Tk * 2 * d;
    }
    a.o = s.o;
    a.ldo = d;
    a.bso = Tq * d;
    a.lse = s.lse;
    a.o_lo = s.o_lo;
    a.kv_len = causal ? text_len : nullptr;
    a.B = B;
    a.H = c->H;
    a.Tq = (int)Tq;
    a.Tk = (int)Tk;
    a.causal = causal ? 1 : 0;
    if (dec_rows && (causal || cross)) {  // a decoder attention of a span-limited step: chunked query rows (+ key rows for self-attention)
      a.q_rows = dec_rows;
      a.k_rows = cross ? nullptr : dec_rows;
      if (dec_rows_fwd) a.q_span = dec_span;  // (the backward sets it itself)
    }
    return OASR_OK;
  }

  int kv_proj(const BlockP& bp, BlockSave& s, const T* xa) {
    const int d = c->d;
    return linear(xa, (long)B * c->Te, d, c->template Wt<T>(bp.cattn.kw), 2 * d, c->aux(bp.cattn.fused_bias) + d, 0, nullptr, s.ca.kv, nullptr);
  }
  int block_fwd(const BlockP& bp, BlockSave& s, const T* x_in, long M, long Tq, const T* xa, bool causal, hipEvent_t kv_ready = nullptr) {
    const int d = c->d;
    s.x_in = const_cast<T*>(x_in);
    RC(launch_layernorm_fwd(x_in, c->P(bp.attn_ln_w), c->P(bp.attn_ln_b), s.sa.ln, s.sa.mean, s.sa.rstd, M, d, st));
    RC(linear(s.sa.ln, M, d, c->template Wt<T>(bp.attn.qw), 3 * d, c->aux(bp.attn.fused_bias), 0, nullptr, s.sa.qkv, nullptr));
    Attn a;
    attn_args(a, s.sa, false, Tq, Tq, causal);
    RC(launch_attention_fwd(a, st));
    RC(linear(s.sa.o, M, d, c->template Wt<T>(bp.attn.ow), d, c->P(bp.attn.ob), 0, x_in, s.x_mid, nullptr));
    const T* xm = s.x_mid;
    if (bp.cross) {
      RC(launch_layernorm_fwd(xm, c->P(bp.cln_w), c->P(bp.cln_b), s.ca.ln, s.ca.mean, s.ca.rstd, M, d, st));
      RC(linear(s.ca.ln, M, d, c->template Wt<T>(bp.cattn.qw), d, c->P(bp.cattn.qb), 0, nullptr, s.ca.qkv, nullptr));
      if (kv_ready)  // (decoder_fwd issued this layer's key|value projection on the side stream)
        OASR_CHECK_HIP(hipStreamWaitEvent(st, kv_ready, 0));
      else
        RC(kv_proj(bp, s, xa));
      attn_args(a, s.ca, true, Tq, c->Te, false);
      RC(launch_attention_fwd(a, st));
      RC(linear(s.ca.o, M, d, c->template Wt<T>(bp.cattn.ow), d, c->P(bp.cattn.ob), 0, xm, s.x_mid2, nullptr));
      xm = s.x_mid2;
    }
    RC(launch_layernorm_fwd(xm, c->P(bp.mlp_ln_w), c->P(bp.mlp_ln_b), s.ln2, s.mean2, s.rstd2, M, d, st));
    RC(linear(s.ln2, M, d, c->template Wt<T>(bp.w1), 4 * d, c->P(bp.b1), train ? 2 : 1, nullptr, s.hg, train ? s.u : nullptr));
    RC(linear(s.hg, M, 4 * d, c->template Wt<T>(bp.w2), d, c->P(bp.b2), 0, xm, s.x_out, nullptr));
    return OASR_OK;
  }

  View conv1_view(const T* mel_tm) const {
    const int nm = c->dims.n_mels;
    return View{mel_tm, nm, c->T1, (long)c->T1 * nm, nm, 3 * nm, 2 * nm};
  }
  View conv2_view(const T* h1) const {
    const int d = c->d;
    return View{h1, 2L * d, c->Te, (long)c->T1 * d, d, 3 * d, 3 * d};
  }

  int encoder_fwd(Plan& p, const float* mel) {
    const int d = c->d;
    const long M1 = (long)B * c->T1, Me = (long)B * c->Te;
    RC(launch_mel_to_time_major(mel, p.mel_tm, B, c->dims.n_mels, c->T1, st, mel_clip_max));
    // Both convolutions run on the direct-to-LDS kernels: the im2col matrix is the input itself read as a PLAIN matrix of
    // overlapping rows (row stride = conv stride * C) that starts one time row before the buffer.  That view is exact
    // except at sample boundaries -- window (b, 0) sees the previous sample's last row (or the zeroed guard row) where
    // the conv pads with zeros, and for conv1 window (b, T1-1) sees the next sample's first row -- so those 2B (conv1) /
    // B (conv2) output rows are recomputed afterwards by a small GEMM over one-row-per-sample window views whose
    // padding IS an out-of-range predicate (OperandView with rpb = 1).
    const int nm = c->dims.n_mels;
    OASR_CHECK_HIP(hipMemsetAsync(p.mel_tm - 256, 0, 256 * sizeof(T), st));
    OASR_CHECK_HIP(hipMemsetAsync(p.mel_tm + M1 * nm, 0, 256 * sizeof(T), st));
    OASR_CHECK_HIP(hipMemsetAsync(p.h1 - d, 0, (size_t)d * sizeof(T), st));
    for (int pass = 0; pass < 3; ++pass) {  // 0: all rows through the plain view; 1: rows (b, 0); 2: rows (b, T1-1)
      Gemm g = gemm_defaults_t<T>();
      long row_off = 0;
      if (pass == 0) {
        g.A = plain_view(p.mel_tm - nm, nm);
        g.M = (int)M1;
        g.ldc = d;
      } else {
        row_off = pass == 1 ? 0 : c->T1 - 1;
        g.A = pass == 1 ? View{p.mel_tm, nm, 1, (long)c->T1 * nm, nm, 3 * nm, 3 * nm}
                        : View{p.mel_tm + (long)(c->T1 - 2) * nm, nm, 1, (long)c->T1 * nm, 0, 3 * nm, 2 * nm};
        g.M = B;
        g.ldc = (long)c->T1 * d;
      }
      g.B = plain_view(c->template w1p<T>(), 256);
      g.N = d;
      g.K = 256;
      g.bias = c->P(c->conv1_b);
      g.act = 1;
      g.out = p.h1 + row_off * d;
      g.out_pre = p.u1 + row_off * d;
      RC(launch_gemm(g, st));
    }
    for (int pass = 0; pass < 2; ++pass) {  // 0: all rows; 1: rows (b, 0)
      Gemm g = gemm_defaults_t<T>();
      if (pass == 0) {
        g.A = plain_view(p.h1 - d, 2L * d);
        g.M = (int)Me;
        g.ldc = d;
        g.pos_period = c->Te;
      } else {
        g.A = View{p.h1, 2L * d, 1, (long)c->T1 * d, d, 3 * d, 3 * d};
        g.M = B;
        g.ldc = (long)c->Te * d;
        g.pos_period = 1;  // every recomputed row is position 0
      }
      g.B = plain_view(c->template w2p<T>(), 3 * d);
      g.N = d;
      g.K = 3 * d;
      g.bias = c->P(c->conv2_b);
      g.act = 1;
      g.pos = c->enc_pos;
      g.out = p.x0;
      g.out_pre = p.u2;
      RC(launch_gemm(g, st));
    }
    const T* x = p.x0;
    for (int i = 0; i < c->L_enc; ++i) {
      RC(block_fwd(c->enc[i], p.enc[i], x, Me, c->Te, nullptr, false));
      x = p.enc[i].x_out;
    }
    RC(launch_layernorm_fwd(x, c->P(c->enc_lnp_w), c->P(c->enc_lnp_b), p.xa, p.mean_p, p.rstd_p, Me, d, st));
    return OASR_OK;
  }

  int decoder_fwd(Plan& p, const int64_t* tokens, bool last_only = false) {
    const int d = c->d;
    const long Md = dec_rows_fwd ? dec_rows_fwd : (long)B * S;  // token rows the row-wise kernels run over
    RC(launch_embedding_fwd(tokens, c->P(c->tok_emb), c->P(c->dec_pos), p.dx0, B, S, d, c->V, st, dec_rows));
    const bool kv_side = (side_mode & 2) && train;  // (training plan: every layer has its own key|value buffer)
    // launch statistics: from here until the backward's last join the main stream shares the chip with side-stream filler (lane 2, "[shared]")
    if (side_mode && train) gemm_profile_lane(2);
    if (kv_side) {
      RC(fork_to(c->side.big));  // p.xa is complete
      OnStream on(st, c->side.big);
      for (int i = 0; i < c->L_dec; ++i) {
        RC(kv_proj(c->dec[i], p.dec[i], p.xa));
        OASR_CHECK_HIP(hipEventRecord(c->side.kv_ready[i], st));
      }
    }
    const T* x = p.dx0;
    for (int i = 0; i < c->L_dec; ++i) {
      RC(block_fwd(c->dec[i], p.dec[i], x, Md, S, p.xa, true, kv_side ? c->side.kv_ready[i] : nullptr));
      x = p.dec[i].x_out;
    }
    RC(launch_layernorm_fwd(x, c->P(c->dec_ln_w), c->P(c->dec_ln_b), p.lnf, p.mean_f, p.rstd_f, Md, d, st));
    if (last_only) {  // greedy decoding only needs position S-1 of every sequence: M = B rows, row stride S*d
      Gemm g = gemm_defaults_t<T>();
      g.A = plain_view(p.lnf + (long)(S - 1) * d, (long)S * d);
      g.B = plain_view(c->template Wt<T>(c->tok_emb), d);
      g.M = B;
      g.N = c->Vp;
      g.K = d;
      g.out = p.logits;
      g.ldc = c->Vp;
      return launch_gemm(g, st);
    }
    RC(linear(p.lnf, Md, d, c->template Wt<T>(c->tok_emb), c->Vp, nullptr, 0, nullptr, p.logits, nullptr));
    return OASR_OK;
  }

  int record(void** ev, int idx) {
    if (ev && ev[idx]) OASR_CHECK_HIP(hipEventRecord((hipEvent_t)ev[idx], st));
    return OASR_OK;
  }

  // dx_out (grad of the block output) -> returns grad of the block input in *dx_in_out (ping-pong ga/gb)
  // Bias gradients of Linears that write into the residual stream (mlp.2, attn.out, cross_attn.out) are column sums of a
  // residual-stream gradient, and every such gradient is produced by a LayerNorm backward -> that kernel accumulates
  // them (its `dsum` output).  The caller's LN backward already filled this block's mlp.2.bias gradient from dx_out;
  // `dsum_next` is the bias gradient the produced dx_in belongs to (previous block's mlp.2.bias, or null).
  int block_bwd(const BlockP& bp, const BlockSave& s, Plan& p, const T* dx_out, T* scratch_a, T* scratch_b, long M,
                long Tq, bool causal, bool first_cross, float* dsum_next, const T** dx_in) {
    const int d = c->d;
    const T* xm = bp.cross ? s.x_mid2 : s.x_mid;
    // ---- MLP -----------------------------------------------------------------------------------------------
    RC(wgrad_side(dx_out, d, M, d, plain_view(s.hg, 4 * d), 4 * d, c->G(bp.w2), 4 * d));
    RC(dgrad(dx_out, M, d, c->template Wt<T>(bp.w2), 4 * d, s.u, nullptr, p.gu, c->G(bp.b1), true));  // s.u = GELU'(u); + fused mlp.0.bias gradient
    RC(wgrad_side(p.gu, 4 * d, M, 4 * d, plain_view(s.ln2, d), d, c->G(bp.w1), d));
    RC(dgrad(p.gu, M, 4 * d, c->template Wt<T>(bp.w1), d, nullptr, nullptr, p.gln));
    RC(join_side());
    RC(launch_layernorm_bwd(p.gln, xm, c->P(bp.mlp_ln_w), s.mean2, s.rstd2, dx_out, scratch_a, c->G(bp.mlp_ln_w), c->G(bp.mlp_ln_b),
                            c->G(bp.cross ? bp.cattn.ob : bp.attn.ob), M, d, st));
    const T* dx = scratch_a;
    T* nxt = scratch_b;
    // ---- cross attention ---------------------------------------------------------------------------------------
    if (bp.cross) {
      const long Mkv = (long)B * c->Te;
      RC(wgrad_side(dx, d, M, d, plain_view(s.ca.o, d), d, c->G(bp.cattn.ow), d));
      RC(dgrad(dx, M, d, c->template Wt<T>(bp.cattn.ow), d, nullptr, nullptr, p.go));
      Attn a;
      attn_args(a, s.ca, true, Tq, c->Te, false);
      a.d_o = p.go;
      a.delta = p.delta;
      a.dq = p.gq;
      a.dk = p.gkv;
      a.dv = p.gkv + d;
      a.dq_colsum = c->G(bp.cattn.qb);  // query / value bias gradients = column sums of dq / dv, fused into the store epilogues
      a.dv_colsum = c->G(bp.cattn.vb);
      a.colsum_scratch = p.cs_scratch;
      // decoder positions the loss ignores have d_o == 0 exactly (three quarters of the 448 on the synthetic lengths): the kernels
      // find those 64-position tiles themselves and skip them (span-limited step: the span says where they are, and the rows past
      // it are not even written)
      a.qtile_flags = dec_span ? nullptr : p.qtile_flags;
      a.q_span = dec_span;
      RC(join_big());  // (the previous layer's key|value gradients still read p.gkv)
      RC(launch_attention_bwd(a, st));
      RC(wgrad_side(p.gq, d, M, d, plain_view(s.ca.ln, d), d, c->G(bp.cattn.qw), d));
      {
        const bool big = (side_mode & 4) != 0;
        if (big) {
          RC(fork_to(c->side.big));
          big_pending = true;
        }
        OnStream on(st, big ? c->side.big : st);
        RC(wgrad(p.gkv, 2 * d, Mkv, 2 * d, plain_view(p.xa, d), d, c->G(bp.cattn.kw), d));
        // d(xa) accumulates over the decoder layers (bf16, like autograd's accumulation into xa.grad)
        RC(dgrad(p.gkv, Mkv, 2 * d, c->template Wt<T>(bp.cattn.kw), d, nullptr, first_cross ? nullptr : p.gxa, p.gxa));
      }
      RC(dgrad(p.gq, M, d, c->template Wt<T>(bp.cattn.qw), d, nullptr, nullptr, p.gln));
      RC(join_side());
      RC(launch_layernorm_bwd(p.gln, s.x_mid, c->P(bp.cln_w), s.ca.mean, s.ca.rstd, dx, nxt, c->G(bp.cln_w), c->G(bp.cln_b),
                              c->G(bp.attn.ob), M, d, st));
      const T* t = dx;
      dx = nxt;
      nxt = const_cast<T*>(t);
    }
    // ---- self attention ----------------------------------------------------------------------------------------
    RC(wgrad_side(dx, d, M, d, plain_view(s.sa.o, d), d, c->G(bp.attn.ow), d));
    RC(dgrad(dx, M, d, c->template Wt<T>(bp.attn.ow), d, nullptr, nullptr, p.go));
    Attn a;
    attn_args(a, s.sa, false, Tq, Tq, causal);
    a.d_o = p.go;
    a.delta = p.delta;
    a.dq = p.gqkv;
    a.dk = p.gqkv + d;
    a.dv = p.gqkv + 2 * d;
    a.dq_colsum = c->G(bp.attn.qb);
    a.dv_colsum = c->G(bp.attn.vb);
    a.colsum_scratch = p.cs_scratch;
    a.qtile_flags = (causal && !dec_span) ? p.qtile_flags : nullptr;  // (decoder blocks only: an encoder block's d_o has no zero rows)
    a.q_span = causal ? dec_span : nullptr;
    RC(launch_attention_bwd(a, st));
    RC(wgrad_side(p.gqkv, 3 * d, M, 3 * d, plain_view(s.sa.ln, d), d, c->G(bp.attn.qw), d));
    RC(dgrad(p.gqkv, M, 3 * d, c->template Wt<T>(bp.attn.qw), d, nullptr, nullptr, p.gln));
    RC(join_side());
    RC(launch_layernorm_bwd(p.gln, s.x_in, c->P(bp.attn_ln_w), s.sa.mean, s.sa.rstd, dx, nxt, c->G(bp.attn_ln_w), c->G(bp.attn_ln_b),
                            dsum_next, M, d, st));
    *dx_in = nxt;
    return OASR_OK;
  }
};

};  // struct Engine

}  // namespace

// ---- small kernels local to the engine -------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fused_bias_kernel(const float* __restrict__ qb, const float* __restrict__ vb, float* __restrict__ out,
                                                        int d) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < 3 * d; i += gridDim.x * 256)
    out[i] = i < d ? qb[i] : (i < 2 * d ? 0.f : vb[i - 2 * d]);
}
namespace {
int check_bound(const oasr_ctx* c, bool need_grads) {
  OASR_REQUIRE(c, "null context");
  if (!c->params || !c->shadow || !c->enc_pos || (need_grads && !c->grads)) {
    oasr_set_error("context not fully bound (oasr_bind / oasr_bind_shadow)");
    return OASR_ESTATE;
  }
  return OASR_OK;
}
}  // namespace

// ================================================ C ABI ============================================================
extern "C" oasr_ctx* oasr_create_ex2(const oasr_dims* dm, int embed_rows, int compute_dtype);
extern "C" oasr_ctx* oasr_create_ex(const oasr_dims* dm, int embed_rows) { return oasr_create_ex2(dm, embed_rows, OASR_DTYPE_BF16); }
extern "C" oasr_ctx* oasr_create(const oasr_dims* dm) { return oasr_create_ex2(dm, dm ? dm->n_vocab + 1 : 0, OASR_DTYPE_BF16); }

// embed_rows: rows of decoder.token_embedding -- n_vocab + 1 for the training model (pad row, olmoasr/model.py:665-667),
// n_vocab for the inference model (olmoasr/inf_model.py:302; checkpoints written by scripts/eval/gen_inf_ckpt.py)
// compute_dtype: OASR_DTYPE_BF16 = the production kernels; OASR_DTYPE_F32 = the fp32 validation kernels on the same
// schedule (reference: precision="float32", scripts/training/train_timestamps.py:2128,2220-2224)
extern "C" oasr_ctx* oasr_create_ex2(const oasr_dims* dm, int embed_rows, int compute_dtype) {
  if (compute_dtype != OASR_DTYPE_BF16 && compute_dtype != OASR_DTYPE_F32) {
    oasr_set_error("oasr_create_ex2: compute_dtype must be OASR_DTYPE_BF16 or OASR_DTYPE_F32");
    return nullptr;
  }
  if (!dm) {
    oasr_set_error("oasr_create: null dims");
    return nullptr;
  }
  if (embed_rows != dm->n_vocab && embed_rows != dm->n_vocab + 1) {
    oasr_set_error("oasr_create_ex: embed_rows must be n_vocab or n_vocab + 1");
    return nullptr;
  }
  const int d = dm->n_audio_state;
  if (dm->n_text_state != d || dm->n_audio_head != dm->n_text_head || d != 64 * dm->n_audio_head || dm->n_mels != 80 ||
      (d % 64) != 0 || d > 2048 || dm->n_text_ctx > 448 || dm->n_text_ctx < 1) {
    oasr_set_error("oasr_create: unsupported dims (need n_audio_state == n_text_state == 64*heads <= 2048, n_mels == 80, n_text_ctx <= 448)");
    return nullptr;
  }
  oasr_ctx* c = new oasr_ctx();
  c->f32 = compute_dtype == OASR_DTYPE_F32;
  c->dims = *dm;
  c->d = d;
  c->H = dm->n_audio_head;
  c->L_enc = dm->n_audio_layer;
  c->L_dec = dm->n_text_layer;
  c->Te = dm->n_audio_ctx;
  c->T1 = 2 * dm->n_audio_ctx;
  c->S_max = dm->n_text_ctx;
  c->V = embed_rows;
  c->Vp = c->f32 ? c->V : (c->V + 127) / 128 * 128;  // (the fp32 kernels need no padded vocabulary)
  c->aux_floats = 0;
  Builder b{c};
  {  // decoder.ln
    const int64_t s0 = b.cur;
    c->dec_ln_w = b.add("decoder.ln.weight", {d});
    c->dec_ln_b = b.add("decoder.ln.bias", {d});
    c->segments.push_back({s0, b.cur - s0});
  }
  c->dec.resize(c->L_dec);
  for (int i = c->L_dec - 1; i >= 0; --i) b.block("decoder.blocks." + std::to_string(i), c->dec[i], d, true);
  {
    const int64_t s0 = b.cur;
    c->dec_pos = b.add("decoder.positional_embedding", {dm->n_text_ctx, d});
    c->segments.push_back({s0, b.cur - s0});
  }
  const size_t emb_seg = c->segments.size();
  c->segments.push_back({0, 0});  // token embedding: becomes final here in time, lives at the arena's end
  {
    const int64_t s0 = b.cur;
    c->enc_lnp_w = b.add("encoder.ln_post.weight", {d});
    c->enc_lnp_b = b.add("encoder.ln_post.bias", {d});
    c->segments.push_back({s0, b.cur - s0});
  }
  c->enc.resize(c->L_enc);
  for (int i = c->L_enc - 1; i >= 0; --i) b.block("encoder.blocks." + std::to_string(i), c->enc[i], d, false);
  {
    const int64_t s0 = b.cur;
    c->conv2_w = b.add("encoder.conv2.weight", {d, d, 3});
    c->conv2_b = b.add("encoder.conv2.bias", {d});
    c->conv1_w = b.add("encoder.conv1.weight", {d, dm->n_mels, 3});
    c->conv1_b = b.add("encoder.conv1.bias", {d});
    c->segments.push_back({s0, b.cur - s0});
  }
  c->tok_emb = b.add("decoder.token_embedding.weight", {c->V, d});
  c->segments[emb_seg] = {c->tok_emb, (int64_t)c->V * d};
  c->numel = b.cur;
  {  // decoder layer 0's tensor offsets (decode_xcd.hip::XLayer order) + the per-layer strides: the one-launch step engine derives every
     // layer's addresses from them, so the blocks must be laid out back to back with one stride -- checked here, engine off otherwise
    auto offs = [](const BlockP& bp) {
      return std::vector<int64_t>{bp.attn_ln_w, bp.attn_ln_b, bp.attn.qw, bp.attn.fused_bias, bp.attn.ow, bp.attn.ob, bp.cln_w, bp.cln_b, bp.cattn.qw,
                                  bp.cattn.qb, bp.cattn.ow, bp.cattn.ob, bp.mlp_ln_w, bp.mlp_ln_b, bp.w1, bp.b1, bp.w2, bp.b2};
    };
    if (c->L_dec >= 1) {
      c->xcd_offsets = offs(c->dec[0]);
      c->xcd_lstride = c->xcd_astride = 0;
      if (c->L_dec >= 2) {
        const std::vector<int64_t> o1 = offs(c->dec[1]);
        c->xcd_lstride = o1[0] - c->xcd_offsets[0];
        c->xcd_astride = o1[3] - c->xcd_offsets[3];
      }
      bool regular = true;
      for (int l = 0; l < c->L_dec; ++l) {
        const std::vector<int64_t> ol = offs(c->dec[l]);
        for (int k = 0; k < 18; ++k) regular = regular && ol[k] == c->xcd_offsets[k] + (int64_t)l * (k == 3 ? c->xcd_astride : c->xcd_lstride);
      }
      if (!regular) c->xcd_offsets.clear();
    }
  }
  // shadow: [bf16 flat arena + zero pad rows for the padded vocab] [W1p d x 256] [W2p d x 3d] [aux fp32]
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  c->sh_flat = 0;
  const size_t esz = c->f32 ? 4 : 2;  // fp32 validation: no shadow of the flat arena (the master weights are the operands)
  size_t off = c->f32 ? 256 : al(((size_t)c->numel + (size_t)(c->Vp - c->V) * d + 64) * 2);
  c->sh_w1p = off;
  off = al(off + (size_t)d * 256 * esz);
  c->sh_w2p = off;
  off = al(off + (size_t)d * 3 * d * esz);
  c->sh_aux = off;
  off = al(off + (size_t)c->aux_floats * 4);
  c->sh_total = off;
  return c;
}
extern "C" void oasr_destroy(oasr_ctx* c) { delete c; }
extern "C" int oasr_compute_dtype(const oasr_ctx* c) { return c && c->f32 ? OASR_DTYPE_F32 : OASR_DTYPE_BF16; }
extern "C" int oasr_param_count(const oasr_ctx* c) { return c ? (int)c->tensors.size() : 0; }
extern "C" int64_t oasr_param_numel(const oasr_ctx* c) { return c ? c->numel : 0; }
extern "C" int oasr_param_info(const oasr_ctx* c, int idx, char* name, int name_cap, int64_t* offset, int64_t* numel, int* ndim,
                               int64_t shape[4]) {
  OASR_REQUIRE(c && idx >= 0 && idx < (int)c->tensors.size(), "param_info: bad index");
  const Tensor& t = c->tensors[idx];
  if (name && name_cap > 0) snprintf(name, name_cap, "%s", t.name.c_str());
  if (offset) *offset = t.off;
  if (numel) *numel = t.numel;
  if (ndim) *ndim = t.ndim;
  if (shape)
    for (int i = 0; i < 4; ++i) shape[i] = t.shape[i];
  return OASR_OK;
}
extern "C" int oasr_segment_count(const oasr_ctx* c) { return c ? (int)c->segments.size() : 0; }
extern "C" int oasr_segment_info(const oasr_ctx* c, int idx, int64_t* offset, int64_t* numel) {
  OASR_REQUIRE(c && idx >= 0 && idx < (int)c->segments.size(), "segment_info: bad index");
  if (offset) *offset = c->segments[idx].off;
  if (numel) *numel = c->segments[idx].numel;
  return OASR_OK;
}
extern "C" int oasr_bind(oasr_ctx* c, float* params, float* grads, float* m, float* v, const float* enc_pos) {
  OASR_REQUIRE(c && params && enc_pos, "oasr_bind: params and enc_pos are required");
  c->params = params;
  c->grads = grads;
  c->m = m;
  c->v = v;
  c->enc_pos = enc_pos;
  return OASR_OK;
}
extern "C" size_t oasr_shadow_bytes(const oasr_ctx* c) { return c ? c->sh_total : 0; }
extern "C" int oasr_bind_shadow(oasr_ctx* c, void* shadow) {
  OASR_REQUIRE(c && shadow, "oasr_bind_shadow: null");
  c->shadow = (char*)shadow;
  return OASR_OK;
}

static int refresh_packed(oasr_ctx* c, hipStream_t st) {
  const int d = c->d;
  if (c->f32) {
    RC(launch_pack_conv_weight(c->P(c->conv1_w), (float*)(c->shadow + c->sh_w1p), d, c->dims.n_mels, 256, st));
    RC(launch_pack_conv_weight(c->P(c->conv2_w), (float*)(c->shadow + c->sh_w2p), d, d, 3 * d, st));
  } else {
    RC(launch_pack_conv_weight(c->P(c->conv1_w), (bf16_t*)(c->shadow + c->sh_w1p), d, c->dims.n_mels, 256, st));
    RC(launch_pack_conv_weight(c->P(c->conv2_w), (bf16_t*)(c->shadow + c->sh_w2p), d, d, 3 * d, st));
  }
  float* aux = (float*)(c->shadow + c->sh_aux);
  auto fb = [&](const AttnP& a) {
    hipLaunchKernelGGL(fused_bias_kernel, dim3(cdiv(3 * d, 256)), dim3(256), 0, st, c->P(a.qb), c->P(a.vb), aux + a.fused_bias, d);
  };
  for (auto& b : c->enc) fb(b.attn);
  for (auto& b : c->dec) {
    fb(b.attn);
    fb(b.cattn);
  }
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}

extern "C" int oasr_refresh_shadow(oasr_ctx* c, void* stream) {
  RC(check_bound(c, false));
  hipStream_t st = (hipStream_t)stream;
  if (!c->f32) {
    bf16_t* flat = (bf16_t*)(c->shadow + c->sh_flat);
    RC(launch_cast_f32_bf16(c->params, flat, c->numel, st));
    OASR_CHECK_HIP(hipMemsetAsync(flat + c->numel, 0, ((size_t)(c->Vp - c->V) * c->d + 64) * 2, st));
  }
  return refresh_packed(c, st);
}

extern "C" size_t oasr_workspace_bytes(const oasr_ctx* c, int B, int S, int mode) {
  if (!c || B <= 0 || S <= 0) return 0;
  Arena A(nullptr, 0);
  if (c->f32) {
    Engine<float>::Plan p;
    Engine<float>::make_plan(c, A, p, B, S, mode == OASR_MODE_TRAIN);
  } else {
    Engine<bf16_t>::Plan p;
    Engine<bf16_t>::make_plan(c, A, p, B, S, mode == OASR_MODE_TRAIN);
  }
  return A.cur + 4096;
}

template <typename T>
static int oasr_forward_impl(oasr_ctx* c, const float* mel, const int64_t* tokens, const int32_t* text_len, int B, int S,
                            float* logits_out, void* xa_out, void* workspace, size_t workspace_bytes, void* stream) {
  RC(check_bound(c, false));
  OASR_REQUIRE(mel && tokens && workspace && B > 0 && S > 0 && S <= c->S_max, "oasr_forward: bad args (B=%d S=%d)", B, S);
  OASR_REQUIRE(workspace_bytes >= oasr_workspace_bytes(c, B, S, OASR_MODE_INFER), "oasr_forward: workspace too small");
  Arena A(workspace, workspace_bytes);
  typename Engine<T>::Plan p;
  Engine<T>::make_plan(c, A, p, B, S, false);
  typename Engine<T>::Runner r{c, (hipStream_t)stream, B, S, text_len};
  RC(r.encoder_fwd(p, mel));
  RC(r.decoder_fwd(p, tokens));
  if (xa_out)
    OASR_CHECK_HIP(hipMemcpyAsync(xa_out, p.xa, (size_t)B * c->Te * c->d * sizeof(T), hipMemcpyDeviceToDevice, r.st));
  if (logits_out) RC(launch_logits_to_f32(p.logits, c->Vp, (long)B * S, c->V, logits_out, r.st));
  return OASR_OK;
}
extern "C" int oasr_forward(oasr_ctx* c, const float* mel, const int64_t* tokens, const int32_t* text_len, int B, int S,
                            float* logits_out, void* xa_out, void* workspace, size_t workspace_bytes, void* stream) {
  OASR_REQUIRE(c, "oasr_forward: null context");
  return c->f32 ? oasr_forward_impl<float>(c, mel, tokens, text_len, B, S, logits_out, xa_out, workspace, workspace_bytes, stream) : oasr_forward_impl<bf16_t>(c, mel, tokens, text_len, B, S, logits_out, xa_out, workspace, workspace_bytes, stream);
}

// AudioEncoder.forward (olmoasr/model.py:571-623): mel -> xa bf16 [B, n_audio_ctx, d]
template <typename T>
static int oasr_encode_impl(oasr_ctx* c, const float* mel, int B, void* xa_out, void* workspace, size_t workspace_bytes, void* stream) {
  RC(check_bound(c, false));
  OASR_REQUIRE(mel && xa_out && workspace && B > 0, "oasr_encode: bad args");
  OASR_REQUIRE(workspace_bytes >= oasr_workspace_bytes(c, B, 1, OASR_MODE_INFER), "oasr_encode: workspace too small");
  Arena A(workspace, workspace_bytes);
  typename Engine<T>::Plan p;
  Engine<T>::make_plan(c, A, p, B, 1, false);
  typename Engine<T>::Runner r{c, (hipStream_t)stream, B, 1, nullptr};
  RC(r.encoder_fwd(p, mel));
  OASR_CHECK_HIP(hipMemcpyAsync(xa_out, p.xa, (size_t)B * c->Te * c->d * sizeof(T), hipMemcpyDeviceToDevice, r.st));
  return OASR_OK;
}
extern "C" int oasr_encode(oasr_ctx* c, const float* mel, int B, void* xa_out, void* workspace, size_t workspace_bytes, void* stream) {
  OASR_REQUIRE(c, "oasr_encode: null context");
  return c->f32 ? oasr_encode_impl<float>(c, mel, B, xa_out, workspace, workspace_bytes, stream) : oasr_encode_impl<bf16_t>(c, mel, B, xa_out, workspace, workspace_bytes, stream);
}

// TextDecoder.forward without kv_cache (olmoasr/model.py:688-775) on given audio features: OLMoASR.logits(tokens, xa).
// last_only != 0: logits_out is f32 [B, rows] for position S-1 only (greedy decode step); else f32 [B, S, rows].
template <typename T>
static int oasr_decode_logits_impl(oasr_ctx* c, const int64_t* tokens, const void* xa, const int32_t* text_len, int B, int S,
                                  int last_only, float* logits_out, void* workspace, size_t workspace_bytes, void* stream) {
  RC(check_bound(c, false));
  OASR_REQUIRE(tokens && xa && logits_out && workspace && B > 0 && S > 0 && S <= c->S_max, "oasr_decode_logits: bad args");
  OASR_REQUIRE(workspace_bytes >= oasr_workspace_bytes(c, B, S, OASR_MODE_INFER), "oasr_decode_logits: workspace too small");
  Arena A(workspace, workspace_bytes);
  typename Engine<T>::Plan p;
  Engine<T>::make_plan(c, A, p, B, S, false);
  typename Engine<T>::Runner r{c, (hipStream_t)stream, B, S, text_len};
  OASR_CHECK_HIP(hipMemcpyAsync(p.xa, xa, (size_t)B * c->Te * c->d * sizeof(T), hipMemcpyDeviceToDevice, r.st));
  RC(r.decoder_fwd(p, tokens, last_only != 0));
  return launch_logits_to_f32(p.logits, c->Vp, last_only ? (long)B : (long)B * S, c->V, logits_out, r.st);
}
extern "C" int oasr_decode_logits(oasr_ctx* c, const int64_t* tokens, const void* xa, const int32_t* text_len, int B, int S,
                                  int last_only, float* logits_out, void* workspace, size_t workspace_bytes, void* stream) {
  OASR_REQUIRE(c, "oasr_decode_logits: null context");
  return c->f32 ? oasr_decode_logits_impl<float>(c, tokens, xa, text_len, B, S, last_only, logits_out, workspace, workspace_bytes, stream) : oasr_decode_logits_impl<bf16_t>(c, tokens, xa, text_len, B, S, last_only, logits_out, workspace, workspace_bytes, stream);
}

// ---- cached greedy decoding (OLMoASR.install_kv_cache_hooks, olmoasr/model.py:925-964 / inf_model.py:422-453) -----------
// The reference caches every key/value Linear output in a dict via forward hooks (self-attention K/V grow by torch.cat per
// token, cross-attention K/V are computed once per window).  Here the cache is one caller-owned buffer:
//   per decoder layer: self Q|K|V [B, n_text_ctx, 3d] | cross KV [B, n_audio_ctx, 2d]   (bf16)
// oasr_decode_begin fills the cross K/V of all layers from xa; oasr_decode_step runs the decoder on ONE new token per
// sequence at position `pos`: ONE fused q|k|v projection writes straight into the cache row of that position (GEMM output
// row stride = one sequence's cache; the q slot is scratch that keeps the three projections in a single launch),
// attention reads q from that row and the first pos+1 cached keys/values through strides.
extern "C" size_t oasr_kv_cache_bytes(const oasr_ctx* c, int B) {
  if (!c || B <= 0) return 0;
  const size_t per_layer = ((size_t)3 * B * c->S_max * c->d + (size_t)B * c->Te * 2 * c->d) * (c->f32 ? 4 : 2);
  return per_layer * c->L_dec + OASR_KV_TAIL_BYTES;
}
namespace {
template <typename T>
struct KvLayer {
  T *qkv, *ckv;  // self [B, S_max, 3d] (q | k | v per position), cross [B, Te, 2d]
};
template <typename T>
KvLayer<T> kv_layer(const oasr_ctx* c, void* cache, int B, int layer) {
  const size_t per_layer = (size_t)3 * B * c->S_max * c->d + (size_t)B * c->Te * 2 * c->d;
  T* base = (T*)cache + per_layer * layer;
  return KvLayer<T>{base, base + (size_t)3 * B * c->S_max * c->d};
}
unsigned* kv_ctrl(const oasr_ctx* c, void* cache, int B) {  // the control tail behind the last layer (oasr_kv_cache_bytes)
  const size_t per_layer = ((size_t)3 * B * c->S_max * c->d + (size_t)B * c->Te * 2 * c->d) * (c->f32 ? 4 : 2);
  return (unsigned*)((char*)cache + per_layer * c->L_dec);
}
// A/B and test switch of the step engine: -1 = default (ONE sequence on the bf16 engine: the chip-wide one-launch engine of decode_wide.hip; 2-4
// sequences: LayerNorm folded into the projections; more: separate kernels), 0 = separate LayerNorm kernels, 1 = LayerNorm folded into the projections'
// operand loads (the round-2/3 default for B <= 4) for every B <= 32, 2 = the one-launch team engine of decode_xcd.hip on one XCD, 3 = that team as 32
// workgroups spread over the chip, 4 = the same with 64, 5 = the chip-wide engine.  0-4 bit-identical, 5 within fp32 summation-order rounding
// (tests/test_gpu_decode_step.py).
int g_decode_ln_fold = -1;
}  // namespace
// Side streams of the supervised-span step (Runner::side_mode): the setter is a testing hook, OASR_SIDE_STREAMS an experiment switch
static int g_side_streams = -1;
static int span_side_streams() {
  if (g_side_streams >= 0) return g_side_streams;
  static const int env = [] {
    const char* e = oasr_experiment_env("OASR_SIDE_STREAMS");
    return e ? atoi(e) : -1;
  }();
  return env >= 0 ? (env & 15) : SIDE_STREAMS_DEFAULT;
}
extern "C" int oasr_span_side_streams(void) { return span_side_streams(); }
extern "C" int oasr_span_set_side_streams(int mode) {
  const char* e = getenv("OASR_TESTING_HOOKS");
  if (!(e && e[0] == '1')) {
    oasr_set_error("oasr_span_set_side_streams: testing hook called without OASR_TESTING_HOOKS=1 (include/oasr_testing.h)");
    return OASR_ESTATE;
  }
  g_side_streams = mode < 0 ? -1 : (mode & 15);
  return OASR_OK;
}
extern "C" int oasr_decode_set_ln_fold(int mode) {
  {  // a testing hook (include/oasr_testing.h): inert without OASR_TESTING_HOOKS=1
    const char* e = getenv("OASR_TESTING_HOOKS");
    if (!(e && e[0] == '1')) {
      oasr_set_error("oasr_decode_set_ln_fold: testing hook called without OASR_TESTING_HOOKS=1 (include/oasr_testing.h)");
      return OASR_ESTATE;
    }
  }
  g_decode_ln_fold = mode < 0 ? -1 : (mode > 5 ? 1 : mode);
  return OASR_OK;
}

extern "C" size_t oasr_decode_step_workspace_bytes(const oasr_ctx* c, int B) {
  if (!c || B <= 0) return 0;
  // x, ln, q, o, x2 (5 * B*d) + u, hg (2 * B*4d) + logits (B*Vp) bf16 + stats
  return ((size_t)B * c->d * 6 + (size_t)B * 8 * c->d + (size_t)B * c->Vp + 9 * 32) * (c->f32 ? 4 : 2) + (size_t)B * c->H * 8 + (size_t)B * 16 + 8192 +
         (B <= 4 ? (decode_xcd_part_floats(B, c->H, c->Te) + decode_wide_part_floats(c->H)) * 4 + 256 + 512 : 0);
}

template <typename T>
static int oasr_decode_begin_impl(oasr_ctx* c, const void* xa, int B, void* kv_cache, void* stream) {
  RC(check_bound(c, false));
  OASR_REQUIRE(xa && kv_cache && B > 0, "oasr_decode_begin: bad args");
  typename Engine<T>::Runner r{c, (hipStream_t)stream, B, 1, nullptr};
  const int d = c->d;
  // the one-launch step engine's control words (barrier counter, error flag, epoch base, XCC mask) live in the cache's 256-byte tail
  OASR_CHECK_HIP(hipMemsetAsync(kv_ctrl(c, kv_cache, B), 0, OASR_KV_TAIL_BYTES, (hipStream_t)stream));
  for (int i = 0; i < c->L_dec; ++i) {
    const BlockP& bp = c->dec[i];
    KvLayer<T> kl = kv_layer<T>(c, kv_cache, B, i);
    RC(r.linear((const T*)xa, (long)B * c->Te, d, c->template Wt<T>(bp.cattn.kw), 2 * d, c->aux(bp.cattn.fused_bias) + d, 0, nullptr, kl.ckv, nullptr));
  }
  return OASR_OK;
}
extern "C" int oasr_decode_begin(oasr_ctx* c, const void* xa, int B, void* kv_cache, void* stream) {
  OASR_REQUIRE(c, "oasr_decode_begin: null context");
  return c->f32 ? oasr_decode_begin_impl<float>(c, xa, B, kv_cache, stream) : oasr_decode_begin_impl<bf16_t>(c, xa, B, kv_cache, stream);
}

// tokens_last i64 [B]: the token at position pos of every sequence.  logits_out f32 [B, rows] for the NEXT position.
template <typename T>
static int oasr_decode_step_impl(oasr_ctx* c, const int64_t* tokens_last, int B, int pos, void* kv_cache, float* logits_out,
                                void* workspace, size_t workspace_bytes, void* stream) {
  RC(check_bound(c, false));
  OASR_REQUIRE(tokens_last && kv_cache && logits_out && workspace && B > 0 && pos >= 0 && pos < c->S_max, "oasr_decode_step: bad args");
  OASR_REQUIRE(workspace_bytes >= oasr_decode_step_workspace_bytes(c, B), "oasr_decode_step: workspace too small");
  const int d = c->d, S_max = c->S_max;
  hipStream_t st = (hipStream_t)stream;
  typename Engine<T>::Runner r{c, st, B, 1, nullptr};
  Arena A(workspace, workspace_bytes);
  T* x = A.template act<T>((size_t)B * d);
  T* ln = A.template act<T>((size_t)B * d);
  T* q = A.template act<T>((size_t)B * d);
  T* o = A.template act<T>((size_t)B * d);
  T* x2 = A.template act<T>((size_t)B * d);
  T* x3 = A.template act<T>((size_t)B * d);
  T* u = A.template act<T>((size_t)B * 4 * d);
  T* hg = A.template act<T>((size_t)B * 4 * d);
  T* logits = A.template act<T>((size_t)B * c->Vp);
  float* lse = A.f32((size_t)B * c->H);
  float* mean = A.f32(B);
  float* rstd = A.f32(B);
  // token + positional embedding of position pos: S = 1 per sequence, positional row offset by pos
  RC(launch_embedding_fwd(tokens_last, c->P(c->tok_emb), c->P(c->dec_pos) + (size_t)pos * d, x, B, 1, d, c->V, st));
  T* cur = x;
  // a few sequences on the bf16 engine: every LayerNorm rides in the operand load of the projection that consumes it and the
  // logits leave as fp32 (8 launches per layer instead of 11; bit-identical to the separate kernels below).  Measured
  // (profiles/r02_decode_step.txt): -5 % per step at B = 1, but every workgroup recomputes the B row statistics, which loses
  // from B = 16 on (+20 %) -- so only small batches take it (oasr_decode_set_ln_fold forces either side for the A/B and the
  // bit-identity test).
  bool folded = false;
  if constexpr (std::is_same<T, bf16_t>::value) {
    folded = d % 64 == 0 && d <= 2048 && g_decode_ln_fold != 0 && (B <= 4 || (g_decode_ln_fold == 1 && B <= 32));
    // one launch for the whole decoder stack (decode_xcd.hip): the default for a handful of sequences
    const int mode = g_decode_ln_fold;
    // (default: ONE sequence -- the timestamp-mode transcribe loop -- on the chip-wide one-launch engine, decode_wide.hip: 0.74 ms per token at medium
    // against 1.76 for the one-XCD team of decode_xcd.hip and 2.44 multi-launch, profiles/r06_decode_wide.txt; at small B = 4 the multi-launch
    // kernels, which spread over the whole chip, win: profiles/r05_decode_xcd_probe_v8.txt.  Mode 5 forces the chip-wide engine, modes 2-4 the
    // team engine up to B = 4.)
    if (c->n_cu == 0) {
      int dev = 0, n = 0;
      OASR_CHECK_HIP(hipGetDevice(&dev));
      OASR_CHECK_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
      c->n_cu = n > 0 ? n : -1;
    }
    const int nwg = c->n_cu >= 256 ? 256 : (c->n_cu > 0 ? c->n_cu & ~3 : 0);
    const bool wide = (mode == -1 || mode == 5) && B == 1 && !c->xcd_disabled && !c->xcd_offsets.empty() &&
                      decode_wide_supports(d, c->H, c->Te, S_max, c->L_dec, B, nwg);
    const bool team = !wide && ((mode == -1 && B == 1) || mode >= 2) && !c->xcd_disabled && !c->xcd_offsets.empty() &&
                      decode_xcd_supports(d, c->H, c->Te, S_max, c->L_dec, B) &&
                      decode_xcd_offsets_ok(c->xcd_offsets.data(), c->xcd_lstride, (long)3 * B * S_max * d + (long)B * c->Te * 2 * d, d, c->Te, c->L_dec, B);
    if (wide || team) {
      DecodeXcdArgs xa;
      xa.wflat = c->template Wt<bf16_t>(0);
      xa.params = c->params;
      xa.aux = c->aux(0);
      xa.cache = (bf16_t*)kv_cache;
      xa.cache_lstride = (long)3 * B * S_max * d + (long)B * c->Te * 2 * d;
      xa.x = x, xa.x2 = x2, xa.x3 = x3, xa.q = q, xa.o = o, xa.hg = hg;
      xa.part = A.f32(decode_xcd_part_floats(B, c->H, c->Te));
      xa.ctrl = kv_ctrl(c, kv_cache, B);
      xa.d = d, xa.H = c->H, xa.Te = c->Te, xa.S_max = S_max, xa.L = c->L_dec, xa.M = B, xa.pos = pos;
      xa.team = mode == 4 ? 64 : 32;
      xa.stride = (mode == 3 || mode == 4) ? 1 : 8;
      if (wide) xa.team = nwg, xa.stride = 1, xa.part = A.f32(decode_wide_part_floats(c->H));
      {  // measurement hooks (scripts/decode_xcd_probe.py; inert without OASR_TESTING_HOOKS=1): experiment flags, in-kernel stamps in the workspace tail
        static const int xflags = [] {
          const char* e = oasr_experiment_env("OASR_XCD_FLAGS");
          return e ? atoi(e) : 0;
        }();
        xa.flags = (xflags & 0xff) | (((xflags >> 9) & 0xff) << 8);  // (bits 9-16: the workgroup whose stamps the chip-wide engine takes)
        xa.stamps = (xflags & 0x100) ? (void*)((char*)workspace + workspace_bytes - 512) : nullptr;
      }
      xa.layer_offsets = c->xcd_offsets.data();
      xa.lstride = c->xcd_lstride, xa.astride = c->xcd_astride;
      if (wide) {  // ... and the final LayerNorm + logits projection as its last phase: one launch per token behind the embedding
        xa.w_logits = c->template Wt<bf16_t>(c->tok_emb), xa.lnf_g = c->P(c->dec_ln_w), xa.lnf_b = c->P(c->dec_ln_b), xa.logits_out = logits_out, xa.V = c->V;
        return launch_decode_wide(xa, st);
      }
      RC(launch_decode_xcd(xa, st));
      return launch_decode_proj(x, B, d, c->template Wt<bf16_t>(c->tok_emb), c->V, c->P(c->dec_ln_w), c->P(c->dec_ln_b), nullptr, 0, nullptr, 0,
                                nullptr, 0, logits_out, c->V, st);
    }
  }
  for (int i = 0; i < c->L_dec; ++i) {
    const BlockP& bp = c->dec[i];
    KvLayer<T> kl = kv_layer<T>(c, kv_cache, B, i);
    if (!folded) RC(launch_layernorm_fwd(cur, c->P(bp.attn_ln_w), c->P(bp.attn_ln_b), ln, mean, rstd, B, d, st));
    if constexpr (std::is_same<T, bf16_t>::value) {
      if (folded)
        RC(launch_decode_proj(cur, B, d, c->template Wt<bf16_t>(bp.attn.qw), 3 * d, c->P(bp.attn_ln_w), c->P(bp.attn_ln_b),
                              c->aux(bp.attn.fused_bias), 0, nullptr, 0, kl.qkv + (size_t)pos * 3 * d, (long)S_max * 3 * d, nullptr, 0, st));
    }
    if (!folded) {  // q | k | v of this position in one launch, straight into the cache: output row b lands at [b, pos, 0:3d]
      GemmArgsT<T> g = gemm_defaults_t<T>();
      g.A = plain_view(ln, d);
      g.B = plain_view(c->template Wt<T>(bp.attn.qw), d);  // query | key | value weights are adjacent in the arena
      g.M = B;
      g.N = 3 * d;
      g.K = d;
      g.bias = c->aux(bp.attn.fused_bias);    // [q_bias | 0 | v_bias]
      g.ldc = (long)S_max * 3 * d;
      g.out = kl.qkv + (size_t)pos * 3 * d;
      RC(launch_gemm(g, st));
    }
    AttnArgsT<T> a;
    memset(&a, 0, sizeof(a));
    a.q = kl.qkv + (size_t)pos * 3 * d;
    a.ldq = 3 * d;
    a.bsq = (long)S_max * 3 * d;
    a.k = kl.qkv + d;
    a.v = kl.qkv + 2 * d;
    a.ldk = a.ldv = 3 * d;
    a.bsk = a.bsv = (long)S_max * 3 * d;
    a.o = o;
    a.ldo = d;
    a.bso = d;
    a.lse = lse;
    a.B = B;
    a.H = c->H;
    a.Tq = 1;
    a.Tk = pos + 1;
    RC(launch_attention_fwd(a, st));
    RC(r.linear(o, B, d, c->template Wt<T>(bp.attn.ow), d, c->P(bp.attn.ob), 0, cur, x2, nullptr));
    if constexpr (std::is_same<T, bf16_t>::value) {
      if (folded)
        RC(launch_decode_proj(x2, B, d, c->template Wt<bf16_t>(bp.cattn.qw), d, c->P(bp.cln_w), c->P(bp.cln_b), c->P(bp.cattn.qb), 0,
                              nullptr, 0, q, d, nullptr, 0, st));
    }
    if (!folded) {
      RC(launch_layernorm_fwd(x2, c->P(bp.cln_w), c->P(bp.cln_b), ln, mean, rstd, B, d, st));
      RC(r.linear(ln, B, d, c->template Wt<T>(bp.cattn.qw), d, c->P(bp.cattn.qb), 0, nullptr, q, nullptr));
    }
    a.q = q;
    a.ldq = d;
    a.bsq = d;
    a.k = kl.ckv;
    a.v = kl.ckv + d;
    a.ldk = a.ldv = 2 * d;
    a.bsk = a.bsv = (long)c->Te * 2 * d;
    a.Tk = c->Te;
    RC(launch_attention_fwd(a, st));
    RC(r.linear(o, B, d, c->template Wt<T>(bp.cattn.ow), d, c->P(bp.cattn.ob), 0, x2, x3, nullptr));
    if constexpr (std::is_same<T, bf16_t>::value) {
      if (folded)
        RC(launch_decode_proj(x3, B, d, c->template Wt<bf16_t>(bp.w1), 4 * d, c->P(bp.mlp_ln_w), c->P(bp.mlp_ln_b), c->P(bp.b1), 1, nullptr,
                              0, hg, 4 * d, nullptr, 0, st));
    }
    if (!folded) {
      RC(launch_layernorm_fwd(x3, c->P(bp.mlp_ln_w), c->P(bp.mlp_ln_b), ln, mean, rstd, B, d, st));
      RC(r.linear(ln, B, d, c->template Wt<T>(bp.w1), 4 * d, c->P(bp.b1), 1, nullptr, hg, u));
    }
    RC(r.linear(hg, B, 4 * d, c->template Wt<T>(bp.w2), d, c->P(bp.b2), 0, x3, cur == x ? x2 : x, nullptr));
    cur = (cur == x) ? x2 : x;
  }
  if constexpr (std::is_same<T, bf16_t>::value) {
    if (folded)
      return launch_decode_proj(cur, B, d, c->template Wt<bf16_t>(c->tok_emb), c->V, c->P(c->dec_ln_w), c->P(c->dec_ln_b), nullptr, 0, nullptr,
                                0, nullptr, 0, logits_out, c->V, st);
  }
  RC(launch_layernorm_fwd(cur, c->P(c->dec_ln_w), c->P(c->dec_ln_b), ln, mean, rstd, B, d, st));
  RC(r.linear(ln, B, d, c->template Wt<T>(c->tok_emb), c->Vp, nullptr, 0, nullptr, logits, nullptr));
  return launch_logits_to_f32(logits, c->Vp, B, c->V, logits_out, st);
}
extern "C" int oasr_decode_step(oasr_ctx* c, const int64_t* tokens_last, int B, int pos, void* kv_cache, float* logits_out,
                                void* workspace, size_t workspace_bytes, void* stream) {
  OASR_REQUIRE(c, "oasr_decode_step: null context");
  return c->f32 ? oasr_decode_step_impl<float>(c, tokens_last, B, pos, kv_cache, logits_out, workspace, workspace_bytes, stream) : oasr_decode_step_impl<bf16_t>(c, tokens_last, B, pos, kv_cache, logits_out, workspace, workspace_bytes, stream);
}

// Synchronises the stream before the caller reads a window's tokens back (the step engines themselves cannot fail once enqueued).
extern "C" int oasr_decode_check(oasr_ctx* c, int B, void* kv_cache, void* stream) {
  OASR_REQUIRE(c && kv_cache && B > 0, "oasr_decode_check: bad args");
  OASR_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  unsigned ctrl[4] = {0, 0, 0, 0};  // the one-launch step engine's control words: a poisoned team barrier / a desynchronised stream is an error
  OASR_CHECK_HIP(hipMemcpy(ctrl, kv_ctrl(c, kv_cache, B), sizeof(ctrl), hipMemcpyDeviceToHost));
  if (ctrl[1] != 0) {
    // The one-launch engine needs its whole team (32 workgroups x ~160 KB of LDS on one XCD) resident at once; a second decoder on the same
    // device, or a CU-masked / partitioned device, can leave part of a team queued behind the rest, and the bounded spin then poisons the
    // barrier instead of hanging.  Nothing is wrong with the cache's K/V rows written before that step, but the window's tokens are: the
    // context falls back to the multi-launch engine for good, the control words are cleared, and the caller re-decodes the window
    // (olmoasr_amd.decoding.decode does; OASR_ERETRY says "same call again").
    c->xcd_disabled = true;
    OASR_CHECK_HIP(hipMemset(kv_ctrl(c, kv_cache, B), 0, OASR_KV_TAIL_BYTES));
    oasr_set_error("oasr_decode_check: the one-launch decoder step reported 0x%x (1 = a team member never reached a barrier -- is the device "
                   "shared or CU-masked? --, 0x1xx = block stream out of step); XCC mask 0x%x.  The one-launch engine is now disabled for this "
                   "context; decode the window again (it will run on the multi-launch engine)", ctrl[1], ctrl[3]);
    return OASR_ERETRY;
  }
  return OASR_OK;
}

extern "C" int oasr_zero_grad(oasr_ctx* c, void* stream) {
  RC(check_bound(c, true));
  OASR_CHECK_HIP(hipMemsetAsync(c->grads, 0, (size_t)c->numel * 4, (hipStream_t)stream));
  return OASR_OK;
}

extern "C" int oasr_train_fwd_bwd(oasr_ctx* c, const float* mel, const int64_t* tokens, const int64_t* targets, const int32_t* text_len,
                                  int B, float loss_scale, float inv_accum, float* loss_out, int accumulate_loss, float* logits_out,
                                  void** ev, void* workspace, size_t workspace_bytes, void* stream) {
  return oasr_train_fwd_bwd_s(c, mel, tokens, targets, text_len, B, c ? c->S_max : 0, loss_scale, inv_accum, loss_out, accumulate_loss,
                              logits_out, ev, workspace, workspace_bytes, stream);
}

// Same step over a decoder context of S <= n_text_ctx positions (tokens / targets are [B, S]).  With S >= max(text_len)
// rounded up, the loss, every gradient and therefore the optimizer step are those of the full padded context: positions
// past the last real token only ever see ignore_index targets, and no real query attends to them (causal mask), so
// the reference spends their share of the decoder on exact zeros (train_timestamps.py:318-329 pads every sample to 448).
// The backward half of a training micro-step: p.logits holds d(loss)/d(logits) (bf16 engine: bf16 [Md, Vp]) on entry -- written in
// place by the fused cross-entropy (oasr_train_fwd_bwd*) or converted from the caller's fp32 tensor (oasr_train_bwd, the
// torch.autograd path) -- and every saved activation of the forward is still in the workspace.
template <typename T>
static int train_backward(oasr_ctx* c, typename Engine<T>::Runner& r, typename Engine<T>::Plan& p, const int64_t* tokens, int B, int S, void** ev) {
  const int d = c->d;
  // Md: the decoder's token rows the backward runs over -- all B*S, or (supervised-span step) the leading rows that hold every
  // position able to carry gradient; the rows behind them are never read or written by the backward
  const long Md = r.dec_rows_bwd ? r.dec_rows_bwd : (long)B * S, Me = (long)B * c->Te, M1 = (long)B * c->T1;
  hipStream_t st = r.st;
  // ---------------- backward: decoder ----------------
  int seg = 0;
  // tied logits: dE += dlogits^T . lnf ; d(lnf) = dlogits . E
  // (V = n_vocab + 1 is odd: the direct-to-LDS kernel wants a multiple of 8 rows, so the pad class gets its own 1-row GEMM)
  {
    const int v8 = c->V & ~7;
    RC(r.wgrad(p.logits, c->Vp, Md, v8, plain_view(p.lnf, d), d, c->G(c->tok_emb), d));
    if (v8 < c->V) RC(r.wgrad(p.logits + v8, c->Vp, Md, c->V - v8, plain_view(p.lnf, d), d, c->G(c->tok_emb) + (long)v8 * d, d));
  }
  RC(r.dgrad(p.logits, Md, c->Vp, c->template Wt<T>(c->tok_emb), d, nullptr, nullptr, p.gln));
  const T* x_last = c->L_dec ? p.dec[c->L_dec - 1].x_out : p.dx0;
  RC(launch_layernorm_bwd(p.gln, x_last, c->P(c->dec_ln_w), p.mean_f, p.rstd_f, nullptr, p.ga, c->G(c->dec_ln_w), c->G(c->dec_ln_b),
                          c->L_dec ? c->G(c->dec[c->L_dec - 1].b2) : nullptr, Md, d, st));
  RC(r.record(ev, seg++));
  const T* dx = p.ga;
  auto others = [&](const T* cur, T** a, T** b) {  // the two stream-gradient buffers that are not `cur`
    T* all[3] = {p.ga, p.gb, p.gc};
    int n = 0;
    T* o[2] = {nullptr, nullptr};
    for (int j = 0; j < 3; ++j)
      if (all[j] != cur && n < 2) o[n++] = all[j];
    *a = o[0];
    *b = o[1];
  };
  for (int i = c->L_dec - 1; i >= 0; --i) {
    T *sa, *sb;
    others(dx, &sa, &sb);
    const T* dx_in = nullptr;
    RC(r.block_bwd(c->dec[i], p.dec[i], p, dx, sa, sb, Md, S, true, i == c->L_dec - 1, i > 0 ? c->G(c->dec[i - 1].b2) : nullptr, &dx_in));
    dx = dx_in;
    // the block's event says "every gradient of this block is complete" (the DDP reducer sends the bucket on it): that includes the
    // key|value weight gradient on the side stream (bit 3, experiments without events: leave it in flight until the next block needs p.gkv)
    if (ev || !(r.side_mode & 8)) RC(r.join_big());
    RC(r.record(ev, seg++));
  }
  RC(r.join_side());
  RC(r.join_big());
  r.side_mode = 0;  // (the encoder's 192k-row GEMMs fill the chip on their own)
  gemm_profile_lane(0);
  RC(launch_embedding_bwd(tokens, dx, c->G(c->tok_emb), c->G(c->dec_pos), B, S, d, PAD_ID, c->V, st, r.dec_rows, r.dec_span));
  RC(r.record(ev, seg++));  // decoder.positional_embedding
  RC(r.record(ev, seg++));  // token embedding (arena tail)

  // ---------------- backward: encoder ----------------
  const T* xe_last = c->L_enc ? p.enc[c->L_enc - 1].x_out : p.x0;
  if (c->L_dec == 0) OASR_CHECK_HIP(hipMemsetAsync(p.gxa, 0, (size_t)Me * d * sizeof(T), st));
  RC(launch_layernorm_bwd(p.gxa, xe_last, c->P(c->enc_lnp_w), p.mean_p, p.rstd_p, nullptr, p.ga, c->G(c->enc_lnp_w), c->G(c->enc_lnp_b),
                          c->L_enc ? c->G(c->enc[c->L_enc - 1].b2) : nullptr, Me, d, st));
  RC(r.record(ev, seg++));
  dx = p.ga;
  for (int i = c->L_enc - 1; i >= 0; --i) {
    T *sa, *sb;
    others(dx, &sa, &sb);
    const T* dx_in = nullptr;
    RC(r.block_bwd(c->enc[i], p.enc[i], p, dx, sa, sb, Me, c->Te, false, false, i > 0 ? c->G(c->enc[i - 1].b2) : nullptr, &dx_in));
    dx = dx_in;
    RC(r.record(ev, seg++));
  }
  // conv stem: x0 = gelu(u2) + pos ; u2 = conv2(h1) ; h1 = gelu(u1) ; u1 = conv1(mel)
  {
    RC(launch_dgelu_mul(dx, p.u2, p.gln, Me * d, st));  // gln = d(u2)
    OASR_CHECK_HIP(hipMemsetAsync(p.tmp_w2p, 0, (size_t)d * 3 * d * 4, st));
    // conv2 weight gradient on the direct-to-LDS kernel: the im2col matrix [B*1500][3d] is h1 itself read as overlapping
    // rows of 3d elements at stride 2d from h1 - d (per-sample stride 3000*d == 1500 rows * 2d, so the view is plain).
    // Only window (b, t = 0) is wrong in its first d elements (it sees the last row of sample b-1, or the zeroed guard row,
    // instead of the left zero padding); that rank-B term is subtracted by a second, tiny GEMM over the B first rows.
    RC(r.wgrad(p.gln, d, Me, d, plain_view(p.h1 - d, 2L * d), 3 * d, p.tmp_w2p, 3 * d));  // (guard row zeroed by the forward)
    {
      GemmArgsT<T> g = gemm_defaults_t<T>();
      g.A = plain_view(p.gln, (long)c->Te * d);          // dY rows (b, t = 0)
      g.ta = 1;
      g.B = plain_view(p.h1 - d, (long)c->T1 * d);       // what those windows wrongly saw as their first tap
      g.tb = 1;
      g.M = d;
      g.N = d;
      g.K = B;
      g.alpha = -1.0f;
      g.out_f32 = p.tmp_w2p;
      g.ldc32 = 3 * d;
      g.atomic = 1;
      RC(launch_gemm(g, st));
    }
    RC(launch_unpack_conv_grad(p.tmp_w2p, c->G(c->conv2_w), d, d, 3 * d, st));
    RC(launch_colsum_accum(p.gln, d, Me, d, c->G(c->conv2_b), st));
    RC(r.dgrad(p.gln, Me, d, c->template w2p<T>(), 3 * d, nullptr, nullptr, p.gA2));
    RC(launch_conv2_col2im_dgelu(p.gA2, p.u1, p.gu, B, c->T1, d, st));  // gu = d(u1) [B*3000, d]
    OASR_CHECK_HIP(hipMemsetAsync(p.tmp_w1p, 0, (size_t)d * 256 * 4, st));
    // conv1 weight gradient, same trick: windows of 3*n_mels (+ junk up to 256, whose gradient columns are never
    // unpacked) at stride n_mels from mel_tm - n_mels; the first tap of every (b, 0) and the last tap of every (b, T1-1)
    // see the neighbouring sample (or a zeroed guard row) instead of the zero padding -> two rank-B corrections.
    {
      const int nm = c->dims.n_mels;
      RC(r.wgrad(p.gu, d, M1, d, plain_view(p.mel_tm - nm, nm), 256, p.tmp_w1p, 256));  // (guard rows zeroed by the forward)
      for (int side = 0; side < 2; ++side) {
        GemmArgsT<T> g = gemm_defaults_t<T>();
        g.A = plain_view(p.gu + (side ? (long)(c->T1 - 1) * d : 0), (long)c->T1 * d);  // dU rows (b, 0) / (b, T1-1)
        g.ta = 1;
        g.B = plain_view(side ? p.mel_tm + (long)c->T1 * nm : p.mel_tm - nm, (long)c->T1 * nm);
        g.tb = 1;
        g.M = d;
        g.N = nm;
        g.K = B;
        g.alpha = -1.0f;
        g.out_f32 = p.tmp_w1p + (side ? 2 * nm : 0);
        g.ldc32 = 256;
        g.atomic = 1;
        RC(launch_gemm(g, st));
      }
    }
    RC(launch_unpack_conv_grad(p.tmp_w1p, c->G(c->conv1_w), d, c->dims.n_mels, 256, st));
    RC(launch_colsum_accum(p.gu, d, M1, d, c->G(c->conv1_b), st));
  }
  RC(r.record(ev, seg++));
  if (seg != (int)c->segments.size()) {
    oasr_set_error("internal: segment count mismatch %d vs %zu", seg, c->segments.size());
    return OASR_ESTATE;
  }
  return OASR_OK;
}

template <typename T>
static int oasr_train_fwd_bwd_s_impl(oasr_ctx* c, const float* mel, const int64_t* tokens, const int64_t* targets,
                                    const int32_t* text_len, int B, int S, float loss_scale, float inv_accum, float* loss_out,
                                    int accumulate_loss, float* logits_out, void** ev, void* workspace, size_t workspace_bytes,
                                    void* stream, const float* mel_clip_max = nullptr) {
  RC(check_bound(c, true));
  OASR_REQUIRE(S > 0 && S <= c->S_max, "oasr_train_fwd_bwd: S=%d outside (0, n_text_ctx=%d]", S, c->S_max);
  OASR_REQUIRE(mel && tokens && targets && text_len && loss_out && workspace && B > 0, "oasr_train_fwd_bwd: bad args");
  OASR_REQUIRE(workspace_bytes >= oasr_workspace_bytes(c, B, S, OASR_MODE_TRAIN), "oasr_train_fwd_bwd: workspace too small");
  const long Md = (long)B * S;
  Arena A(workspace, workspace_bytes);
  typename Engine<T>::Plan p;
  Engine<T>::make_plan(c, A, p, B, S, true);
  typename Engine<T>::Runner r{c, (hipStream_t)stream, B, S, text_len};
  r.train = true;
  r.cs_scratch = p.gemm_cs_scratch;
  r.mel_clip_max = mel_clip_max;  // un-finalized log-mel (oasr_log_mel_raw): the floor / scale lines ride in the encoder's transpose
  hipStream_t st = r.st;
  // ---------------- forward ----------------
  RC(r.encoder_fwd(p, mel));
  RC(r.decoder_fwd(p, tokens));
  if (logits_out) RC(launch_logits_to_f32(p.logits, c->Vp, Md, c->V, logits_out, st));
  RC(launch_count_valid(targets, Md, PAD_ID, c->V, p.n_valid, st));
  RC(launch_cross_entropy(p.logits, c->Vp, c->V, targets, Md, PAD_ID, loss_scale * inv_accum, p.n_valid, p.row_loss, 1, st));
  RC(launch_loss_reduce(p.row_loss, Md, p.n_valid, inv_accum, loss_out, accumulate_loss, st));
  return train_backward<T>(c, r, p, tokens, B, S, ev);
}

// ---- the supervised-span micro-step ---------------------------------------------------------------------------------------------
// Same forward (all n_text_ctx positions, as the reference pads them: train_timestamps.py:318-329), same loss, same gradients; what
// changes is WHERE the decoder's token rows live and how much of the backward is executed.  span_host[b] (host memory, known to the
// data loader: train_timestamps.py:238-343 builds the token sequences on the host) bounds the positions of sample b that can carry
// gradient: every target at or past it is ignore_index (train_timestamps.py:1444) and it is >= text_len[b], the first masked key
// column (:314-315).  Rows of every decoder-side gradient past the span are exactly zero in the reference's computation -- the loss
// ignores them, no supervised query attends to them -- so:
//   * the decoder's activations are laid out in 64-position CHUNKS, every chunk with a position < span first (kernels.h:
//     AttnArgs.q_rows; only the embedding, the attention kernels and the target gather know about the permutation -- LayerNorm, the
//     GEMMs and their epilogues are row-wise and see plain matrices);
//   * the backward of the decoder (dgrad / wgrad GEMMs, LayerNorm, attention, cross-entropy gradient, embedding scatter) runs on the
//     leading R = sum_b ceil64(span[b]) rows only -- on the synthetic lengths 1/3 of the 448 * B.
// The results differ from oasr_train_fwd_bwd's only by fp32 summation order (weight gradients sum over fewer, re-ordered token rows).
template <typename T>
static int oasr_train_fwd_bwd_span_impl(oasr_ctx* c, const float* mel, const int64_t* tokens, const int64_t* targets, const int32_t* text_len,
                                       const int32_t* span_host, int forward_rows, const float* mel_clip_max, int B, float loss_scale,
                                       float inv_accum, float* loss_out, int accumulate_loss, void** ev, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  const int S = c->S_max;
  const long Md = (long)B * S;
  Arena A(workspace, workspace_bytes);
  typename Engine<T>::Plan p;
  Engine<T>::make_plan(c, A, p, B, S, true);
  typename Engine<T>::Runner r{c, (hipStream_t)stream, B, S, text_len};
  r.train = true;
  r.cs_scratch = p.gemm_cs_scratch;
  hipStream_t st = r.st;
  long R = 0;
  RC(launch_build_span_tables(span_host, B, S, targets, PAD_ID, p.rows, p.span_dev, p.targets_phys, &R, st));
  OASR_REQUIRE(R > 0, "oasr_train_fwd_bwd_span: no position of the micro-batch carries gradient (every span is 0)");
  r.dec_rows = p.rows;
  r.dec_span = p.span_dev;
  r.dec_rows_bwd = R;
  r.dec_rows_fwd = forward_rows == OASR_SPAN_FORWARD_ACTIVE ? R : 0;
  r.mel_clip_max = mel_clip_max;
  if (const int mode = span_side_streams()) RC(r.side_begin(mode));
  // ---------------- forward (every position, unless the caller opted out of the padded ones) ----------------
  RC(r.encoder_fwd(p, mel));
  RC(r.decoder_fwd(p, tokens));
  // loss over the active rows (the other rows' targets are ignore_index: they add nothing to the sum and nothing to the count)
  RC(launch_count_valid(targets, Md, PAD_ID, c->V, p.n_valid, st));
  RC(launch_cross_entropy(p.logits, c->Vp, c->V, p.targets_phys, R, PAD_ID, loss_scale * inv_accum, p.n_valid, p.row_loss, 1, st));
  RC(launch_loss_reduce(p.row_loss, R, p.n_valid, inv_accum, loss_out, accumulate_loss, st));
  return train_backward<T>(c, r, p, tokens, B, S, ev);
}
extern "C" int oasr_train_fwd_bwd_span(oasr_ctx* c, const float* mel, const int64_t* tokens, const int64_t* targets, const int32_t* text_len,
                                       const int32_t* span_host, int forward_rows, const float* mel_clip_max, int B, float loss_scale,
                                       float inv_accum, float* loss_out, int accumulate_loss, void** ev, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  RC(check_bound(c, true));
  OASR_REQUIRE(mel && tokens && targets && text_len && span_host && loss_out && workspace && B > 0, "oasr_train_fwd_bwd_span: bad args");
  OASR_REQUIRE(forward_rows == OASR_SPAN_FORWARD_ALL || forward_rows == OASR_SPAN_FORWARD_ACTIVE, "oasr_train_fwd_bwd_span: forward_rows");
  OASR_REQUIRE(workspace_bytes >= oasr_workspace_bytes(c, B, c->S_max, OASR_MODE_TRAIN), "oasr_train_fwd_bwd_span: workspace too small");
  if ((c->S_max % 64) != 0 || c->S_max > 64 * OASR_ROWTAB || B > 512) {  // no chunking for this shape: the plain step (same results)
    return c->f32 ? oasr_train_fwd_bwd_s_impl<float>(c, mel, tokens, targets, text_len, B, c->S_max, loss_scale, inv_accum, loss_out,
                                                     accumulate_loss, nullptr, ev, workspace, workspace_bytes, stream, mel_clip_max)
                  : oasr_train_fwd_bwd_s_impl<bf16_t>(c, mel, tokens, targets, text_len, B, c->S_max, loss_scale, inv_accum, loss_out,
                                                      accumulate_loss, nullptr, ev, workspace, workspace_bytes, stream, mel_clip_max);
  }
  return c->f32 ? oasr_train_fwd_bwd_span_impl<float>(c, mel, tokens, targets, text_len, span_host, forward_rows, mel_clip_max, B, loss_scale,
                                                      inv_accum, loss_out, accumulate_loss, ev, workspace, workspace_bytes, stream)
                : oasr_train_fwd_bwd_span_impl<bf16_t>(c, mel, tokens, targets, text_len, span_host, forward_rows, mel_clip_max, B, loss_scale,
                                                       inv_accum, loss_out, accumulate_loss, ev, workspace, workspace_bytes, stream);
}

// ---- the same micro-step cut at the logits, for torch.autograd (OLMoASR.forward in training mode, olmoasr/model.py:856-887 followed
// by the caller's own loss, train_timestamps.py:1440-1454): oasr_train_fwd returns fp32 logits [B, S, rows] and leaves every saved
// activation in the workspace; oasr_train_bwd takes d(loss)/d(logits) (fp32, same shape; rounded to the engine's activation type
// exactly where autocast's backward would round it) and accumulates the parameter gradients into the bound arena.  The workspace
// must not be used for anything else in between; B, S, tokens and text_len must be the forward's.
template <typename T>
static int oasr_train_fwd_impl(oasr_ctx* c, const float* mel, const int64_t* tokens, const int32_t* text_len, int B, int S,
                               float* logits_out, void* workspace, size_t workspace_bytes, void* stream) {
  Arena A(workspace, workspace_bytes);
  typename Engine<T>::Plan p;
  Engine<T>::make_plan(c, A, p, B, S, true);
  typename Engine<T>::Runner r{c, (hipStream_t)stream, B, S, text_len};
  r.train = true;
  r.cs_scratch = p.gemm_cs_scratch;
  RC(r.encoder_fwd(p, mel));
  RC(r.decoder_fwd(p, tokens));
  return launch_logits_to_f32(p.logits, c->Vp, (long)B * S, c->V, logits_out, r.st);
}
template <typename T>
static int oasr_train_bwd_impl(oasr_ctx* c, const int64_t* tokens, const int32_t* text_len, const float* dlogits, int B, int S, void** ev,
                               void* workspace, size_t workspace_bytes, void* stream) {
  Arena A(workspace, workspace_bytes);
  typename Engine<T>::Plan p;
  Engine<T>::make_plan(c, A, p, B, S, true);
  typename Engine<T>::Runner r{c, (hipStream_t)stream, B, S, text_len};
  r.train = true;
  r.cs_scratch = p.gemm_cs_scratch;
  RC(launch_dlogits_from_f32(dlogits, c->V, (long)B * S, c->Vp, p.logits, r.st));
  return train_backward<T>(c, r, p, tokens, B, S, ev);
}
extern "C" int oasr_train_fwd(oasr_ctx* c, const float* mel, const int64_t* tokens, const int32_t* text_len, int B, int S, float* logits_out,
                              void* workspace, size_t workspace_bytes, void* stream) {
  RC(check_bound(c, true));
  OASR_REQUIRE(S > 0 && S <= c->S_max, "oasr_train_fwd: S=%d outside (0, n_text_ctx=%d]", S, c->S_max);
  OASR_REQUIRE(mel && tokens && text_len && logits_out && workspace && B > 0, "oasr_train_fwd: bad args");
  OASR_REQUIRE(workspace_bytes >= oasr_workspace_bytes(c, B, S, OASR_MODE_TRAIN), "oasr_train_fwd: workspace too small");
  return c->f32 ? oasr_train_fwd_impl<float>(c, mel, tokens, text_len, B, S, logits_out, workspace, workspace_bytes, stream)
                : oasr_train_fwd_impl<bf16_t>(c, mel, tokens, text_len, B, S, logits_out, workspace, workspace_bytes, stream);
}
extern "C" int oasr_train_bwd(oasr_ctx* c, const int64_t* tokens, const int32_t* text_len, const float* dlogits, int B, int S, void** ev,
                              void* workspace, size_t workspace_bytes, void* stream) {
  RC(check_bound(c, true));
  OASR_REQUIRE(S > 0 && S <= c->S_max, "oasr_train_bwd: S=%d outside (0, n_text_ctx=%d]", S, c->S_max);
  OASR_REQUIRE(tokens && text_len && dlogits && workspace && B > 0, "oasr_train_bwd: bad args");
  OASR_REQUIRE(workspace_bytes >= oasr_workspace_bytes(c, B, S, OASR_MODE_TRAIN), "oasr_train_bwd: workspace too small");
  return c->f32 ? oasr_train_bwd_impl<float>(c, tokens, text_len, dlogits, B, S, ev, workspace, workspace_bytes, stream)
                : oasr_train_bwd_impl<bf16_t>(c, tokens, text_len, dlogits, B, S, ev, workspace, workspace_bytes, stream);
}
extern "C" int oasr_train_fwd_bwd_s(oasr_ctx* c, const float* mel, const int64_t* tokens, const int64_t* targets,
                                    const int32_t* text_len, int B, int S, float loss_scale, float inv_accum, float* loss_out,
                                    int accumulate_loss, float* logits_out, void** ev, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  OASR_REQUIRE(c, "oasr_train_fwd_bwd_s: null context");
  return c->f32 ? oasr_train_fwd_bwd_s_impl<float>(c, mel, tokens, targets, text_len, B, S, loss_scale, inv_accum, loss_out, accumulate_loss, logits_out, ev, workspace, workspace_bytes, stream) : oasr_train_fwd_bwd_s_impl<bf16_t>(c, mel, tokens, targets, text_len, B, S, loss_scale, inv_accum, loss_out, accumulate_loss, logits_out, ev, workspace, workspace_bytes, stream);
}

extern "C" int oasr_optim_step(oasr_ctx* c, float inv_loss_scale, float max_grad_norm, float lr, float beta1, float beta2, float eps,
                               float weight_decay, int64_t step, float* stats_out, void* scratch, void* stream) {
  RC(check_bound(c, true));
  OASR_REQUIRE(c->m && c->v && stats_out && scratch && step >= 1, "oasr_optim_step: bad args");
  hipStream_t st = (hipStream_t)stream;
  RC(launch_grad_stats(c->grads, c->numel, (double*)scratch, stats_out, st));
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  const float bc2 = (float)(1.0 - pow((double)beta2, (double)step));
  RC(launch_adamw(c->params, c->grads, c->m, c->v, c->f32 ? nullptr : (bf16_t*)(c->shadow + c->sh_flat), c->numel, stats_out, inv_loss_scale, max_grad_norm,
                  lr, beta1, beta2, eps, weight_decay, bc1, bc2, st));
  return refresh_packed(c, st);
}

// ---- ZeRO-1 building blocks: the optimizer over a contiguous RANGE of the flat arenas --------------------------------------
// The reference's sharded variant is FSDP (scripts/training/train_fsdp_timestamps.py:2665-2719).  Here optimizer-state
// sharding follows from the flat arena: rank r owns one contiguous range, keeps exp_avg / exp_avg_sq for that range only,
// and the step is  reduce-scatter(grads) -> partial sum of squares per rank -> all-reduce of 2 floats -> AdamW on the owned
// range -> all-gather(params)  (olmoasr_amd/zero.py).  Two entry points: the gradient statistics of a range and the step of a
// range given GLOBAL statistics.
extern "C" int oasr_grad_sumsq_range(oasr_ctx* c, int64_t off, int64_t numel, float* stats_out, void* scratch, void* stream) {
  RC(check_bound(c, true));
  OASR_REQUIRE(stats_out && scratch && off >= 0 && numel > 0 && off + numel <= c->numel && (off % 4) == 0 && (numel % 4) == 0,
               "oasr_grad_sumsq_range: bad range [%lld, +%lld) (multiples of 4 inside the arena)", (long long)off, (long long)numel);
  return launch_grad_stats(c->grads + off, numel, (double*)scratch, stats_out, (hipStream_t)stream);
}

// stats: device f32[2] = [sum of squares of the scaled gradients over the WHOLE arena, non-finite flag] (already reduced
// over ranks); m_shard / v_shard: this range's exp_avg / exp_avg_sq (numel floats each, shard-local buffers).
extern "C" int oasr_optim_step_range(oasr_ctx* c, int64_t off, int64_t numel, float* m_shard, float* v_shard, const float* stats,
                                     float inv_loss_scale, float max_grad_norm, float lr, float beta1, float beta2, float eps,
                                     float weight_decay, int64_t step, void* stream) {
  RC(check_bound(c, true));
  OASR_REQUIRE(m_shard && v_shard && stats && step >= 1 && off >= 0 && numel > 0 && off + numel <= c->numel && (off % 4) == 0 &&
                   (numel % 4) == 0,
               "oasr_optim_step_range: bad args");
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  const float bc2 = (float)(1.0 - pow((double)beta2, (double)step));
  return launch_adamw(c->params + off, c->grads + off, m_shard, v_shard, c->f32 ? nullptr : (bf16_t*)(c->shadow + c->sh_flat) + off, numel,
                      stats, inv_loss_scale, max_grad_norm, lr, beta1, beta2, eps, weight_decay, bc1, bc2, (hipStream_t)stream);
}

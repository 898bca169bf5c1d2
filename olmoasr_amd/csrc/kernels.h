// Internal launch interface between the kernel translation units and the engine (not part of the C ABI).
#pragma once
#include "common.h"

// A row-major bf16 operand, optionally viewed as overlapping convolution windows:
//   plain   : element (r, k) at  r*ld + k
//   windows : rpb > 0.  r = b*rpb + t ; element (r,k) at  b*bstride + t*ld - lead + k, and it is an implicit
//             zero when k >= kvalid, or (t == 0 && k < lead), or (t == rpb-1 && k >= trail_from).
//             conv1 (k3,s1,p1) over time-major mel [B][3000][80]  : ld=80,  rpb=3000, lead=80, kvalid=240, trail_from=160
//             conv2 (k3,s2,p1) over time-major h1 [B][3000][d]    : ld=2d,  rpb=1500, lead=d,  kvalid=3d,  trail_from=3d
// T = bf16_t: the production kernels.  T = float: the fp32 VALIDATION kernels (fp32ref.hip) that run the same engine
// schedule with fp32 activations, fp32 operands and fp32 accumulation -- the reference's precision="float32" path
// (scripts/training/train_timestamps.py:2128,2220-2224), used to hold the engine to 1e-3 against the fp32 CPU oracle.
template <typename T>
struct OperandViewT {
  const T* ptr;
  long ld;
  int rpb;
  long bstride;
  int lead;
  int kvalid;
  int trail_from;
};
typedef OperandViewT<bf16_t> OperandView;
typedef OperandViewT<float> OperandViewF;
template <typename T>
static inline OperandViewT<T> plain_view(const T* p, long ld) {
  return OperandViewT<T>{p, ld, 0, 0, 0, 0, 0};
}

// C[M,N] = epilogue( alpha * sum_k A(m,k) * B(n,k) )
//   ta == 0 : A stored [M][K] (k contiguous);  ta == 1 : A stored [K][M] (m contiguous)
//   tb == 0 : B stored [N][K];                 tb == 1 : B stored [K][N]
// Epilogue, in this order, every pointer optional (fp32 math on the accumulator):
//   v = alpha*acc (+ bias[n]) ; out_pre <- bf16(v) ; if act: v = gelu(bf16(v)) ;
//   (act == 2: as act == 1, but out_pre <- bf16(gelu'(bf16(v))) -- the training forward of mlp.0 saves the derivative the
//    backward needs in the slot of the pre-activation, same bytes, so the dgrad epilogue is one multiply instead of ~25 VALU)
//   if pos: v = bf16(v) + pos[(m % pos_period)*N + n] ; if dgelu_u: v = bf16(v) * (dgelu_deriv ? u[m,n] : gelu'(u[m,n])) ;
//   if resid: v = bf16(v) + resid[m,n] ; out <- bf16(v) ; out_f32 <- (atomic ? += v : beta*out_f32 + v)
template <typename T>
struct GemmArgsT {
  OperandViewT<T> A, B;
  int M, N, K;
  int ta, tb;
  float alpha;
  const float* bias;
  int act;  // 0 none, 1 exact-erf GELU, 2 exact-erf GELU with out_pre = GELU'(pre-activation)
  const float* pos;
  int pos_period;
  const T* dgelu_u;
  long ldu;
  const T* resid;
  long ldr;
  T* out;
  T* out_pre;
  long ldc;
  float* out_f32;
  long ldc32;
  float beta;
  float* colsum;  // optional [N]: += column sums of the bf16 values stored to `out` (fused bias gradient)
  int atomic;   // out_f32 += v with hardware fp32 atomics (split-K safe)
  int split_k;  // >= 1; > 1 requires atomic out_f32 and no other output
  int raster_gm;  // fast path: tile rows per L2 group (0 = choose from residency)
  int dgelu_deriv;  // dgelu_u already holds GELU'(u) (written by an act == 2 forward)
  int stagger, stagger_phases;  // ping-pong kernel: first-wave phase stagger in units of s_sleep(127) (0 = off)
  int atomic_on_pp;             // split-K / atomic output on the 256x256 ping-pong kernel instead of the 256x128 one
  int vgrid;                    // set by the launcher: blocks of the virtual grid a persistent ping-pong launch walks
  int epi_flags;                // epilogue memory policy: bit 0 non-temporal output stores, bit 1 non-temporal side-input loads
  float* colsum_scratch;        // optional, with colsum on the direct-to-LDS kernels: fp32 [2 * ceil(M/256)][N] partial rows (one per
                                // 128-row wave block) that the launcher reduces into colsum -- instead of fp32 atomics from every wave
};
typedef GemmArgsT<bf16_t> GemmArgs;
typedef GemmArgsT<float> GemmArgsF;
int launch_gemm(const GemmArgs& a, hipStream_t stream);
int launch_gemm(const GemmArgsF& a, hipStream_t stream);  // fp32 validation kernel: any shape / view, split_k ignored
// bench-only: time every GEMM launch with HIP events on its stream; collect() sums per variant (2*ta+tb)
void gemm_profile_enable(int on);
int gemm_profile_lane(int lane);  // tag the following launches' profile records (0 = main stream, 1 = side stream); returns the old tag
void gemm_force_general(int on);  // tests: disable the direct-to-LDS fast path
void gemm_set_variant(int dma_in_mma);  // experiments: ping-pong kernel issues its DMA pieces between the MFMAs
void gemm_set_stagger(int sleeps, int phases);  // experiments: first-wave phase stagger of the ping-pong kernel
int gemm_profile_collect(double ms[4], double flops[4], long count[4], char* by_symbol, int cap);
template <typename T>
static inline GemmArgsT<T> gemm_defaults_t() {
  GemmArgsT<T> g;
  memset(&g, 0, sizeof(g));
  g.alpha = 1.0f;
  g.split_k = 1;
  return g;
}
static inline GemmArgs gemm_defaults() { return gemm_defaults_t<bf16_t>(); }

// LayerNorm over the last dim (eps 1e-5, fp32 internals, bf16 in/out; reference model.py:39)
int launch_layernorm_fwd(const bf16_t* x, const float* gamma, const float* beta, bf16_t* y, float* mean, float* rstd,
                         long rows, int d, hipStream_t s);
// dx = LN'(dy) (+ dres) ; dgamma/dbeta are ACCUMULATED (atomic fp32) into the grad arena; dsum (optional) += column
// sums of the produced dx (= the bias gradient of the Linear whose output this residual-stream gradient belongs to)
int launch_layernorm_bwd(const bf16_t* dy, const bf16_t* x, const float* gamma, const float* mean, const float* rstd,
                         const bf16_t* dres, bf16_t* dx, float* dgamma, float* dbeta, float* dsum, long rows, int d, hipStream_t s);
int launch_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, long rows, int d,
                         hipStream_t s);
int launch_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, const float* dres,
                         float* dx, float* dgamma, float* dbeta, float* dsum, long rows, int d, hipStream_t s);

// Flash attention over head_dim 64.  Q/K/V are strided views [B, T, H, 64] (row strides in elements);
// key j visible to query i iff j < kv_len[b] (nullptr -> Tk) and (!causal || j <= i).  scale = 1/8.
template <typename T>
struct AttnArgsT {
  const T *q, *k, *v;
  long ldq, ldk, ldv;        // token strides
  long bsq, bsk, bsv;        // batch strides
  T* o;                      // [B, Tq, H*64]
  long ldo, bso;
  float* lse;                // [B, H, Tq] natural-log-sum-exp of scaled scores
  T* o_lo;                   // optional bf16 rounding residual of o (same strides): fwd writes it, bwd uses o + o_lo for delta = rowsum(dO*O)
  const int32_t* kv_len;     // [B] or null
  int B, H, Tq, Tk, causal;
  // backward only
  const T* d_o;              // [B, Tq, H*64], same strides as o
  float* delta;              // [B, H, Tq] workspace: rowsum(dO * O)
  T *dq, *dk, *dv;           // same strides as q/k/v
  // optional [H*64] fp32, ACCUMULATED: column sums over all (b, t) rows of the stored dq / dv = the gradients of the
  // query / value projection biases (key has none, olmoasr/model.py:259), fused into the backward kernels' store epilogues
  float *dq_colsum, *dv_colsum;
  float* colsum_scratch;     // required with either: fp32 [(B*ceil(Tq/128) + B*ceil(Tk/128)) * H*64] per-workgroup partial rows
  // optional int32 [B, H, ceil(Tq/64)] workspace: the dQ kernel records which 64-query tiles of d_o hold any non-zero value, the
  // dK/dV kernel stops at the last such tile (decoder side of a padded batch: the loss ignores the padded positions, their d_o rows
  // are exactly zero and contribute exactly nothing -- three quarters of the 448 positions on average).  Results are bit-identical.
  int32_t* qtile_flags;
  // ---- chunked token rows (the decoder side of a span-limited training step, engine.hip "supervised span") -----------------
  // q_rows / k_rows: optional int32 [B][OASR_ROWTAB]: entry c = first token row (relative to the tensor base pointers, batch
  // strides unused) of the 64-position chunk c of sample b; q_rows addresses q / o / o_lo / d_o / dq, k_rows k / v / dk / dv.
  // lse / delta / qtile_flags keep their logical [B, H, Tq] indexing.  Needs Tq (Tk) % 64 == 0 and <= 64 * OASR_ROWTAB.
  const int32_t *q_rows, *k_rows;
  // q_span: optional int32 [B], multiples of 64 (backward only): d_o of sample b is zero at every query position >= q_span[b];
  // those rows of d_o / o are not read, dq (and, for self-attention, dk / dv: their keys see no supervised query) not written.
  const int32_t* q_span;
};
#ifndef OASR_ROWTAB
#define OASR_ROWTAB 16  // (also defined by include/oasr.h)
#endif
static inline size_t attn_colsum_scratch_floats(int B, int H, int Tq, int Tk) {
  return (size_t)B * ((size_t)(Tq + 127) / 128 + (size_t)(Tk + 127) / 128) * H * 64;
}
typedef AttnArgsT<bf16_t> AttnArgs;
typedef AttnArgsT<float> AttnArgsF;
int launch_attention_fwd(const AttnArgs& a, hipStream_t s);
int launch_attention_bwd(const AttnArgs& a, hipStream_t s);
// qk of MultiHeadAttention.qkv_attention (model.py:347-442): fp32 pre-softmax scaled scores [B, H, Tq, Tk], masked entries -inf (scores.hip)
int launch_attention_scores(const AttnArgs& a, float* out, hipStream_t s);
void attention_set_pingpong(int on);  // testing hook: 0 = general kernels for the unmasked case too
int launch_attention_fwd(const AttnArgsF& a, hipStream_t s);  // fp32 validation kernels (o_lo unused: O is fp32)
int launch_attention_bwd(const AttnArgsF& a, hipStream_t s);
int launch_attention_scores(const AttnArgsF& a, float* out, hipStream_t s);

// ---- LayerNorm-folded decode projection (decode_proj.hip) ------------------------------------------------
int launch_decode_proj(const bf16_t* x, int M, int K, const bf16_t* W, int N, const float* ln_g, const float* ln_b, const float* bias,
                       int gelu, const bf16_t* resid, long ldr, bf16_t* out, long ldc, float* out_f32, long ldf, hipStream_t s);

// ---- one-launch decoder step for <= 4 sequences (decode_xcd.hip) -----------------------------------------------
struct DecodeXcdArgs {
  const bf16_t* wflat;  // bf16 shadow of the flat parameter arena
  const float* params;  // fp32 parameter arena (LayerNorm parameters, biases)
  const float* aux;     // fused [q_bias | 0 | v_bias] rows
  bf16_t* cache;        // KV cache (per layer: self q|k|v [M, S_max, 3d], cross k|v [M, Te, 2d])
  long cache_lstride;   // elements per layer
  bf16_t *x, *x2, *x3, *q, *o, *hg;  // activation rows: x holds the embedded token on entry and the last block's output on return
  float* part;          // decode_xcd_part_floats() floats
  unsigned* ctrl;       // 4 words, zeroed once per cache (oasr_decode_begin): barrier counter, error flag, epoch base, XCC mask
  int d, H, Te, S_max, L, M, pos;
  int flags;            // experiments (decode_xcd.hip::XArgs::flags), 0 in production
  void* stamps;         // optional 512-byte device buffer for in-kernel cycle stamps (measurement), or null
  int team, stride;     // `team` workgroups; stride 8 = one per CU of ONE XCD (grid 8 x team, blockIdx % 8 == 0), 1 = spread over the chip
  // chip-wide engine only: the final LayerNorm + tied logits projection as the launch's last phase (null logits_out: not done there)
  const bf16_t* w_logits = nullptr;  // bf16 token embedding [V][d]
  const float *lnf_g = nullptr, *lnf_b = nullptr;
  float* logits_out = nullptr;       // f32 [V]
  int V = 0;
  const int64_t* layer_offsets;   // HOST: [18] element offsets of decoder layer 0 in decode_xcd.hip::XLayer order
  long lstride, astride;          // layer l = layer 0 + l * lstride (arena / shadow elements), + l * astride for the aux entry
};
int launch_decode_xcd(const DecodeXcdArgs& a, hipStream_t s);
bool decode_xcd_supports(int d, int H, int Te, int S_max, int L, int M);
// every 32-bit buffer offset of the launch stays below 2 GiB (else: the multi-launch step)
bool decode_xcd_offsets_ok(const int64_t* layer0, long lstride, long cache_lstride, int d, int Te, int L, int M);
size_t decode_xcd_part_floats(int M, int H, int Te);
// chip-wide one-launch step for ONE sequence (decode_wide.hip): `team` = workgroups (one per CU), ctrl = the cache's control tail
// (OASR_KV_TAIL_BYTES: words [1] error flag, [3] XCC mask, [4] epoch base; the per-workgroup flag array 1 KB in)
int launch_decode_wide(const DecodeXcdArgs& a, hipStream_t s);
bool decode_wide_supports(int d, int H, int Te, int S_max, int L, int M, int nwg);
size_t decode_wide_part_floats(int H);

// ---- elementwise / reductions -------------------------------------------------------------------------
int launch_cast_f32_bf16(const float* src, bf16_t* dst, long n, hipStream_t s);
// conv weight [co][ci][3] f32 -> bf16 [co][ldk] with k = kk*ci_n + ci, zero padded to ldk
int launch_pack_conv_weight(const float* w, bf16_t* dst, int co, int ci, int ldk, hipStream_t s);
// grad [co][ldk] f32 (k = kk*ci_n+ci) accumulated into dw [co][ci][3]
int launch_unpack_conv_grad(const float* g, float* dw, int co, int ci, int ldk, hipStream_t s);
// token embedding rows f32 [rows][d] -> bf16 [rows_pad][d], rows >= rows zero
int launch_pack_embedding(const float* e, bf16_t* dst, int rows, int rows_pad, int d, hipStream_t s);
// mel f32 [B][80][T] -> bf16 time-major [B][T][80]
// clip_max (optional [B]): mel is the un-finalized log10 mel power of oasr_log_mel_raw; max(x, clip_max[b] - 8), (x + 4) / 4 on the fly
int launch_mel_to_time_major(const float* mel, bf16_t* out, int B, int n_mels, int T, hipStream_t s, const float* clip_max = nullptr);
// x[b,s,:] = bf16(E[tok[b,s]] + pos[s]);  rows (optional): chunk-row table [B][OASR_ROWTAB] -- x row of (b, s) =
// rows[b][s >> 6] + (s & 63) instead of b*S + s
int launch_embedding_fwd(const int64_t* tok, const float* E, const float* pos, bf16_t* x, int B, int S, int d, long n_embed,
                         hipStream_t s, const int32_t* rows = nullptr);
// dE[tok] += dx (skipping pad_id), dpos[s] += sum_b dx;  span (optional, with rows): positions s >= span[b] hold no gradient and are skipped
int launch_embedding_bwd(const int64_t* tok, const bf16_t* dx, float* dE, float* dpos, int B, int S, int d, long pad_id, long n_embed,
                         hipStream_t s, const int32_t* rows = nullptr, const int32_t* span = nullptr);
// Supervised-span tables of one decoder micro-batch (engine.hip): span_host[b] (host, <= S) -> on the device
//   rows [B][OASR_ROWTAB]: the chunk-row table -- the active chunks (64*c < span[b]) of all samples first, position-block-major,
//                          then the inactive ones; span_dev [B] = span rounded up to 64; targets_phys [B*S]: targets in row order
// Returns the number of active token rows through *active_rows (host).  S % 64 == 0, S <= 64 * OASR_ROWTAB, B <= 512.
int launch_build_span_tables(const int32_t* span_host, int B, int S, const int64_t* targets, long ignore, int32_t* rows, int32_t* span_dev,
                             int64_t* targets_phys, long* active_rows, hipStream_t s);
// out[n] += sum_m x[m, n]   (x bf16 [M, ld], columns [0, ncols))
int launch_colsum_accum(const bf16_t* x, long ld, long M, int ncols, float* out, hipStream_t s);
// conv2 input-gradient fold + conv1 GELU backward: dpre1[b,t,c] = gelu'(u1) * sum of the dA windows covering t
int launch_conv2_col2im_dgelu(const bf16_t* dA /*[B*T2][3d]*/, const bf16_t* u1 /*[B*T1][d]*/, bf16_t* dpre1, int B, int T1,
                              int d, hipStream_t s);
// dst(f32) += src(f32) over n  (grad of the fp32 sinusoid buffer is not needed; used for misc accumulations)
int launch_axpy_f32(const float* src, float* dst, long n, float a, hipStream_t s);
// fp32 validation overloads (fp32ref.hip): same contracts with fp32 activations
int launch_pack_conv_weight(const float* w, float* dst, int co, int ci, int ldk, hipStream_t s);
int launch_mel_to_time_major(const float* mel, float* out, int B, int n_mels, int T, hipStream_t s, const float* clip_max = nullptr);
int launch_embedding_fwd(const int64_t* tok, const float* E, const float* pos, float* x, int B, int S, int d, long n_embed, hipStream_t s,
                         const int32_t* rows = nullptr);
int launch_embedding_bwd(const int64_t* tok, const float* dx, float* dE, float* dpos, int B, int S, int d, long pad_id, long n_embed,
                         hipStream_t s, const int32_t* rows = nullptr, const int32_t* span = nullptr);
int launch_colsum_accum(const float* x, long ld, long M, int ncols, float* out, hipStream_t s);
int launch_conv2_col2im_dgelu(const float* dA, const float* u1, float* dpre1, int B, int T1, int d, hipStream_t s);
int launch_dgelu_mul(const bf16_t* dy, const bf16_t* u, bf16_t* out, long n, hipStream_t s);
int launch_dgelu_mul(const float* dy, const float* u, float* out, long n, hipStream_t s);
int launch_logits_to_f32(const bf16_t* logits, long ld, long rows, int V, float* out, hipStream_t s);
int launch_logits_to_f32(const float* logits, long ld, long rows, int V, float* out, hipStream_t s);
int launch_dlogits_from_f32(const float* src, int V, long rows, long ld, bf16_t* dst, hipStream_t s);  // fp32 [rows, V] -> padded [rows, ld]
int launch_dlogits_from_f32(const float* src, int V, long rows, long ld, float* dst, hipStream_t s);

// ---- loss -------------------------------------------------------------------------------------------------
// logits bf16 [rows][ld] (first V entries valid).  Writes, in place, dlogits = (softmax - onehot) * gscale / n_valid
// (zeros for ignored rows and pad columns), row_loss[r] = lse - logit[target] (0 for ignored rows).
// n_valid_dev: device int32 (count of targets != ignore).  loss_out += sum(row_loss)/n_valid * loss_mul.
int launch_count_valid(const int64_t* targets, long rows, long ignore, int V, int32_t* n_valid_dev, hipStream_t s);
int launch_cross_entropy(bf16_t* logits, long ld, int V, const int64_t* targets, long rows, long ignore, float gscale,
                         const int32_t* n_valid_dev, float* row_loss, int write_grad, hipStream_t s);
int launch_cross_entropy(float* logits, long ld, int V, const int64_t* targets, long rows, long ignore, float gscale,
                         const int32_t* n_valid_dev, float* row_loss, int write_grad, hipStream_t s);  // fp32 validation
int launch_loss_reduce(const float* row_loss, long rows, const int32_t* n_valid_dev, float mul, float* loss_out, int accumulate,
                       hipStream_t s);

// tok[r] = argmax_c (logits[r][c] + mask[c] + mask2[c]) (lowest index on ties), logprob[r] = log_softmax of that entry (optional)
int launch_pick_tokens(const float* logits, long ld, int V, long rows, const float* mask, const float* mask2, int64_t* tok, float* logprob,
                       hipStream_t s);
// + whisper's ApplyTimestampRules evaluated from the device-resident sampled history (loss.hip::pick_ts_kernel)
int launch_pick_tokens_ts(const float* logits, long ld, int V, long rows, const float* mask, const float* mask2, const int64_t* hist,
                          long hist_ld, int n_hist, int ts_begin, int eot, int no_ts, int max_initial_index, int64_t* tok, float* logprob,
                          hipStream_t s);

// the K best (log_softmax, token) pairs per row over the same filtered distribution (n_hist < 0: masks only, no timestamp rules):
// tok / logprob [rows][K] -- BeamSearchDecoder.update's topk(beam_size + 1)
int launch_topk_tokens_ts(const float* logits, long ld, int V, long rows, const float* mask, const float* mask2, const int64_t* hist,
                          long hist_ld, int n_hist, int ts_begin, int eot, int no_ts, int max_initial_index, int K, int64_t* tok,
                          float* logprob, hipStream_t s);
// one draw per row from softmax(filtered logits / temperature) by inverse CDF on u[row] in [0, 1); logprob at temperature 1
int launch_sample_tokens_ts(const float* logits, long ld, int V, long rows, const float* mask, const float* mask2, const int64_t* hist,
                            long hist_ld, int n_hist, int ts_begin, int eot, int no_ts, int max_initial_index, float temperature,
                            const float* u, int64_t* tok, float* logprob, hipStream_t s);

// ---- optimizer (flat fp32 arenas) ---------------------------------------------------------------------------
// stats[0] = sum g^2 (of the *scaled* grads), stats[1] = found_inf flag (nonzero if any non-finite)
int launch_grad_stats(const float* g, long n, double* partial /*[1024]*/, float* stats /*[2]*/, hipStream_t s);
// Fused unscale + clip + AdamW (decoupled decay) + bf16 shadow emit.  Skips everything when stats[1] != 0.
int launch_adamw(float* p, const float* g, float* m, float* v, bf16_t* shadow, long n, const float* stats, float inv_scale,
                 float max_norm, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2, hipStream_t s);

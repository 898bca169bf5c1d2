// C ABI plumbing: error string, unit-operator wrappers over the internal launchers, and a one-wave probe that
// dumps what ds_read_b64_tr_b16 returns (used by tests to pin the transpose-read lane mapping the GEMM and
// attention kernels rely on).
#include <stdarg.h>

#include "../../include/oasr.h"
#include "kernels.h"

static thread_local char g_err[512] = "";

void oasr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* oasr_last_error(void) { return g_err; }
extern "C" int oasr_version(void) { return OASR_ABI_VERSION; }
extern "C" size_t oasr_sizeof_attn_args(void) { return sizeof(oasr_attn_args); }

static OperandView to_view(const oasr_operand& o) {
  return OperandView{(const bf16_t*)o.ptr, (long)o.ld, o.rpb, (long)o.bstride, o.lead, o.kvalid, o.trail_from};
}

extern "C" int oasr_gemm(const oasr_gemm_args* a, void* stream) {
  OASR_REQUIRE(a, "oasr_gemm: null args");
  GemmArgs g = gemm_defaults();
  g.A = to_view(a->A);
  g.B = to_view(a->B);
  g.M = a->M;
  g.N = a->N;
  g.K = a->K;
  g.ta = a->ta;
  g.tb = a->tb;
  g.alpha = a->alpha;
  g.bias = a->bias;
  g.act = a->act;
  g.pos = a->pos;
  g.pos_period = a->pos_period;
  g.dgelu_u = (const bf16_t*)a->dgelu_u;
  g.ldu = a->ldu;
  g.resid = (const bf16_t*)a->resid;
  g.ldr = a->ldr;
  g.out = (bf16_t*)a->out;
  g.out_pre = (bf16_t*)a->out_pre;
  g.ldc = a->ldc;
  g.out_f32 = a->out_f32;
  g.ldc32 = a->ldc32;
  g.beta = a->beta;
  g.colsum = a->colsum;
  g.atomic = a->atomic;
  g.split_k = a->split_k < 1 ? 1 : a->split_k;
  g.dgelu_deriv = a->dgelu_deriv;
  OASR_REQUIRE(a->act >= 0 && a->act <= 2, "oasr_gemm: act must be 0 (none), 1 (GELU) or 2 (GELU, out_pre = GELU')");
  return launch_gemm(g, (hipStream_t)stream);
}

// The setters below change process-wide kernel-selection state (A/B experiments, forcing a kernel path in a parity test).  They live in
// the same library as the product ABI, so they are inert unless the process opts in: OASR_TESTING_HOOKS=1 in the environment
// (tests/conftest.py and the scripts/ that use them set it); without it they fail and change nothing.
static int hooks_enabled(const char* what) {
  const char* e = getenv("OASR_TESTING_HOOKS");
  if (e && e[0] == '1') return OASR_OK;
  oasr_set_error("%s: testing hook called without OASR_TESTING_HOOKS=1 (include/oasr_testing.h)", what);
  return OASR_ESTATE;
}
#define OASR_HOOK_GATE(name)          \
  do {                                \
    const int g_ = hooks_enabled(name); \
    if (g_) return g_;                \
  } while (0)

extern "C" int oasr_profile_gemm(int enable) {
  gemm_profile_enable(enable);
  return OASR_OK;
}
extern "C" int oasr_gemm_set_variant(int dma_in_mma) {
  OASR_HOOK_GATE("oasr_gemm_set_variant");
  gemm_set_variant(dma_in_mma);
  return OASR_OK;
}
extern "C" int oasr_gemm_set_stagger(int sleeps, int phases) {
  OASR_HOOK_GATE("oasr_gemm_set_stagger");
  gemm_set_stagger(sleeps, phases);
  return OASR_OK;
}
extern "C" int oasr_attention_set_pingpong(int on) {
  OASR_HOOK_GATE("oasr_attention_set_pingpong");
  attention_set_pingpong(on);
  return OASR_OK;
}
extern "C" int oasr_gemm_force_general(int on) {
  OASR_HOOK_GATE("oasr_gemm_force_general");
  gemm_force_general(on);
  return OASR_OK;
}
extern "C" int oasr_profile_gemm_collect(double* ms4, double* flops4, int64_t* count4, char* by_symbol, int cap) {
  OASR_REQUIRE(ms4 && flops4 && count4, "profile_collect: null");
  long c[4];
  int rc = gemm_profile_collect(ms4, flops4, c, by_symbol, cap);
  for (int i = 0; i < 4; ++i) count4[i] = c[i];
  return rc;
}

extern "C" int oasr_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int64_t rows,
                                  int d, void* stream) {
  return launch_layernorm_fwd((const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, rows, d, (hipStream_t)stream);
}
extern "C" int oasr_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                  const void* dres, void* dx, float* dgamma, float* dbeta, int64_t rows, int d, void* stream) {
  return launch_layernorm_bwd((const bf16_t*)dy, (const bf16_t*)x, gamma, mean, rstd, (const bf16_t*)dres, (bf16_t*)dx, dgamma, dbeta,
                              nullptr, rows, d, (hipStream_t)stream);
}

static AttnArgs to_attn(const oasr_attn_args* a) {
  AttnArgs r;
  memset(&r, 0, sizeof(r));
  r.q = (const bf16_t*)a->q;
  r.k = (const bf16_t*)a->k;
  r.v = (const bf16_t*)a->v;
  r.ldq = a->ldq;
  r.ldk = a->ldk;
  r.ldv = a->ldv;
  r.bsq = a->bsq;
  r.bsk = a->bsk;
  r.bsv = a->bsv;
  r.o = (bf16_t*)a->o;
  r.ldo = a->ldo;
  r.bso = a->bso;
  r.lse = a->lse;
  r.o_lo = (bf16_t*)a->o_lo;
  r.kv_len = a->kv_len;
  r.B = a->B;
  r.H = a->H;
  r.Tq = a->Tq;
  r.Tk = a->Tk;
  r.causal = a->causal;
  r.d_o = (const bf16_t*)a->d_o;
  r.delta = a->delta;
  r.dq = (bf16_t*)a->dq;
  r.dk = (bf16_t*)a->dk;
  r.dv = (bf16_t*)a->dv;
  r.dq_colsum = a->dq_colsum;
  r.dv_colsum = a->dv_colsum;
  r.colsum_scratch = a->colsum_scratch;
  r.qtile_flags = a->qtile_flags;
  r.q_rows = a->q_rows;
  r.k_rows = a->k_rows;
  r.q_span = a->q_span;
  return r;
}
extern "C" int oasr_attention_fwd(const oasr_attn_args* a, void* stream) {
  OASR_REQUIRE(a, "oasr_attention_fwd: null args");
  return launch_attention_fwd(to_attn(a), (hipStream_t)stream);
}
extern "C" int oasr_attention_bwd(const oasr_attn_args* a, void* stream) {
  OASR_REQUIRE(a, "oasr_attention_bwd: null args");
  return launch_attention_bwd(to_attn(a), (hipStream_t)stream);
}
extern "C" int oasr_attention_scores(const oasr_attn_args* a, int dtype, float* scores, void* stream) {
  OASR_REQUIRE(a && scores, "oasr_attention_scores: null args");
  OASR_REQUIRE(dtype == OASR_DTYPE_BF16 || dtype == OASR_DTYPE_F32, "oasr_attention_scores: dtype %d (0 = bf16 operands, 1 = fp32 operands)", dtype);
  if (dtype == OASR_DTYPE_BF16) return launch_attention_scores(to_attn(a), scores, (hipStream_t)stream);
  AttnArgsF f;
  memset(&f, 0, sizeof(f));
  f.q = (const float*)a->q, f.k = (const float*)a->k;
  f.ldq = a->ldq, f.ldk = a->ldk, f.bsq = a->bsq, f.bsk = a->bsk;
  f.kv_len = a->kv_len;
  f.B = a->B, f.H = a->H, f.Tq = a->Tq, f.Tk = a->Tk, f.causal = a->causal;
  f.q_rows = a->q_rows, f.k_rows = a->k_rows;
  return launch_attention_scores(f, scores, (hipStream_t)stream);
}

extern "C" int oasr_test_span_tables(const int32_t* span_host, int B, int S, const int64_t* targets, int32_t* rows_out, int32_t* span_out,
                                     int64_t* targets_rows_out, int64_t* active_rows_out, void* stream) {
  OASR_REQUIRE(active_rows_out, "oasr_test_span_tables: null");
  long act = 0;
  const int rc = launch_build_span_tables(span_host, B, S, targets, 51864, rows_out, span_out, targets_rows_out, &act, (hipStream_t)stream);
  *active_rows_out = act;
  return rc;
}

extern "C" int oasr_cross_entropy(void* logits, int64_t ld, int V, const int64_t* targets, int64_t rows, int64_t ignore, float gscale,
                                  int32_t* n_valid_dev, float* row_loss, float* loss_out, int write_grad, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  int rc = launch_count_valid(targets, rows, ignore, V, n_valid_dev, st);
  if (rc) return rc;
  rc = launch_cross_entropy((bf16_t*)logits, ld, V, targets, rows, ignore, gscale, n_valid_dev, row_loss, write_grad, st);
  if (rc) return rc;
  if (loss_out) rc = launch_loss_reduce(row_loss, rows, n_valid_dev, 1.0f, loss_out, 0, st);
  return rc;
}

extern "C" int oasr_pick_tokens(const float* logits, int64_t ld, int V, int64_t rows, const float* mask, const float* mask2, int64_t* tok,
                                float* logprob, void* stream) {
  return launch_pick_tokens(logits, ld, V, rows, mask, mask2, tok, logprob, (hipStream_t)stream);
}

extern "C" int oasr_pick_tokens_ts(const float* logits, int64_t ld, int V, int64_t rows, const float* mask, const float* mask2,
                                   const int64_t* history, int64_t history_ld, int n_history, int timestamp_begin, int eot, int no_timestamps,
                                   int max_initial_index, int64_t* tok, float* logprob, void* stream) {
  return launch_pick_tokens_ts(logits, ld, V, rows, mask, mask2, history, history_ld, n_history, timestamp_begin, eot, no_timestamps,
                               max_initial_index, tok, logprob, (hipStream_t)stream);
}

extern "C" int oasr_topk_tokens(const float* logits, int64_t ld, int V, int64_t rows, const float* mask, const float* mask2,
                                const int64_t* history, int64_t history_ld, int n_history, int timestamp_begin, int eot, int no_timestamps,
                                int max_initial_index, int K, int64_t* tok, float* logprob, void* stream) {
  return launch_topk_tokens_ts(logits, ld, V, rows, mask, mask2, history, history_ld, n_history, timestamp_begin, eot, no_timestamps,
                               max_initial_index, K, tok, logprob, (hipStream_t)stream);
}
extern "C" int oasr_sample_tokens(const float* logits, int64_t ld, int V, int64_t rows, const float* mask, const float* mask2,
                                  const int64_t* history, int64_t history_ld, int n_history, int timestamp_begin, int eot, int no_timestamps,
                                  int max_initial_index, float temperature, const float* uniforms, int64_t* tok, float* logprob, void* stream) {
  return launch_sample_tokens_ts(logits, ld, V, rows, mask, mask2, history, history_ld, n_history, timestamp_begin, eot, no_timestamps,
                                 max_initial_index, temperature, uniforms, tok, logprob, (hipStream_t)stream);
}

extern "C" int oasr_cast_f32_bf16(const float* src, void* dst, int64_t n, void* stream) {
  return launch_cast_f32_bf16(src, (bf16_t*)dst, n, (hipStream_t)stream);
}

// One wave: LDS image = src [16 rows][64 cols] bf16 (128-byte rows); lane l reads 8 bytes at
// row (l >> 4)*4 + ((l & 15) >> 2), col ((l & 15) & 3) * 4 through ds_read_b64_tr_b16 and stores its 4 results.
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr_t;
__global__ void probe_tr16_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[16 * 64];
  const int l = threadIdx.x;
  for (int i = l; i < 16 * 64; i += 64) tile[i] = src[i];
  __syncthreads();
  const int row = (l >> 4) * 4 + ((l & 15) >> 2), col = ((l & 15) & 3) * 4;
  const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr_t)(tile + row * 64 + col));
  for (int j = 0; j < 4; ++j) dst[l * 4 + j] = (bf16_t)v[j];
}
// One wave: LDS pre-filled with 0xAAAA; lanes < 32 issue an in-range 16-byte buffer_load..lds, lanes >= 32 an
// out-of-range one (offset beyond num_records).  dst[64*8] u16 shows what the hardware writes for OOB lanes.
typedef __attribute__((address_space(3))) void* lds_void_ptr_t;
__global__ void probe_lds_oob_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[64 * 8];
  const int l = threadIdx.x;
  for (int i = l; i < 64 * 8; i += 64) tile[i] = 0xAAAA;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(src), 0, 64 * 16, 0x00020000);
  const unsigned off = l < 32 ? l * 16 : 0x80000000u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_ptr_t)tile, 16, off, 0, 0, 0);
  __syncthreads();
  for (int i = l; i < 64 * 8; i += 64) dst[i] = tile[i];
}
extern "C" int oasr_probe_lds_oob(const void* src, void* dst, void* stream) {
  hipLaunchKernelGGL(probe_lds_oob_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const bf16_t*)src, (bf16_t*)dst);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}
extern "C" int oasr_probe_tr16(const void* src, void* dst, void* stream) {
  hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const bf16_t*)src, (bf16_t*)dst);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}

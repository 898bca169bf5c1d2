/* liboasr -- C ABI of the MI355X-native OLMoASR training hot path.
 *
 * The reference (allenai/OLMoASR) is pure Python: it has no FFI.  Its "boundary" for this path is the Python surface
 * of olmoasr/model.py + scripts/training/train_timestamps.py; every entry point below names the reference call site
 * it replaces.  olmoasr_amd/ (Python, ctypes) mirrors that surface on top of this ABI; INTEGRATION.md shows the stub a
 * reference maintainer would add.
 *
 * Conventions: all pointers are DEVICE pointers owned by the caller unless stated; every function is asynchronous on
 * `stream` (a hipStream_t passed as void*), never synchronises, returns 0 on success or a negative OASR_E* code and
 * leaves a message retrievable by oasr_last_error().  One host thread per process (rank) drives a context.
 * bf16 tensors are raw uint16 bit patterns.  No exceptions cross the ABI.
 */
#ifndef OASR_H
#define OASR_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define OASR_OK 0
#define OASR_EINVAL (-1)
#define OASR_EHIP (-2)
#define OASR_ESTATE (-3)
#define OASR_ERETRY (-4) /* oasr_decode_check: the window's steps must be enqueued again (the context changed engines; see there) */

typedef struct oasr_ctx oasr_ctx;

/* olmoasr/config/model_dims.py:4-25 (ModelDimensions) */
typedef struct oasr_dims {
  int n_mels, n_audio_ctx, n_audio_state, n_audio_head, n_audio_layer;
  int n_vocab, n_text_ctx, n_text_state, n_text_head, n_text_layer;
} oasr_dims;

const char* oasr_last_error(void);
/* ABI version: 100 * major + minor.  Structs passed by pointer (oasr_attn_args, oasr_gemm_args) only grow at the end and only with a
 * major bump; olmoasr_amd/_native.py refuses to drive a library whose version differs from OASR_ABI_VERSION. */
#define OASR_ABI_VERSION 211
int oasr_version(void);

/* ---- log-mel front end: whisper.audio.log_mel_spectrogram as called at train_timestamps.py:196,214 and
 *      olmoasr/transcribe.py:148 (re-exported olmoasr/__init__.py:21).  pcm_dtype 0 = f32 waveform, 1 = int16 PCM
 *      (scaled by 1/32768 like train_timestamps.py:196).  mel: f32 [B, 80, n_samples/160].  The dynamic-range floor
 *      (max - 8) is per clip.  workspace: oasr_log_mel_workspace_bytes(B) bytes. */
size_t oasr_log_mel_workspace_bytes(int B);
int oasr_log_mel(const void* pcm, int pcm_dtype, int B, int n_samples, float* mel, void* workspace, void* stream);
/* The same front end without its last pass over the tensor: mel_raw = log10(max(mel power, 1e-10)), clip_max f32 [B] = each clip's maximum
 * of it.  whisper.audio's last two lines -- max(x, x.max() - 8), (x + 4) / 4 -- are then applied by the consumer while it reads the
 * tensor anyway (oasr_train_fwd_bwd_span's mel_clip_max): half the HBM traffic of this front end, bit-identical encoder input. */
int oasr_log_mel_raw(const void* pcm, int pcm_dtype, int B, int n_samples, float* mel_raw, float* clip_max, void* workspace, void* stream);
/* HOST helper: the slaney 80 x 201 filterbank (whisper assets/mel_filters.npz) into a host buffer. */
int oasr_mel_filterbank(float* out_host);

/* ---- model context: olmoasr.model.OLMoASR(dims) (olmoasr/model.py:778-813) -------------------------------------------- */
oasr_ctx* oasr_create(const oasr_dims* dims); /* training model: n_vocab + 1 embedding rows (pad row, model.py:665-667) */
/* embed_rows = n_vocab selects olmoasr.inf_model.OLMoASR's layout (inf_model.py:302; scripts/eval/gen_inf_ckpt.py) */
oasr_ctx* oasr_create_ex(const oasr_dims* dims, int embed_rows);
/* compute_dtype selects the arithmetic of the whole context -- the reference's --precision flag
 * (scripts/training/train_timestamps.py:2128: "bfloat16" | "float32"; autocast + dtype policy :1414, :2220-2224):
 *   OASR_DTYPE_BF16  production: bf16 MFMA operands / activations, fp32 accumulation and master weights (= autocast(bfloat16))
 *   OASR_DTYPE_F32   validation: the same engine schedule on plain fp32 kernels (fp32 activations, operands, softmax,
 *                    residual stream; exact-erf GELU).  Every `void*` activation the ABI exchanges (xa, kv_cache) then holds
 *                    fp32 instead of bf16.  This is the mode the "logits within 1e-3" criterion is tested in. */
#define OASR_DTYPE_BF16 0
#define OASR_DTYPE_F32 1
oasr_ctx* oasr_create_ex2(const oasr_dims* dims, int embed_rows, int compute_dtype);
int oasr_compute_dtype(const oasr_ctx*);
void oasr_destroy(oasr_ctx*);

/* Parameter table: one flat fp32 arena in gradient-ready (reverse-backward) order; the Python modules expose
 * nn.Parameter views of it under the reference's state_dict names. */
int oasr_param_count(const oasr_ctx*);
int oasr_param_info(const oasr_ctx*, int idx, char* name, int name_cap, int64_t* offset, int64_t* numel, int* ndim,
                    int64_t shape[4]);
int64_t oasr_param_numel(const oasr_ctx*);
/* Gradient segments (arena ranges that become final together during backward), in the order they complete. */
int oasr_segment_count(const oasr_ctx*);
int oasr_segment_info(const oasr_ctx*, int idx, int64_t* offset, int64_t* numel);

/* Bind the flat arenas (params/grads/exp_avg/exp_avg_sq: fp32 [numel]) and the encoder's sinusoid buffer
 * (encoder.positional_embedding, fp32 [n_audio_ctx, d], olmoasr/model.py:199-230,565). */
int oasr_bind(oasr_ctx*, float* params, float* grads, float* exp_avg, float* exp_avg_sq, const float* enc_pos);
/* bf16 compute copies of the weights (+ packed conv kernels, fused qkv biases). */
size_t oasr_shadow_bytes(const oasr_ctx*);
int oasr_bind_shadow(oasr_ctx*, void* shadow);
int oasr_refresh_shadow(oasr_ctx*, void* stream); /* after params changed outside oasr_optim_step */

#define OASR_MODE_INFER 0
#define OASR_MODE_TRAIN 1
size_t oasr_workspace_bytes(const oasr_ctx*, int B, int S, int mode);

/* OLMoASR.forward(mel, tokens, padding_mask) (olmoasr/model.py:856-887).  mel f32 [B,80,2*n_audio_ctx]; tokens i64 [B,S];
 * text_len i32 [B] = first padded key column of the reference's column-only padding mask (train_timestamps.py:314-315),
 * NULL = no padding mask (causal only).  logits_out f32 [B,S,n_vocab+1] (or NULL); xa_out bf16 [B,n_audio_ctx,d] (or NULL). */
int oasr_forward(oasr_ctx*, const float* mel, const int64_t* tokens, const int32_t* text_len, int B, int S, float* logits_out,
                 void* xa_out, void* workspace, size_t workspace_bytes, void* stream);

/* OLMoASR.embed_audio(mel) (model.py:815): xa_out bf16 [B, n_audio_ctx, d]. */
int oasr_encode(oasr_ctx*, const float* mel, int B, void* xa_out, void* workspace, size_t workspace_bytes, void* stream);
/* OLMoASR.logits(tokens, audio_features) (model.py:818-854): decoder on given xa.  last_only != 0 -> logits_out f32
 * [B, rows] of position S-1 only (the greedy step of whisper.decoding / notebooks/ow_decoding.py:50-72). */
int oasr_decode_logits(oasr_ctx*, const int64_t* tokens, const void* xa, const int32_t* text_len, int B, int S, int last_only,
                       float* logits_out, void* workspace, size_t workspace_bytes, void* stream);

/* Cached greedy decoding = the reference's install_kv_cache_hooks (model.py:925-964 / inf_model.py:422-453) + one
 * TextDecoder step per token.  kv_cache: oasr_kv_cache_bytes(B) bytes, caller owned, valid for one 30 s window batch.
 * decode_begin computes the cross-attention K/V of every layer from xa (bf16 [B, n_audio_ctx, d]); decode_step consumes
 * the token at position `pos` of each sequence (tokens_last i64 [B]) and returns f32 logits [B, rows] for position pos+1.
 * Engines (one arithmetic, csrc/decode_shared.h): ONE sequence on the bf16 engine -> one persistent launch for the whole decoder stack
 * on every CU of the device (csrc/decode_wide.hip: a few weight rows per workgroup, rows exchanged through per-workgroup phase flags;
 * agrees with the other engines to the fp32 rounding of a differently ordered K sum); where that engine does not apply, the one-XCD team
 * of csrc/decode_xcd.hip; 2-4 sequences -> LayerNorm folded into the projections' operand loads; more -> separate kernels (these three
 * are bit-identical).  The one-launch engines keep their control words and phase flags in the cache's last OASR_KV_TAIL_BYTES bytes
 * (zeroed by decode_begin; the cache is oasr_kv_cache_bytes(B) bytes INCLUDING that tail -- ABI 211; a caller that re-packs a cache,
 * e.g. the beam re-gather of whisper's rearrange_kv_cache, zeroes the tail of the new buffer).  A one-launch engine needs all of its
 * workgroups resident at once, i.e. the device to itself: a workgroup that never arrives (a second decoder on the device, a CU mask)
 * poisons a flag instead of hanging; oasr_decode_check then clears it, switches the CONTEXT to the multi-launch engine for good and
 * returns OASR_ERETRY: the caller decodes the window again. */
#define OASR_KV_TAIL_BYTES 327680
size_t oasr_kv_cache_bytes(const oasr_ctx*, int B);
size_t oasr_decode_step_workspace_bytes(const oasr_ctx*, int B);
int oasr_decode_begin(oasr_ctx*, const void* xa, int B, void* kv_cache, void* stream);
int oasr_decode_step(oasr_ctx*, const int64_t* tokens_last, int B, int pos, void* kv_cache, float* logits_out, void* workspace,
                     size_t workspace_bytes, void* stream);
/* Synchronises the stream and checks the one-launch engine's error flag: call once per decoded window, where the caller reads the
 * tokens back.  OASR_OK, or OASR_ERETRY (see above: enqueue the window's begin / steps again; at most once per context). */
int oasr_decode_check(oasr_ctx*, int B, void* kv_cache, void* stream);

/* One micro-step of train() (train_timestamps.py:1440-1454): forward, CE(ignore_index=pad)/accum, backward.
 * Gradients of the loss scaled by loss_scale are ACCUMULATED into the bound grad arena (zero it with oasr_zero_grad
 * at the start of an accumulation window).  loss_out (device f32): unscaled loss/accum, overwritten or accumulated.
 * seg_events: NULL or oasr_segment_count() hipEvent_t handles; event i is recorded when segment i's gradient is final.
 * logits_out: NULL, or f32 [B,S,n_vocab+1] (parity mode; costs an extra pass). */
int oasr_train_fwd_bwd(oasr_ctx*, const float* mel, const int64_t* tokens, const int64_t* targets, const int32_t* text_len, int B,
                       float loss_scale, float inv_accum, float* loss_out, int accumulate_loss, float* logits_out,
                       void** seg_events, void* workspace, size_t workspace_bytes, void* stream);
/* The same step over the first S <= n_text_ctx decoder positions (tokens/targets [B,S]; logits_out [B,S,n_vocab+1]).
 * For S >= max(text_len) the loss and all gradients equal the full-context ones (the trimmed positions are pure
 * padding: ignore_index targets, never attended to by a real query) -- an opt-in the reference does not have. */
int oasr_train_fwd_bwd_s(oasr_ctx*, const float* mel, const int64_t* tokens, const int64_t* targets, const int32_t* text_len, int B,
                         int S, float loss_scale, float inv_accum, float* loss_out, int accumulate_loss, float* logits_out,
                         void** seg_events, void* workspace, size_t workspace_bytes, void* stream);

/* The same micro-step with the decoder's BACKWARD limited to the supervised span -- exact, and the forward still covers all
 * n_text_ctx positions like the reference's (train_timestamps.py:318-329 pads every sample to 448; :1444 then ignores the padding).
 * span_host: HOST int32 [B] (the data loader builds the token sequences on the host, train_timestamps.py:238-343):
 * every target of sample b at a position >= span_host[b] is ignore_index, and span_host[b] >= text_len[b].  Gradient rows past the
 * span are exactly zero in the reference's computation, so the decoder's token rows are stored in 64-position chunks with the
 * chunks that can carry gradient first and the decoder's backward GEMMs / LayerNorms / attention run over those rows only
 * (olmoasr_amd/csrc/engine.hip).  Loss and gradients equal oasr_train_fwd_bwd's up to fp32 summation order.  Falls back to the
 * plain step when n_text_ctx is not a multiple of 64 or B > 512. */
#define OASR_SPAN_FORWARD_ALL 0    /* the reference's shape: the decoder's forward covers all n_text_ctx positions */
#define OASR_SPAN_FORWARD_ACTIVE 1 /* opt-in: the forward leaves the positions past the span out too -- their logits exist in the reference
                                    * (model.py:768-770 over the padded context) but nothing reads them: loss and gradients unchanged */
/* mel_clip_max: NULL (mel is finished log-mel, as everywhere else), or device f32 [B] with mel = oasr_log_mel_raw's output.
 * Streams: asynchronous on `stream` like everything else.  Part of the decoder's backward (weight-gradient GEMMs) runs on two lowest-priority
 * streams the context owns (created on the first call, destroyed by oasr_destroy), forked from and joined back into `stream` with events inside
 * the call: when the call returns, everything it enqueued is ordered before whatever the caller enqueues on `stream` next, and each
 * seg_events[i] still means "every gradient of segment i is complete". */
int oasr_train_fwd_bwd_span(oasr_ctx*, const float* mel, const int64_t* tokens, const int64_t* targets, const int32_t* text_len,
                            const int32_t* span_host, int forward_rows, const float* mel_clip_max, int B, float loss_scale, float inv_accum,
                            float* loss_out, int accumulate_loss, void** seg_events, void* workspace, size_t workspace_bytes, void* stream);

/* The same micro-step cut at the logits, for torch.autograd: OLMoASR.forward in training mode (olmoasr/model.py:856-887) followed by
 * the CALLER's loss and .backward() (train_timestamps.py:1440-1454 unchanged).  train_fwd: fp32 logits [B, S, rows], every saved
 * activation stays in the workspace.  train_bwd: dlogits = d(loss)/d(logits), fp32, same shape -> parameter gradients ACCUMULATED into
 * the bound arena (seg_events as above).  Same B, S, tokens, text_len for the pair; the workspace must not be reused in between. */
int oasr_train_fwd(oasr_ctx*, const float* mel, const int64_t* tokens, const int32_t* text_len, int B, int S, float* logits_out,
                   void* workspace, size_t workspace_bytes, void* stream);
int oasr_train_bwd(oasr_ctx*, const int64_t* tokens, const int32_t* text_len, const float* dlogits, int B, int S, void** seg_events,
                   void* workspace, size_t workspace_bytes, void* stream);
int oasr_zero_grad(oasr_ctx*, void* stream);

/* scaler.unscale_ + clip_grad_norm_(max_norm) + AdamW.step + bf16 shadow refresh (train_timestamps.py:1509-1512).
 * step is 1-based.  stats_out (device f32[2]): [0] = sum of squares of the SCALED grads, [1] = non-finite flag
 * (step skipped when nonzero, like GradScaler).  scratch: >= 8192 bytes device. */
int oasr_optim_step(oasr_ctx*, float inv_loss_scale, float max_grad_norm, float lr, float beta1, float beta2, float eps,
                    float weight_decay, int64_t step, float* stats_out, void* scratch, void* stream);

/* ZeRO-1 (optimizer-state sharding; the reference's sharded variant is FSDP, scripts/training/train_fsdp_timestamps.py:2665-2719):
 * the same fused step over a contiguous range [off, off+numel) of the arenas (multiples of 4).  sumsq_range: stats_out[0] = sum of
 * squares of the scaled gradients in the range, [1] = non-finite flag.  step_range: `stats` = those two numbers summed over ALL
 * ranges (all-reduce them), exp_avg / exp_avg_sq are shard-local buffers of numel floats.  After every rank has stepped its range
 * and the parameters are all-gathered, call oasr_refresh_shadow.  Host side: olmoasr_amd/zero.py. */
int oasr_grad_sumsq_range(oasr_ctx*, int64_t off, int64_t numel, float* stats_out, void* scratch, void* stream);
int oasr_optim_step_range(oasr_ctx*, int64_t off, int64_t numel, float* exp_avg_shard, float* exp_avg_sq_shard, const float* stats,
                          float inv_loss_scale, float max_grad_norm, float lr, float beta1, float beta2, float eps, float weight_decay,
                          int64_t step, void* stream);

/* ---- unit operators (exposed for op-level parity tests; the engine calls the same launchers) ---------------------- */
typedef struct oasr_operand { /* bf16 matrix, optionally a conv-window view: see olmoasr_amd/csrc/kernels.h */
  const void* ptr; int64_t ld; int rpb; int64_t bstride; int lead; int kvalid; int trail_from;
} oasr_operand;
typedef struct oasr_gemm_args {
  oasr_operand A, B; int M, N, K; int ta, tb; float alpha;
  const float* bias; int act; const float* pos; int pos_period;
  const void* dgelu_u; int64_t ldu; const void* resid; int64_t ldr;
  void* out; void* out_pre; int64_t ldc; float* out_f32; int64_t ldc32; float beta; float* colsum; int atomic; int split_k;
  int dgelu_deriv; /* dgelu_u holds GELU'(u) itself (written by an act == 2 forward: out_pre = GELU'(pre)) */
} oasr_gemm_args;
int oasr_gemm(const oasr_gemm_args*, void* stream);
int oasr_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int64_t rows, int d, void* stream);
int oasr_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, const void* dres,
                       void* dx, float* dgamma, float* dbeta, int64_t rows, int d, void* stream);
typedef struct oasr_attn_args {
  const void *q, *k, *v; int64_t ldq, ldk, ldv, bsq, bsk, bsv; void* o; int64_t ldo, bso; float* lse; void* o_lo; const int32_t* kv_len;
  int B, H, Tq, Tk, causal; const void* d_o; float* delta; void *dq, *dk, *dv;
  float *dq_colsum, *dv_colsum; /* optional [H*64], accumulated: column sums of dq / dv = query / value bias gradients */
  float* colsum_scratch;        /* with either of them: B * (ceil(Tq/128) + ceil(Tk/128)) * H*64 floats of scratch */
  int32_t* qtile_flags;         /* optional [B, H, ceil(Tq/64)] workspace (backward): 64-query tiles of d_o that are all zero -- the padded
                                 * positions of a decoder batch -- are recorded by the dQ kernel and skipped by both; bit-identical */
  /* ABI 200: chunked token rows (the decoder of oasr_train_fwd_bwd_span).  q_rows / k_rows: optional int32 [B][OASR_ROWTAB]: first
   * token row -- relative to the base pointers, batch strides unused -- of every 64-position chunk of sample b (q_rows: q, o, o_lo,
   * d_o, dq; k_rows: k, v, dk, dv); q_span: optional int32 [B], multiples of 64 (backward): d_o is zero at query positions >=
   * q_span[b]; those rows are not read and their dq (self-attention: dk / dv too) not written. */
  const int32_t *q_rows, *k_rows, *q_span;
} oasr_attn_args;
#define OASR_ROWTAB 16
/* sizeof(oasr_attn_args) as the library was built: a binding compares it with its own before the first call */
size_t oasr_sizeof_attn_args(void);
int oasr_attention_fwd(const oasr_attn_args*, void* stream);
int oasr_attention_bwd(const oasr_attn_args*, void* stream);
/* The score matrix on request = `qk` of MultiHeadAttention.qkv_attention (olmoasr/model.py:347-442, returned by forward :327,:345 on the
 * manual path; inf_model.py:172-196): scores f32 [B, H, Tq, Tk] = (q * 64^-1/4) . (k * 64^-1/4), pre-softmax, with -inf where the
 * reference's additive mask puts it (causal: j > i; key padding: j >= kv_len[b]).  Reads q, k, their strides, kv_len, B, H, Tq, Tk, causal
 * of the argument block; dtype = OASR_DTYPE_BF16 / OASR_DTYPE_F32 of the operands.  The training and decoding kernels never form this
 * matrix; word-level timestamp alignment (whisper.timing.find_alignment, called from olmoasr/transcribe.py:410-419) reads it from the
 * cross-attention of the upper decoder layers. */
int oasr_attention_scores(const oasr_attn_args*, int dtype, float* scores, void* stream);
int oasr_cross_entropy(void* logits_bf16, int64_t ld, int V, const int64_t* targets, int64_t rows, int64_t ignore, float gscale,
                       int32_t* n_valid_dev, float* row_loss, float* loss_out, int write_grad, void* stream);
int oasr_cast_f32_bf16(const float* src, void* dst, int64_t n, void* stream);
/* Per-row token pick over fp32 logits [rows, ld] (first V columns): tok = argmax(logits + mask + mask2) with the lowest index
 * among equal maxima, logprob = log_softmax(logits + masks)[tok] (NULL to skip).  masks: additive f32 [V] (0 / -inf) or NULL.
 * = the tail of whisper.decoding GreedyDecoder.update (argmax + log_softmax gather) after SuppressBlank / SuppressTokens,
 * and gen_pred's argmax over teacher-forced logits (scripts/training/train_timestamps.py:1096-1098). */
int oasr_pick_tokens(const float* logits, int64_t ld, int V, int64_t rows, const float* mask, const float* mask2, int64_t* tok,
                     float* logprob, void* stream);
/* The same pick in TIMESTAMP mode -- the reference's default for transcribe() (olmoasr/transcribe.py:212: DecodingOptions without
 * without_timestamps) -- with whisper.decoding.ApplyTimestampRules evaluated on the device from the sampled history
 * (history i64 [rows, history_ld], n_history tokens sampled so far per row; max_initial_index < 0: no limit on the first timestamp):
 * pairs must be closed, text must follow a pair, timestamps do not decrease, the first token is a timestamp <= max_initial, and a
 * timestamp is forced when the timestamp mass beats every text token.  Replaces ApplyTimestampRules.apply + GreedyDecoder.update. */
int oasr_pick_tokens_ts(const float* logits, int64_t ld, int V, int64_t rows, const float* mask, const float* mask2,
                        const int64_t* history, int64_t history_ld, int n_history, int timestamp_begin, int eot, int no_timestamps,
                        int max_initial_index, int64_t* tok, float* logprob, void* stream);
/* Beam search and sampling on the SAME filtered distribution (suppress masks; with n_history >= 0 also ApplyTimestampRules from the
 * device-resident history, as above; n_history < 0: masks only -- the without_timestamps mode).
 * topk: tok / logprob [rows, K] = the K (<= 16) largest log_softmax values and their tokens, descending, ties to the lower id --
 *       whisper.decoding.BeamSearchDecoder.update's logprobs.topk(beam_size + 1) (the reference's long-form eval is beam 5,
 *       scripts/eval/eval.py:2077-2084).
 * sample: one draw per row from softmax(filtered logits / temperature) by inverse CDF on uniforms[row] in [0, 1) (the caller's
 *       generator), logprob = the draw's log_softmax at temperature 1 -- GreedyDecoder.update at temperature > 0 (the fallback
 *       temperatures of olmoasr/transcribe.py:193-233). */
int oasr_topk_tokens(const float* logits, int64_t ld, int V, int64_t rows, const float* mask, const float* mask2, const int64_t* history,
                     int64_t history_ld, int n_history, int timestamp_begin, int eot, int no_timestamps, int max_initial_index, int K,
                     int64_t* tok, float* logprob, void* stream);
int oasr_sample_tokens(const float* logits, int64_t ld, int V, int64_t rows, const float* mask, const float* mask2, const int64_t* history,
                       int64_t history_ld, int n_history, int timestamp_begin, int eot, int no_timestamps, int max_initial_index,
                       float temperature, const float* uniforms, int64_t* tok, float* logprob, void* stream);
/* Measurement / test hooks (GEMM launch timing for bench.py, kernel-path forcing, hardware probes) are declared in
 * include/oasr_testing.h: they are exported by the same library but are not part of the product surface. */

#ifdef __cplusplus
}
#endif
#endif

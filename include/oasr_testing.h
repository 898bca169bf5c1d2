/* liboasr -- measurement and test hooks.  Exported by liboasr.so next to the product ABI of include/oasr.h, but NOT part of
 * the drop-in surface: nothing on the training / decoding path calls them.  Users: bench.py (live GEMM roofline), tests/
 * (kernel-path forcing, hardware-behaviour probes that pin what the kernels rely on), scripts/ (A/B experiments).
 * The setters that change process-wide kernel selection (oasr_gemm_set_variant / _set_stagger / _force_general,
 * oasr_attention_set_pingpong, oasr_decode_set_ln_fold, oasr_span_set_side_streams) are INERT unless the process opts in with OASR_TESTING_HOOKS=1 in its
 * environment: without it they return OASR_ESTATE and change nothing, so a production process cannot be steered through them. */
#ifndef OASR_TESTING_H
#define OASR_TESTING_H
#include "oasr.h"
#ifdef __cplusplus
extern "C" {
#endif

/* bench.py's live roofline measurement: when enabled every GEMM launch is bracketed by HIP events on ITS stream;
 * collect() synchronises and returns, per kernel variant (index 2*ta+tb: 0 = NT forward, 1 = NN dgrad, 3 = TN wgrad),
 * summed milliseconds, summed algorithmic flops (2*M*N*K, conv windows at their real width) and launch count; by_symbol
 * receives the same sums keyed by the kernel symbol rocprofv3 prints, so the two can be compared line by line. */
int oasr_profile_gemm(int enable);
/* experiments on the 256x256 kernel.  v < 0: defaults.  bits 0-3: schedule variant (8 = per-layout default); bits 4-5: 1 = plain
 * launches, 2 = persistent launches (next tile's prologue ahead of the epilogue); bit 6 / 7: non-temporal epilogue stores / side loads */
int oasr_gemm_set_variant(int v);
/* tests / A-B of the KV-cached step engine: -1 = default (ONE sequence on the bf16 engine: the chip-wide one-launch engine of
 * csrc/decode_wide.hip; 2-4: LayerNorm folded into the projections; more: separate kernels), 0 = separate LayerNorm kernels, 1 = LayerNorm
 * folded into the projections' operand loads for every B <= 32, 2 = the one-launch TEAM engine of csrc/decode_xcd.hip on the 32 CUs of one
 * XCD (B <= 4), 3 / 4 = that team as 32 / 64 workgroups spread over the chip, 5 = the chip-wide engine (one sequence; more: as 2).
 * 0-4 are bit-identical; 5 agrees with them to the fp32 rounding of differently ordered K sums (tests/test_gpu_decode_step.py). */
int oasr_decode_set_ln_fold(int mode);
/* tests / A-B: side streams of the supervised-span step (oasr_train_fwd_bwd_span; csrc/engine.hip: Runner::side_mode).  Bit 0: the decoder
 * backward's weight gradients over the R active rows, bit 2: the cross-attention key|value weight gradient and d(xa) -- run on lowest-priority
 * streams beside the data-gradient chain; bit 1: the key|value projections of the decoder forward likewise; bit 3: without segment events leave
 * the key|value gradients in flight across blocks.  -1 = the library default.  Gradients differ by fp32 atomic order only
 * (tests/test_gpu_span.py).  The getter returns the mode in effect. */
int oasr_span_set_side_streams(int mode);
int oasr_span_side_streams(void);
/* tests (CPU): the static block stream of workgroup `wg` of the one-launch step engine as its cursors generate it: out[4 i ..] = layer,
 * segment (0 qkv, 1 attn.out, 2 cross q, 3 cross K/V, 4 cross out, 5 mlp.0, 6 mlp.2), tile / item ordinal, block; returns the count. */
int oasr_xcd_plan_debug(int d, int H, int Te, int M, int L, int team, int wg, int* out, int max_blocks);
/* tests (CPU): 1 when every 32-bit buffer offset of a one-launch step stays below 2 GiB (layer0 = the 18 element offsets of decoder layer 0
 * in csrc/decode_xcd.hip::XLayer order, strides in elements), 0 when the engine must take the multi-launch step instead. */
/* the chip-wide step engine (csrc/decode_wide.hip): does the shape run on it with `nwg` workgroups; the (row, 512-element K span) units of compute wave
 * `wave` (1 .. 8) of workgroup `wg` for an [N x K] projection of a width-d model: out[2 i] = row, out[2 i + 1] = span; returns the count (-1: bad arguments or
 * past the kernel's unrolled bound) */
int oasr_wide_supports_debug(int d, int H, int Te, int S_max, int L, int M, int nwg);
int oasr_wide_plan_debug(int d, int nwg, int N, int K, int wg, int wave, int* out, int max_units);
int oasr_xcd_offsets_ok_debug(const int64_t* layer0, long long lstride, long long cache_lstride, int d, int Te, int L, int M);
/* tests / A-B: 1 (default) = the unmasked attention cases (encoder self-, cross-attention) run the 8-wave ping-pong kernels,
 * 0 = the general (maskable) kernels run everything.  Same results up to accumulation order (tests/test_gpu_ops.py). */
int oasr_attention_set_pingpong(int on);
int oasr_gemm_set_stagger(int sleeps, int phases); /* experiments: first-wave phase stagger of the 256x256 kernel (0 = off) */
int oasr_gemm_force_general(int on); /* tests: route every GEMM through the register-staged general kernel */
int oasr_profile_gemm_collect(double* ms4, double* flops4, int64_t* count4, char* by_symbol /* "symbol\tlaunches\tms\tflops\n"... or NULL */, int cap);
int oasr_probe_lds_oob(const void* src_u16 /*[512]*/, void* dst_u16 /*[512]*/, void* stream);
int oasr_probe_tr16(const void* src_bf16 /*[16][64]*/, void* dst_bf16 /*[64 lanes][4]*/, void* stream);
/* tests: the tables oasr_train_fwd_bwd_span builds for one micro-batch -- span_host: HOST int32 [B]; rows_out: device int32
 * [B][OASR_ROWTAB] chunk-row table; span_out: device int32 [B] spans rounded up to 64; targets_rows_out: device int64 [B*S] targets in
 * row order (active rows only); active_rows_out: HOST int64, the number of leading rows the decoder's backward runs over. */
int oasr_test_span_tables(const int32_t* span_host, int B, int S, const int64_t* targets, int32_t* rows_out, int32_t* span_out,
                          int64_t* targets_rows_out, int64_t* active_rows_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif

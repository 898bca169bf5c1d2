// Cross-entropy over the BPE vocabulary, forward + backward in one pass (gfx950, HBM-bound).
// Reference: F.cross_entropy(logits.view(-1, V+1), text_y.view(-1), ignore_index=51864) / accumulation_steps
// (scripts/training/train_timestamps.py:1444-1450) on logits that the reference produced in bf16 and then
// .float()-ed (olmoasr/model.py:768-770); the gradient it back-propagates into the tied-logits matmul is bf16.
// One 1024-thread workgroup per token row: the 51865 bf16 logits (104 KB) are read ONCE into registers
// (7 x 16 bytes per thread), softmax statistics in fp32, and (softmax - onehot) * g is written back in place
// as bf16 -> 2 * 2 * V bytes of HBM traffic per row, no fp32 logits tensor ever exists.
#include "kernels.h"

namespace {

// Workgroup width (scripts/ce_bench.py, [57344, 51968] bf16, every row valid, one MI355X): 256 threads x 26 chunks = 146 VGPRs, three rows
// per CU: 2.66 ms; 512 x 13: 2.54 ms; 1024 x 7 (62 VGPRs, 32 waves per CU): 2.42 ms = 4.93 TB/s = 0.62 of 8 TB/s.  (Round 2: 3.34 ms, with a
// per-element column test, a per-element onehot compare and the natural-base exponential in both passes.)
constexpr int CE_THREADS = 1024;
constexpr int CE_CHUNKS = 7;     // 7 * 1024 threads * 8 = 57344 >= 51968

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = is_max ? wave_max(v) : wave_sum(v);
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return is_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : (red[0] + red[1]) + (red[2] + red[3]);
}
// the same over the 8 waves of a cross-entropy workgroup (fixed combination order: deterministic)
__device__ __forceinline__ float block_reduce8(float v, float* red, bool is_max) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = is_max ? wave_max(v) : wave_sum(v);
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
#pragma unroll
  for (int w = 1; w < CE_THREADS / 64; ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
  return r;
}

__global__ __launch_bounds__(CE_THREADS) void ce_kernel(bf16_t* __restrict__ logits, long ld, int V, const int64_t* __restrict__ targets,
                                                 long ignore, float gscale, const int32_t* __restrict__ n_valid_dev,
                                                 float* __restrict__ row_loss, int write_grad) {
  __shared__ float red[CE_THREADS / 64];
  const long row = blockIdx.x;
  bf16_t* lr = logits + row * ld;
  const long tgt = targets[row];
  const int nchunk = (int)(ld >> 3);
  // ignored row: zero gradient, zero loss (uniform branch).  A target outside [0, V) (F.cross_entropy would raise) is
  // treated the same way instead of reading / writing outside the row; hosts that want the raise validate ids up front.
  if (tgt == ignore || tgt < 0 || tgt >= V) {
    if (write_grad) {
      const u32x4_t z = {0u, 0u, 0u, 0u};
      for (int ch = threadIdx.x; ch < nchunk; ch += CE_THREADS) *(u32x4_t*)(lr + ch * 8) = z;
    }
    if (threadIdx.x == 0) row_loss[row] = 0.f;
    return;
  }
  // The row is VALU-heavy once it sits in registers (two exponentials per logit), so the per-element work is kept minimal: columns
  // past V (the 128-padding of the tied head) are overwritten with -inf ONCE after the load -- they then drop out of the maximum,
  // contribute exp(-inf) = 0 to the sum and get a zero gradient without any per-element test -- the exponent runs in the log2
  // domain (one fma + v_exp_f32), and the "- onehot" of the target column is patched into the one chunk that holds it.
  constexpr float L2E = 1.4426950408889634f;
  u32x4_t v[CE_CHUNKS];
  float mx = -3.0e38f;
#pragma unroll
  for (int c = 0; c < CE_CHUNKS; ++c) {
    const int ch = threadIdx.x + CE_THREADS * c;
    if (ch < nchunk) {
      v[c] = *(const u32x4_t*)(lr + ch * 8);
      if (ch * 8 + 8 > V) {  // (only the last ~13 chunks of a row)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int col = ch * 8 + 2 * i;
          if (col >= V) v[c][i] = (v[c][i] & 0xffff0000u) | 0x0000ff80u;
          if (col + 1 >= V) v[c][i] = (v[c][i] & 0x0000ffffu) | 0xff800000u;
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) mx = fmaxf(mx, fmaxf(bf_lo(v[c][i]), bf_hi(v[c][i])));
    }
  }
  mx = block_reduce8(mx, red, true);
  const float nm2 = -mx * L2E;
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < CE_CHUNKS; ++c) {
    const int ch = threadIdx.x + CE_THREADS * c;
    if (ch < nchunk) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        sum += __builtin_amdgcn_exp2f(fmaf(bf_lo(v[c][i]), L2E, nm2)) + __builtin_amdgcn_exp2f(fmaf(bf_hi(v[c][i]), L2E, nm2));
    }
  }
  sum = block_reduce8(sum, red, false);
  const float lse = mx + __logf(sum);
  if (threadIdx.x == 0) row_loss[row] = lse - bf2f(lr[tgt]);
  if (!write_grad) return;
  __syncthreads();  // lr[tgt] read above must precede the in-place overwrite
  const float g = gscale / (float)max(1, *n_valid_dev);
  const float sg = g / sum;  // softmax * g = exp2(..) * sg
  const int tch = (int)(tgt >> 3), tpos = (int)(tgt & 7);
#pragma unroll
  for (int c = 0; c < CE_CHUNKS; ++c) {
    const int ch = threadIdx.x + CE_THREADS * c;
    if (ch < nchunk) {
      float e[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        e[2 * i] = __builtin_amdgcn_exp2f(fmaf(bf_lo(v[c][i]), L2E, nm2)) * sg;
        e[2 * i + 1] = __builtin_amdgcn_exp2f(fmaf(bf_hi(v[c][i]), L2E, nm2)) * sg;
      }
      if (ch == tch) {  // (one thread of the row)
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (i == tpos) e[i] -= g;
      }
      u32x4_t o;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = pack_bf2(e[2 * i], e[2 * i + 1]);
      *(u32x4_t*)(lr + ch * 8) = o;
    }
  }
}

__global__ __launch_bounds__(256) void count_valid_kernel(const int64_t* __restrict__ t, long rows, long ignore, int V, int32_t* out) {
  __shared__ int red[4];
  int c = 0;
  // F.cross_entropy's denominator; the predicate is ce_kernel's (an id outside [0, V) contributes neither loss nor count)
  for (long i = threadIdx.x; i < rows; i += 256) c += (t[i] != ignore && t[i] >= 0 && t[i] < V) ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) *out = red[0] + red[1] + red[2] + red[3];
}

// deterministic single-block reduction (rows <= a few hundred thousand)
__global__ __launch_bounds__(256) void loss_reduce_kernel(const float* __restrict__ row_loss, long rows,
                                                          const int32_t* __restrict__ n_valid_dev, float mul,
                                                          float* __restrict__ loss_out, int accumulate) {
  __shared__ double red[4];
  double s = 0.0;
  for (long i = threadIdx.x; i < rows; i += 256) s += (double)row_loss[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double tot = (red[0] + red[1]) + (red[2] + red[3]);
    const float v = (float)(tot / (double)max(1, *n_valid_dev)) * mul;
    *loss_out = accumulate ? (*loss_out + v) : v;
  }
}

// Token pick over the vocabulary (the per-step tail of whisper.decoding's GreedyDecoder.update, and gen_pred's argmax,
// train_timestamps.py:1096-1098): one workgroup per row of fp32 logits;
//   x[c] = logits[c] + mask[c] (+ mask2[c])          (additive 0 / -inf suppress masks, optional)
//   tok = argmax_c x[c] (lowest index among equal maxima), logprob = x[tok] - logsumexp(x)
__global__ __launch_bounds__(256) void pick_kernel(const float* __restrict__ logits, long ld, int V, const float* __restrict__ mask,
                                                  const float* __restrict__ mask2, int64_t* __restrict__ tok, float* __restrict__ logprob) {
  __shared__ float red[4];
  __shared__ float bv[4];
  __shared__ int bi[4];
  const float* lr = logits + (long)blockIdx.x * ld;
  float best = -INFINITY;
  int besti = 0x7fffffff;
  for (int c = threadIdx.x; c < V; c += 256) {
    float x = lr[c];
    if (mask) x += mask[c];
    if (mask2) x += mask2[c];
    if (x > best) {  // strict: keeps the lowest index of this thread's stride
      best = x;
      besti = c;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(besti, o, 64);
    if (ov > best || (ov == best && oi < besti)) {
      best = ov;
      besti = oi;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    bv[wave] = best;
    bi[wave] = besti;
  }
  __syncthreads();
  best = bv[0];
  besti = bi[0];
#pragma unroll
  for (int w = 1; w < 4; ++w)
    if (bv[w] > best || (bv[w] == best && bi[w] < besti)) {
      best = bv[w];
      besti = bi[w];
    }
  float sum = 0.f;
  if (logprob) {
    for (int c = threadIdx.x; c < V; c += 256) {
      float x = lr[c];
      if (mask) x += mask[c];
      if (mask2) x += mask2[c];
      sum += __expf(x - best);
    }
    sum = block_reduce(sum, red, false);
  }
  if (threadIdx.x == 0) {
    tok[blockIdx.x] = besti == 0x7fffffff ? 0 : besti;
    if (logprob) logprob[blockIdx.x] = -__logf(sum);
  }
}


// The same pick with whisper.decoding.ApplyTimestampRules folded in (timestamp mode is the reference's default for transcribe(),
// olmoasr/transcribe.py:212; whisper/decoding.py ApplyTimestampRules.apply): the rules are a function of the row's sampled history,
// which lives on the device (int64 [rows, hist_ld], n_hist tokens sampled so far), so the greedy loop needs no host round trip per
// token.  Column c of row r is masked (-inf) when
//   c == <|notimestamps|>;
//   the last sampled token is a timestamp and  (the one before it too, or there is none)  -> c >= ts_begin   (text must follow a pair)
//                                              (the one before it is text)                -> c <  eot         (a pair must be closed)
//   a timestamp was sampled before: ts_begin <= c < last_ts (+ 1 unless the last token is an unpaired timestamp)   (monotonic, > 0 s)
//   nothing sampled yet: c < ts_begin, and c > ts_begin + max_initial_index when that is >= 0;
// then, on log_softmax of what is left: if logsumexp(timestamp part) > max(text part) every text column is masked too.
// tok = argmax of the survivors (lowest index on ties), logprob = its log_softmax over the survivors (GreedyDecoder.update).
// One pass: online (max, argmax, sum-exp) for the text part and the timestamp part separately.
struct PickPart {
  float m, s;
  int i;
};
__device__ __forceinline__ void pick_merge(PickPart& a, const PickPart& b) {
  if (b.m > a.m || (b.m == a.m && b.i < a.i)) {
    a.s = a.s * __expf(a.m - b.m) + b.s;  // (a.m = -inf: exp(-inf) = 0; both -inf: handled by the caller's guard)
    a.m = b.m;
    a.i = b.i;
  } else if (b.m > -INFINITY) {
    a.s += b.s * __expf(b.m - a.m);
  }
}
__device__ __forceinline__ PickPart pick_block_reduce(PickPart p, PickPart* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    PickPart q;
    q.m = __shfl_xor(p.m, o, 64);
    q.s = __shfl_xor(p.s, o, 64);
    q.i = __shfl_xor(p.i, o, 64);
    pick_merge(p, q);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = p;
  __syncthreads();
  p = sh[0];
#pragma unroll
  for (int w = 1; w < 4; ++w) pick_merge(p, sh[w]);
  return p;
}
__global__ __launch_bounds__(256) void pick_ts_kernel(const float* __restrict__ logits, long ld, int V, const float* __restrict__ mask,
                                                     const float* __restrict__ mask2, const int64_t* __restrict__ hist, long hist_ld,
                                                     int n_hist, int ts_begin, int eot, int no_ts, int max_initial_index,
                                                     int64_t* __restrict__ tok, float* __restrict__ logprob) {
  __shared__ PickPart sh[4];
  __shared__ int sh_last[4];
  const float* lr = logits + (long)blockIdx.x * ld;
  const int64_t* h = hist + (long)blockIdx.x * hist_ld;
  // position of the last sampled timestamp (history is at most n_text_ctx tokens)
  int last_pos = -1;
  for (int t = threadIdx.x; t < n_hist; t += 256)
    if (h[t] >= ts_begin) last_pos = t;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) last_pos = max(last_pos, __shfl_xor(last_pos, o, 64));
  if ((threadIdx.x & 63) == 0) sh_last[threadIdx.x >> 6] = last_pos;
  __syncthreads();
  last_pos = max(max(sh_last[0], sh_last[1]), max(sh_last[2], sh_last[3]));
  const bool last_was_ts = n_hist >= 1 && h[n_hist - 1] >= ts_begin;
  const bool pen_was_ts = n_hist < 2 || h[n_hist - 2] >= ts_begin;
  int ts_lo = -1;  // timestamps below this are masked
  if (last_pos >= 0) ts_lo = (int)h[last_pos] + ((last_was_ts && !pen_was_ts) ? 0 : 1);
  const bool first = n_hist == 0;
  const int ts_hi = (first && max_initial_index >= 0) ? ts_begin + max_initial_index : 0x7fffffff;  // timestamps above this are masked

  PickPart text{-INFINITY, 0.f, 0x7fffffff}, stamp{-INFINITY, 0.f, 0x7fffffff};
  for (int c = threadIdx.x; c < V; c += 256) {
    float x = lr[c];
    if (mask) x += mask[c];
    if (mask2) x += mask2[c];
    const bool is_ts = c >= ts_begin;
    bool dead = c == no_ts;
    if (last_was_ts) dead = dead || (pen_was_ts ? is_ts : c < eot);
    if (is_ts) dead = dead || c < ts_lo || c > ts_hi;
    else dead = dead || first;
    if (dead || !(x > -INFINITY)) continue;
    PickPart& p = is_ts ? stamp : text;
    if (x > p.m) {  // strict: lowest index of this thread's stride
      p.s = p.s * __expf(p.m - x) + 1.f;
      p.m = x;
      p.i = c;
    } else {
      p.s += __expf(x - p.m);
    }
  }
  text = pick_block_reduce(text, sh);
  stamp = pick_block_reduce(stamp, sh);
  if (threadIdx.x == 0) {
    const float M = fmaxf(text.m, stamp.m);
    int pick = 0;
    float lp = -INFINITY;
    if (M > -INFINITY) {
      const float st = text.m > -INFINITY ? text.s * __expf(text.m - M) : 0.f;
      const float ss = stamp.m > -INFINITY ? stamp.s * __expf(stamp.m - M) : 0.f;
      const float lse = __logf(st + ss);
      const bool force = stamp.m > -INFINITY && (text.m == -INFINITY || (__logf(ss) - lse) > ((text.m - M) - lse));
      if (force) {
        pick = stamp.i;
        lp = -__logf(stamp.s);
      } else {
        const bool from_text = text.m >= stamp.m;  // equal maxima: text ids are the lower ones
        pick = from_text ? text.i : stamp.i;
        lp = ((from_text ? text.m : stamp.m) - M) - lse;
      }
    }
    tok[blockIdx.x] = pick;
    if (logprob) logprob[blockIdx.x] = lp;
  }
}

// ---- beam search / sampling on the same filtered distribution ---------------------------------------------------------------------
// whisper.decoding: BeamSearchDecoder.update takes, per beam, the beam_size + 1 best log-probabilities of log_softmax(filtered logits);
// GreedyDecoder.update at temperature > 0 draws from Categorical(filtered logits / T) and scores the draw with the UN-tempered
// log_softmax.  "Filtered" = SuppressBlank / SuppressTokens (the additive masks) and, in timestamp mode, ApplyTimestampRules -- the
// predicate of pick_ts_kernel, restated once here as a struct so the two new kernels cannot drift apart (n_hist < 0: no timestamp rules).
struct TsRules {
  bool on, last_was_ts, pen_was_ts, first;
  int ts_begin, eot, no_ts, ts_lo, ts_hi;
  __device__ __forceinline__ bool dead(int c) const {
    if (!on) return false;
    const bool is_ts = c >= ts_begin;
    bool d = c == no_ts;
    if (last_was_ts) d = d || (pen_was_ts ? is_ts : c < eot);
    if (is_ts) d = d || c < ts_lo || c > ts_hi;
    else d = d || first;
    return d;
  }
};
// (all 256 threads call it; sh_last: 4 ints of LDS)
__device__ __forceinline__ TsRules ts_rules(const int64_t* h, int n_hist, int ts_begin, int eot, int no_ts, int max_initial_index, int* sh_last) {
  TsRules r;
  r.on = n_hist >= 0;
  r.ts_begin = ts_begin;
  r.eot = eot;
  r.no_ts = no_ts;
  r.last_was_ts = r.pen_was_ts = r.first = false;
  r.ts_lo = -1;
  r.ts_hi = 0x7fffffff;
  if (!r.on) return r;
  int last_pos = -1;
  for (int t = threadIdx.x; t < n_hist; t += 256)
    if (h[t] >= ts_begin) last_pos = t;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) last_pos = max(last_pos, __shfl_xor(last_pos, o, 64));
  if ((threadIdx.x & 63) == 0) sh_last[threadIdx.x >> 6] = last_pos;
  __syncthreads();
  last_pos = max(max(sh_last[0], sh_last[1]), max(sh_last[2], sh_last[3]));
  r.last_was_ts = n_hist >= 1 && h[n_hist - 1] >= ts_begin;
  r.pen_was_ts = n_hist < 2 || h[n_hist - 2] >= ts_begin;
  if (last_pos >= 0) r.ts_lo = (int)h[last_pos] + ((r.last_was_ts && !r.pen_was_ts) ? 0 : 1);
  r.first = n_hist == 0;
  if (r.first && max_initial_index >= 0) r.ts_hi = ts_begin + max_initial_index;
  return r;
}
// Filtered value of column c (-inf when masked)
__device__ __forceinline__ float filt(const float* lr, const float* mask, const float* mask2, const TsRules& R, int c) {
  float x = lr[c];
  if (mask) x += mask[c];
  if (mask2) x += mask2[c];
  return (R.dead(c) || !(x > -INFINITY)) ? -INFINITY : x;
}
// pass 1 of both kernels: (max, sum-exp, argmax) of the text and the timestamp part -> M = row maximum, lse = log sum exp(x - M) over the
// survivors, text_off = whether the "timestamp mass beats every text token" rule removes the text part.  Returns false for an empty row.
struct RowStat {
  float M, lse;
  bool text_off, any;
};
__device__ __forceinline__ RowStat row_stat(const float* lr, int V, const float* mask, const float* mask2, const TsRules& R, PickPart* sh) {
  PickPart text{-INFINITY, 0.f, 0x7fffffff}, stamp{-INFINITY, 0.f, 0x7fffffff};
  for (int c = threadIdx.x; c < V; c += 256) {
    const float x = filt(lr, mask, mask2, R, c);
    if (!(x > -INFINITY)) continue;
    PickPart& p = (R.on && c >= R.ts_begin) ? stamp : text;
    if (x > p.m) {
      p.s = p.s * __expf(p.m - x) + 1.f;
      p.m = x;
      p.i = c;
    } else {
      p.s += __expf(x - p.m);
    }
  }
  text = pick_block_reduce(text, sh);
  stamp = pick_block_reduce(stamp, sh);
  RowStat st;
  st.M = fmaxf(text.m, stamp.m);
  st.any = st.M > -INFINITY;
  st.text_off = false;
  st.lse = 0.f;
  if (st.any) {
    const float a = text.m > -INFINITY ? text.s * __expf(text.m - st.M) : 0.f;
    const float b = stamp.m > -INFINITY ? stamp.s * __expf(stamp.m - st.M) : 0.f;
    st.lse = __logf(a + b);
    st.text_off = stamp.m > -INFINITY && (text.m == -INFINITY || (__logf(b) - st.lse) > ((text.m - st.M) - st.lse));
    if (st.text_off) {
      st.M = stamp.m;
      st.lse = __logf(stamp.s);
    }
  }
  return st;
}

// K <= 16 best (log_softmax value, token) pairs of each row, descending, ties to the lower token id.  K + 1 passes over the row (it
// stays in L2): pass j finds the best survivor strictly after pick j-1 in (value desc, id asc) order.
__global__ __launch_bounds__(256) void topk_ts_kernel(const float* __restrict__ logits, long ld, int V, const float* __restrict__ mask,
                                                     const float* __restrict__ mask2, const int64_t* __restrict__ hist, long hist_ld,
                                                     int n_hist, int ts_begin, int eot, int no_ts, int max_initial_index, int K,
                                                     int64_t* __restrict__ tok, float* __restrict__ logprob) {
  __shared__ PickPart sh[4];
  __shared__ int sh_last[4];
  __shared__ float bv[4];
  __shared__ int bi[4];
  const float* lr = logits + (long)blockIdx.x * ld;
  const TsRules R = ts_rules(hist ? hist + (long)blockIdx.x * hist_ld : nullptr, n_hist, ts_begin, eot, no_ts, max_initial_index, sh_last);
  const RowStat st = row_stat(lr, V, mask, mask2, R, sh);
  float prev_v = INFINITY;
  int prev_i = -1;
  for (int j = 0; j < K; ++j) {
    float best = -INFINITY;
    int besti = 0x7fffffff;
    if (st.any) {
      for (int c = threadIdx.x; c < V; c += 256) {
        if (st.text_off && c < R.ts_begin) continue;
        const float x = filt(lr, mask, mask2, R, c);
        if (!(x > -INFINITY)) continue;
        if (!(x < prev_v || (x == prev_v && c > prev_i))) continue;  // not after the previous pick
        if (x > best) {  // (ascending c within a thread: the first of equal values wins)
          best = x;
          besti = c;
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(besti, o, 64);
      if (ov > best || (ov == best && oi < besti)) {
        best = ov;
        besti = oi;
      }
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
      bv[threadIdx.x >> 6] = best;
      bi[threadIdx.x >> 6] = besti;
    }
    __syncthreads();
    best = bv[0];
    besti = bi[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
      if (bv[w] > best || (bv[w] == best && bi[w] < besti)) {
        best = bv[w];
        besti = bi[w];
      }
    if (threadIdx.x == 0) {
      const bool ok = best > -INFINITY;
      tok[(long)blockIdx.x * K + j] = ok ? besti : 0;
      logprob[(long)blockIdx.x * K + j] = ok ? (best - st.M) - st.lse : -INFINITY;
    }
    prev_v = best;
    prev_i = besti;
  }
}

// One draw per row from softmax(filtered logits / temperature) by inverse CDF on the caller's uniform u[row] in [0, 1): every thread
// owns a contiguous range of columns, the block scans the range masses, the owner of u * total walks its range.  logprob = the draw's
// log_softmax at temperature 1 (what GreedyDecoder.update accumulates).
__global__ __launch_bounds__(256) void sample_ts_kernel(const float* __restrict__ logits, long ld, int V, const float* __restrict__ mask,
                                                       const float* __restrict__ mask2, const int64_t* __restrict__ hist, long hist_ld,
                                                       int n_hist, int ts_begin, int eot, int no_ts, int max_initial_index, float inv_temp,
                                                       const float* __restrict__ u, int64_t* __restrict__ tok, float* __restrict__ logprob) {
  __shared__ PickPart sh[4];
  __shared__ int sh_last[4];
  __shared__ float part[256];
  __shared__ int pick_sh;
  const float* lr = logits + (long)blockIdx.x * ld;
  const TsRules R = ts_rules(hist ? hist + (long)blockIdx.x * hist_ld : nullptr, n_hist, ts_begin, eot, no_ts, max_initial_index, sh_last);
  const RowStat st = row_stat(lr, V, mask, mask2, R, sh);
  const int per = (V + 255) / 256, c0 = threadIdx.x * per, c1 = min(V, c0 + per);
  float mass = 0.f;
  if (st.any)
    for (int c = c0; c < c1; ++c) {
      if (st.text_off && c < R.ts_begin) continue;
      const float x = filt(lr, mask, mask2, R, c);
      if (x > -INFINITY) mass += __expf((x - st.M) * inv_temp);
    }
  part[threadIdx.x] = mass;
  if (threadIdx.x == 0) pick_sh = -1;
  __syncthreads();
  if (threadIdx.x == 0 && st.any) {  // (256 additions: not worth a parallel scan)
    float total = 0.f;
    for (int t = 0; t < 256; ++t) total += part[t];
    const float target = u[blockIdx.x] * total;
    float acc = 0.f;
    int owner = -1, last_nz = -1;
    for (int t = 0; t < 256; ++t) {
      if (part[t] > 0.f) last_nz = t;
      if (owner < 0 && part[t] > 0.f && target < acc + part[t]) owner = t;
      if (owner < 0) acc += part[t];
    }
    if (owner < 0) {  // rounding pushed the target past the end: the last range with any mass
      owner = last_nz;
      acc = total - part[last_nz];
    }
    // walk the owner's range
    const int a0 = owner * per, a1 = min(V, a0 + per);
    int pick = -1, last = -1;
    float run = acc;
    for (int c = a0; c < a1 && pick < 0; ++c) {
      if (st.text_off && c < R.ts_begin) continue;
      const float x = filt(lr, mask, mask2, R, c);
      if (!(x > -INFINITY)) continue;
      last = c;
      run += __expf((x - st.M) * inv_temp);
      if (target < run) pick = c;
    }
    pick_sh = pick >= 0 ? pick : last;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int pick = pick_sh;
    tok[blockIdx.x] = pick >= 0 ? pick : 0;
    if (logprob) logprob[blockIdx.x] = pick >= 0 ? (filt(lr, mask, mask2, R, pick) - st.M) - st.lse : -INFINITY;
  }
}

}  // namespace

int launch_topk_tokens_ts(const float* logits, long ld, int V, long rows, const float* mask, const float* mask2, const int64_t* hist,
                          long hist_ld, int n_hist, int ts_begin, int eot, int no_ts, int max_initial_index, int K, int64_t* tok,
                          float* logprob, hipStream_t s) {
  OASR_REQUIRE(logits && tok && logprob && V > 0 && ld >= V && K >= 1 && K <= 16 && K <= V, "topk_tokens_ts: bad args (K=%d)", K);
  OASR_REQUIRE(n_hist < 0 || ((n_hist == 0 || (hist && hist_ld >= n_hist)) && 0 <= eot && eot < ts_begin && ts_begin < V), "topk_tokens_ts: bad history / token ids");
  if (rows <= 0) return OASR_OK;
  hipLaunchKernelGGL(topk_ts_kernel, dim3((unsigned)rows), dim3(256), 0, s, logits, ld, V, mask, mask2, hist, hist_ld, n_hist, ts_begin, eot,
                     no_ts, max_initial_index, K, tok, logprob);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}
int launch_sample_tokens_ts(const float* logits, long ld, int V, long rows, const float* mask, const float* mask2, const int64_t* hist,
                            long hist_ld, int n_hist, int ts_begin, int eot, int no_ts, int max_initial_index, float temperature,
                            const float* u, int64_t* tok, float* logprob, hipStream_t s) {
  OASR_REQUIRE(logits && tok && u && V > 0 && ld >= V && temperature > 0.f, "sample_tokens_ts: bad args");
  OASR_REQUIRE(n_hist < 0 || ((n_hist == 0 || (hist && hist_ld >= n_hist)) && 0 <= eot && eot < ts_begin && ts_begin < V), "sample_tokens_ts: bad history / token ids");
  if (rows <= 0) return OASR_OK;
  hipLaunchKernelGGL(sample_ts_kernel, dim3((unsigned)rows), dim3(256), 0, s, logits, ld, V, mask, mask2, hist, hist_ld, n_hist, ts_begin, eot,
                     no_ts, max_initial_index, 1.0f / temperature, u, tok, logprob);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}

int launch_pick_tokens(const float* logits, long ld, int V, long rows, const float* mask, const float* mask2, int64_t* tok, float* logprob,
                       hipStream_t s) {
  OASR_REQUIRE(logits && tok && V > 0 && ld >= V, "pick_tokens: bad args");
  if (rows <= 0) return OASR_OK;
  hipLaunchKernelGGL(pick_kernel, dim3((unsigned)rows), dim3(256), 0, s, logits, ld, V, mask, mask2, tok, logprob);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}

int launch_count_valid(const int64_t* targets, long rows, long ignore, int V, int32_t* n_valid_dev, hipStream_t s) {
  OASR_REQUIRE(targets && n_valid_dev && rows > 0, "count_valid: bad args");
  hipLaunchKernelGGL(count_valid_kernel, dim3(1), dim3(256), 0, s, targets, rows, ignore, V, n_valid_dev);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}

int launch_cross_entropy(bf16_t* logits, long ld, int V, const int64_t* targets, long rows, long ignore, float gscale,
                         const int32_t* n_valid_dev, float* row_loss, int write_grad, hipStream_t s) {
  OASR_REQUIRE(logits && targets && n_valid_dev && row_loss, "cross_entropy: null pointer");
  OASR_REQUIRE(ld % 8 == 0 && V <= ld && ld <= CE_CHUNKS * CE_THREADS * 8, "cross_entropy: ld=%ld V=%d unsupported", ld, V);
  if (rows <= 0) return OASR_OK;
  hipLaunchKernelGGL(ce_kernel, dim3((unsigned)rows), dim3(CE_THREADS), 0, s, logits, ld, V, targets, ignore, gscale, n_valid_dev,
                     row_loss, write_grad);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}

int launch_loss_reduce(const float* row_loss, long rows, const int32_t* n_valid_dev, float mul, float* loss_out, int accumulate,
                       hipStream_t s) {
  OASR_REQUIRE(row_loss && n_valid_dev && loss_out, "loss_reduce: null pointer");
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, s, row_loss, rows, n_valid_dev, mul, loss_out, accumulate);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}

int launch_pick_tokens_ts(const float* logits, long ld, int V, long rows, const float* mask, const float* mask2, const int64_t* hist,
                          long hist_ld, int n_hist, int ts_begin, int eot, int no_ts, int max_initial_index, int64_t* tok, float* logprob,
                          hipStream_t s) {
  OASR_REQUIRE(logits && tok && V > 0 && ld >= V && n_hist >= 0 && (n_hist == 0 || (hist && hist_ld >= n_hist)), "pick_tokens_ts: bad args");
  OASR_REQUIRE(0 <= eot && eot < ts_begin && ts_begin < V, "pick_tokens_ts: token ids out of range");
  if (rows <= 0) return OASR_OK;
  hipLaunchKernelGGL(pick_ts_kernel, dim3((unsigned)rows), dim3(256), 0, s, logits, ld, V, mask, mask2, hist, hist_ld, n_hist, ts_begin, eot,
                     no_ts, max_initial_index, tok, logprob);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}

// Log-mel front end on gfx950: PCM (int16 or f32) -> log-mel [B, 80, n_frames] f32.
//
// Replaces whisper.audio.log_mel_spectrogram as called by the reference at
// scripts/training/train_timestamps.py:196,214 and olmoasr/transcribe.py:148.
//   hann(400) STFT, hop 160, center/reflect pad, drop last frame -> |X|^2 [201, n/160]
//   -> 80x201 slaney mel filterbank -> log10(clamp 1e-10) -> max(x, clipmax-8) -> (x+4)/4
//
// Design: one workgroup = 64 frames of one clip.  The windowed DFT runs on the exact-f32 matrix pipe
// (v_mfma_f32_16x16x4_f32, k-ordered fmaf chain == f32 accuracy; bf16 MFMA cannot hold the 80 dB dynamic range the
// -8 floor needs) with the frame FOLDED about its centre first: the periodic hann window and cos are even, sin is odd
// under n -> 400 - n, so   Re X[k] = sum_{n=0..200} w[n] c[k][n] (x[n] + x[400-n])   (n = 0 and 200 counted once),
//                          Im X[k] = sum_{n=1..199} w[n] s[k][n] (x[n] - x[400-n]),
// i.e. two [64 x 208] x [208 x 208] products instead of one [64 x 400] x [400 x 416]: half the MFMA work of the
// dense DFT, same arithmetic.  The hann window is folded into the basis.  Power and the mel projection stay on chip (LDS), only
// PCM is read and log-mel written: algorithmic HBM bytes = 2*n (i16) + 4*80*n/160 per clip.
// The per-clip max (for the -8 floor) is an ordered-uint atomicMax; a second tiny pass applies it.
#include <stdlib.h>

#include <vector>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int NFFT = 400, HOP = 160, NFREQ = 201, NMEL = 80;
constexpr int NBH = 208;           // bins padded to 13 tiles of 16
constexpr int NB = 2 * NBH;        // cos | sin columns
constexpr int FT = 64;             // frames per workgroup
constexpr int NS = FT * HOP + (NFFT - HOP);  // 10480 samples per workgroup
constexpr int SLAB_LD = 432;       // basis slab row stride (floats): 432 % 32 == 16 -> conflict-free b32 reads
constexpr int P_LD = 212;          // power row stride (floats): 53 16-B slots -> conflict-free b128 reads
constexpr int KS = 16;             // DFT k per slab
constexpr int NK = 208;            // folded sample index n = 0..200, padded to 13 slabs (rows > 200 of the basis are zero)

__device__ __forceinline__ unsigned f2ord(float f) {
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__device__ __forceinline__ int lds_sample_addr(int s) { return s + 4 * (s / HOP); }

template <typename PCM>
__device__ __forceinline__ float load_pcm(const PCM* p, long i);
template <>
__device__ __forceinline__ float load_pcm<int16_t>(const int16_t* p, long i) { return (float)p[i] * (1.0f / 32768.0f); }
template <>
__device__ __forceinline__ float load_pcm<float>(const float* p, long i) { return p[i]; }

template <typename PCM>
__global__ __launch_bounds__(256) void logmel_main(const PCM* __restrict__ pcm, int n_samples, int n_frames,
                                                   const float* __restrict__ basis,    // [208][416] folded basis: w[n] cos | w[n] sin
                                                   const float* __restrict__ melfilt,  // [208][80]
                                                   float* __restrict__ out,            // [B][80][n_frames] log10 values
                                                   unsigned* __restrict__ clipmax) {   // [B] ordered-uint max
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // region 0: samples (10480 + 4*66 floats) -- later aliased by the per-wave power tiles
  // region 1: basis slab [16][432]
  constexpr int SAMP_FLOATS = NS + 4 * (NS / HOP + 1);
  constexpr int REG0 = (SAMP_FLOATS > 4 * 16 * P_LD ? SAMP_FLOATS : 4 * 16 * P_LD);
  float* samp = smem;
  float* slab = smem + ((REG0 + 3) & ~3);

  const int b = blockIdx.y;
  const int f0 = blockIdx.x * FT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const PCM* clip = pcm + (long)b * n_samples;

  // ---- stage the samples this frame block touches (reflect padding of torch.stft(center=True)) ----
  const long s_begin = (long)f0 * HOP - NFFT / 2;
  for (int s = tid; s < NS; s += 256) {
    long idx = s_begin + s;
    if (idx < 0) idx = -idx;
    if (idx >= n_samples) idx = 2L * (n_samples - 1) - idx;
    idx = idx < 0 ? 0 : (idx >= n_samples ? n_samples - 1 : idx);
    samp[lds_sample_addr(s)] = load_pcm<PCM>(clip, idx);
  }

  f32x4_t acc[26];
#pragma unroll
  for (int i = 0; i < 26; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int frame_l = wave * 16 + c;  // A-operand row of this lane
  // basis slabs go L2 -> registers -> LDS one slab ahead: the loads of slab kb+1 are in flight under slab kb's MFMAs
  constexpr int SLAB_V4 = KS * (NB / 4), PER_T = (SLAB_V4 + 255) / 256;  // 1664 16-byte pieces, 7 per thread
  f32x4_t pre[PER_T];
  auto slab_fetch = [&](int kb) {
#pragma unroll
    for (int j = 0; j < PER_T; ++j) {
      const int i = tid + j * 256;
      if (i < SLAB_V4) pre[j] = *(const f32x4_t*)(basis + (long)(kb * KS + i / (NB / 4)) * NB + (i % (NB / 4)) * 4);
    }
  };
  slab_fetch(0);
  for (int kb = 0; kb < NK / KS; ++kb) {
    __syncthreads();  // previous slab fully consumed (and, first time, samples staged)
#pragma unroll
    for (int j = 0; j < PER_T; ++j) {
      const int i = tid + j * 256;
      if (i < SLAB_V4) *(f32x4_t*)(slab + (i / (NB / 4)) * SLAB_LD + (i % (NB / 4)) * 4) = pre[j];
    }
    __syncthreads();
    if (kb + 1 < NK / KS) slab_fetch(kb + 1);
    // this lane's 4 folded samples n0..n0+3: x[n] and its mirror x[400-n] (two aligned 16-byte reads around 400-n0)
    const int n0 = kb * KS + 4 * g;
    const f32x4_t xa = *(const f32x4_t*)(samp + lds_sample_addr(frame_l * HOP + n0));
    const f32x4_t g1 = *(const f32x4_t*)(samp + lds_sample_addr(frame_l * HOP + NFFT - n0 - 4));
    const float x_m = samp[lds_sample_addr(frame_l * HOP + NFFT - n0)];  // x[400 - n0]: outside the frame when n0 == 0
    const bool once = n0 == 0 || n0 == NFFT / 2;                           // n = 0 and n = 200 are their own mirror
    const float xr[4] = {once ? 0.f : x_m, g1[3], g1[2], g1[1]};
    float ev[4], od[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ev[i] = xa[i] + xr[i];
      od[i] = xa[i] - xr[i];
    }
#pragma unroll
    for (int nt = 0; nt < 13; ++nt) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float bc = slab[(4 * g + i) * SLAB_LD + nt * 16 + c];
        const float bs = slab[(4 * g + i) * SLAB_LD + NBH + nt * 16 + c];
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ev[i], bc, acc[nt], 0, 0, 0);
        acc[13 + nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(od[i], bs, acc[13 + nt], 0, 0, 0);
      }
    }
  }
  __syncthreads();  // all waves done reading samples: region 0 becomes the power tiles

  // ---- power spectrum -> LDS [wave][16 frames][212] ------------------------------------------------
  float* pw = smem + wave * 16 * P_LD;
#pragma unroll
  for (int nt = 0; nt < 13; ++nt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float re = acc[nt][r], im = acc[13 + nt][r];
      pw[(g * 4 + r) * P_LD + nt * 16 + c] = re * re + im * im;
    }
  }
  __syncthreads();

  // ---- mel projection: [16 frames x 208] x [208 x 80] ---------------------------------------------
  f32x4_t macc[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) macc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < NBH; k0 += 16) {
    const f32x4_t a4 = *(const f32x4_t*)(pw + c * P_LD + k0 + 4 * g);
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float bv = melfilt[(k0 + 4 * g + i) * NMEL + nt * 16 + c];
        macc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i], bv, macc[nt], 0, 0, 0);
      }
    }
  }

  // ---- log10, per-clip max, store ------------------------------------------------------------------
  float vmax = -1e30f;
  const int t0 = f0 + wave * 16 + g * 4;
#pragma unroll
  for (int nt = 0; nt < 5; ++nt) {
    const int m = nt * 16 + c;
    float* dst = out + ((long)b * NMEL + m) * n_frames + t0;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = log10f(fmaxf(macc[nt][r], 1e-10f));
      if (t0 + r < n_frames) vmax = fmaxf(vmax, v[r]);
    }
    if (t0 + 3 < n_frames && ((n_frames & 3) == 0)) {
      *(f32x4_t*)dst = (f32x4_t){v[0], v[1], v[2], v[3]};
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (t0 + r < n_frames) dst[r] = v[r];
    }
  }
  vmax = wave_max(vmax);
  if (lane == 0 && vmax > -1e29f) atomicMax(clipmax + b, f2ord(vmax));
}

// ---- FFT front end (the default) -----------------------------------------------------------------------------------------------
// The dense DFT above spends 0.55 GFLOP per clip on the fp32 matrix pipe (3.5 us/clip at its peak, 6.7 measured) for a kernel
// whose algorithmic HBM traffic (1.92 MB/clip) needs 0.3 us.  A 400-point real FFT needs ~10 kFLOP per frame = 30 MFLOP per clip:
// a workgroup takes 32 frames of one clip through
//     pack      z[n] = w[2n] x[2n] + i w[2n+1] x[2n+1], n < 200   (a real 400-point transform as a complex 200-point one)
//     stage A   25 x radix-8 over n1 (n = 25 n1 + n2), twiddle W200^(n2 k1)                       -> A[k1][n2]
//     stage B    8 x 25-point DFT over n2 as 5 x 5 (n2 = 5a + b, k2 = c + 5e; twiddle W25^(b c))   -> Z[k1 + 8 k2]
//     unpack    X[k] = (Z[k] + conj Z[200-k]) / 2 - i W400^k (Z[k] - conj Z[200-k]) / 2, k <= 200 -> |X[k]|^2
//     mel       80 triangular filters as sparse rows (402 non-zeros of 16080), log10, per-clip max
// with every stage laid out as (32 frames) x (8 items per pass) over the 256 threads: the lane index is the frame, so all 32
// lanes of a half-wave run the same butterfly on different frames -- no divergence, table reads are broadcasts, and every LDS
// row stride (162 floats of samples, 201 complex / 201 floats per frame) is odd in its access width: conflict-free.  fp32 throughout
// (twiddles rounded once from double): |error| ~ 1e-7 of the frame's amplitude, same as the k-ordered fmaf chains of the MFMA form.
// (frames per workgroup: template parameter T of logmel_fft, 16 by default)
constexpr int XROW = 201;                     // per-frame row length (complex for Z, float for the power spectrum)
// constant tables, resident in LDS: W200 (200 complex) | hann[0..200] (+3 pad) | W400 (201 complex) | mel: (first bin, first weight) x 81 | weights
// (the power spectrum overwrites the Z rows in place -- each thread holds its 26 values across a barrier -- so samples + Z + tables
// fit 80 KiB: two workgroups per CU)
constexpr int XTAB_A = 400 + 204, XTAB_P = 402, XTAB_M_MAX = 604;
constexpr int XTAB_OFF_P = XTAB_A, XTAB_OFF_M = XTAB_A + XTAB_P, XTAB = XTAB_A + XTAB_P + XTAB_M_MAX;

struct cf {
  float x, y;
};
__device__ __forceinline__ cf cadd(cf a, cf b) { return cf{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cf csub(cf a, cf b) { return cf{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cf cmul(cf a, cf b) { return cf{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cf mni(cf a) { return cf{a.y, -a.x}; }  // -i a
__device__ __forceinline__ cf pli(cf a) { return cf{-a.y, a.x}; }  // +i a
__device__ __forceinline__ cf cscale(cf a, float s) { return cf{a.x * s, a.y * s}; }

__device__ __forceinline__ void fft8(cf (&v)[8]) {  // forward, natural order in and out
  constexpr float R2 = 0.70710678118654752f;
  const cf a0 = cadd(v[0], v[4]), a1 = csub(v[0], v[4]), a2 = cadd(v[2], v[6]), a3 = csub(v[2], v[6]);
  const cf a4 = cadd(v[1], v[5]), a5 = csub(v[1], v[5]), a6 = cadd(v[3], v[7]), a7 = csub(v[3], v[7]);
  const cf E0 = cadd(a0, a2), E2 = csub(a0, a2), E1 = cadd(a1, mni(a3)), E3 = cadd(a1, pli(a3));
  const cf O0 = cadd(a4, a6), O2 = csub(a4, a6), O1 = cadd(a5, mni(a7)), O3 = cadd(a5, pli(a7));
  const cf w1 = cf{(O1.x + O1.y) * R2, (O1.y - O1.x) * R2};
  const cf w2 = mni(O2);
  const cf w3 = cf{(O3.y - O3.x) * R2, (-O3.x - O3.y) * R2};
  v[0] = cadd(E0, O0);
  v[1] = cadd(E1, w1);
  v[2] = cadd(E2, w2);
  v[3] = cadd(E3, w3);
  v[4] = csub(E0, O0);
  v[5] = csub(E1, w1);
  v[6] = csub(E2, w2);
  v[7] = csub(E3, w3);
}
__device__ __forceinline__ void dft5(cf x0, cf x1, cf x2, cf x3, cf x4, cf (&y)[5]) {  // forward
  constexpr float C1 = 0.30901699437494742f, C2 = -0.80901699437494742f, S1 = 0.95105651629515357f, S2 = 0.58778525229247313f;
  const cf t1 = cadd(x1, x4), t2 = cadd(x2, x3), t3 = csub(x1, x4), t4 = csub(x2, x3);
  y[0] = cadd(x0, cadd(t1, t2));
  const cf m1 = cf{x0.x + C1 * t1.x + C2 * t2.x, x0.y + C1 * t1.y + C2 * t2.y};
  const cf m2 = cf{x0.x + C2 * t1.x + C1 * t2.x, x0.y + C2 * t1.y + C1 * t2.y};
  const cf s1 = cf{S1 * t3.x + S2 * t4.x, S1 * t3.y + S2 * t4.y};
  const cf s2 = cf{S2 * t3.x - S1 * t4.x, S2 * t3.y - S1 * t4.y};
  y[1] = cadd(m1, mni(s1));
  y[2] = cadd(m2, mni(s2));
  y[3] = cadd(m2, pli(s2));
  y[4] = cadd(m1, pli(s1));
}
__device__ __forceinline__ int xsamp_addr(int s) { return s + 2 * (s / HOP); }  // 162 floats between frames: 81 8-byte slots (odd)

// T = frames per workgroup (32: lane = frame over a half wave, 8 item groups, two workgroups per CU; 16: 16 item groups, 43 KiB of LDS,
// three workgroups per CU -- half the serial work per thread in the sample / unpack / mel stages and 12 instead of 8 waves per CU to hide the
// LDS round trips this kernel is bound by, profiles/r03_logmel_fft.txt)
template <typename PCM, int T>
__global__ __launch_bounds__(256, T == 32 ? 2 : 3) void logmel_fft(const PCM* __restrict__ pcm, int n_samples, int n_frames,
                                                     const float* __restrict__ tab,  // [XTAB]: the three stage tables
                                                     int mel_words,                  // words of the mel table (162 + non-zeros)
                                                     int max_span,                   // taps of the widest filter
                                                     float* __restrict__ out, unsigned* __restrict__ clipmax) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NG = 256 / T;                    // item groups: thread (f = tid % T, j0 = tid / T)
  constexpr int TNS = T * HOP + (NFFT - HOP);    // samples of T frames
  constexpr int TR0 = TNS + 2 * (TNS / HOP + 1);
  float* samp = smem;                       // region 0: samples
  cf* cb = (cf*)(smem + ((TR0 + 3) & ~3));  // [T][XROW] complex: A, then Z, then (as floats, same row stride) the power spectrum
  float* pw = (float*)cb;
  float* tw = smem + ((TR0 + 3) & ~3) + 2 * T * XROW;
  const cf* tw200 = (const cf*)tw;
  const float* win = tw + 400;
  const cf* tw400 = (const cf*)(tw + XTAB_OFF_P);
  // 1-D grid, XCD-contiguous: neighbouring frame blocks of a clip run on the same XCD at the same time, so the 128-byte row
  // segments they write (row pitch 12000 B: never line aligned) meet in ONE L2 and leave it as whole lines, and the 240 samples
  // two neighbours share are fetched once
  const int nblk = (n_frames + T - 1) / T;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int b = bid / nblk, f0 = (bid - b * nblk) * T, tid = threadIdx.x;
  const int f = tid & (T - 1), j0 = tid / T;  // lane = frame, NG items per pass
  const PCM* clip = pcm + (long)b * n_samples;

  for (int i = tid; i < (XTAB_OFF_M + mel_words + 3) / 4; i += 256) ((f32x4_t*)tw)[i] = ((const f32x4_t*)tab)[i];  // (XTAB floats allocated)
  // ---- samples of these 32 frames (reflect padding of torch.stft(center=True) at the clip's ends)
  const long s_begin = (long)f0 * HOP - NFFT / 2;
  const bool interior = s_begin >= 0 && s_begin + TNS <= n_samples && ((size_t)(clip + s_begin) & 15) == 0;
  if (interior) {  // (wave-uniform) 16-byte loads; a vector never straddles a hop boundary (160 % 8 == 0), so its floats stay adjacent
    constexpr int PER = 16 / (int)sizeof(PCM);  // 8 int16 / 4 float samples per load
    for (int v = tid; v < TNS / PER; v += 256) {
      const u32x4_t raw = *(const u32x4_t*)(clip + s_begin + (long)v * PER);
      float* dst = samp + xsamp_addr(v * PER);
      if (sizeof(PCM) == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int w = (int)raw[i];
          *(float2*)(dst + 2 * i) = float2{(float)(short)(w & 0xffff) * (1.0f / 32768.0f), (float)(w >> 16) * (1.0f / 32768.0f)};
        }
      } else {
        *(float2*)dst = float2{__uint_as_float(raw[0]), __uint_as_float(raw[1])};
        *(float2*)(dst + 2) = float2{__uint_as_float(raw[2]), __uint_as_float(raw[3])};
      }
    }
  } else {
    for (int s = tid; s < TNS; s += 256) {
      long idx = s_begin + s;
      if (idx < 0) idx = -idx;
      if (idx >= n_samples) idx = 2L * (n_samples - 1) - idx;
      idx = idx < 0 ? 0 : (idx >= n_samples ? n_samples - 1 : idx);
      samp[xsamp_addr(s)] = load_pcm<PCM>(clip, idx);
    }
  }
  __syncthreads();

  // ---- stage A
#pragma unroll 1
  for (int it = 0; it < (25 + NG - 1) / NG; ++it) {
    const int n2 = it * NG + j0;
    if (n2 < 25) {
      cf v[8];
#pragma unroll
      for (int n1 = 0; n1 < 8; ++n1) {
        const int n = 25 * n1 + n2, i0 = 2 * n;
        const float2 x = *(const float2*)(samp + xsamp_addr(f * HOP + i0));
        const float w0 = win[i0 <= 200 ? i0 : NFFT - i0], w1 = win[i0 + 1 <= 200 ? i0 + 1 : NFFT - i0 - 1];
        v[n1] = cf{x.x * w0, x.y * w1};
      }
      fft8(v);
      cf* dst = cb + f * XROW + n2;
      dst[0] = v[0];
#pragma unroll
      for (int k1 = 1; k1 < 8; ++k1) dst[k1 * 25] = cmul(v[k1], tw200[n2 * k1]);  // (n2 k1 <= 168)
    }
  }
  __syncthreads();

  // ---- stage B: k1 = j0 (the groups past 7 idle here when NG = 16)
  {
    const int k1 = j0 & 7;
    const bool act = j0 < 8;
    cf a[25];
    const cf* src = cb + f * XROW + k1 * 25;
    if (act) {
#pragma unroll
      for (int i = 0; i < 25; ++i) a[i] = src[i];
    }
    __syncthreads();  // every thread holds its inputs: the rows can be overwritten in natural order
    if (act) {
      cf bm[5][5];  // [b][c]
#pragma unroll
      for (int bb = 0; bb < 5; ++bb) {
        cf y[5];
        dft5(a[bb], a[5 + bb], a[10 + bb], a[15 + bb], a[20 + bb], y);
        bm[bb][0] = y[0];
#pragma unroll
        for (int c = 1; c < 5; ++c) bm[bb][c] = bb == 0 ? y[c] : cmul(y[c], tw200[8 * bb * c]);  // W25^(b c) = W200^(8 b c)
      }
      cf* dst = cb + f * XROW + k1;
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        cf y[5];
        dft5(bm[0][c], bm[1][c], bm[2][c], bm[3][c], bm[4][c], y);
#pragma unroll
        for (int e = 0; e < 5; ++e) dst[8 * (c + 5 * e)] = y[e];
      }
    }
  }
  __syncthreads();

  // ---- unpack + power: every thread first computes its 26 bins of frame f into registers, then (behind a barrier: all reads of
  // the Z rows are done) writes them over the row, as floats
  {
    constexpr int NU = (201 + NG - 1) / NG;
    float pv[NU];
#pragma unroll
    for (int it = 0; it < NU; ++it) {
      const int k = it * NG + j0;
      pv[it] = 0.f;
      if (k <= 200) {
        const cf zk = cb[f * XROW + (k == 200 ? 0 : k)];
        cf zm = cb[f * XROW + (k == 0 ? 0 : 200 - k)];
        zm.y = -zm.y;
        const cf e = cscale(cadd(zk, zm), 0.5f), o = cscale(mni(csub(zk, zm)), 0.5f);
        const cf x = cadd(e, cmul(tw400[k], o));
        pv[it] = x.x * x.x + x.y * x.y;
      }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NU; ++it) {
      const int k = it * NG + j0;
      if (k <= 200) pw[f * (2 * XROW) + k] = pv[it];
    }
  }
  __syncthreads();

  // ---- mel filters (sparse rows), log10, per-clip max, store
  const int* mel_idx = (const int*)(tw + XTAB_OFF_M);  // [81][2]: first bin, first weight (row 80 carries the total)
  const float* mel_val = tw + XTAB_OFF_M + 162;
  float vmax = -1e30f;
  const int t = f0 + f;
  {
    // this thread's 10 filters (m = 8 it + j0) advance together, one tap per step: 10 independent LDS-read + fma chains in flight
    // instead of one serial chain per filter (each step is two LDS round trips; the filters have 2..14 taps)
    constexpr int NF = NMEL / NG, NH = (NF + 1) / 2;  // filters per thread (m = NG it + j0); they advance in pairs (a, a + NH)
    int lo[NF], p0[NF], cnt[NF];
    float acc[NF];
#pragma unroll
    for (int it = 0; it < NF; ++it) {
      const int m = it * NG + j0;
      lo[it] = mel_idx[2 * m];
      p0[it] = mel_idx[2 * m + 1];
      cnt[it] = mel_idx[2 * m + 3] - p0[it];
      acc[it] = 0.f;
    }
    const float* prow = pw + f * (2 * XROW);
    // filters a and a + NH advance together (two independent chains); the step count of a pair is the widest filter any lane of
    // the wave holds for it (filter width grows with the index)
#pragma unroll
    for (int a = 0; a < NH; ++a) {
      const int a2 = a + NH < NF ? a + NH : a;  // (odd NF: the last one runs alone)
      int n = cnt[a] > cnt[a2] ? cnt[a] : cnt[a2];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) n = max(n, __shfl_xor(n, o, 64));
      n = __builtin_amdgcn_readfirstlane(n);
      for (int i = 0; i < n; i += 4) {  // 16 LDS reads in flight per trip (reads past a filter's end stay inside the row / the table)
        float x0[4], w0[4], x1[4], w1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          x0[u] = prow[lo[a] + i + u];
          w0[u] = mel_val[p0[a] + i + u];
          x1[u] = prow[lo[a2] + i + u];
          w1[u] = mel_val[p0[a2] + i + u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc[a] = (i + u < cnt[a]) ? fmaf(x0[u], w0[u], acc[a]) : acc[a];
          if (a2 != a) acc[a2] = (i + u < cnt[a2]) ? fmaf(x1[u], w1[u], acc[a2]) : acc[a2];
        }
      }
    }
    if (t < n_frames) {
#pragma unroll
      for (int it = 0; it < NF; ++it) {
        const float v = log10f(fmaxf(acc[it], 1e-10f));
        out[((long)b * NMEL + it * NG + j0) * n_frames + t] = v;
        vmax = fmaxf(vmax, v);
      }
    }
  }
  // one ordered-uint atomicMax per workgroup (the tables are dead: their first words carry the four wave maxima)
  vmax = wave_max(vmax);
  __syncthreads();
  if ((tid & 63) == 0) tw[tid >> 6] = vmax;
  __syncthreads();
  if (tid == 0) {
    const float m4 = fmaxf(fmaxf(tw[0], tw[1]), fmaxf(tw[2], tw[3]));
    if (m4 > -1e29f) atomicMax(clipmax + b, f2ord(m4));
  }
}

#include "logmel_quad.h"

__global__ __launch_bounds__(256) void logmel_finalize(float* __restrict__ mel, const unsigned* __restrict__ clipmax,
                                                       long per_clip) {
  const int b = blockIdx.y;
  const float floor_v = ord2f(clipmax[b]) - 8.0f;
  float* p = mel + (long)b * per_clip;
  const long n4 = per_clip >> 2;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4_t v = ((f32x4_t*)p)[i];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (fmaxf(v[r], floor_v) + 4.0f) * 0.25f;
    ((f32x4_t*)p)[i] = v;
  }
  if (blockIdx.x == 0)
    for (long i = (n4 << 2) + threadIdx.x; i < per_clip; i += 256) p[i] = (fmaxf(p[i], floor_v) + 4.0f) * 0.25f;
}

// ---- host-built constant tables (double precision, rounded once to f32) -----------------------------
struct MelTables {
  float* basis = nullptr;    // [208][416] (folded)
  float* melfilt = nullptr;  // [208][80]
  float* fft_tab = nullptr;  // [XTAB] stage tables of the FFT kernel (twiddles, window, sparse mel filters)
  float* quad_tab = nullptr; // [QTAB] per-lane tables of the quad-lane kernel (logmel_quad.h)
  int mel_words = 0;         // used words of its mel table
  int mel_span = 0;          // taps of the widest filter
  int device = -1;
};
MelTables g_tables[16];

double hz_to_mel(double f) {
  const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
  return f >= min_log_hz ? min_log_mel + log(f / min_log_hz) / logstep : f / f_sp;
}
double mel_to_hz(double m) {
  const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
  return m >= min_log_mel ? min_log_hz * exp(logstep * (m - min_log_mel)) : f_sp * m;
}

}  // namespace

// Slaney-scale / slaney-normalised filterbank == librosa.filters.mel(sr=16000, n_fft=400, n_mels=80),
// the content of whisper's assets/mel_filters.npz.  Exported so the host mirror can hand it to tests.
extern "C" int oasr_mel_filterbank(float* out /*[80][201]*/) {
  const int sr = 16000;
  double pts[NMEL + 2];
  const double m0 = hz_to_mel(0.0), m1 = hz_to_mel(sr / 2.0);
  for (int i = 0; i < NMEL + 2; ++i) pts[i] = mel_to_hz(m0 + (m1 - m0) * i / (NMEL + 1));
  for (int m = 0; m < NMEL; ++m) {
    const double enorm = 2.0 / (pts[m + 2] - pts[m]);
    for (int f = 0; f < NFREQ; ++f) {
      const double hz = (sr / 2.0) * f / (NFREQ - 1);
      const double lower = (hz - pts[m]) / (pts[m + 1] - pts[m]);
      const double upper = (pts[m + 2] - hz) / (pts[m + 2] - pts[m + 1]);
      double w = lower < upper ? lower : upper;
      if (w < 0) w = 0;
      out[m * NFREQ + f] = (float)(w * enorm);
    }
  }
  return OASR_OK;
}

static int ensure_tables(int device, MelTables** t_out) {
  OASR_REQUIRE(device >= 0 && device < 16, "device index %d out of range", device);
  MelTables& t = g_tables[device];
  if (t.basis == nullptr) {
    float* hb = (float*)calloc((size_t)NK * NB, sizeof(float));
    float* hf = (float*)calloc((size_t)NBH * NMEL, sizeof(float));
    float* fb = (float*)malloc(sizeof(float) * NMEL * NFREQ);
    if (!hb || !hf || !fb) {
      oasr_set_error("host alloc failed");
      return OASR_EHIP;
    }
    const double PI = 3.14159265358979323846;
    for (int j = 0; j <= NFFT / 2; ++j) {  // folded rows; rows 201..207 stay zero
      const double w = 0.5 - 0.5 * cos(2.0 * PI * j / NFFT);  // torch.hann_window(400), periodic: w[400 - j] == w[j]
      for (int f = 0; f < NFREQ; ++f) {
        const int ph = (int)(((long)j * f) % NFFT);  // exact argument reduction
        const double ang = 2.0 * PI * ph / NFFT;
        hb[(size_t)j * NB + f] = (float)(w * cos(ang));
        if (j >= 1 && j < NFFT / 2) hb[(size_t)j * NB + NBH + f] = (float)(w * sin(ang));
      }
    }
    oasr_mel_filterbank(fb);
    for (int m = 0; m < NMEL; ++m)
      for (int f = 0; f < NFREQ; ++f) hf[(size_t)f * NMEL + m] = fb[m * NFREQ + f];
    {  // FFT kernel tables
      std::vector<float> ft((size_t)XTAB, 0.f);
      for (int j = 0; j < 200; ++j) {
        ft[(size_t)2 * j] = (float)cos(2.0 * PI * j / 200.0);
        ft[(size_t)2 * j + 1] = (float)(-sin(2.0 * PI * j / 200.0));
      }
      for (int j = 0; j <= 200; ++j) ft[(size_t)400 + j] = (float)(0.5 - 0.5 * cos(2.0 * PI * j / NFFT));
      for (int k = 0; k <= 200; ++k) {
        ft[(size_t)XTAB_OFF_P + 2 * k] = (float)cos(2.0 * PI * k / 400.0);
        ft[(size_t)XTAB_OFF_P + 2 * k + 1] = (float)(-sin(2.0 * PI * k / 400.0));
      }
      std::vector<int> idx((size_t)2 * (NMEL + 1), 0);
      std::vector<float> val;
      oasr_mel_filterbank(fb);
      for (int m = 0; m < NMEL; ++m) {
        int lo = NFREQ, hi = -1;
        for (int f = 0; f < NFREQ; ++f)
          if (fb[m * NFREQ + f] != 0.f) {
            lo = f < lo ? f : lo;
            hi = f;
          }
        if (hi < lo) lo = hi = 0;
        idx[(size_t)2 * m] = lo;
        idx[(size_t)2 * m + 1] = (int)val.size();
        t.mel_span = hi - lo + 1 > t.mel_span ? hi - lo + 1 : t.mel_span;
        for (int f = lo; f <= hi; ++f) val.push_back(fb[m * NFREQ + f]);
      }
      idx[(size_t)2 * NMEL] = 0;
      idx[(size_t)2 * NMEL + 1] = (int)val.size();
      if (idx.size() + val.size() > (size_t)XTAB_M_MAX) {
        oasr_set_error("log-mel: sparse filterbank (%zu words) exceeds its LDS slot", idx.size() + val.size());
        return OASR_ESTATE;
      }
      memcpy(&ft[(size_t)XTAB_OFF_M], idx.data(), idx.size() * sizeof(int));
      memcpy(&ft[(size_t)XTAB_OFF_M + idx.size()], val.data(), val.size() * sizeof(float));
      t.mel_words = (int)(idx.size() + val.size());
      OASR_CHECK_HIP(hipMalloc((void**)&t.fft_tab, sizeof(float) * XTAB));
      OASR_CHECK_HIP(hipMemcpy(t.fft_tab, ft.data(), sizeof(float) * XTAB, hipMemcpyHostToDevice));
    }
    {  // quad-lane kernel tables: lane l of a quad holds k1 = {0, 2, 1, 3}[l] (bit-reversed) after the cross-lane radix-4
      const int K1[4] = {0, 2, 1, 3};
      std::vector<float> qt((size_t)QTAB, 0.f);
      for (int s = 0; s < NFFT; ++s) qt[(size_t)QTAB_WIN + s] = (float)(0.5 - 0.5 * cos(2.0 * PI * s / NFFT));
      for (int m = 0; m < 50; ++m)
        for (int l = 0; l < 4; ++l) {
          const int e200 = (m * K1[l]) % 200, e400 = K1[l] + 4 * m;  // W200^(m k1); W400^(k1 + 4 k2) with k2 = m
          const double sgn = (l == 1 || l == 2) ? -1.0 : 1.0;       // sg1 sg2 of lane l: the sign its one-instruction butterflies leave (logmel_quad.h)
          qt[(size_t)QTAB_TW200 + 2 * (4 * m + l)] = (float)(sgn * cos(2.0 * PI * e200 / 200.0));
          qt[(size_t)QTAB_TW200 + 2 * (4 * m + l) + 1] = (float)(-sgn * sin(2.0 * PI * e200 / 200.0));
          qt[(size_t)QTAB_TW400 + 2 * (4 * m + l)] = (float)cos(2.0 * PI * e400 / 400.0);
          qt[(size_t)QTAB_TW400 + 2 * (4 * m + l) + 1] = (float)(-sin(2.0 * PI * e400 / 400.0));
        }
      oasr_mel_filterbank(fb);
      for (int m = 0; m < NMEL; ++m) {
        int lo = NFREQ, hi = -1;
        for (int f = 0; f < NFREQ; ++f)
          if (fb[m * NFREQ + f] != 0.f) {
            lo = f < lo ? f : lo;
            hi = f;
          }
        // the compile-time slot structure (logmel_quad_tables.h, generated from the same formulas) must describe THIS filterbank
        if (hi < lo || lo / 4 != QMEL_K2LO[m] || hi / 4 - lo / 4 + 1 != QMEL_NSLOT[m] || hi > 199) {
          oasr_set_error("log-mel: filter %d spans bins %d..%d, logmel_quad_tables.h says k2 %d + %d: regenerate (scripts/gen_logmel_quad_tables.py)",
                         m, lo, hi, QMEL_K2LO[m], QMEL_NSLOT[m]);
          return OASR_ESTATE;
        }
        for (int sl = 0; sl < QMEL_NSLOT[m]; ++sl)
          for (int l = 0; l < 4; ++l) {
            const int k = K1[l] + 4 * (QMEL_K2LO[m] + sl);
            qt[(size_t)QTAB_MEL + 4 * (QMEL_OFF[m] + sl) + l] = k <= 199 ? 0.25f * fb[m * NFREQ + k] : 0.f;  // (P holds 4 |X|^2)
          }
      }
      OASR_CHECK_HIP(hipMalloc((void**)&t.quad_tab, sizeof(float) * QTAB));
      OASR_CHECK_HIP(hipMemcpy(t.quad_tab, qt.data(), sizeof(float) * QTAB, hipMemcpyHostToDevice));
    }
    OASR_CHECK_HIP(hipMalloc((void**)&t.basis, sizeof(float) * NK * NB));
    OASR_CHECK_HIP(hipMalloc((void**)&t.melfilt, sizeof(float) * NBH * NMEL));
    OASR_CHECK_HIP(hipMemcpy(t.basis, hb, sizeof(float) * NK * NB, hipMemcpyHostToDevice));
    OASR_CHECK_HIP(hipMemcpy(t.melfilt, hf, sizeof(float) * NBH * NMEL, hipMemcpyHostToDevice));
    free(hb);
    free(hf);
    free(fb);
    t.device = device;
  }
  *t_out = &t;
  return OASR_OK;
}

extern "C" size_t oasr_log_mel_workspace_bytes(int B) { return (size_t)((B * 4 + 255) / 256) * 256; }

__global__ __launch_bounds__(256) void logmel_clipmax_kernel(const unsigned* __restrict__ clipmax, float* __restrict__ out, int B) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b < B) out[b] = ord2f(clipmax[b]);
}

static int log_mel_impl(const void* pcm, int pcm_dtype, int B, int n_samples, float* mel, void* workspace, hipStream_t stream,
                        float* clip_max_out);
extern "C" int oasr_log_mel(const void* pcm, int pcm_dtype, int B, int n_samples, float* mel, void* workspace,
                            hipStream_t stream) {
  return log_mel_impl(pcm, pcm_dtype, B, n_samples, mel, workspace, stream, nullptr);
}
// The same front end WITHOUT its last pass: mel_raw = log10(max(mel power, 1e-10)) and clip_max[b] = the clip's maximum of it (device
// f32 [B]).  whisper's last two lines -- log_spec = max(log_spec, log_spec.max() - 8); (log_spec + 4) / 4 -- need the clip maximum, i.e.
// a second pass over the tensor; a consumer that reads the tensor anyway applies them on the fly instead
// (oasr_train_fwd_bwd_span's mel_clip_max: the encoder's time-major transpose), which halves this front end's HBM traffic.
extern "C" int oasr_log_mel_raw(const void* pcm, int pcm_dtype, int B, int n_samples, float* mel_raw, float* clip_max, void* workspace,
                                hipStream_t stream) {
  OASR_REQUIRE(clip_max, "oasr_log_mel_raw: null clip_max");
  return log_mel_impl(pcm, pcm_dtype, B, n_samples, mel_raw, workspace, stream, clip_max);
}
static int log_mel_impl(const void* pcm, int pcm_dtype, int B, int n_samples, float* mel, void* workspace, hipStream_t stream,
                        float* clip_max_out) {
  OASR_REQUIRE(pcm && mel && workspace, "null pointer");
  OASR_REQUIRE(pcm_dtype == 0 || pcm_dtype == 1, "pcm_dtype must be 0 (f32) or 1 (i16)");
  OASR_REQUIRE(B > 0 && n_samples > NFFT / 2, "need B > 0 and n_samples > 200 (reflect padding), got %d, %d", B, n_samples);
  const int n_frames = n_samples / HOP;
  OASR_REQUIRE(n_frames > 0, "n_samples %d shorter than one hop", n_samples);
  int device = 0;
  OASR_CHECK_HIP(hipGetDevice(&device));
  MelTables* t = nullptr;
  int rc = ensure_tables(device, &t);
  if (rc) return rc;
  unsigned* clipmax = (unsigned*)workspace;
  OASR_CHECK_HIP(hipMemsetAsync(clipmax, 0, sizeof(unsigned) * B, stream));
  static const bool use_mfma_dft = [] {
    const char* e = oasr_experiment_env("OASR_LOGMEL");  // "mfma": the dense folded DFT on the fp32 matrix pipe (round 1-2 kernel; A/B and cross-check)
    return e && strcmp(e, "mfma") == 0;
  }();
  const long per_clip = (long)NMEL * n_frames;
  const dim3 g2((unsigned)((per_clip / 4 + 255) / 256 > 64 ? 64 : (per_clip / 4 + 255) / 256 + 1), B);
  static const bool use_lds_fft = [] {  // "fft" / "fft32": the round-3/5 LDS FFT kernel (A/B and cross-check); default: the quad-lane register kernel
    const char* e = oasr_experiment_env("OASR_LOGMEL");
    return e && (strcmp(e, "fft") == 0 || strcmp(e, "fft32") == 0);
  }();
  if (!use_mfma_dft && !use_lds_fft) {
    const size_t qlds = pcm_dtype == 1 ? quad_lds_bytes<int16_t>() : quad_lds_bytes<float>();
    const dim3 qgrid((unsigned)(cdiv(n_frames, QT) * B));
    static LdsAttrOnce qattr16, qattr32;
    if (pcm_dtype == 1) {
      { const int rc_ = ensure_dynamic_lds(qattr16, (const void*)logmel_quad<int16_t>, (int)qlds); if (rc_) return rc_; }
      hipLaunchKernelGGL(logmel_quad<int16_t>, qgrid, dim3(256), qlds, stream, (const int16_t*)pcm, n_samples, n_frames, t->quad_tab, mel, clipmax);
    } else {
      { const int rc_ = ensure_dynamic_lds(qattr32, (const void*)logmel_quad<float>, (int)qlds); if (rc_) return rc_; }
      hipLaunchKernelGGL(logmel_quad<float>, qgrid, dim3(256), qlds, stream, (const float*)pcm, n_samples, n_frames, t->quad_tab, mel, clipmax);
    }
    OASR_LAUNCH_CHECK();
    if (clip_max_out) hipLaunchKernelGGL(logmel_clipmax_kernel, dim3(cdiv(B, 256)), dim3(256), 0, stream, clipmax, clip_max_out, B);
    else hipLaunchKernelGGL(logmel_finalize, g2, dim3(256), 0, stream, mel, clipmax, per_clip);
    OASR_LAUNCH_CHECK();
    return OASR_OK;
  }
  if (!use_mfma_dft) {
    static const int frames_per_wg = [] {  // A/B: OASR_LOGMEL=fft32 is the round-3/4 geometry
      const char* e = oasr_experiment_env("OASR_LOGMEL");
      return (e && strcmp(e, "fft32") == 0) ? 32 : 16;
    }();
    auto lds_floats = [](int T) { const int tns = T * HOP + (NFFT - HOP); const int r0 = tns + 2 * (tns / HOP + 1); return ((r0 + 3) & ~3) + 2 * T * XROW + XTAB; };
    const size_t xlds = sizeof(float) * lds_floats(frames_per_wg);
    const dim3 xgrid((unsigned)(cdiv(n_frames, frames_per_wg) * B));
    static LdsAttrOnce attr16_32, attr32_32, attr16_16, attr32_16;
#define OASR_LOGMEL_LAUNCH(PCM_T, TT, ATTR)                                                                                       \
  do {                                                                                                                          \
    const int rc_ = ensure_dynamic_lds(ATTR, (const void*)logmel_fft<PCM_T, TT>, (int)xlds);                                    \
    if (rc_) return rc_;                                                                                                        \
    hipLaunchKernelGGL((logmel_fft<PCM_T, TT>), xgrid, dim3(256), xlds, stream, (const PCM_T*)pcm, n_samples, n_frames, t->fft_tab, \
                       t->mel_words, t->mel_span, mel, clipmax);                                                                \
  } while (0)
    if (pcm_dtype == 1) {
      if (frames_per_wg == 32) OASR_LOGMEL_LAUNCH(int16_t, 32, attr16_32);
      else OASR_LOGMEL_LAUNCH(int16_t, 16, attr16_16);
    } else {
      if (frames_per_wg == 32) OASR_LOGMEL_LAUNCH(float, 32, attr32_32);
      else OASR_LOGMEL_LAUNCH(float, 16, attr32_16);
    }
#undef OASR_LOGMEL_LAUNCH
    OASR_LAUNCH_CHECK();
    if (clip_max_out) hipLaunchKernelGGL(logmel_clipmax_kernel, dim3(cdiv(B, 256)), dim3(256), 0, stream, clipmax, clip_max_out, B);
    else hipLaunchKernelGGL(logmel_finalize, g2, dim3(256), 0, stream, mel, clipmax, per_clip);
    OASR_LAUNCH_CHECK();
    return OASR_OK;
  }
  constexpr int SAMP_FLOATS = NS + 4 * (NS / HOP + 1);
  constexpr int REG0 = (SAMP_FLOATS > 4 * 16 * P_LD ? SAMP_FLOATS : 4 * 16 * P_LD);
  const size_t lds = sizeof(float) * (((REG0 + 3) & ~3) + KS * SLAB_LD);
  dim3 grid(cdiv(n_frames, FT), B);
  if (pcm_dtype == 1) {
    static LdsAttrOnce attr;
    { const int rc_ = ensure_dynamic_lds(attr, (const void*)logmel_main<int16_t>, (int)lds); if (rc_) return rc_; }
    hipLaunchKernelGGL(logmel_main<int16_t>, grid, dim3(256), lds, stream, (const int16_t*)pcm, n_samples, n_frames,
                       t->basis, t->melfilt, mel, clipmax);
  } else {
    static LdsAttrOnce attr;
    { const int rc_ = ensure_dynamic_lds(attr, (const void*)logmel_main<float>, (int)lds); if (rc_) return rc_; }
    hipLaunchKernelGGL(logmel_main<float>, grid, dim3(256), lds, stream, (const float*)pcm, n_samples, n_frames, t->basis,
                       t->melfilt, mel, clipmax);
  }
  OASR_LAUNCH_CHECK();
  if (clip_max_out) hipLaunchKernelGGL(logmel_clipmax_kernel, dim3(cdiv(B, 256)), dim3(256), 0, stream, clipmax, clip_max_out, B);
  else hipLaunchKernelGGL(logmel_finalize, g2, dim3(256), 0, stream, mel, clipmax, per_clip);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}

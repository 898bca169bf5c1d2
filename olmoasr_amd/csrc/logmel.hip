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

extern "C" int oasr_log_mel(const void* pcm, int pcm_dtype, int B, int n_samples, float* mel, void* workspace,
                            hipStream_t stream) {
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
  constexpr int SAMP_FLOATS = NS + 4 * (NS / HOP + 1);
  constexpr int REG0 = (SAMP_FLOATS > 4 * 16 * P_LD ? SAMP_FLOATS : 4 * 16 * P_LD);
  const size_t lds = sizeof(float) * (((REG0 + 3) & ~3) + KS * SLAB_LD);
  dim3 grid(cdiv(n_frames, FT), B);
  if (pcm_dtype == 1) {
    static bool attr = false;
    if (!attr) {
      OASR_CHECK_HIP(hipFuncSetAttribute((const void*)logmel_main<int16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr = true;
    }
    hipLaunchKernelGGL(logmel_main<int16_t>, grid, dim3(256), lds, stream, (const int16_t*)pcm, n_samples, n_frames,
                       t->basis, t->melfilt, mel, clipmax);
  } else {
    static bool attr = false;
    if (!attr) {
      OASR_CHECK_HIP(hipFuncSetAttribute((const void*)logmel_main<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr = true;
    }
    hipLaunchKernelGGL(logmel_main<float>, grid, dim3(256), lds, stream, (const float*)pcm, n_samples, n_frames, t->basis,
                       t->melfilt, mel, clipmax);
  }
  OASR_LAUNCH_CHECK();
  const long per_clip = (long)NMEL * n_frames;
  dim3 g2((unsigned)((per_clip / 4 + 255) / 256 > 64 ? 64 : (per_clip / 4 + 255) / 256 + 1), B);
  hipLaunchKernelGGL(logmel_finalize, g2, dim3(256), 0, stream, mel, clipmax, per_clip);
  OASR_LAUNCH_CHECK();
  return OASR_OK;
}

// Shared device/host helpers for the gfx950 (CDNA4, wave64) kernels of liboasr.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef uint16_t bf16_t;  // raw bf16 bits; all arithmetic goes through float

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;

#define OASR_OK 0
#define OASR_EINVAL (-1)
#define OASR_EHIP (-2)
#define OASR_ESTATE (-3)
#define OASR_ERETRY (-4)

void oasr_set_error(const char* fmt, ...);

#define OASR_CHECK_HIP(expr)                                                              \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      oasr_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return OASR_EHIP;                                                                   \
    }                                                                                     \
  } while (0)

#define OASR_REQUIRE(cond, ...)  \
  do {                           \
    if (!(cond)) {               \
      oasr_set_error(__VA_ARGS__); \
      return OASR_EINVAL;        \
    }                            \
  } while (0)

#define OASR_LAUNCH_CHECK() OASR_CHECK_HIP(hipGetLastError())

// Experiment / A-B switches read from the environment (OASR_PP_*, OASR_GEMM_*, OASR_LOGMEL, OASR_PROF_SHAPES, OASR_XCD_FLAGS) are part of the
// measurement tooling, not of the product: like the setters of include/oasr_testing.h they are INERT unless the process opted in with
// OASR_TESTING_HOOKS=1 (scripts/ and tests/ do), so a production process cannot be steered through its environment.
static inline const char* oasr_experiment_env(const char* name) {
  const char* h = getenv("OASR_TESTING_HOOKS");
  return (h && h[0] == '1') ? getenv(name) : nullptr;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE property of a kernel: a process that touches a second GPU must set
// it there too.  One of these per launch site (a function-local static); setting it again is harmless, so two host threads racing on
// a slot only repeat the call.
struct LdsAttrOnce {
  bool done[64] = {};
};
static inline int ensure_dynamic_lds(LdsAttrOnce& st, const void* fn, int bytes) {
  int dev = 0;
  OASR_CHECK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !st.done[dev]) {
    OASR_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    if (dev >= 0 && dev < 64) st.done[dev] = true;
  }
  return OASR_OK;
}

// ---- bf16 <-> f32 (round-to-nearest-even, NaN preserved) -------------------------------------------
__host__ __device__ __forceinline__ float bf2f(bf16_t h) {
  union { uint32_t u; float f; } c;
  c.u = ((uint32_t)h) << 16;
  return c.f;
}
__host__ __device__ __forceinline__ bf16_t f2bf(float f) {
  union { uint32_t u; float f; } c;
  c.f = f;
  uint32_t u = c.u;
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // quiet NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
typedef __attribute__((ext_vector_type(2))) float oasr_f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 oasr_bf16x2_t;
#if defined(__HIP_DEVICE_COMPILE__)
// gfx950 has a hardware RNE converter (v_cvt_pk_bf16_f32, 2 values per instruction); the bit-twiddling f2bf above is
// the host-side / reference definition and costs ~6 VALU per value on the device.
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  const oasr_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, oasr_bf16x2_t));
}
__device__ __forceinline__ float bf_round(float f) { return (float)(__bf16)f; }
__device__ __forceinline__ bf16_t f2bf_dev(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
#else
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) { return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16); }
__device__ __forceinline__ float bf_round(float f) { return bf2f(f2bf(f)); }
__device__ __forceinline__ bf16_t f2bf_dev(float f) { return f2bf(f); }
#endif
__device__ __forceinline__ float bf_lo(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(uint32_t p) { return __uint_as_float(p & 0xffff0000u); }

// exact-erf GELU (nn.GELU() default) and its derivative.  erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, far below
// the bf16 rounding every consumer applies) on v_rcp_f32 / v_exp_f32: ~15 VALU per value instead of libm erff's 34,
// and GELU' reuses the same exponential (pdf = exp(-x^2/2)/sqrt(2 pi)).
struct GeluParts {
  float cdf, e;  // Phi(x), exp(-x^2/2)
};
__device__ __forceinline__ GeluParts gelu_parts(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170f);
  const float half_erfc = 0.5f * poly * e;  // 0.5 * erfc(|x|/sqrt2)
  GeluParts g;
  g.cdf = x >= 0.f ? 1.0f - half_erfc : half_erfc;
  g.e = e;
  return g;
}
// gelu(x) = x * Phi(x) = max(x, 0) - |x| * (0.5 erfc(|x|/sqrt2)): no sign select, 3 VALU fewer than x * cdf
__device__ __forceinline__ float gelu_f(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  const float half_poly = t * (0.127414796f + t * (-0.142248368f + t * (0.7107068705f + t * (-0.7265760135f + t * 0.5307027145f))));
  const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170f);
  return fmaf(-fabsf(x), half_poly * e, fmaxf(x, 0.f));
}
// both at once (the act == 2 epilogue): gelu = x * Phi(x), gelu' = Phi(x) + x * pdf(x)
__device__ __forceinline__ void gelu_and_deriv(float x, float& g, float& d) {
  const GeluParts p = gelu_parts(x);
  g = x * p.cdf;
  d = fmaf(x * p.e, 0.3989422804014327f, p.cdf);
}
__device__ __forceinline__ float dgelu_f(float x) {
  const GeluParts g = gelu_parts(x);
  return g.cdf + x * g.e * 0.3989422804014327f;
}

// ---- wave64 reductions ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Bijective XCD-aware remap of a linear workgroup id (block b runs on XCD b % 8): gives each XCD a
// contiguous chunk of the logical grid so neighbouring tiles share one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// VALU issue-rate probe (gfx950): how many cycles does one SIMD spend per wave64 instruction of a given kind, with 1..4 waves per SIMD?
// Each kernel is a straight chain-free instruction stream (16 independent accumulators per lane), 256 workgroups of 64 * W * 4 threads.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate scripts/probes/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int NACC = 16, UNROLL = 8;

template <int KIND>
__global__ void probe(float* out, int iters, float a, float b) {
  float acc[NACC];
  f2 pacc[NACC / 2];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (float)(threadIdx.x + i);
#pragma unroll
  for (int i = 0; i < NACC / 2; ++i) pacc[i] = f2{(float)threadIdx.x, (float)i};
  const f2 pa = {a, a}, pb = {b, b};
  const unsigned long long mask = __builtin_amdgcn_ballot_w64(threadIdx.x & 1);
  if (KIND == 11) asm volatile("v_cmp_gt_f32 vcc, %0, %1" ::"v"(acc[0]), "v"(a) : "vcc");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
        if (KIND == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
        if (KIND == 2) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc[i]));
        if (KIND == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(acc[i]) : "v"(a) : );
        if (KIND == 4 && i < NACC / 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pacc[i]) : "v"(pa), "v"(pb));
        if (KIND == 5) asm volatile("v_log_f32 %0, %0" : "+v"(acc[i]));
        if (KIND == 6) asm volatile("v_add_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc[i]) : "v"(a));
        if (KIND == 7) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "s"(mask));
        if (KIND == 8) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(acc[i]) : "v"(acc[(i + 1) % NACC]), "v"(a), "s"(mask));
        if (KIND == 9) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
        if (KIND == 11) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(acc[i]) : "v"(acc[(i + 1) % NACC]), "v"(a) : );
        if (KIND == 12) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
        if (KIND == 13) asm volatile("v_max_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
        if (KIND == 14) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(acc[i]));
        if (KIND == 10) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
#pragma unroll
  for (int i = 0; i < NACC / 2; ++i) s += pacc[i][0] + pacc[i][1];
  if (s == 12345.678f) out[0] = s;
}

template <int KIND>
void run(const char* name, int per_iter, float* out) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 2000;
  for (int w = 1; w <= 4; ++w) {  // waves per SIMD
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(64 * 4 * w), 0, 0, out, iters, 1.0001f, 0.5f);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double instr_per_simd = (double)iters * UNROLL * per_iter * w;
    printf("%-28s %d wave(s)/SIMD: %7.3f ms -> %.2f cycles per wave64 instruction per SIMD at 2.4 GHz\n", name, w, best, best * 1e-3 * 2.4e9 / instr_per_simd);
  }
}
int main() {
  float* out; CK(hipMalloc(&out, 4));
  run<0>("v_fma_f32", NACC, out);
  run<1>("v_add_f32", NACC, out);
  run<2>("v_mov_b32_dpp quad_perm", NACC, out);
  run<3>("v_cndmask_b32", NACC, out);
  run<4>("v_pk_fma_f32", NACC / 2, out);
  run<5>("v_log_f32", NACC, out);
  run<6>("v_add_f32_dpp quad_perm", NACC, out);
  run<7>("v_cndmask_b32_e64 (sgpr mask)", NACC, out);
  run<8>("v_cndmask_e64, dst != src", NACC, out);
  run<9>("v_mul_f32", NACC, out);
  run<10>("v_bfi_b32", NACC, out);
  run<11>("v_cndmask_b32 vcc (v_cmp first)", NACC, out);
  run<12>("v_xor_b32", NACC, out);
  run<13>("v_max_f32", NACC, out);
  run<14>("v_cvt_f32_i32", NACC, out);
  return 0;
}

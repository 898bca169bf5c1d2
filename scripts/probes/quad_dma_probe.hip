// Upper bound of the instruction stream of a 4-wave x 128x128 GEMM K loop with LDS-DMA staging on gfx950 (verdict item "4-wave 128x128 form", the variant
// round 3 did not build): one wave per SIMD issues, per 64-deep K-tile of a 256 x 256 output tile, 64 MFMAs (32x32x16 bf16, 256 accumulator registers),
// 32 ds_read_b128 fragment reads (double-buffered over the 4 k-steps), its 16 direct-to-LDS 1 KiB pieces of the next K-tile and one workgroup barrier.
// Addresses are conflict-free and the data is garbage: this measures issue / latency structure only, with real HBM streaming (every workgroup reads its own 4 MiB).
// MODE bits: 1 MFMAs, 2 fragment reads, 4 DMA pieces, 8 barrier + counted wait.   build: hipcc --offload-arch=gfx950 -O3 quad_dma_probe.hip -o quad_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define FENCE() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ void glds16(const u32x4 rs, unsigned lds_addr, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs) : "memory");
}

template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64, 1) void quad(const char* __restrict__ src, float* __restrict__ sink, int nt, int shared_src) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned smem_a = (unsigned)(size_t)smem;
  const unsigned long addr = (unsigned long)(src + (shared_src ? 0 : (size_t)blockIdx.x * ((size_t)nt << 16)));
  u32x4 rs;
  rs[0] = (unsigned)addr;
  rs[1] = (unsigned)(addr >> 32) & 0xffffu;
  rs[2] = (unsigned)nt << 16;
  rs[3] = 0x00020000u;
  constexpr int NB = NW == 4 ? 4 : 2;   // B tiles (32 columns each) per wave: 128 x 128 or 128 x 64 outputs per wave
  constexpr int NP = 64 / NW;           // DMA pieces per wave and K-tile
  f32x16 acc[4 * NB];
#pragma unroll
  for (int i = 0; i < 4 * NB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 fa[2][4], fb[2][NB];
  // fragment of k-step ks, operand o (0 = A rows of this wave, 1 = B rows), tile i: 1 KiB, lane-linear (conflict-free)
  auto frag = [&](int stage, int o, int i, int ks) { return *(const bf16x8*)(smem + stage * 65536 + o * 32768 + (((o ? wave >> 1 : wave) & 1) * 4 + i) * 4096 + ks * 1024 + lane * 16); };
  auto piece = [&](int t, int j) {  // this wave's j-th of 16 pieces of K-tile t
    if (MODE & 4) glds16(rs, smem_a + (t & 1) * 65536 + (wave * NP + j) * 1024, (unsigned)((shared_src ? (t & 15) : t) << 16) + (wave * NP + j) * 1024 + lane * 16);
  };
  for (int j = 0; j < NP; ++j) piece(0, j);
  if (MODE & 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) fa[0][i] = frag(0, 0, i, 0);
#pragma unroll
  for (int i = 0; i < NB; ++i) fb[0][i] = frag(0, 1, i, 0);
  for (int t = 0; t < nt; ++t) {
    const int st = t & 1;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      FENCE();
      if (MODE & 2) {
        if (ks < 3) {
#pragma unroll
          for (int i = 0; i < 4; ++i) fa[nxt][i] = frag(st, 0, i, ks + 1);
#pragma unroll
          for (int i = 0; i < NB; ++i) fb[nxt][i] = frag(st, 1, i, ks + 1);
        }
      }
      FENCE();
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (MODE & 1) {
#pragma unroll
          for (int j = 0; j < NB; ++j) acc[g * NB + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][g], fb[cur][j], acc[g * NB + j], 0, 0, 0);
        }
        FENCE();
        if (t + 1 < nt && (NW == 4 || (g & 1))) piece(t + 1, NW == 4 ? ks * 4 + g : ks * 2 + (g >> 1));
        FENCE();
      }
    }
    if (MODE & 8) {
      if (MODE & 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next tile's pieces (issued over this whole tile) have landed
      FENCE();
      asm volatile("s_barrier" ::: "memory");
      FENCE();
    }
    if (MODE & 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[0][i] = frag(st ^ 1, 0, i, 0);  // first k-step of the next tile (exposed here; a real kernel reads it under the last k-step)
#pragma unroll
      for (int i = 0; i < NB; ++i) fb[0][i] = frag(st ^ 1, 1, i, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4 * NB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) sink[threadIdx.x] = s;
}

// sustained rate under the package power cap: back-to-back launches for ~seconds (the clock settles to what the stream's energy per flop allows)
template <int MODE, int NW>
static void sustained(const char* name, const char* src, float* sink, int nt, float seconds) {
  CK(hipFuncSetAttribute((const void*)quad<MODE, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((quad<MODE, NW>), dim3(256), dim3(NW * 64), 131072, 0, src, sink, nt, 1);
  CK(hipDeviceSynchronize());
  const int n = (int)(seconds / 0.33e-3f);
  // first half warms the package up, second half is timed
  for (int i = 0; i < n / 2; ++i) hipLaunchKernelGGL((quad<MODE, NW>), dim3(256), dim3(NW * 64), 131072, 0, src, sink, nt, 1);
  CK(hipEventRecord(e0));
  for (int i = 0; i < n / 2; ++i) hipLaunchKernelGGL((quad<MODE, NW>), dim3(256), dim3(NW * 64), 131072, 0, src, sink, nt, 1);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double flops = 256.0 * nt * 256.0 * 256.0 * 64.0 * 2.0 * (n / 2);
  printf("%-64s %8.1f ms for %d launches  %7.0f TFLOP/s-equivalent sustained\n", name, ms, n / 2, flops / ms / 1e9);
}

template <int MODE, int NW>
static void run(const char* name, const char* src, float* sink, int nt, int shared_src) {
  CK(hipFuncSetAttribute((const void*)quad<MODE, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int it = 0; it < 6; ++it) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((quad<MODE, NW>), dim3(256), dim3(NW * 64), 131072, 0, src, sink, nt, shared_src);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (it && ms < best) best = ms;
  }
  const double flops = 256.0 * nt * 256.0 * 256.0 * 64.0 * 2.0;
  printf("%-64s %8.3f ms  %7.0f TFLOP/s-equivalent  (%.0f cycles per K-tile at 2.4 GHz; 64 MFMAs = 2048)\n", name, best, flops / best / 1e9, best * 1e-3 * 2.4e9 / nt);
}

int main() {
  const int nt = 256;  // K = 16384: long enough for the clocks to settle
  char* src;
  float* sink;
  CK(hipMalloc(&src, (size_t)256 * nt << 16));
  CK(hipMemset(src, 0, (size_t)256 * nt << 16));
  {  // the shared 1 MiB that the L2-resident runs read: random bf16 in +-[0.5, 1) -- matrix-core power depends on the operands' toggling, zeros would flatter every stream
    unsigned short* h = (unsigned short*)malloc(1 << 20);
    unsigned x = 12345u;
    for (int i = 0; i < (1 << 19); ++i) {
      x = x * 1664525u + 1013904223u;
      h[i] = (unsigned short)(((x >> 16) & 0x807Fu) | 0x3F00u);
    }
    CK(hipMemcpy(src, h, 1 << 20, hipMemcpyHostToDevice));
    free(h);
  }
  CK(hipMalloc(&sink, 4096));
  printf("operand tiles SHARED by all workgroups (1 MiB, L2-resident: the reuse a GEMM's rasterisation gives)\n");
  printf("-- 4 waves x 128x128 outputs (one wave per SIMD: 64 MFMAs, 32 fragment reads, 16 DMA pieces per wave and K-tile)\n");
  run<1, 4>("MFMAs only", src, sink, nt, 1);
  run<3, 4>("MFMAs + fragment reads", src, sink, nt, 1);
  run<5, 4>("MFMAs + DMA pieces", src, sink, nt, 1);
  run<15, 4>("all: MFMAs + fragment reads + DMA pieces + barrier/wait", src, sink, nt, 1);
  run<14, 4>("no MFMAs: fragment reads + DMA pieces + barrier/wait", src, sink, nt, 1);
  printf("-- 8 waves x 128x64 outputs (two waves per SIMD, free-running: 32 MFMAs, 24 fragment reads, 8 DMA pieces per wave and K-tile)\n");
  run<1, 8>("MFMAs only", src, sink, nt, 1);
  run<3, 8>("MFMAs + fragment reads", src, sink, nt, 1);
  run<5, 8>("MFMAs + DMA pieces", src, sink, nt, 1);
  run<15, 8>("all: MFMAs + fragment reads + DMA pieces + barrier/wait", src, sink, nt, 1);
  run<14, 8>("no MFMAs: fragment reads + DMA pieces + barrier/wait", src, sink, nt, 1);
  printf("-- again, 4 waves then 8 waves, everything on (order effects / clock drift)\n");
  run<15, 4>("4 waves, all", src, sink, nt, 1);
  run<15, 8>("8 waves, all", src, sink, nt, 1);
  run<15, 4>("4 waves, all", src, sink, nt, 1);
  run<15, 8>("8 waves, all", src, sink, nt, 1);
  printf("-- sustained (3 s each, second half timed; L2-resident operands): what the power cap leaves of each stream\n");
  sustained<1, 4>("4 waves, MFMAs only", src, sink, nt, 3.0f);
  sustained<1, 8>("8 waves, MFMAs only", src, sink, nt, 3.0f);
  sustained<15, 4>("4 waves, all", src, sink, nt, 3.0f);
  sustained<15, 8>("8 waves, all", src, sink, nt, 3.0f);
  sustained<15, 4>("4 waves, all", src, sink, nt, 3.0f);
  sustained<15, 8>("8 waves, all", src, sink, nt, 3.0f);
  printf("operand tiles private per workgroup (4 MiB each: pure HBM streaming, 1 GiB per launch)\n");
  run<15, 4>("4 waves, all", src, sink, nt, 0);
  run<15, 8>("8 waves, all", src, sink, nt, 0);
  return 0;
}

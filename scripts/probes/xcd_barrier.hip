// micro-benchmark: barrier among the 32 workgroups that land on XCD 0 (blockIdx % 8 == 0) vs among all 256
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>  // 0: agent-scope release/acquire fences (round-2 barrier); 1: vmcnt(0) + relaxed atomics + buffer_inv sc1; 2: vmcnt(0) + relaxed atomics, data via sc1 loads
__device__ __forceinline__ void bar(unsigned* counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 22)) break;
    }
    if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    else if (MODE == 1) asm volatile("buffer_inv sc1" ::: "memory");
  }
  __syncthreads();
}

// each phase: every participating workgroup writes its slot of buf[phase & 1], after the barrier reads ALL slots written in this phase and
// checks them (detects stale data); participants = workgroups with (blockIdx.x & mask) == 0
template <int MODE>
__global__ void k(unsigned* counter, unsigned* buf, unsigned* bad, int nphase, int mask, int nwg) {
  if (blockIdx.x & mask) return;
  const int me = mask ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  unsigned errs = 0;
  for (int p = 0; p < nphase; ++p) {
    unsigned* b = buf + (p & 1) * 4096;
    if (threadIdx.x < 16) b[me * 16 + threadIdx.x] = (unsigned)(p * 1000003 + me * 16 + threadIdx.x);
    bar<MODE>(counter, (unsigned)(p + 1) * nwg);
    for (int i = threadIdx.x; i < nwg * 16; i += blockDim.x) {
      unsigned v;
      if (MODE == 2) v = __hip_atomic_load(b + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else v = b[i];
      if (v != (unsigned)(p * 1000003 + i)) ++errs;
    }
    // (second barrier so that nobody overwrites buf[(p+2)&1] while a slow reader is still on it: folded into the next phase's barrier
    //  because the buffers alternate)
  }
  if (errs) atomicAdd(bad, errs);
}

template <int MODE>
void run(const char* name, int mask) {
  unsigned *counter, *buf, *bad;
  CK(hipMalloc(&counter, 64)); CK(hipMalloc(&buf, 2 * 4096 * 4)); CK(hipMalloc(&bad, 4));
  const int nphase = 200, nwg = mask ? 32 : 256;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f; unsigned hbad = 0;
  for (int it = 0; it < 5; ++it) {
    CK(hipMemset(counter, 0, 64)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(buf, 0xff, 2 * 4096 * 4));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, counter, buf, bad, nphase, mask, nwg);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    unsigned b; CK(hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost)); hbad += b;
  }
  printf("%-70s %3d workgroups: %7.2f us per phase (write + barrier + read-all), stale reads %u\n", name, nwg, best * 1000.f / nphase, hbad);
}
int main() {
  run<0>("agent release/acquire fences, all 256 workgroups", 0);
  run<0>("agent release/acquire fences, XCD 0 only", 7);
  run<1>("vmcnt(0) + relaxed counter + buffer_inv sc1, XCD 0 only", 7);
  run<2>("vmcnt(0) + relaxed counter, data through sc1 (agent-scope) loads, XCD 0 only", 7);
  run<1>("vmcnt(0) + relaxed counter + buffer_inv sc1, all 256 (expected: stale reads)", 0);
  return 0;
}

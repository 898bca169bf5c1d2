"""Writes a copy of olmoasr_amd/csrc/gemm.hip with s_memtime / s_memrealtime / HW_ID stamps in the ping-pong kernel (one record per
workgroup in a __device__ buffer + an exported reader).  The copy is compiled into a scratch library next to the product objects and
loaded through OASR_LIB (scripts/probes/gemm_tile_stamps.py reads the records): profiles/r03_gemm_tile_stamps.txt.
usage: python scripts/probes/gemm_tile_stamps_patch.py /tmp/gemm_dbg.hip"""
import sys
s=open('/root/repo/olmoasr_amd/csrc/gemm.hip').read()
def rep(a,b):
    global s
    assert a in s, a[:60]
    s=s.replace(a,b,1)
rep('int g_stagger = 0, g_stagger_phases = 2;','__device__ unsigned long long g_gemm_dbg[8 * 16384];\nextern "C" int oasr_gemm_dbg_read(void* dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_gemm_dbg), bytes); }\nint g_stagger = 0, g_stagger_phases = 2;')
rep('''  int v = blockIdx.x;
  OASR_PP_TILE(v);
  if (nt <= 0) return;  // (uneven split-K; the host never launches those persistent)''','''  const unsigned long long dbg_r0 = __builtin_amdgcn_s_memrealtime(), dbg_t0 = __builtin_amdgcn_s_memtime();
  unsigned long long dbg_t1 = 0, dbg_t2 = 0;
  int v = blockIdx.x;
  OASR_PP_TILE(v);
  if (nt <= 0) return;  // (uneven split-K; the host never launches those persistent)''')
rep('''  OASR_PP_BARRIER();
  if (wm == 1 && !(VAR & 8)) OASR_PP_BARRIER();  // the upper half trails by one barrier from here on (VAR bit 3: lockstep experiment)''','''  OASR_PP_BARRIER();
  dbg_t1 = __builtin_amdgcn_s_memtime();
  if (wm == 1 && !(VAR & 8)) OASR_PP_BARRIER();  // the upper half trails by one barrier from here on (VAR bit 3: lockstep experiment)''')
rep('''  if (wm == 0 && !(VAR & 8)) OASR_PP_BARRIER();  // re-join: the trailing half has finished its LDS reads after this
''','''  if (wm == 0 && !(VAR & 8)) OASR_PP_BARRIER();  // re-join: the trailing half has finished its LDS reads after this
  dbg_t2 = __builtin_amdgcn_s_memtime();
''')
rep('''  fast_epilogue<SWAP, CSUM>(p, acc, stg, (float*)(smem + 2 * BUF + wave * 256), em0, en0, wm, wn, lane);
  if (!has_next) break;''','''  fast_epilogue<SWAP, CSUM>(p, acc, stg, (float*)(smem + 2 * BUF + wave * 256), em0, en0, wm, wn, lane);
  if (tid == 0 && blockIdx.y == 0 && blockIdx.x < 16384) {
    const unsigned long long t3 = __builtin_amdgcn_s_memtime(), r3 = __builtin_amdgcn_s_memrealtime();
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* d = g_gemm_dbg + (long)blockIdx.x * 8;
    d[0] = dbg_r0; d[1] = r3; d[2] = dbg_t1 - dbg_t0; d[3] = dbg_t2 - dbg_t1; d[4] = t3 - dbg_t2; d[5] = ((unsigned long long)xcc << 32) | hw; d[6] = t3 - dbg_t0;
  }
  if (!has_next) break;''')
open(sys.argv[1],'w').write(s)

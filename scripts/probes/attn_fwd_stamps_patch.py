"""Writes a copy of olmoasr_amd/csrc/attention.hip whose unmasked forward kernel (attn_fwd_kernel<false, false>) carries s_memtime stamps at its phase
boundaries, summed per wave over the key tiles and written to a __device__ buffer (+ an exported reader): where do the ~3900 cycles per 64-key tile go?
  seg 0  loop top -> the 8 S^T MFMAs issued (K fragment reads + MFMA issue)
  seg 1  -> row maximum known (first use of the MFMA results: waits for the matrix pipe; 16 max3 + the lane exchange + the rescale test)
  seg 2  -> exponentials, sums and packs done (VALU only)
  seg 3  -> the 8 O^T MFMAs issued (V^T fragment reads + MFMA issue)
  seg 4  -> next tile committed to LDS (waits for the global prefetch)
  seg 5  -> workgroup barrier passed
Every stamp is an s_memtime + s_waitcnt lgkmcnt(0) between scheduling fences: it also drains the LDS reads in flight, so the stamped build runs slower than the product
(reported).  usage: python scripts/probes/attn_fwd_stamps_patch.py /tmp/attn_stamps.hip ; scripts/probes/build_variant_lib.sh /tmp/attn_stamps.hip stamps"""
import sys
s = open('/root/repo/olmoasr_amd/csrc/attention.hip').read()


def rep(a, b, count=1):
    global s
    assert s.count(a) >= 1, a[:70]
    s = s.replace(a, b, count)


STAMP = lambda i: f"""    if (!CAUSAL && !ROWS) {{ __builtin_amdgcn_sched_barrier(0); const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); dbg_acc[{i}] += tn_ - dbg_t; dbg_t = tn_; __builtin_amdgcn_sched_barrier(0); }}
"""
rep('namespace {\n', '__device__ unsigned long long g_attn_dbg[8 * 4 * 32768];\nextern "C" int oasr_attn_dbg_read(void* dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_attn_dbg), bytes); }\nnamespace {\n')
# loop top
rep('''  for (int t = 0; t < ntiles; ++t) {
    const char* kb = smem + (t & 1) * 2 * TILE;
    const char* vb = kb + TILE;
    const bool more = t + 1 < ntiles;
    if (more) {
      const int r0 = krow0(t + 1);''', '''  unsigned long long dbg_acc[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long dbg_t = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < ntiles; ++t) {
    const char* kb = smem + (t & 1) * 2 * TILE;
    const char* vb = kb + TILE;
    const bool more = t + 1 < ntiles;
    if (more) {
      const int r0 = krow0(t + 1);''')
rep('''      for (int ds = 0; ds < 4; ++ds) sT[kt] = MFMA(frag_rows(kb, kt * 32, ds, lane), qf[ds], sT[kt]);
    }
    // Only boundary tiles''', '''      for (int ds = 0; ds < 4; ++ds) sT[kt] = MFMA(frag_rows(kb, kt * 32, ds, lane), qf[ds], sT[kt]);
    }
''' + STAMP(0) + '''    // Only boundary tiles''')
rep('''    {
      const float nm = -m_run;
      float ps0 = 0.f, ps1 = 0.f;''', STAMP(1) + '''    {
      const float nm = -m_run;
      float ps0 = 0.f, ps1 = 0.f;''')
rep('''      l_run2[0] += ps0;
      l_run2[1] += ps1;
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const bf16x8_t pf = pack_half(sT[kt], u);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) oT[dt] = MFMA(frag_cols(vb, kt * 32 + 16 * u, dt * 32, lane), pf, oT[dt]);
      }
    if (more) {
      char* nb = smem + ((t + 1) & 1) * 2 * TILE;
      tile_commit(nb, tid, rk);
      tile_commit(nb + TILE, tid, rv);
    }
    __syncthreads();
  }
''', '''      l_run2[0] += ps0;
      l_run2[1] += ps1;
    }
    bf16x8_t dbg_pf[2][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int u = 0; u < 2; ++u) dbg_pf[kt][u] = pack_half(sT[kt], u);
    asm volatile("" : "+v"(dbg_pf[0][0]), "+v"(dbg_pf[0][1]), "+v"(dbg_pf[1][0]), "+v"(dbg_pf[1][1]));
''' + STAMP(2) + '''#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const bf16x8_t pf = dbg_pf[kt][u];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) oT[dt] = MFMA(frag_cols(vb, kt * 32 + 16 * u, dt * 32, lane), pf, oT[dt]);
      }
''' + STAMP(3) + '''    if (more) {
      char* nb = smem + ((t + 1) & 1) * 2 * TILE;
      tile_commit(nb, tid, rk);
      tile_commit(nb + TILE, tid, rv);
    }
''' + STAMP(4) + '''    __syncthreads();
''' + STAMP(5) + '''  }
  if (!CAUSAL && !ROWS && lane == 0) {
    const long wid = (long)blockIdx.x * 4 + wave;
    if (wid < 4 * 32768) {
#pragma unroll
      for (int i = 0; i < 6; ++i) g_attn_dbg[wid * 8 + i] = dbg_acc[i];
      g_attn_dbg[wid * 8 + 6] = (unsigned long long)ntiles;
    }
  }
''')
open(sys.argv[1], 'w').write(s)

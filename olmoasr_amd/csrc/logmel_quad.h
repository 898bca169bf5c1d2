// Quad-lane, register-resident log-mel kernel (included by logmel.hip after its cf / dft5 helpers).
//
// The LDS FFT kernel above (logmel_fft) keeps a frame's 200 complex points in LDS and walks them in three barrier-separated stages; with
// 1.6 KB of LDS per frame only 12 waves fit a CU and every stage runs at LDS round-trip latency (profiles/r03_logmel_fft.txt).  Here a frame
// lives in the REGISTERS of four adjacent lanes (a DPP quad), 50 complex points each, and never touches LDS between its samples and its 80
// log-mel values:
//     pack      z[n] = w[2n] x[2n] + i w[2n+1] x[2n+1], n = 50 l + m: lane l reads the 100 consecutive samples 100 l .. 100 l + 99 of the frame
//     stage 1   radix-4 ACROSS the four lanes (n1 = lane): two DPP quad_perm exchanges (xor 2, xor 1) with per-lane signs and one -i rotation
//               in lane 3; lane l ends up with A[k1(l)][m], k1(l) = bit-reversed l = {0, 2, 1, 3}; twiddle W200^(m k1) from a per-lane table
//     stage 2   one 50-point FFT per lane, entirely in registers (2 x 25, 25 = 5 x 5, compile-time twiddles): Z[k1 + 4 k2], k2 = 0..49
//     unpack    X[k] = (Z[k] + conj Z[200-k]) / 2 - i W400^k (Z[k] - conj Z[200-k]) / 2: the partner bin 200 - k sits in register
//               (50 - k2) % 50 of the same lane (k1 = 0) or in register 49 - k2 of the lane holding 4 - k1 (one more quad_perm);
//               |X[k]|^2 for k = 0..199 (bin 200, like bin 0, has weight zero in every slaney filter and is never formed)
//     mel       filter m covers bins lo..hi = registers k2 = lo/4 .. hi/4 of all four lanes: each lane multiplies its <= 5 registers by its
//               own weights (zero where its bin lies outside the triangle) and the quad sums with two more DPP adds; lane l keeps the filters
//               m = l mod 4, takes log10 and stores
// 64 frames per 256-thread workgroup; LDS holds only the workgroup's samples (as fp32, converted once while staging) and 7.4 KB of per-lane
// tables: 49 KB, three workgroups per CU, no barrier after the staging one.  fp32 throughout, twiddles rounded once from double.
#pragma once
#include <type_traits>

#include "logmel_quad_tables.h"

constexpr int QT = 64;                              // frames per workgroup
constexpr int QNS = QT * HOP + (NFFT - HOP);        // samples of 64 frames: 10480
constexpr int QTAB_WIN = 0, QTAB_TW200 = 400, QTAB_TW400 = 800, QTAB_MEL = 1200, QTAB = 1200 + 4 * QMEL_SLOTS;  // floats

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// quad_perm controls: lane i reads lane perm[i] of its quad
constexpr int DPP_XOR1 = 1 | (0 << 2) | (3 << 4) | (2 << 6), DPP_XOR2 = 2 | (3 << 2) | (0 << 4) | (1 << 6), DPP_SWAP23 = 0 | (1 << 2) | (3 << 4) | (2 << 6);
template <int CTRL>
__device__ __forceinline__ float qperm(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ cf qperm(cf v) { return cf{qperm<CTRL>(v.x), qperm<CTRL>(v.y)}; }

// acc + quad_perm(x) * c in ONE instruction (v_fmac_f32_dpp: a DPP instruction issues in ~4.2 cycles whatever it computes, so the exchange and
// the butterfly's add cost what the exchange alone did; hipcc's DPP combiner does not form it from fmaf(mov_dpp(x), c, acc)).  Inline asm: the
// VALU-write -> DPP-read hazard (2 wait states, cdna_hip_programming.md 5.7 item 2) is padded inside the string -- x usually comes straight out of
// the multiply in front.  x and acc may be the same value: every lane reads before any lane writes.
template <int CTRL>
__device__ __forceinline__ float fmac_qperm(float acc, float x, float c) {
  static_assert(CTRL == DPP_XOR1 || CTRL == DPP_XOR2, "spelled-out controls only");
  if constexpr (CTRL == DPP_XOR1)
    asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(x), "v"(c));
  else
    asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(x), "v"(c));
  return acc;
}

// Per-lane selects as BIT selects on an all-ones / all-zeros mask register (v_bfi_b32): 195 fewer VALU instructions per lane than the
// v_cmp-free ternaries compile to (sign flips fold into v_xor), same speed (profiles/r06_logmel_quad.txt).  The masks pass through an
// empty asm so that the compiler cannot turn the and / or back into a select.
__device__ __forceinline__ unsigned lane_mask(bool p) {
  unsigned m = p ? 0xffffffffu : 0u;
  asm volatile("" : "+v"(m));
  return m;
}
__device__ __forceinline__ float bsel(unsigned m, float a, float b) {  // m ? a : b
  return __uint_as_float((__float_as_uint(a) & m) | (__float_as_uint(b) & ~m));
}
__device__ __forceinline__ float fneg(float a) { return __uint_as_float(__float_as_uint(a) ^ 0x80000000u); }

// 25-point DFT of x[0..24] (stride-1 registers) into out[0..24], natural order: b = 5 b1 + b2, e = e1 + 5 e2
__device__ __forceinline__ void dft25(const cf (&x)[25], cf (&out)[25]) {
  cf s[5][5];  // [b2][e1]
  static_for<0, 5>([&](auto b2c) {
    constexpr int b2 = decltype(b2c)::value;
    cf y[5];
    dft5(x[b2], x[5 + b2], x[10 + b2], x[15 + b2], x[20 + b2], y);
    static_for<0, 5>([&](auto e1c) {
      constexpr int e1 = decltype(e1c)::value;
      if constexpr (b2 * e1 == 0) {
        s[b2][e1] = y[e1];
      } else {
        constexpr float wr = QW25_RE[b2 * e1], wi = QW25_IM[b2 * e1];
        s[b2][e1] = cmul(y[e1], cf{wr, wi});
      }
    });
  });
  static_for<0, 5>([&](auto e1c) {
    constexpr int e1 = decltype(e1c)::value;
    cf y[5];
    dft5(s[0][e1], s[1][e1], s[2][e1], s[3][e1], s[4][e1], y);
    static_for<0, 5>([&](auto e2c) { out[e1 + 5 * decltype(e2c)::value] = y[decltype(e2c)::value]; });
  });
}

#ifndef QUAD_WAVES
#define QUAD_WAVES 3
#endif
#ifndef QUAD_ABLATE
#define QUAD_ABLATE 0  // timing experiments only (results garbage): bit 0 no cross-lane radix-4, 1 no 50-point FFT, 2 no unpack, 3 no mel reduction / pick
#endif
#ifndef QUAD_STAGE16
#define QUAD_STAGE16 0  // 1: int16 PCM stays int16 in LDS (21 KB instead of 42: five workgroups per CU) and every lane converts its own 100 samples
#endif
template <typename PCM>
constexpr bool quad_stage16() { return QUAD_STAGE16 && sizeof(PCM) == 2; }
template <typename PCM>
constexpr size_t quad_lds_bytes() { return (quad_stage16<PCM>() ? (size_t)QNS * 2 : (size_t)QNS * 4) + (size_t)QTAB * 4; }
template <typename PCM>
__global__ __launch_bounds__(256, QUAD_WAVES) void logmel_quad(const PCM* __restrict__ pcm, int n_samples, int n_frames,
                                                      const float* __restrict__ tab,  // [QTAB]: window | W200 per lane | W400 per lane | mel weights per lane
                                                      float* __restrict__ out, unsigned* __restrict__ clipmax) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr bool S16 = quad_stage16<PCM>();
  float* samp = smem;                          // [QNS] samples of this workgroup's 64 frames: fp32 (converted once here), or raw int16 (S16)
  float* tw = smem + (S16 ? QNS / 2 : QNS);    // [QTAB]
  const int nblk = (n_frames + QT - 1) / QT;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);  // XCD-contiguous: neighbouring frame blocks of a clip share an L2 (see logmel_fft)
  const int b = bid / nblk, f0 = (bid - b * nblk) * QT, tid = threadIdx.x;
  const int l = tid & 3, f = tid >> 2;
  const PCM* clip = pcm + (long)b * n_samples;

  // ---- staging: every global load of this thread is issued before the first LDS write waits for one (the plain loop form compiles to
  // load / s_waitcnt vmcnt(0) / ds_write per trip: 6-11 dependent memory round trips per workgroup, as long as the whole FFT)
  constexpr int NTAB = (QTAB / 4 + 255) / 256;
  f32x4_t tv[NTAB];
#pragma unroll
  for (int i = 0; i < NTAB; ++i) {
    const int q = tid + 256 * i;
    if (q < QTAB / 4) tv[i] = ((const f32x4_t*)tab)[q];
  }
  // samples of these 64 frames (reflect padding of torch.stft(center=True) at the clip's ends)
  const long s_begin = (long)f0 * HOP - NFFT / 2;
  const bool interior = s_begin >= 0 && s_begin + QNS <= n_samples && ((size_t)(clip + s_begin) & 15) == 0;
  if (interior) {  // (wave-uniform)
    constexpr int PER = 16 / (int)sizeof(PCM), NVEC = QNS / PER, NV = (NVEC + 255) / 256;
    u32x4_t raw[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = tid + 256 * i;
      if (v < NVEC) raw[i] = *(const u32x4_t*)(clip + s_begin + (long)v * PER);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = tid + 256 * i;
      if (v < NVEC) {
        if constexpr (S16) {
          ((u32x4_t*)samp)[v] = raw[i];
        } else if constexpr (sizeof(PCM) == 2) {  // fp32 staging: converted once here
          float* dst = samp + v * PER;
          f32x4_t lo4, hi4;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int w0 = (int)raw[i][h], w1 = (int)raw[i][2 + h];
            lo4[2 * h] = (float)(short)(w0 & 0xffff) * (1.0f / 32768.0f);
            lo4[2 * h + 1] = (float)(w0 >> 16) * (1.0f / 32768.0f);
            hi4[2 * h] = (float)(short)(w1 & 0xffff) * (1.0f / 32768.0f);
            hi4[2 * h + 1] = (float)(w1 >> 16) * (1.0f / 32768.0f);
          }
          *(f32x4_t*)dst = lo4;
          *(f32x4_t*)(dst + 4) = hi4;
        } else {
          ((u32x4_t*)samp)[v] = raw[i];
        }
      }
    }
  } else {
    for (int s = tid; s < QNS; s += 256) {
      long idx = s_begin + s;
      if (idx < 0) idx = -idx;
      if (idx >= n_samples) idx = 2L * (n_samples - 1) - idx;
      idx = idx < 0 ? 0 : (idx >= n_samples ? n_samples - 1 : idx);
      if constexpr (S16) ((PCM*)samp)[s] = clip[idx];
      else samp[s] = load_pcm<PCM>(clip, idx);
    }
  }
#pragma unroll
  for (int i = 0; i < NTAB; ++i) {
    const int q = tid + 256 * i;
    if (q < QTAB / 4) {
      f32x4_t v4 = tv[i];
      if (S16 && q < NFFT / 4) v4 *= (1.0f / 32768.0f);  // int16 samples meet a window that carries the 2^-15 (exact: same products as x / 32768 * w)
      ((f32x4_t*)tw)[q] = v4;
    }
  }
  __syncthreads();

  // per-lane constants of the cross-lane radix-4
  const float sg1 = (l & 2) ? -1.f : 1.f;  // x + / - its xor-2 partner
  const float sg2 = (l & 1) ? -1.f : 1.f;  // r + / - its xor-1 partner
  const float sg12 = sg1 * sg2;            // what the one-instruction butterflies leave on the result (folded into the W200 rows; applied to m = 0 here)
  const unsigned rot = lane_mask(l == 3);  // lane 3 carries (x1 - x3): times -i before the second exchange
  const unsigned m_self = lane_mask(l == 0), m_odd = lane_mask(l & 1), m_hi = lane_mask(l & 2);

  // ---- pack + window + stage 1 + W200 twiddle: v[m] = A[k1(l)][m] W200^(m k1(l))
  cf v[50];
  {
    const float* xs = samp + f * HOP + 100 * l;
    const int* xs16 = (const int*)samp + (f * HOP + 100 * l) / 2;  // (S16) two samples per word, 8-byte aligned runs
    const float* ws = tw + QTAB_WIN + 100 * l;
    const cf* t200 = (const cf*)(tw + QTAB_TW200) + l;
    static_for<0, 25>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      f32x4_t x4;
      if constexpr (S16) {
        const int w0 = xs16[2 * j], w1 = xs16[2 * j + 1];
        x4 = f32x4_t{(float)(short)(w0 & 0xffff), (float)(w0 >> 16), (float)(short)(w1 & 0xffff), (float)(w1 >> 16)};
      } else {
        x4 = *(const f32x4_t*)(xs + 4 * j);
      }
      const f32x4_t w4 = *(const f32x4_t*)(ws + 4 * j);
      const cf z[2] = {cf{x4[0] * w4[0], x4[1] * w4[1]}, cf{x4[2] * w4[2], x4[3] * w4[3]}};
      static_for<0, 2>([&](auto hc) {
        constexpr int m = 2 * j + decltype(hc)::value;
        const cf zz = z[decltype(hc)::value];
        if constexpr (QUAD_ABLATE & 1) {
          v[m] = zz;
          return;
        }
        // u' = z + sg1 P2(z) = sg1 u;  r' = rot(u') = sg1 r;  y' = r' + sg2 P1(r') = sg1 sg2 y: the lane's sign sg1 sg2 rides in its twiddle row
        const cf u = cf{fmac_qperm<DPP_XOR2>(zz.x, zz.x, sg1), fmac_qperm<DPP_XOR2>(zz.y, zz.y, sg1)};
        const cf r = cf{bsel(rot, u.y, u.x), bsel(rot, fneg(u.x), u.y)};
        const cf y = cf{fmac_qperm<DPP_XOR1>(r.x, r.x, sg2), fmac_qperm<DPP_XOR1>(r.y, r.y, sg2)};
        if constexpr (m == 0) v[m] = cf{y.x * sg12, y.y * sg12};
        else v[m] = cmul(y, t200[4 * m]);
      });
    });
  }

  __builtin_amdgcn_sched_barrier(0);  // (phase fences: the scheduler otherwise hoists the next phase's table reads over this one and spills)
  // ---- stage 2: 50-point FFT over m in registers -> z[k2] = Z[k1(l) + 4 k2]
  cf z[50];
  if constexpr (QUAD_ABLATE & 2) {
    static_for<0, 50>([&](auto kc) { z[decltype(kc)::value] = v[decltype(kc)::value]; });
  } else {
    cf t0[25], t1[25], g0[25], g1[25];
    static_for<0, 25>([&](auto bc) {
      constexpr int bb = decltype(bc)::value;
      t0[bb] = cadd(v[bb], v[25 + bb]);
      const cf d = csub(v[bb], v[25 + bb]);
      if constexpr (bb == 0) {
        t1[bb] = d;
      } else {
        constexpr float wr = QW50_RE[bb], wi = QW50_IM[bb];
        t1[bb] = cmul(d, cf{wr, wi});
      }
    });
    dft25(t0, g0);
    dft25(t1, g1);
    static_for<0, 25>([&](auto ec) {
      constexpr int e = decltype(ec)::value;
      z[2 * e] = g0[e];
      z[2 * e + 1] = g1[e];
    });
  }

  __builtin_amdgcn_sched_barrier(0);
  // ---- unpack + power: P[k2] = 4 |X[k1(l) + 4 k2]|^2 (the 1/4 rides in the mel weights).  Bins k2 = i and 49 - i go together: between them
  // they are the last readers of z[i] and (one step later) z[49 - i], so the z registers die as fast as the P registers are born
  float P[50];
  {
    const cf* t400 = (const cf*)(tw + QTAB_TW400) + l;
    // (m_self: k1 = 0, the partner bin lives in this lane)
    auto bin = [&](auto kc) {
      constexpr int k2 = decltype(kc)::value;
      const cf own = z[k2];
      if constexpr (QUAD_ABLATE & 4) {
        P[k2] = own.x * own.x + own.y * own.y;
        return;
      }
      const cf pa = z[(50 - k2) % 50];
      const cf pb = qperm<DPP_SWAP23>(z[49 - k2]);
      const cf part = cf{bsel(m_self, pa.x, pb.x), bsel(m_self, pa.y, pb.y)};  // Z[200 - k]; its conjugate is (c, -d)
      const float ex = own.x + part.x, ey = own.y - part.y;        // e = zk + conj(zm')
      const float ox = own.y + part.y, oy = part.x - own.x;        // -i (zk - conj(zm')) = (b + d, c - a)
      const cf w = t400[4 * k2];
      const float xr = ex + (w.x * ox - w.y * oy), xi = ey + (w.x * oy + w.y * ox);
      P[k2] = xr * xr + xi * xi;
    };
    static_for<0, 25>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      bin(std::integral_constant<int, i>{});
      bin(std::integral_constant<int, 49 - i>{});
      // (the pin orders the ARITHMETIC: instruction selection keeps chained operations -- LDS reads, fences -- in source order but is free to
      // sink pure arithmetic to its use, which left all 100 twiddle registers and 50 DPP results of this phase live until the mel phase)
      asm volatile("" : "+v"(P[i]), "+v"(P[49 - i]));
      if constexpr (i % 5 == 4) __builtin_amdgcn_sched_barrier(0);
    });
  }

  // ---- mel filters: per-lane partial sums over <= 5 registers, quad sum, this lane keeps m = l mod 4
  const int t = f0 + f;
  const bool live = t < n_frames;
  float vmax = -1e30f;
  {
    const float* mw = tw + QTAB_MEL + l;
    float val[NMEL / 4];
    static_for<0, NMEL / 4>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      float s4[4];
      static_for<0, 4>([&](auto ic) {
        constexpr int m = 4 * j + decltype(ic)::value;
        constexpr int lo = QMEL_K2LO[m], ns = QMEL_NSLOT[m], off = QMEL_OFF[m];
        float a = mw[4 * off] * P[lo];
        static_for<1, ns>([&](auto sc) { a = fmaf(mw[4 * (off + decltype(sc)::value)], P[lo + decltype(sc)::value], a); });
        if constexpr (!(QUAD_ABLATE & 8)) {
          a += qperm<DPP_XOR1>(a);
          a += qperm<DPP_XOR2>(a);
        }
        s4[decltype(ic)::value] = a;
      });
      const float mine = (QUAD_ABLATE & 8) ? (s4[0] + s4[1]) + (s4[2] + s4[3]) : bsel(m_hi, bsel(m_odd, s4[3], s4[2]), bsel(m_odd, s4[1], s4[0]));
      // log10 = log2 * log10(2) on the hardware log2 (v_log_f32, ~1 ulp; the argument is a normal number >= 1e-10): libm's log10f spends a dozen
      // VALU on denormal scaling and a correction step this kernel has no use for (20 values per lane)
      val[j] = __builtin_amdgcn_logf(fmaxf(mine, 1e-10f)) * 0.30102999566398120f;
      if constexpr (j % 4 == 3) __builtin_amdgcn_sched_barrier(0);
    });
    if (live) {
#pragma unroll
      for (int j = 0; j < NMEL / 4; ++j) vmax = fmaxf(vmax, val[j]);
    }
    // The 80 x 64 tile leaves through LDS (over the samples, which every wave has consumed by now) so that each store instruction
    // writes one mel row's 64 frames = 256 contiguous bytes instead of four 64-byte pieces of four rows
    __syncthreads();
    constexpr int OLD_ = QT + 1;  // row stride of the tile (odd: conflict-free row reads)
    float* tile = smem;
#pragma unroll
    for (int j = 0; j < NMEL / 4; ++j) tile[(4 * j + l) * OLD_ + f] = val[j];
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    if (f0 + lane < n_frames) {
      float* ocol = out + (long)b * NMEL * n_frames + f0 + lane;
#pragma unroll
      for (int i = 0; i < NMEL / 4; ++i) {
        const int m = wave * (NMEL / 4) + i;
        ocol[(long)m * n_frames] = tile[m * OLD_ + lane];
      }
    }
  }
  // one ordered-uint atomicMax per workgroup (the tables are dead: their first words carry the four wave maxima)
  vmax = wave_max(vmax);
  if ((tid & 63) == 0) tw[tid >> 6] = vmax;
  __syncthreads();
  if (tid == 0) {
    const float m4 = fmaxf(fmaxf(tw[0], tw[1]), fmaxf(tw[2], tw[3]));
    if (m4 > -1e29f) atomicMax(clipmax + b, f2ord(m4));
  }
}

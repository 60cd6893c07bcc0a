// kern_line32k.hpp — a complex line of N = 2^15 points in ONE workgroup (r02).
//
// 256 KB per line: more than the 160 KB of LDS, less than the 512 KB of vector registers of a CU.  The line lives in the
// registers of a 512-thread workgroup (64 values per thread = two radix-32 butterflies per stage, 32 * 32 * 32 = 2^15) and
// each of the two Stockham exchanges goes through LDS in two halves of 128 KB: outputs q < 16 of every butterfly first, then
// q >= 16 — the index algebra below makes the consumers of a half exactly the butterflies whose own index has the matching
// bit, so each thread owns one consumer per half and picks up its 32 inputs from a run of consecutive LDS slots.
// One HBM round trip per point (16 B) where the four-step routes (xcd-solo) make two; replaces, for this length, the
// reference's S = 5 passes of `stockham_stage.js:17-106` (plan.js:1250-1259).
//
// Stockham, radix 32, Ns_prev = 1, 32, 1024 (kern_lines.hpp conventions): butterfly j (0..1023) of stage s reads
// idx = j + 1024 q, multiplies by w^(q k) of order 32 Ns_prev (k = j mod Ns_prev) and writes blk * 32 Ns_prev + k + q Ns_prev.
//   exchange 1:  (j, q)  -> butterfly j'  = 32 (j mod 32) + q   at slot q'  = j div 32
//   exchange 2:  (j', q) -> butterfly j'' = 32 q + (j' mod 32)  at slot q'' = j' div 32
// Ownership: stage 0 and stage 2 butterflies t and t + 512 (global accesses run along t), stage 1 butterflies
// a = 32 (t div 16) + (t mod 16) and a + 16.
#pragma once
#include "kern_mixed.hpp"

namespace mi355 {

struct Line32kCfg {
  static constexpr int N = 32768, THREADS = 512;
  static constexpr int HALF_ELEMS = 16 * 1024;           // one half of an exchange
  static constexpr int TW1_ELEMS = 31 * 32;              // stage 1 roots [q-1][k], order 1024
  static constexpr int LDS_BYTES = (HALF_ELEMS + TW1_ELEMS) * 8;
  // table buffer handed in by the planner: stage 1 roots, then LO[1024] and HI[32] of order 2^15 (root(m) = HI[m >> 10] * LO[m & 1023])
  static constexpr int TAB_LO = TW1_ELEMS, TAB_HI = TW1_ELEMS + 1024, TAB_ELEMS = TW1_ELEMS + 1024 + 32;
};

template <bool INV>
__global__ void __launch_bounds__(Line32kCfg::THREADS) fft_line32k_kernel(const MixedArgs a) {
  using C = Line32kCfg;
  MI_SMEM_DECL(smem);
  cf* xb = reinterpret_cast<cf*>(smem);
  cf* tw1 = xb + C::HALF_ELEMS;
  const int t = threadIdx.x;
  for (int i = t; i < C::TW1_ELEMS; i += C::THREADS) tw1[i] = a.tw[i];
  const cf* lo = a.tw + C::TAB_LO;
  const cf* hi = a.tw + C::TAB_HI;
  // exchange-1 layout: (j, q16) at j * 16 + ((q16 + j) & 15): the rotation spreads a writer wave (stride 16) over the banks and
  // keeps a reader's 16 lanes (same j, q16 = lane) on one 16-slot row
  const auto slot1 = [](int j, int q16) { return j * 16 + ((q16 + j) & 15); };
  // stage-1 butterflies of this thread
  const int ja = 32 * (t >> 4) + (t & 15), jb = ja + 16;
  __syncthreads();
  for (long long line = blockIdx.x; line < a.lines; line += gridDim.x) {
    const cf* in = a.in + line * (long long)C::N;
    cf* out = a.out + line * (long long)C::N;
    cf v0[32], v1[32];
    // ---- stage 0: butterflies t and t + 512 ----
#pragma unroll
    for (int q = 0; q < 32; ++q) { v0[q] = cswap_if<INV>(in[t + 1024 * q]); v1[q] = cswap_if<INV>(in[t + 512 + 1024 * q]); }
    fft_radix<32>(v0);
    fft_radix<32>(v1);
    // ---- exchange 1 in two halves, consumers ja (q = t mod 16) then jb (q = 16 + t mod 16) ----
    cf w0[32], w1[32];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int q = 0; q < 16; ++q) { xb[slot1(t, q)] = v0[16 * h + q]; xb[slot1(t + 512, q)] = v1[16 * h + q]; }
      __syncthreads();
      // consumer j' = ja (h = 0) / jb (h = 1): inputs from j = (j' div 32) + 32 q' = (t div 16) + 32 q', q16 = t mod 16
#pragma unroll
      for (int qp = 0; qp < 32; ++qp) {
        const cf x = xb[slot1((t >> 4) + 32 * qp, t & 15)];
        if (h == 0) w0[qp] = x; else w1[qp] = x;
      }
      __syncthreads();
    }
    // ---- stage 1: roots of order 1024 from LDS, k = j' mod 32 ----
    {
      const int ka = ja & 31, kb = jb & 31;
#pragma unroll
      for (int q = 1; q < 32; ++q) { w0[q] = cmul(w0[q], tw1[(q - 1) * 32 + ka]); w1[q] = cmul(w1[q], tw1[(q - 1) * 32 + kb]); }
    }
    fft_radix<32>(w0);
    fft_radix<32>(w1);
    // ---- exchange 2 in two halves: (j', q) -> j'' = 32 q + (j' mod 32), slot j' div 32; layout (q16, j') at q16 * 1024 + j' ----
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int q = 0; q < 16; ++q) { xb[q * 1024 + ja] = w0[16 * h + q]; xb[q * 1024 + jb] = w1[16 * h + q]; }
      __syncthreads();
      // consumer j'' = t + 512 h: q = j'' div 32 = (t div 32) + 16 h -> q16 = t div 32; inputs from j' = 32 q'' + (t mod 32)
#pragma unroll
      for (int qp = 0; qp < 32; ++qp) {
        const cf x = xb[(t >> 5) * 1024 + 32 * qp + (t & 31)];
        if (h == 0) v0[qp] = x; else v1[qp] = x;
      }
      __syncthreads();
    }
    // ---- stage 2: roots w^(q k) of order 2^15, k = j'' = t (+ 512): anchors every 8th q from the HI / LO tables, a recurrence between ----
    const auto root = [&](unsigned m) { return cmul(hi[m >> 10], lo[m & 1023u]); };
    const auto twiddle = [&](cf (&x)[32], unsigned k) {
      const cf step = root(k);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        cf w = root(k * (unsigned)(8 * g));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (8 * g + j > 0) x[8 * g + j] = cmul(x[8 * g + j], w);
          if (j < 7) w = cmul(w, step);
        }
      }
    };
    twiddle(v0, (unsigned)t);
    twiddle(v1, (unsigned)t + 512u);
    fft_radix<32>(v0);
    fft_radix<32>(v1);
#pragma unroll
    for (int q = 0; q < 32; ++q) {
      out[t + 1024 * q] = cswap_if<INV>(v0[q] * a.scale);
      out[t + 512 + 1024 * q] = cswap_if<INV>(v1[q] * a.scale);
    }
  }
}

}  // namespace mi355

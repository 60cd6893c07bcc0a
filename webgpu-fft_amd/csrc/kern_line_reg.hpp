// kern_line_reg.hpp — complex lines of N = 2^13, 2^14, 2^15 points held in the REGISTERS of one workgroup (r02).
//
// N / 64 threads, 64 values per thread (64 / R butterflies of radix R per stage), three Stockham stages R0 * R1 * R2 with
// R0 = 32.  Each of the two exchanges goes through LDS in two halves of N/2 values: outputs q < R/2 of every butterfly first,
// then q >= R/2 — the index algebra below makes the consumers of a half exactly the butterflies whose own index has the
// matching bit, so every thread owns as many consumers in one half as in the other and picks up a butterfly's inputs from a run
// of consecutive LDS slots.  For 2^15 (256 KB per line, more than the LDS) this is what makes a single-workgroup line possible
// at all — one HBM round trip where the four-step routes (xcd-solo) make two; replaces, for these lengths, the reference's
// S = 5 passes of `stockham_stage.js:17-106` (plan.js:1250-1259).
//
// Stockham conventions of kern_lines.hpp: butterfly j of stage s (NB_s = N / R_s of them, Ns_prev = 1, R0, R0 R1) reads
// idx = j + NB_s q, multiplies by w^(q k) of order R_s Ns_prev (k = j mod Ns_prev) and writes blk R_s Ns_prev + k + q Ns_prev.
//   exchange 1:  (j, q)  -> butterfly j'  = R0 (j mod R2) + q     at slot q'  = j div R2          (halves: q < R0/2, j' mod R0 < R0/2)
//   exchange 2:  (j', q) -> butterfly j'' = R0 q + (j' mod R0)    at slot q'' = j' div R0         (halves: q < R1/2, j'' < NB_2 / 2)
// Ownership: stage 0 and stage 2 butterflies t + m THREADS (global accesses run along t); stage 1: consumer e = t + i THREADS of
// half A is j' = R0 (e div 16) + (e mod 16), its partner in half B j' + 16.
#pragma once
#include "kern_mixed.hpp"

namespace mi355 {

template <int N_, int R1_, int R2_>
struct LineRegCfg {
  static constexpr int N = N_, R0 = 32, R1 = R1_, R2 = R2_;
  static_assert(R0 * R1 * R2 == N && (R1 == 16 || R1 == 32) && (R2 == 8 || R2 == 16 || R2 == 32), "radix plan");
  static constexpr int VPT = 64, THREADS = N / VPT;
  static constexpr int NB0 = N / R0, NB1 = N / R1, NB2 = N / R2;
  static constexpr int BPT0 = VPT / R0, BPT1 = VPT / R1, BPT2 = VPT / R2;
  static constexpr int HALF_ELEMS = N / 2;                 // one half of an exchange
  static constexpr int TW1_ELEMS = (R1 - 1) * R0;          // stage 1 roots [q-1][k], order R0 R1
  static constexpr int LDS_BYTES = (HALF_ELEMS + TW1_ELEMS) * 8;
  // table buffer handed in by the planner: stage 1 roots, then LO[1024] and HI[N/1024] of order N (root(m) = HI[m >> 10] * LO[m & 1023])
  static constexpr int TAB_LO = TW1_ELEMS, TAB_HI = TW1_ELEMS + 1024, TAB_ELEMS = TW1_ELEMS + 1024 + N / 1024;
  static_assert((NB2 / 2) % THREADS == 0 && (NB1 / 2) % THREADS == 0, "halves split the owners evenly");
};

template <class C, bool INV>
__global__ void __launch_bounds__(C::THREADS) fft_line_reg_kernel(const MixedArgs a) {
  constexpr int R0 = C::R0, R1 = C::R1, R2 = C::R2, TH = C::THREADS;
  MI_SMEM_DECL(smem);
  cf* xb = reinterpret_cast<cf*>(smem);
  cf* tw1 = xb + C::HALF_ELEMS;
  const int t = threadIdx.x;
  for (int i = t; i < C::TW1_ELEMS; i += TH) tw1[i] = a.tw[i];
  const cf* lo = a.tw + C::TAB_LO;
  const cf* hi = a.tw + C::TAB_HI;
  // exchange-1 layout: (j, q16) at j * 16 + ((q16 + j) & 15): the rotation spreads a writer wave (stride 16) over the banks and
  // keeps a reader's 16 lanes (same j, q16 = lane) on one 16-slot row
  const auto slot1 = [](int j, int q16) { return j * 16 + ((q16 + j) & 15); };
  const auto root = [&](unsigned m) { return cmul(hi[m >> 10], lo[m & 1023u]); };
  __syncthreads();
  for (long long line = blockIdx.x; line < a.lines; line += gridDim.x) {
    const cf* in = a.in + line * (long long)C::N;
    cf* out = a.out + line * (long long)C::N;
    cf v[C::VPT], w[C::VPT];
    // ---- stage 0: butterflies t + m THREADS ----
#pragma unroll
    for (int m = 0; m < C::BPT0; ++m) {
#pragma unroll
      for (int q = 0; q < R0; ++q) v[m * R0 + q] = cswap_if<INV>(in[t + m * TH + C::NB0 * q]);
    }
#pragma unroll
    for (int m = 0; m < C::BPT0; ++m) {
      cf x[R0];
#pragma unroll
      for (int q = 0; q < R0; ++q) x[q] = v[m * R0 + q];
      fft_radix<R0>(x);
#pragma unroll
      for (int q = 0; q < R0; ++q) v[m * R0 + q] = x[q];
    }
    // ---- exchange 1, halves h: consumers j' with (j' mod 32) div 16 == h; stage-1 butterfly b = 2 i + h of this thread ----
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int m = 0; m < C::BPT0; ++m) {
#pragma unroll
        for (int q = 0; q < 16; ++q) xb[slot1(t + m * TH, q)] = v[m * R0 + 16 * h + q];
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < C::BPT1 / 2; ++i) {
        const int e = t + i * TH, c = e >> 4, r = e & 15;      // j' = 32 c + r (+ 16 h); sources j = c + R2 q', q16 = r
#pragma unroll
        for (int qp = 0; qp < R1; ++qp) w[(2 * i + h) * R1 + qp] = xb[slot1(c + R2 * qp, r)];
      }
      __syncthreads();
    }
    // ---- stage 1: roots of order R0 R1 from LDS, k = j' mod 32 ----
#pragma unroll
    for (int b = 0; b < C::BPT1; ++b) {
      const int e = t + (b >> 1) * TH, k = (e & 15) + 16 * (b & 1);
      cf x[R1];
#pragma unroll
      for (int q = 0; q < R1; ++q) x[q] = q ? cmul(w[b * R1 + q], tw1[(q - 1) * R0 + k]) : w[b * R1];
      fft_radix<R1>(x);
#pragma unroll
      for (int q = 0; q < R1; ++q) w[b * R1 + q] = x[q];
    }
    // ---- exchange 2, halves h: (j', q) -> j'' = 32 q + (j' mod 32), slot j' div 32; layout (qh, j') at qh * NB1 + j' ----
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int b = 0; b < C::BPT1; ++b) {
        const int e = t + (b >> 1) * TH, jp = 32 * (e >> 4) + (e & 15) + 16 * (b & 1);
#pragma unroll
        for (int q = 0; q < R1 / 2; ++q) xb[q * C::NB1 + jp] = w[b * R1 + (R1 / 2) * h + q];
      }
      __syncthreads();
#pragma unroll
      for (int mm = 0; mm < C::BPT2 / 2; ++mm) {
        const int m = h * (C::BPT2 / 2) + mm, jpp = t + m * TH;     // half h: j'' in [h NB2/2, (h+1) NB2/2)
        const int qh = (jpp >> 5) - h * (R1 / 2);
#pragma unroll
        for (int qp = 0; qp < R2; ++qp) v[m * R2 + qp] = xb[qh * C::NB1 + 32 * qp + (jpp & 31)];
      }
      __syncthreads();
    }
    // ---- stage 2: roots w^(q k) of order N, k = j'': anchors every 8th q from the HI / LO tables, a recurrence between ----
#pragma unroll
    for (int m = 0; m < C::BPT2; ++m) {
      const unsigned k = (unsigned)(t + m * TH);
      cf x[R2];
#pragma unroll
      for (int q = 0; q < R2; ++q) x[q] = v[m * R2 + q];
      const cf step = root(k);
#pragma unroll
      for (int g = 0; g < R2 / 8; ++g) {
        cf ww = root(k * (unsigned)(8 * g));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (8 * g + j > 0) x[8 * g + j] = cmul(x[8 * g + j], ww);
          if (j < 7) ww = cmul(ww, step);
        }
      }
      fft_radix<R2>(x);
#pragma unroll
      for (int q = 0; q < R2; ++q) out[k + (unsigned)(C::NB2 * q)] = cswap_if<INV>(x[q] * a.scale);
    }
  }
}

}  // namespace mi355

// kern_xcd_real.hpp — XCD-fused four-step r2c: a real line of N = N1*N2 points to its N/2+1 packed bins in ONE persistent
// launch, 16 B of fabric traffic per real point (the half-length route of DESIGN.md 4.4 pays 24: a complex FFT of the
// packed line, then a separate split pass).  Replaces the reference's expand -> full complex FFT -> pack
// (`r2c.js:1518-1557`) for dense power-of-two lines.
//
// x[n1*N2 + n2] -> X[k1 + N1*k2]:
//   phase A  the N2 real columns are read two at a time as ONE complex column (a real row is read as N2/2 complex
//            numbers: the load code of the c2c PASS_A tiles, 128-B segments), the length-N1 complex FFT runs in LDS, and the
//            two real columns' spectra are separated before they leave the workgroup:
//              Y[k1][2c] = (Z[k1] + conj Z[N1-k1])/2,  Y[k1][2c+1] = (Z[k1] - conj Z[N1-k1])/(2i),   k1 = 0..N1/2
//            (columns of a real matrix have Hermitian spectra: rows k1 > N1/2 of Y are never formed).  W = Y[0..N1/2][N2].
//   group barrier (kern_xcd.hpp)
//   phase B  rows k1 = 0..N1/2: four-step roots e^{-2 pi i k1 n2/N}, length-N2 complex FFT, transposed store.  Bins
//            k = k1 + N1*k2 <= N/2 are stored as they are; the others are stored conjugated at N-k, which is where the rows
//            that were never formed would have put them.  Rows 0 and N1/2 only contribute their first half.
#pragma once
#include "kern_xcd.hpp"

namespace mi355 {

typedef float cf4 __attribute__((ext_vector_type(4)));

template <class CA, class CB>
__global__ void __launch_bounds__(CA::THREADS) fft_xcd_r2c_kernel(const XcdFusedArgs f) {
  static_assert(CA::THREADS == CB::THREADS, "both passes run in the same workgroup");
  static_assert(CA::IN_COL && CA::OUT_COL && !CB::IN_COL && CB::OUT_COL, "PASS_A then PASS_B");
  static_assert(CA::NSTAGES >= 2 && !CA::SWAP_IN && !CB::SWAP_OUT, "forward only; the separation needs the LDS line buffer");
  MI_SMEM_DECL(smem);
  cf* lds = reinterpret_cast<cf*>(smem);
  constexpr int DATA = CA::DATA_ELEMS > CB::DATA_ELEMS ? CA::DATA_ELEMS : CB::DATA_ELEMS;
  using TB = XcdTables<CA, CB>;
  cf* tw_a = lds + DATA;
  cf* tw_b = TB::SHARED ? tw_a : tw_a + CA::TW_ELEMS;
  unsigned* s_words = reinterpret_cast<unsigned*>(tw_a + TB::ELEMS);
  const int t = threadIdx.x;
  for (int i = t; i < CA::TW_ELEMS; i += CA::THREADS) tw_a[i] = f.tw_a[i];
  if constexpr (!TB::SHARED) { for (int i = t; i < CB::TW_ELEMS; i += CA::THREADS) tw_b[i] = f.tw_b[i]; }

  if (!xcd_register(f.ctl, f.split, f.spin_limit, f.sticky_error, s_words)) return;
  const unsigned gslot = s_words[0], rank = s_words[1], gsize = s_words[2], gidx = s_words[4], groups = s_words[5];

  constexpr int N1 = CA::N, N2 = CB::N, ROWS = N1 / 2 + 1;
  const long long wsize = (long long)ROWS * N2;               // complex elements of one workspace slot
  LineArgs aa{}, ab{};
  aa.tw = f.tw_a; aa.num_tiles = (N2 / 2) / CA::T; aa.num_lines = N2 / 2;
  aa.in_S = N2 / 2; aa.in_outer_stride = f.N / 2; aa.out_S = N2 / 2; aa.out_outer_stride = f.N / 2; aa.scale = 1.0f; aa.fs_group = 1;
  ab.tw = f.tw_b; ab.num_tiles = (ROWS + CB::T - 1) / CB::T; ab.num_lines = ROWS;
  ab.in_S = 1; ab.in_outer_stride = N2; ab.out_S = N1; ab.out_outer_stride = f.N; ab.scale = f.scale; ab.fs_group = N1;
  const bool two_slots = f.slots != 1u;
  cf* const W0 = f.wslots + (size_t)((two_slots ? 2u : 1u) * gslot) * (size_t)wsize;
  unsigned k = 0;
  for (long long tr = gidx; tr < f.num_transforms; tr += groups, ++k) {
    cf* const W = W0 + (size_t)(two_slots ? (k & 1u) : 0u) * (size_t)wsize;
    // ---- phase A: pairs of real columns as complex columns, FFT, separation, rows 0..N1/2 of W ----
    aa.in = f.in + tr * f.in_pitch; aa.out = W;
    for (long long i = 0;; ++i) {
      const long long tile = xcd_tile<CA>(i, rank, gsize);
      if (tile - (i % PairOf<CA>::C) >= aa.num_tiles) break;
      if (tile >= aa.num_tiles) continue;
      cf v[CA::E];
      stage_read<CA, 0, PairOf<CA, false>::NT>(v, aa, tile, t, lds);
      stage_compute_write<CA, 0>(v, aa, tile, t, lds, tw_a, nullptr);
      __syncthreads();
      stage_read<CA, 1>(v, aa, tile, t, lds);
      __syncthreads();
      stage_compute_write<CA, 1, false, true>(v, aa, tile, t, lds, tw_a, nullptr);
      if constexpr (CA::NSTAGES == 3) {
        __syncthreads();
        stage_read<CA, 2>(v, aa, tile, t, lds);
        __syncthreads();
        stage_compute_write<CA, 2, false, true>(v, aa, tile, t, lds, tw_a, nullptr);
      }
      __syncthreads();
      // Z[k1][c] sits at lds[k1*T + c]; one lane per (k1, c) writes the two separated values as one 16-byte store:
      // T consecutive lanes cover 2T consecutive complex columns of row k1
      cf* const wt = W + tile * (2 * CA::T);
      for (int p = t; p < ROWS * CA::T; p += CA::THREADS) {
        const int k1 = p / CA::T, c = p - k1 * CA::T;
        const cf a = lds[k1 * CA::T + c];
        const cf b = lds[((N1 - k1) & (N1 - 1)) * CA::T + c];
        cf4 o;
        o.x = 0.5f * (a.x + b.x); o.y = 0.5f * (a.y - b.y);        // even column
        o.z = 0.5f * (a.y + b.y); o.w = 0.5f * (b.x - a.x);        // odd column: -i/2 * (a - conj b)
        *reinterpret_cast<cf4*>(wt + (size_t)k1 * N2 + 2 * c) = o;
      }
      __syncthreads();   // LDS is re-used by the next tile
    }
    xcd_arrive(&f.ctl->bar[gslot][0]);
    if (!xcd_wait(&f.ctl->bar[gslot][0], (k + 1u) * gsize, f.spin_limit, f.sticky_error, &s_words[6])) return;
    // ---- phase B: rows 0..N1/2, four-step roots, row FFT, transposed store with the Hermitian mirror ----
    ab.in = W;
    cf* const po = f.out + tr * f.out_pitch;
    for (long long i = 0;; ++i) {
      const long long tile = xcd_tile<CB>(i, rank, gsize);
      if (tile - (i % PairOf<CB>::C) >= ab.num_tiles) break;
      if (tile >= ab.num_tiles) continue;
      cf v[CB::E];
      stage_read<CB, 0>(v, ab, tile, t, lds);
      {
        int line, u; thread_map<CB, 0>(t, line, u);
        fourstep_apply_chain<CB>(v, f, (unsigned)(tile * CB::T + line), u);
      }
      constexpr int LASTS = CB::NSTAGES - 1;
      stage_compute_write<CB, 0>(v, ab, tile, t, lds, tw_b, nullptr);
      __syncthreads();
      stage_read<CB, 1>(v, ab, tile, t, lds);
      if constexpr (CB::NSTAGES == 3) {
        __syncthreads();
        stage_compute_write<CB, 1>(v, ab, tile, t, lds, tw_b, nullptr);
        __syncthreads();
        stage_read<CB, 2>(v, ab, tile, t, lds);
      }
      // last stage with the mirrored sink (stage_compute_write's arithmetic, different addresses)
      {
        using I = StageInfo<CB, LASTS>;
        static_assert(I::R % 2 == 0, "the last radix splits the bins k2 < N2/2 from the rest");
        int line, u; thread_map<CB, LASTS>(t, line, u);
        const int k1 = (int)(tile * CB::T) + line;
        const bool live = k1 < ROWS;
        const bool edge = k1 == 0 || k1 == N1 / 2;            // rows whose second half duplicates their first
        const int voff = k1 + u * N1;
#pragma unroll
        for (int b = 0; b < I::NB; ++b) {
          const int j = u + b * CB::TPL;
          const int kk = j % I::NSP;
          cf w[I::R];
#pragma unroll
          for (int q = 0; q < I::R; ++q) w[q] = v[b * I::R + q];
#pragma unroll
          for (int q = 1; q < I::R; ++q) w[q] = cmul(w[q], tw_b[I::TW_OFF + (q - 1) * I::NSP + kk]);
          fft_radix<I::R>(w);
#pragma unroll
          for (int q = 0; q < I::R; ++q) {
            cf r = w[q];
            if (ab.scale != 1.0f) r = r * ab.scale;
            const int uni = (b * CB::TPL + q * I::NSP) * N1;     // uniform part of k = k1 + N1*k2, k2 = j + q*(N2/R)
            if (q < I::R / 2) {
              if (live) st_stream<PairOf<CB, false>::NT>(po + (uni + voff), r);
            } else {
              // k > N/2 (or = N/2 for k1 = 0, k2 = N2/2, which is its own mirror and stays unconjugated)
              const bool nyq = k1 == 0 && q == I::R / 2 && j == 0;
              cf m; m.x = r.x; m.y = nyq ? r.y : -r.y;
              if (live && (!edge || nyq)) st_stream<PairOf<CB, false>::NT>(po + ((int)f.N - uni - voff), m);
            }
          }
        }
      }
      __syncthreads();
    }
    if (!two_slots) {
      xcd_arrive(&f.ctl->bar[gslot][1]);
      if (!xcd_wait(&f.ctl->bar[gslot][1], (k + 1u) * gsize, f.spin_limit, f.sticky_error, &s_words[6])) return;
    }
  }
}

}  // namespace mi355

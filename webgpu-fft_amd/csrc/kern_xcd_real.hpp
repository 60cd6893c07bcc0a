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

// The packed side of the real transforms has rows of N/2 + 1 bins: every 128-byte run starts on an odd 8-byte offset and shares its first and last
// cache line with the neighbouring tiles' runs.  r03 (profiles/r03_real_packed_side_nt_ab.log): temporal accesses on that side let those partial lines
// meet in the L2.
#ifndef MI355_XCD_R2C_NT_OUT
#define MI355_XCD_R2C_NT_OUT 0   /* temporal: r2c 2^17 265 -> 314, 2^18 257 -> 338, 2^19 297 -> 358, 2^20 318 -> 331 G real points/s */
#endif
#ifndef MI355_XCD_C2R_NT_IN
#define MI355_XCD_C2R_NT_IN 1    /* nontemporal stays: temporal loads measured 0 ... -5 % on the LDS-resident c2r kernels (the register-tile 2^22 kernel, whose tiles are larger, gains 6 % and uses them) */
#endif

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

  if (f.solo) {   // one workgroup per transform (kern_xcd.hpp): no registration, no cross-workgroup barrier
    if (t == 0) { s_words[0] = blockIdx.x; s_words[1] = 0; s_words[2] = 1; s_words[3] = 1; s_words[4] = blockIdx.x; s_words[5] = gridDim.x; }
    __syncthreads();
  } else if (!xcd_register(f.ctl, f.split, f.spin_limit, f.sticky_error, s_words)) return;
  const unsigned gslot = s_words[0], rank = s_words[1], gsize = s_words[2], gidx = s_words[4], groups = s_words[5];

  constexpr int N1 = CA::N, N2 = CB::N, ROWS = N1 / 2 + 1;
  const long long wsize = (long long)ROWS * N2;               // complex elements of one workspace slot
  LineArgs aa{}, ab{};
  aa.tw = f.tw_a; aa.num_tiles = (N2 / 2) / CA::T; aa.num_lines = N2 / 2;
  aa.in_S = N2 / 2; aa.in_outer_stride = f.N / 2; aa.out_S = N2 / 2; aa.out_outer_stride = f.N / 2; aa.scale = 1.0f; aa.fs_group = 1;
  ab.tw = f.tw_b; ab.num_tiles = (ROWS + CB::T - 1) / CB::T; ab.num_lines = ROWS;
  ab.in_S = 1; ab.in_outer_stride = N2; ab.out_S = N1; ab.out_outer_stride = f.N; ab.scale = f.scale; ab.fs_group = N1;
  const bool two_slots = f.slots != 1u && !f.solo;
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
    if (f.solo) xcd_local_handoff();
    else {
      xcd_arrive(&f.ctl->bar[gslot][0]);
      if (!xcd_wait(&f.ctl->bar[gslot][0], (k + 1u) * gsize, f.spin_limit, f.sticky_error, &s_words[6])) return;
    }
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
              if (live) st_stream<PairOf<CB, false>::NT && MI355_XCD_R2C_NT_OUT>(po + (uni + voff), r);
            } else {
              // k > N/2 (or = N/2 for k1 = 0, k2 = N2/2, which is its own mirror and stays unconjugated)
              const bool nyq = k1 == 0 && q == I::R / 2 && j == 0;
              cf m; m.x = r.x; m.y = nyq ? r.y : -r.y;
              if (live && (!edge || nyq)) st_stream<PairOf<CB, false>::NT && MI355_XCD_R2C_NT_OUT>(po + ((int)f.N - uni - voff), m);
            }
          }
        }
      }
      __syncthreads();
    }
    if (f.solo) __syncthreads();
    else if (!two_slots) {
      xcd_arrive(&f.ctl->bar[gslot][1]);
      if (!xcd_wait(&f.ctl->bar[gslot][1], (k + 1u) * gsize, f.spin_limit, f.sticky_error, &s_words[6])) return;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// c2r: N/2+1 packed bins X of a real line -> the N real samples, one persistent launch, 16 B of traffic per real sample.
// Replaces the reference's Hermitian unpack -> full complex inverse -> real part (`c2r.js:1743-1763`, `real_complex.js:116-201`)
// and this library's half-length route (c2r_pre_kernel + inverse c2c) for dense power-of-two lines.
//
// x real  =>  x = IFFT(X) = FFT(conj X): a FORWARD four-step over the Hermitian sequence Xt = conj(X_full), input index
// n = n1*N2 + n2 (bins), output index k1 + N1*k2 (samples).
//   phase A  columns n2 = 0..N2/2 only (column N2-n2 is determined by column n2).  The upper half of a column (n1 < N1/2)
//            is read conjugated from the stored bins; the lower half is the UNconjugated upper half of column N2-n2 walked
//            backwards (Xt[n] = conj Xt[N-n]); column 0 mirrors onto itself one row lower.  Length-N1 complex FFT in LDS,
//            stored as W[k1][n2], n2 <= N2/2.
//   phase B  every row of W, times the four-step roots, is a Hermitian sequence in n2 whose transform is real: rows 2p and
//            2p+1 are extended to full length (z[N2-n2] = conj z[n2]) and packed as z_a + i*z_b while they are loaded, one
//            complex row FFT produces samples k1 = 2p (real part) and 2p+1 (imaginary part), which are ADJACENT in the
//            output: the transposed store of the c2c pass B with N1/2 complex rows writes the real line directly.
template <class CA, class CB>
__global__ void __launch_bounds__(CA::THREADS) fft_xcd_c2r_kernel(const XcdFusedArgs f) {
  static_assert(CA::THREADS == CB::THREADS, "both passes run in the same workgroup");
  static_assert(CA::IN_COL && CA::OUT_COL && !CB::IN_COL && CB::OUT_COL, "PASS_A then PASS_B");
  static_assert(!CA::SWAP_IN && !CB::SWAP_OUT, "the conjugation is part of the loads");
  static_assert(StageInfo<CA, 0>::NB == 1 && StageInfo<CB, 0>::NB == 1 && StageInfo<CA, 0>::R % 2 == 0 && StageInfo<CB, 0>::R % 2 == 0,
                "first radix = values per thread, even");
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

  if (f.solo) {   // one workgroup per transform (kern_xcd.hpp): no registration, no cross-workgroup barrier
    if (t == 0) { s_words[0] = blockIdx.x; s_words[1] = 0; s_words[2] = 1; s_words[3] = 1; s_words[4] = blockIdx.x; s_words[5] = gridDim.x; }
    __syncthreads();
  } else if (!xcd_register(f.ctl, f.split, f.spin_limit, f.sticky_error, s_words)) return;
  const unsigned gslot = s_words[0], rank = s_words[1], gsize = s_words[2], gidx = s_words[4], groups = s_words[5];

  constexpr int N1 = CA::N, N2 = CB::N, COLS = N2 / 2 + 1, WP = N2 / 2 + 16;   // live columns, row pitch of W
  const long long wsize = (long long)N1 * WP;
  LineArgs aa{}, ab{};
  aa.tw = f.tw_a; aa.num_tiles = (COLS + CA::T - 1) / CA::T; aa.num_lines = COLS;
  aa.out_S = WP; aa.out_outer_stride = wsize; aa.scale = 1.0f; aa.fs_group = 1;
  ab.tw = f.tw_b; ab.num_tiles = (N1 / 2) / CB::T; ab.num_lines = N1 / 2;
  ab.out_S = N1 / 2; ab.out_outer_stride = f.N / 2; ab.scale = f.scale; ab.fs_group = N1;
  const bool two_slots = f.slots != 1u && !f.solo;
  cf* const W0 = f.wslots + (size_t)((two_slots ? 2u : 1u) * gslot) * (size_t)wsize;
  unsigned k = 0;
  for (long long tr = gidx; tr < f.num_transforms; tr += groups, ++k) {
    cf* const W = W0 + (size_t)(two_slots ? (k & 1u) : 0u) * (size_t)wsize;
    // ---- phase A ----
    const cf* const X = f.in + tr * f.in_pitch;
    aa.out = W;
    for (long long i = 0;; ++i) {
      const long long tile = xcd_tile<CA>(i, rank, gsize);
      if (tile - (i % PairOf<CA>::C) >= aa.num_tiles) break;
      if (tile >= aa.num_tiles) continue;
      cf v[CA::E];
      {
        using I = StageInfo<CA, 0>;
        int line, u; thread_map<CA, 0>(t, line, u);
        int n2 = (int)(tile * CA::T) + line;
        if (n2 > N2 / 2) n2 = N2 / 2;                       // padding lines of the last tile re-read its live column
        const int up = u * N2 + n2;                          // n1 = u + q*(N1/R): upper half, conjugated
        const int lo = n2 ? (N1 - 1 - u) * N2 + (N2 - n2) : (N1 - u) * N2;   // lower half: mirrored element, as stored
#pragma unroll
        for (int q = 0; q < I::R; ++q) {
          const int step = q * (N1 / I::R) * N2;             // uniform
          if (q < I::R / 2) { const cf x = ld_stream<PairOf<CA, false>::NT && MI355_XCD_C2R_NT_IN>(X + (up + step)); v[q] = cf{x.x, -x.y}; }
          else v[q] = ld_stream<PairOf<CA, false>::NT && MI355_XCD_C2R_NT_IN>(X + (lo - step));
        }
      }
      stage_compute_write<CA, 0>(v, aa, tile, t, lds, tw_a, nullptr);
      __syncthreads();
      stage_read<CA, 1>(v, aa, tile, t, lds);
      __syncthreads();
      stage_compute_write<CA, 1>(v, aa, tile, t, lds, tw_a, nullptr);
      if constexpr (CA::NSTAGES == 3) {
        __syncthreads();
        stage_read<CA, 2>(v, aa, tile, t, lds);
        __syncthreads();
        stage_compute_write<CA, 2>(v, aa, tile, t, lds, tw_a, nullptr);
      }
      __syncthreads();
    }
    if (f.solo) xcd_local_handoff();
    else {
      xcd_arrive(&f.ctl->bar[gslot][0]);
      if (!xcd_wait(&f.ctl->bar[gslot][0], (k + 1u) * gsize, f.spin_limit, f.sticky_error, &s_words[6])) return;
    }
    // ---- phase B ----
    ab.out = f.out + tr * f.out_pitch;
    for (long long i = 0;; ++i) {
      const long long tile = xcd_tile<CB>(i, rank, gsize);
      if (tile - (i % PairOf<CB>::C) >= ab.num_tiles) break;
      if (tile >= ab.num_tiles) continue;
      cf v[CB::E];
      {
        using I = StageInfo<CB, 0>;
        constexpr int H = I::R / 2, STR = N2 / I::R;
        int line, u; thread_map<CB, 0>(t, line, u);
        const unsigned ka = 2u * (unsigned)(tile * CB::T + line), kb = ka + 1u;
        const cf* const wa = W + (size_t)ka * WP;
        const cf* const wb = wa + WP;
        const auto root = [&](unsigned m) { return cmul(f.tw_hi[m >> f.fs_shift], f.tw_lo[m & f.fs_lo_mask]); };
        const cf step_a = root(ka * (unsigned)STR), step_b = root(kb * (unsigned)STR);
        // first half of the thread's elements: n2 = u + q*STR < N2/2, read as stored.  Roots e^{-2 pi i k1 n2/N}: exact
        // lookups every 8th element, the recurrence with the per-row step between
#pragma unroll
        for (int g = 0; g < H; g += 8) {
          cf ra = root(ka * (unsigned)(u + g * STR)), rb = root(kb * (unsigned)(u + g * STR));
#pragma unroll
          for (int q = g; q < g + 8 && q < H; ++q) {
            const cf a = cmul(wa[u + q * STR], ra), b = cmul(wb[u + q * STR], rb);
            v[q] = cf{a.x - b.y, a.y + b.x};
            ra = cmul(ra, step_a); rb = cmul(rb, step_b);
          }
          MI_SCHED_FENCE();
        }
        // second half: n2 = u + (H + q')*STR > N2/2 (= N2/2 for u = 0, q' = 0) is the conjugate of column
        // m = N2 - n2 = (STR - u) + q*STR with q = H - 1 - q'
#pragma unroll
        for (int g = 0; g < H; g += 8) {
          cf ra = root(ka * (unsigned)((STR - u) + g * STR)), rb = root(kb * (unsigned)((STR - u) + g * STR));
#pragma unroll
          for (int q = g; q < g + 8 && q < H; ++q) {
            const int m = (STR - u) + q * STR;
            const cf a = cmul(wa[m], ra), b = cmul(wb[m], rb);
            v[I::R - 1 - q] = cf{a.x + b.y, b.x - a.y};     // conj(a) + i*conj(b)
            ra = cmul(ra, step_a); rb = cmul(rb, step_b);
          }
          MI_SCHED_FENCE();
        }
      }
      stage_compute_write<CB, 0, PairOf<CB, false>::NT>(v, ab, tile, t, lds, tw_b, nullptr);
      __syncthreads();
      stage_read<CB, 1>(v, ab, tile, t, lds);
      __syncthreads();
      stage_compute_write<CB, 1, PairOf<CB, false>::NT>(v, ab, tile, t, lds, tw_b, nullptr);
      if constexpr (CB::NSTAGES == 3) {
        __syncthreads();
        stage_read<CB, 2>(v, ab, tile, t, lds);
        __syncthreads();
        stage_compute_write<CB, 2, PairOf<CB, false>::NT>(v, ab, tile, t, lds, tw_b, nullptr);
      }
      __syncthreads();
    }
    if (f.solo) __syncthreads();
    else if (!two_slots) {
      xcd_arrive(&f.ctl->bar[gslot][1]);
      if (!xcd_wait(&f.ctl->bar[gslot][1], (k + 1u) * gsize, f.spin_limit, f.sticky_error, &s_words[6])) return;
    }
  }
}

}  // namespace mi355

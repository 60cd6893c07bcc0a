// kern_regtile.hpp — four-step passes over 2048-point sides on REGISTER-resident tiles of 16 lines (r03).
//
// Why: a 16 x 2048 tile is 2^15 points = 256 KB, more than the 160 KB of LDS, so the LDS-resident PASS_A / PASS_B kernels of
// kern_lines.hpp take 2048-point sides as 8-line tiles — 64-byte segments on their strided side, which the fabric moves 15-20 %
// slower than 128-byte ones (profiles/r02_size_sweep.log: c2c 2^21 163, 2^22 112 GPoints/s against 176-190 for 2^18 .. 2^20).
// Here the tile lives in the registers of a 512-thread workgroup (64 values per thread, the scheme of kern_line32k.hpp) and only
// the ONE exchange of the plan 2048 = 64 * 32 goes through LDS, in two halves of 128 KB:
//   stage 0+1  a thread owns (line, u), u = 0..31: inputs x[u + 32 m], m = 0..63, and forms their 64-point DFT in registers
//              (32 radix-2 butterflies, the odd half times e^{-2 pi i m/64} — literals —, two radix-32 DFTs): output p of producer
//              u is element 64 u + p of the Stockham intermediate;
//   exchange   consumer butterfly j2 = p (0..63) of the same line reads element j2 + 64 q2 = output p of producer u = q2: the
//              outputs p < 32 of every producer travel first, then p >= 32, so a thread owns one consumer per half;
//   stage 2    radix 32 with roots e^{-2 pi i q2 j2/2048} from an LDS table, outputs at j2 + 64 q.
// Replaces, for these sides, the reference's per-stage passes (src/kernels/stockham_stage.js:17-106) and its transposes
// (src/kernels/transpose.js:1-51, src/plan.js:375-384) exactly as the LDS-resident passes do: the transposes are the address maps.
//
// Thread maps (every global access is a run of 16 lanes over 16 adjacent lines = 128 bytes, or of 32 lanes along a row):
//   column side:  line = t mod 16, u (or jj) = t div 16        rows as input:  line = t div 32, u = t mod 32
// Exchange layout: slot(line, p', u) = u * 512 + p' * 16 + ((line + u) mod 16), p' = p mod 32.  Writers (fixed line-or-u, 16 lanes
// over the other) hit 16 different 8-byte slots mod 16, readers (32 lanes = 16 lines x 2 adjacent p') 32 different ones mod 32:
// no LDS bank conflicts on either side for either input map.
#pragma once
#include "kern_xcd_real.hpp"
#include "kern_xcd_res.hpp"   // sgpr_base(): uniform 64-bit bases stay in SGPRs, lanes add a 32-bit offset

namespace mi355 {

#ifndef MI355_RT_NT_OUT
#define MI355_RT_NT_OUT 0   /* same-box A/B (profiles/r03_regtile_ab.log): temporal output stores r2c 2^22 272 vs 253 G real points/s (its rows start on odd 8-byte
                               offsets: out pitch N/2 + 1, so neighbouring tiles' partial lines have to meet in the L2), c2c 2^22 171 vs 167 */
#endif
#ifndef MI355_RT_EXP_SKIP_LAST
#define MI355_RT_EXP_SKIP_LAST 0   /* timing-only: r2c phase B without its 65th row tile (row N1/2 alone) */
#endif
#ifndef MI355_RT_EXP_ALIGN
#define MI355_RT_EXP_ALIGN 0       /* timing-only: r2c output rows on a pitch of N/2 (128-byte aligned segments) */
#endif
#ifndef MI355_RT_EXP_ASC
#define MI355_RT_EXP_ASC 0         /* timing-only: the mirrored stores of r2c phase B in ascending lane order (wrong rows) */
#endif
#ifndef MI355_RT_ONLY_PHASE
#define MI355_RT_ONLY_PHASE 0   /* timing-only builds: 1 = phase A alone, 2 = phase B alone (results wrong by construction) */
#endif
#ifndef MI355_RT_NT_IN
#define MI355_RT_NT_IN MI355_XCD_NT
#endif
constexpr bool RT_NT_OUT = MI355_RT_NT_OUT != 0, RT_NT_IN = MI355_RT_NT_IN != 0;   // nontemporal output stores / input loads of the register-tile kernels

struct RtCfg {
  static constexpr int N = 2048, T = 16, THREADS = 512;
  static constexpr int HALF_ELEMS = T * 32 * 32;   // one half of the exchange
  static constexpr int TW2_ELEMS = 31 * 64;        // stage-2 roots, rows q2 = 1..31, j2 = 0..63 fastest
};

// e^{-2 pi i m/64}, m = 0..31 (cos / sin of 2 pi m/64 rounded from an 80-bit evaluation)
struct Root64 {
  static constexpr double c[32] = {1.0, 0.9951847266721969, 0.9807852804032304, 0.9569403357322088, 0.9238795325112867, 0.881921264348355, 0.8314696123025452, 0.773010453362737, 0.7071067811865476, 0.6343932841636455, 0.5555702330196022, 0.47139673682599764, 0.3826834323650898, 0.2902846772544624, 0.19509032201612828, 0.0980171403295606, 0.0, -0.0980171403295606, -0.19509032201612828, -0.2902846772544624, -0.3826834323650898, -0.47139673682599764, -0.5555702330196022, -0.6343932841636455, -0.7071067811865476, -0.773010453362737, -0.8314696123025452, -0.881921264348355, -0.9238795325112867, -0.9569403357322088, -0.9807852804032304, -0.9951847266721969};
  static constexpr double s[32] = {0.0, 0.0980171403295606, 0.19509032201612828, 0.2902846772544624, 0.3826834323650898, 0.47139673682599764, 0.5555702330196022, 0.6343932841636455, 0.7071067811865476, 0.773010453362737, 0.8314696123025452, 0.881921264348355, 0.9238795325112867, 0.9569403357322088, 0.9807852804032304, 0.9951847266721969, 1.0, 0.9951847266721969, 0.9807852804032304, 0.9569403357322088, 0.9238795325112867, 0.881921264348355, 0.8314696123025452, 0.773010453362737, 0.7071067811865476, 0.6343932841636455, 0.5555702330196022, 0.47139673682599764, 0.3826834323650898, 0.2902846772544624, 0.19509032201612828, 0.0980171403295606};
};

// 64-point DFT of v[m], m = 0..63, in registers: output p = 2 q lands in v[q], p = 2 q + 1 in v[32 + q].
MI_DEV void rt_dft64(cf (&v)[64], cf (&v0)[32], cf (&v1)[32]) {
#pragma unroll
  for (int m = 0; m < 32; ++m) {
    const cf a = v[m] + v[m + 32], b = v[m] - v[m + 32];
    v0[m] = a;
    if (m == 0) v1[m] = b;
    else if (m == 16) v1[m] = mul_neg_i(b);
    else { cf w; w.x = (float)Root64::c[m]; w.y = (float)(-Root64::s[m]); v1[m] = cmul(b, w); }
  }
  fft_radix<32>(v0);
  fft_radix<32>(v1);
}

MI_DEV int rt_slot(int line, int pl, int u) { return u * 512 + pl * 16 + ((line + u) & 15); }

// The exchange.  Producer side: this thread's (line, u) and its 64 outputs F[2q] = v0[q], F[2q+1] = v1[q].  Consumer side: this
// thread's line `rl` and its two consumers, p' = pl0 in the first half (j2 = pl0) and p' = pl1 in the second (j2 = 32 + pl1).
// rt_exchange_first leaves consumer 0's inputs q2 = 0..31 in w0 and the SECOND half in LDS, so the caller can finish consumer 0
// (stage 2 + stores: 32 live values instead of 64) before rt_exchange_second picks up consumer 1's inputs; the caller ends the
// tile with a workgroup barrier (the next tile's first half overwrites the LDS).
// PARITY: the halves are the even outputs p = 2q (v0) and the odd ones (v1) instead of p < 32 and p >= 32 — a thread's two
// consumers are then j2 = 2 pl0 and 2 pl1 + 1 (adjacent lines of the result, which the c2r pass packs into one complex row).
template <bool PARITY = false>
MI_DEV void rt_exchange_first(const cf (&v0)[32], const cf (&v1)[32], cf (&w0)[32], cf* xb, int line, int u, int rl, int pl0) {
  const int wb = u * 512 + ((line + u) & 15);
  if constexpr (PARITY) {
#pragma unroll
    for (int q = 0; q < 32; ++q) xb[wb + q * 16] = v0[q];
  } else {
#pragma unroll
    for (int qq = 0; qq < 16; ++qq) { xb[wb + (2 * qq) * 16] = v0[qq]; xb[wb + (2 * qq + 1) * 16] = v1[qq]; }
  }
  __syncthreads();
#pragma unroll
  for (int uu = 0; uu < 32; ++uu) w0[uu] = xb[uu * 512 + pl0 * 16 + ((rl + uu) & 15)];
  __syncthreads();
  if constexpr (PARITY) {
#pragma unroll
    for (int q = 0; q < 32; ++q) xb[wb + q * 16] = v1[q];
  } else {
#pragma unroll
    for (int qq = 0; qq < 16; ++qq) { xb[wb + (2 * qq) * 16] = v0[16 + qq]; xb[wb + (2 * qq + 1) * 16] = v1[16 + qq]; }
  }
  __syncthreads();
  MI_SCHED_FENCE();
}
MI_DEV void rt_exchange_second(cf (&w1)[32], const cf* xb, int rl, int pl1) {
  MI_SCHED_FENCE();
#pragma unroll
  for (int uu = 0; uu < 32; ++uu) w1[uu] = xb[uu * 512 + pl1 * 16 + ((rl + uu) & 15)];
}

// stage 2 of consumer j2: roots e^{-2 pi i q2 j2/2048} from the LDS table, radix 32; output q is element j2 + 64 q of the line
#ifndef MI355_RT_TW_FENCE
#define MI355_RT_TW_FENCE 0
#endif
MI_DEV void rt_stage2(cf (&w)[32], const cf* tw2, int j2) {
  MI_OPAQUE_LANE_INT(j2);   // the 31 roots of a thread are the same for every tile: hoisted out of the tile loop they would pin 62 registers
#pragma unroll
  for (int q = 1; q < 32; ++q) {
    w[q] = cmul(w[q], tw2[(q - 1) * 64 + j2]);
#if MI355_RT_TW_FENCE
    if (q % MI355_RT_TW_FENCE == 0) MI_SCHED_FENCE();   // keeps the scheduler from requesting all 31 roots at once (62 registers)
#endif
  }
  fft_radix<32>(w);
}

// four-step roots e^{-2 pi i k1 n2/N} on the 64 inputs n2 = u + 32 m of row k1: exact lookups every 8th m, a recurrence between
// (kern_xcd.hpp fourstep_apply_chain, for the register tile's input map)
MI_DEV void rt_fourstep(cf (&v)[64], const XcdFusedArgs& a, unsigned k1, int u) {
  const auto root = [&](unsigned m) { return cmul(a.tw_hi[m >> a.fs_shift], a.tw_lo[m & a.fs_lo_mask]); };
  const cf step = root(k1 * 32u);
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    cf w = root(k1 * (unsigned)(u + 256 * g));
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[8 * g + j] = cmul(v[8 * g + j], w);
      if (j < 7) w = cmul(w, step);
    }
  }
}

// ---- one tile of a pass ------------------------------------------------------------------------------------------------------
// Every pass is a load half (64 values per thread into v) and a finish half (64-point DFTs, exchange, stage 2, stores).  The
// finish half calls `next(v)` as soon as the first exchange half has left the registers that held v: with MI355_RT_PREFETCH the
// kernels issue the NEXT tile's loads there, so that they are in flight while this tile's second stage computes and stores.
#ifndef MI355_RT_PREFETCH
#define MI355_RT_PREFETCH 0      /* number of the 64 loads per thread that are requested one tile ahead (0, 16, 32, 48, 64) */
#endif
constexpr int RT_PF = MI355_RT_PREFETCH;
constexpr bool RT_PREFETCH = RT_PF != 0;

// Column pass (four-step pass A of a [2048][S] matrix): 16 adjacent columns starting at c0, x -> W in place-layout.
template <bool SWAP_IN, bool NT_IN, int M0 = 0, int M1 = 64>
MI_DEV void rt_load_cols(cf (&v)[64], const cf* in, unsigned S, unsigned c0, int t) {
  const int line = t & 15, u = t >> 4;
  const cf* p = in + c0;
  const unsigned voff = (unsigned)u * S + (unsigned)line;
#pragma unroll
  for (int m = M0; m < M1; ++m) v[m] = cswap_if<SWAP_IN>(ld_stream<NT_IN>(p + (unsigned)(32 * m) * S + voff));
}
template <class Next>
MI_DEV void rt_finish_cols(cf (&v)[64], cf* W, unsigned S, unsigned c0, int t, cf* xb, const cf* tw2, Next&& next) {
  const int line = t & 15, u = t >> 4;
  cf v0[32], v1[32], w[32];
  rt_dft64(v, v0, v1);
  rt_exchange_first(v0, v1, w, xb, line, u, line, u);        // consumers j2 = u and u + 32 of the same column
  next(v);
  cf* po = W + c0;
  rt_stage2(w, tw2, u);
#pragma unroll
  for (int q = 0; q < 32; ++q) po[(unsigned)(u + 64 * q) * S + (unsigned)line] = w[q];
  rt_exchange_second(w, xb, line, u);
  rt_stage2(w, tw2, u + 32);
#pragma unroll
  for (int q = 0; q < 32; ++q) po[(unsigned)(u + 32 + 64 * q) * S + (unsigned)line] = w[q];
  __syncthreads();
}

// Row pass (four-step pass B): 16 adjacent rows k1 = r0 .. r0+15 of W[N1][2048], roots on load, transposed store out[k1 + N1 k2].
template <int M0 = 0, int M1 = 64>
MI_DEV void rt_load_rows(cf (&v)[64], const cf* W, unsigned r0, int t) {
  const int line = t >> 5, u = t & 31;
  const cf* p = W + (size_t)r0 * RtCfg::N;
  const unsigned voff = (unsigned)line * RtCfg::N + (unsigned)u;
#pragma unroll
  for (int m = M0; m < M1; ++m) v[m] = p[(unsigned)(32 * m) + voff];
}
template <bool SWAP_OUT, bool NT_OUT, bool FOURSTEP = true, class Next>
MI_DEV void rt_finish_rows(cf (&v)[64], cf* out, const XcdFusedArgs& f, unsigned N1, unsigned r0, float scale, int t, cf* xb, const cf* tw2, Next&& next) {
  const int line = t >> 5, u = t & 31;
  cf v0[32], v1[32], w[32];
  if constexpr (FOURSTEP) rt_fourstep(v, f, r0 + (unsigned)line, u);
  rt_dft64(v, v0, v1);
  const int rl = t & 15, jj = t >> 4;
  rt_exchange_first(v0, v1, w, xb, line, u, rl, jj);
  next(v);
  cf* po = out + r0;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    if (c == 1) rt_exchange_second(w, xb, rl, jj);
    rt_stage2(w, tw2, jj + 32 * c);
#pragma unroll
    for (int q = 0; q < 32; ++q) {
      cf r = w[q];
      if (scale != 1.0f) r = r * scale;
      st_stream<NT_OUT>(po + (unsigned)(jj + 32 * c + 64 * q) * N1 + (unsigned)rl, cswap_if<SWAP_OUT>(r));
    }
  }
  __syncthreads();
}

// ---- the fused kernel: kern_xcd.hpp's group / barrier machinery around register-tile passes ---------------------------------
// N = N1 * 2048.  N1 = 2048: both passes on register tiles.  N1 = 1024: pass A is the LDS-resident PASS_A configuration CA of
// kern_lines.hpp (16 x 1024 tiles fit the LDS), pass B the register tile.
template <int N1_, bool INV> struct XcdRtCfg {
  static constexpr int N1 = N1_, N2 = RtCfg::N, THREADS = RtCfg::THREADS;
  static constexpr bool A_RT = N1 == RtCfg::N;
  using CA = LineCfg<A_RT ? 1024 : N1, 32, (A_RT ? 1024 : N1) / 32, 1, 16, true, true, INV, false, TWID_NONE>;
  static_assert(A_RT || CA::THREADS == THREADS, "both passes run in the same workgroup");
  static constexpr int DATA_A = A_RT ? 0 : CA::DATA_ELEMS, TW_A = A_RT ? 0 : CA::TW_ELEMS;
  static constexpr int DATA = DATA_A > RtCfg::HALF_ELEMS ? DATA_A : RtCfg::HALF_ELEMS;
  static constexpr int LDS_BYTES = (DATA + TW_A + RtCfg::TW2_ELEMS) * 8 + 64;
  static_assert(LDS_BYTES <= 160 * 1024, "does not fit LDS");
};

template <int N1_, bool INV>
__global__ void __launch_bounds__(RtCfg::THREADS) fft_xcd_rt_kernel(const XcdFusedArgs f) {
  using X = XcdRtCfg<N1_, INV>;
  using CA = typename X::CA;
  MI_SMEM_DECL(smem);
  cf* lds = reinterpret_cast<cf*>(smem);
  cf* tw_a = lds + X::DATA;
  cf* tw2 = tw_a + X::TW_A;
  unsigned* s_words = reinterpret_cast<unsigned*>(tw2 + RtCfg::TW2_ELEMS);
  const int t = threadIdx.x;
  if constexpr (!X::A_RT) { for (int i = t; i < X::TW_A; i += X::THREADS) tw_a[i] = f.tw_a[i]; }
  for (int i = t; i < RtCfg::TW2_ELEMS; i += X::THREADS) tw2[i] = f.tw_b[i];
  if (!xcd_register(f.ctl, f.split, f.spin_limit, f.sticky_error, s_words)) return;
  const unsigned gslot = s_words[0], rank = s_words[1], gsize = s_words[2], gidx = s_words[4], groups = s_words[5];

  constexpr unsigned N1 = X::N1, N2 = X::N2;
  LineArgs aa{};
  aa.tw = f.tw_a; aa.num_tiles = N2 / 16; aa.num_lines = N2;
  aa.in_S = N2; aa.in_outer_stride = f.N; aa.out_S = N2; aa.out_outer_stride = f.N; aa.scale = 1.0f; aa.fs_group = 1;
  const bool two_slots = f.slots != 1u;
  cf* const W0 = f.wslots + (size_t)((two_slots ? 2u : 1u) * gslot) * (size_t)f.N;
  unsigned k = 0;
  for (long long tr = gidx; tr < f.num_transforms; tr += groups, ++k) {
    cf* const W = W0 + (size_t)(two_slots ? (k & 1u) : 0u) * (size_t)f.N;
    const cf* const x = f.in + tr * f.in_pitch;
    // ---- phase A: column FFTs of length N1 over 16-column tiles ----
    if constexpr (MI355_RT_ONLY_PHASE == 2) {
    } else if constexpr (X::A_RT) {
      cf v[64];
      if constexpr (RT_PREFETCH) { if (rank < N2 / 16) rt_load_cols<INV, RT_NT_IN, 0, RT_PF>(v, x, N2, rank * 16u, t); }
      for (unsigned tile = rank; tile < N2 / 16; tile += gsize) {
        const unsigned nx = tile + gsize;
        if constexpr (RT_PF < 64) rt_load_cols<INV, RT_NT_IN, RT_PF, 64>(v, x, N2, tile * 16u, t);
        rt_finish_cols(v, W, N2, tile * 16u, t, lds, tw2, [&](cf (&vv)[64]) { if (RT_PREFETCH && nx < N2 / 16) rt_load_cols<INV, RT_NT_IN, 0, RT_PF>(vv, x, N2, nx * 16u, t); });
      }
    } else {
      for (unsigned tile = rank; tile < N2 / 16; tile += gsize) {
        aa.in = x; aa.out = W;
        cf v[CA::E];
        stage_read<CA, 0, RT_NT_IN>(v, aa, tile, t, lds);
        stage_compute_write<CA, 0>(v, aa, tile, t, lds, tw_a, nullptr);
        __syncthreads();
        stage_read<CA, 1>(v, aa, tile, t, lds);
        __syncthreads();
        stage_compute_write<CA, 1>(v, aa, tile, t, lds, tw_a, nullptr);
        __syncthreads();
      }
    }
    xcd_arrive(&f.ctl->bar[gslot][0]);
    if (!xcd_wait(&f.ctl->bar[gslot][0], (k + 1u) * gsize, f.spin_limit, f.sticky_error, &s_words[6])) return;
    // ---- phase B: four-step roots on load, row FFTs of length 2048, transposed store ----
    cf* const y = f.out + tr * f.out_pitch;
    if constexpr (MI355_RT_ONLY_PHASE != 1) {
      cf v[64];
      if constexpr (RT_PREFETCH) { if (rank < N1 / 16) rt_load_rows<0, RT_PF>(v, W, rank * 16u, t); }
      for (unsigned tile = rank; tile < N1 / 16; tile += gsize) {
        const unsigned nx = tile + gsize;
        if constexpr (RT_PF < 64) rt_load_rows<RT_PF, 64>(v, W, tile * 16u, t);
        rt_finish_rows<INV, RT_NT_OUT, true>(v, y, f, N1, tile * 16u, f.scale, t, lds, tw2, [&](cf (&vv)[64]) { if (RT_PREFETCH && nx < N1 / 16) rt_load_rows<0, RT_PF>(vv, W, nx * 16u, t); });
      }
    }
    if (!two_slots) {
      xcd_arrive(&f.ctl->bar[gslot][1]);
      if (!xcd_wait(&f.ctl->bar[gslot][1], (k + 1u) * gsize, f.spin_limit, f.sticky_error, &s_words[6])) return;
    }
  }
}

// ---- r2c: the real four-step of kern_xcd_real.hpp on register tiles, N = 2048 x 2048 real points (config 5's line) ------------
// phase A  16 complex columns (= 32 real columns) per tile; the stage-2 consumers of a thread are j2 = jj and its MIRROR 64 - jj
//          (jj = 0: consumers 0 and 32, which mirror onto themselves), so that Z[k1] and Z[N1 - k1] of a column meet in one
//          thread's registers and the two real columns' spectra are separated there (kern_xcd_real.hpp: Y[k1][2c] = (Z[k1] +
//          conj Z[N1-k1])/2, Y[k1][2c+1] = (Z[k1] - conj Z[N1-k1])/(2i)); rows k1 = 0..N1/2 leave as 16-byte stores.
// phase B  rows 0..N1/2 of W: four-step roots, row FFT, bins k = k1 + N1 k2 <= N/2 stored directly, the others conjugated at
//          N - k (rows 0 and N1/2 contribute their first half only, the Nyquist bin stays unconjugated).
MI_DEV void rt_separate_store(cf* wt, unsigned k1, unsigned N2, cf a, cf b) {
  cf4 o;
  o.x = 0.5f * (a.x + b.x); o.y = 0.5f * (a.y - b.y);        // even real column
  o.z = 0.5f * (a.y + b.y); o.w = 0.5f * (b.x - a.x);        // odd real column: -i/2 * (a - conj b)
  *reinterpret_cast<cf4*>(wt + (size_t)k1 * N2) = o;
}

template <int N1_>     // (a template so that only the translation unit of its instance carries the device code)
__global__ void __launch_bounds__(RtCfg::THREADS) fft_xcd_rt_r2c_kernel(const XcdFusedArgs f) {
  static_assert(N1_ == 2048 || N1_ == 1024, "2048 x 2048 (both passes on register tiles) or 1024 x 2048 (pass A on the LDS-resident 16 x 1024 tiles of kern_xcd_real.hpp)");
  using XC = XcdRtCfg<N1_, false>;
  using CA = typename XC::CA;
  MI_SMEM_DECL(smem);
  cf* xb = reinterpret_cast<cf*>(smem);
  cf* tw_a = xb + XC::DATA;
  cf* tw2 = tw_a + XC::TW_A;
  unsigned* s_words = reinterpret_cast<unsigned*>(tw2 + RtCfg::TW2_ELEMS);
  const int t = threadIdx.x;
  if constexpr (!XC::A_RT) { for (int i = t; i < XC::TW_A; i += RtCfg::THREADS) tw_a[i] = f.tw_a[i]; }
  for (int i = t; i < RtCfg::TW2_ELEMS; i += RtCfg::THREADS) tw2[i] = f.tw_b[i];
  if (!xcd_register(f.ctl, f.split, f.spin_limit, f.sticky_error, s_words)) return;
  const unsigned gslot = s_words[0], rank = s_words[1], gsize = s_words[2], gidx = s_words[4], groups = s_words[5];

  constexpr unsigned N1 = N1_, N2 = 2048, ROWS = N1 / 2 + 1, NREAL = N1 * N2;
  LineArgs aa{};
  aa.tw = f.tw_a; aa.num_tiles = (N2 / 2) / 16; aa.num_lines = N2 / 2;
  aa.in_S = N2 / 2; aa.in_outer_stride = NREAL / 2; aa.out_S = N2 / 2; aa.out_outer_stride = NREAL / 2; aa.scale = 1.0f; aa.fs_group = 1;
  constexpr size_t wsize = (size_t)ROWS * N2;
  const bool two_slots = f.slots != 1u;
  cf* const W0 = f.wslots + (size_t)((two_slots ? 2u : 1u) * gslot) * wsize;
  unsigned k = 0;
  for (long long tr = gidx; tr < f.num_transforms; tr += groups, ++k) {
    cf* const W = W0 + (size_t)(two_slots ? (k & 1u) : 0u) * wsize;
    const cf* const x = f.in + tr * f.in_pitch;
    // ---- phase A ----
    if constexpr (!XC::A_RT) {
      // LDS-resident pass A (kern_xcd_real.hpp): 16 complex columns of length 1024, the finished columns kept in LDS, separation from there
      aa.in = x; aa.out = W;
      for (unsigned tile = rank; tile < (N2 / 2) / 16; tile += gsize) {
        cf v[CA::E];
        stage_read<CA, 0, RT_NT_IN>(v, aa, tile, t, xb);
        stage_compute_write<CA, 0>(v, aa, tile, t, xb, tw_a, nullptr);
        __syncthreads();
        stage_read<CA, 1>(v, aa, tile, t, xb);
        __syncthreads();
        stage_compute_write<CA, 1, false, true>(v, aa, tile, t, xb, tw_a, nullptr);
        __syncthreads();
        cf* const wt = W + tile * 32u;
        for (int p = t; p < (int)ROWS * 16; p += RtCfg::THREADS) {
          const int k1 = p >> 4, c = p & 15;
          rt_separate_store(wt + 2 * c, (unsigned)k1, N2, xb[k1 * 16 + c], xb[((N1 - k1) & (N1 - 1)) * 16 + c]);
        }
        __syncthreads();
      }
    } else
    for (unsigned tile = rank; tile < (MI355_RT_ONLY_PHASE == 2 ? 0u : (N2 / 2) / 16); tile += gsize) {
      const int line = t & 15, jj = t >> 4;
      cf v[64], v0[32], v1[32];
      {
        const cf* p = x + tile * 16u;
        const unsigned voff = (unsigned)jj * (N2 / 2) + (unsigned)line;
#pragma unroll
        for (int m = 0; m < 64; ++m) v[m] = ld_stream<RT_NT_IN>(p + (unsigned)(32 * m) * (N2 / 2) + voff);
      }
      rt_dft64(v, v0, v1);
      cf wa[32], wb[32];
      const int plb = (32 - jj) & 31;
      rt_exchange_first(v0, v1, wa, xb, line, jj, line, jj);
      rt_stage2(wa, tw2, jj);
      rt_exchange_second(wb, xb, line, plb);
      rt_stage2(wb, tw2, 32 + plb);
      cf* const wt = W + tile * 32u + 2u * (unsigned)line;
      if (jj != 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) rt_separate_store(wt, (unsigned)(jj + 64 * q), N2, wa[q], wb[31 - q]);
#pragma unroll
        for (int q = 0; q < 16; ++q) rt_separate_store(wt, (unsigned)(64 - jj + 64 * q), N2, wb[q], wa[31 - q]);
      } else {
#pragma unroll
        for (int q = 0; q <= 16; ++q) rt_separate_store(wt, (unsigned)(64 * q), N2, wa[q], wa[(32 - q) & 31]);
#pragma unroll
        for (int q = 0; q < 16; ++q) rt_separate_store(wt, (unsigned)(32 + 64 * q), N2, wb[q], wb[31 - q]);
      }
      __syncthreads();
    }
    xcd_arrive(&f.ctl->bar[gslot][0]);
    if (!xcd_wait(&f.ctl->bar[gslot][0], (k + 1u) * gsize, f.spin_limit, f.sticky_error, &s_words[6])) return;
    // ---- phase B ----
    cf* const po = f.out + tr * (f.out_pitch - MI355_RT_EXP_ALIGN);
    // (The packed rows start on odd 8-byte offsets — pitch N/2 + 1 — so every 128-byte run of these stores straddles two cache
    // lines.  Shifting the row tiles of a transform by its offset so that the direct stores land on whole lines measured no gain,
    // 275.5 vs 278 G real points/s: profiles/r03_regtile_ab.log; the mirrored stores sit one element further and cannot be aligned
    // at the same time.  Temporal stores, which let the partial lines of neighbouring tiles meet in the L2, are what helps: +7 %.)
    constexpr int sh = 0;
    for (unsigned tile = rank; tile < (MI355_RT_ONLY_PHASE == 1 ? 0u : (ROWS + 15) / 16 - MI355_RT_EXP_SKIP_LAST); tile += gsize) {
      cf v[64], v0[32], v1[32];
      const int r0 = (int)(tile * 16u) - sh;
      const int line = t >> 5, u = t & 31;
      {
        const int rr = r0 + line;
        const unsigned row = rr < 0 ? 0u : (rr < (int)ROWS ? (unsigned)rr : ROWS - 1);   // padding rows re-read a live row
        const cf* p = W + (size_t)row * N2 + u;
#pragma unroll
        for (int m = 0; m < 64; ++m) v[m] = p[32 * m];
        rt_fourstep(v, f, row, u);
      }
      rt_dft64(v, v0, v1);
      cf w[32];
      const int rl = t & 15, jj = t >> 4;
      const int k1s = r0 + rl;
      const unsigned k1 = (unsigned)k1s;
      const bool live = k1s >= 0 && k1s < (int)ROWS, edge = k1 == 0 || k1 == N1 / 2;
      rt_exchange_first(v0, v1, w, xb, line, u, rl, jj);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int j2 = jj + 32 * c;
        if (c == 1) rt_exchange_second(w, xb, rl, jj);
        rt_stage2(w, tw2, j2);
        if (f.scale != 1.0f) {
#pragma unroll
          for (int q = 0; q < 32; ++q) w[q] = w[q] * f.scale;
        }
        const unsigned kb = k1 + N1 * (unsigned)j2;                     // k = k1 + N1 k2, k2 = j2 + 64 q
        if (live) {
#pragma unroll
          for (int q = 0; q < 16; ++q) st_stream<RT_NT_OUT>(po + (kb + (unsigned)(64 * q) * N1), w[q]);
        }
        if (live && !edge) {                                            // k > N/2: conjugated at N - k
#pragma unroll
          for (int q = 16; q < 32; ++q) { cf mm; mm.x = w[q].x; mm.y = -w[q].y; st_stream<RT_NT_OUT>(po + (NREAL - MI355_RT_EXP_ALIGN - (MI355_RT_EXP_ASC ? kb + 15u - 2u * (unsigned)rl : kb) - (unsigned)(64 * q) * N1), mm); }
        }
        if (c == 0 && k1 == 0 && jj == 0) st_stream<RT_NT_OUT>(po + NREAL / 2, w[16]);   // k = N/2 (k1 = 0, k2 = N2/2): its own mirror, unconjugated
      }
      __syncthreads();
    }
    if (!two_slots) {
      xcd_arrive(&f.ctl->bar[gslot][1]);
      if (!xcd_wait(&f.ctl->bar[gslot][1], (k + 1u) * gsize, f.spin_limit, f.sticky_error, &s_words[6])) return;
    }
  }
}
struct XcdRtR2cCfg {
  static constexpr int THREADS = RtCfg::THREADS, LDS_BYTES = (RtCfg::HALF_ELEMS + RtCfg::TW2_ELEMS) * 8 + 64;
};
template <int N1_> struct XcdRtR2cCfgN { static constexpr int THREADS = RtCfg::THREADS, LDS_BYTES = XcdRtCfg<N1_, false>::LDS_BYTES; };

// ---- c2r: the Hermitian four-step of kern_xcd_real.hpp on register tiles, N = 2048 x 2048 real samples -----------------------
// x = IFFT(X) = FFT(conj X) for real x: a FORWARD four-step over Xt = conj(X_full), input index n1*N2 + n2 (bins), output
// k1 + N1*k2 (samples).
// phase A  columns n2 = 0..N2/2 (65 tiles of 16, the last with one live column); a column's upper half is read conjugated from the
//          stored bins, its lower half is the unconjugated upper half of column N2 - n2 walked backwards.  The exchange halves are
//          the even and the odd outputs, so a thread ends up with the ADJACENT results k1 = 2i + 64q and 2i + 1 + 64q of its
//          column: it multiplies them by the four-step roots e^{-2 pi i k1 n2/N} and packs the pair as z = a + i b into row
//          p = k1/2 of W at n2 — and, because every such row is Hermitian in n2, its extension conj(a) + i conj(b) at N2 - n2.
//          W = [N1/2][N2] holds full-length packed rows.  (Packing in pass B instead, as kern_xcd_real.hpp does, makes every
//          thread of pass B fetch two rows through two root chains in register-bounded batches: 175 vs 218 G real samples/s for
//          the half-length route, profiles/r03_regtile_ab.log.)
// phase B  a plain length-N2 row FFT of the N1/2 packed rows; real part = sample k1 = 2p, imaginary part = 2p + 1: adjacent in
//          the output, i.e. the transposed store of the c2c pass B with N1/2 complex rows writes the real line directly.
template <int N1_>
__global__ void __launch_bounds__(RtCfg::THREADS) fft_xcd_rt_c2r_kernel(const XcdFusedArgs f) {
  static_assert(N1_ == 2048 || N1_ == 1024, "2048 x 2048, or 1024 x 2048 with pass A on LDS-resident 16 x 1024 tiles (the packing then walks the LDS image)");
  using XC = XcdRtCfg<N1_, false>;
  using CA = typename XC::CA;
  MI_SMEM_DECL(smem);
  cf* xb = reinterpret_cast<cf*>(smem);
  cf* tw_a = xb + XC::DATA;
  cf* tw2 = tw_a + XC::TW_A;
  unsigned* s_words = reinterpret_cast<unsigned*>(tw2 + RtCfg::TW2_ELEMS);
  const int t = threadIdx.x;
  if constexpr (!XC::A_RT) { for (int i = t; i < XC::TW_A; i += RtCfg::THREADS) tw_a[i] = f.tw_a[i]; }
  for (int i = t; i < RtCfg::TW2_ELEMS; i += RtCfg::THREADS) tw2[i] = f.tw_b[i];
  if (!xcd_register(f.ctl, f.split, f.spin_limit, f.sticky_error, s_words)) return;
  const unsigned gslot = s_words[0], rank = s_words[1], gsize = s_words[2], gidx = s_words[4], groups = s_words[5];

  constexpr int N1 = N1_, N2 = 2048, COLS = N2 / 2 + 1;
  LineArgs aa{};
  aa.tw = f.tw_a; aa.num_tiles = (COLS + 15) / 16; aa.num_lines = COLS; aa.out_S = 16; aa.out_outer_stride = 0; aa.scale = 1.0f; aa.fs_group = 1;
  constexpr size_t wsize = (size_t)(N1 / 2) * N2;
  const bool two_slots = f.slots != 1u;
  cf* const W0 = f.wslots + (size_t)((two_slots ? 2u : 1u) * gslot) * wsize;
  const auto root = [&](unsigned m) { return cmul(f.tw_hi[m >> f.fs_shift], f.tw_lo[m & f.fs_lo_mask]); };
  unsigned k = 0;
  for (long long tr = gidx; tr < f.num_transforms; tr += groups, ++k) {
    cf* const W = W0 + (size_t)(two_slots ? (k & 1u) : 0u) * wsize;
    const cf* const X = f.in + tr * f.in_pitch;
    // ---- phase A ----
    if constexpr (!XC::A_RT) {
      for (unsigned tile = rank; tile < (COLS + 15) / 16; tile += gsize) {
        cf v[CA::E];
        {
          const int line = t & 15, u = t >> 4;                             // thread_map<CA, 0>: 32 butterflies of 32 points per column
          const int n2r = (int)(tile * 16u) + line, n2 = n2r > N2 / 2 ? N2 / 2 : n2r;
          const int up = u * N2 + n2;
          const int lo = n2 ? (N1 - 1 - u) * N2 + (N2 - n2) : (N1 - u) * N2;
#pragma unroll
          for (int q = 0; q < 32; ++q) {
            const int step = q * (N1 / 32) * N2;
            if (q < 16) { const cf x = X[up + step]; v[q] = cf{x.x, -x.y}; }
            else v[q] = X[lo - step];
          }
        }
        stage_compute_write<CA, 0>(v, aa, tile, t, xb, tw_a, nullptr);
        __syncthreads();
        stage_read<CA, 1>(v, aa, tile, t, xb);
        __syncthreads();
        stage_compute_write<CA, 1, false, true>(v, aa, tile, t, xb, tw_a, nullptr);      // finished columns stay in LDS: [k1][16]
        __syncthreads();
        // adjacent results k1 = 2i, 2i + 1 of column n2: four-step roots, packed as a + i b at n2 and conj(a) + i conj(b) at N2 - n2
        for (int p = t; p < (N1 / 2) * 16; p += RtCfg::THREADS) {
          const int i = p >> 4, c = p & 15;
          const int n2r = (int)(tile * 16u) + c;
          if (n2r > N2 / 2) continue;
          const cf a = cmul(xb[(2 * i) * 16 + c], root((unsigned)(2 * i) * (unsigned)n2r));
          const cf b = cmul(xb[(2 * i + 1) * 16 + c], root((unsigned)(2 * i + 1) * (unsigned)n2r));
          W[(size_t)i * N2 + n2r] = cf{a.x - b.y, a.y + b.x};
          if (n2r >= 1 && n2r < N2 / 2) W[(size_t)i * N2 + (N2 - n2r)] = cf{a.x + b.y, b.x - a.y};
        }
        __syncthreads();
      }
    } else
    for (unsigned tile = rank; tile < (COLS + 15) / 16; tile += gsize) {
      const int line = t & 15, i = t >> 4;
      cf v[64], v0[32], v1[32];
      const int n2r = (int)(tile * 16u) + line;
      const int n2 = n2r > N2 / 2 ? N2 / 2 : n2r;                        // padding lines of the last tile re-read its live column
      {
        const int up = i * N2 + n2;                                      // n1 = i + 32 m < N1/2: stored bin, conjugated
        const int lo = n2 ? (N1 - 1 - i) * N2 + (N2 - n2) : (N1 - i) * N2;   // n1 >= N1/2: bin N - n, as stored
#pragma unroll
        // (temporal loads: the packed rows start on odd 8-byte offsets — pitch N/2 + 1 — so every 128-byte run straddles two cache
        // lines whose other parts the neighbouring tiles read; nontemporal: 232 vs 245 G real samples/s, profiles/r03_regtile_ab.log)
        for (int m = 0; m < 32; ++m) { const cf x = X[up + (32 * m) * N2]; v[m] = cf{x.x, -x.y}; }
#pragma unroll
        for (int m = 32; m < 64; ++m) v[m] = X[lo - (32 * m) * N2];
      }
      rt_dft64(v, v0, v1);
      cf wa[32], wb[32];
      rt_exchange_first<true>(v0, v1, wa, xb, line, i, line, i);
      rt_stage2(wa, tw2, 2 * i);
      rt_exchange_second(wb, xb, line, i);
      rt_stage2(wb, tw2, 2 * i + 1);
      // roots e^{-2 pi i k1 n2/N}, k1 = 2i (+1) + 64 q: exact lookups every 8th q, the step e^{-2 pi i 64 n2/N} between; pack and store
      const bool live = n2r <= N2 / 2, mirror = live && n2r >= 1 && n2r < N2 / 2;
      const cf step = root(64u * (unsigned)n2);
      cf* const pd = W + (unsigned)i * N2 + (unsigned)n2;
      cf* const pm = W + (unsigned)i * N2 + (unsigned)(N2 - n2);
#pragma unroll
      for (int g = 0; g < 32; g += 8) {
        cf ra = root((unsigned)(2 * i + 64 * g) * (unsigned)n2), rb = root((unsigned)(2 * i + 1 + 64 * g) * (unsigned)n2);
#pragma unroll
        for (int q = g; q < g + 8; ++q) {
          const cf a = cmul(wa[q], ra), b = cmul(wb[q], rb);
          if (live) pd[(unsigned)(32 * q) * N2] = cf{a.x - b.y, a.y + b.x};       // a + i b
          if (mirror) pm[(unsigned)(32 * q) * N2] = cf{a.x + b.y, b.x - a.y};     // conj(a) + i conj(b)
          ra = cmul(ra, step); rb = cmul(rb, step);
        }
      }
      __syncthreads();
    }
    xcd_arrive(&f.ctl->bar[gslot][0]);
    if (!xcd_wait(&f.ctl->bar[gslot][0], (k + 1u) * gsize, f.spin_limit, f.sticky_error, &s_words[6])) return;
    // ---- phase B ----
    cf* const y = f.out + tr * f.out_pitch;
    for (unsigned tile = rank; tile < (N1 / 2) / 16; tile += gsize) {
      cf v[64];
      rt_load_rows(v, W, tile * 16u, t);
      rt_finish_rows<false, RT_NT_OUT, false>(v, y, f, N1 / 2, tile * 16u, f.scale, t, xb, tw2, [](cf (&)[64]) {});
    }
    if (!two_slots) {
      xcd_arrive(&f.ctl->bar[gslot][1]);
      if (!xcd_wait(&f.ctl->bar[gslot][1], (k + 1u) * gsize, f.spin_limit, f.sticky_error, &s_words[6])) return;
    }
  }
}

// ---- N = 1024 x 1024 (the headline's size) with TWO workgroups per CU -----------------------------------------------------------
// The LDS-resident fused kernel (kern_xcd.hpp) holds a 16 x 1024 tile in 128 KB of LDS: one 512-thread workgroup per CU, whose
// load, compute and store phases cannot overlap (the arithmetic costs it 9 %: profiles/r02_no_math_bound.log).  Here the tile
// lives in registers (32 values per thread) and its one exchange goes through LDS in two halves of 64 KB — outputs p < 16 of every
// radix-32 butterfly first (consumers j2 < 16: the threads of waves 0..3), then p >= 16 (waves 4..7) — so a workgroup needs 72 KB
// of LDS and at most 128 VGPRs, and two of them share a CU: while one computes, the other streams.
#ifndef MI355_HX_WAVES
#define MI355_HX_WAVES 4
#endif
struct HxCfg {
  static constexpr int N = 1024, T = 16, THREADS = 512;
  static constexpr int HALF_ELEMS = T * 32 * 16;
  static constexpr int TW1_ELEMS = 31 * 32;        // stage-1 roots e^{-2 pi i q k/1024}, rows q = 1..31, k = 0..31 fastest (a LineCfg<1024,32,32> table)
  static constexpr int LDS_BYTES = (HALF_ELEMS + TW1_ELEMS) * 8 + 64;
};

// producer (line, u): outputs v[p] of its radix-32 butterfly; consumer (rl, jj): inputs q2 = 0..31 of butterfly j2 = jj.
// `first` (wave-uniform, in an SGPR) says in which half this wave's consumers are served.  The two cases are separate straight-line
// paths with the same barrier sequence: one path with conditionally executed reads leaves w partly undefined for the optimiser,
// which then keeps 64 registers alive across the whole tile loop.
MI_DEV void hx_exchange(const cf (&v)[32], cf (&w)[32], cf* xb, int line, int u, int rl, int jj, bool first) {
  const int wb = u * 256 + ((line + u) & 15);
  const int rb = (jj & 15) * 16;
  if (first) {
#pragma unroll
    for (int q = 0; q < 16; ++q) xb[wb + q * 16] = v[q];
    __syncthreads();
#pragma unroll
    for (int uu = 0; uu < 32; ++uu) w[uu] = xb[uu * 256 + rb + ((rl + uu) & 15)];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) xb[wb + q * 16] = v[16 + q];
    __syncthreads();
    __syncthreads();
  } else {
#pragma unroll
    for (int q = 0; q < 16; ++q) xb[wb + q * 16] = v[q];
    __syncthreads();
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) xb[wb + q * 16] = v[16 + q];
    __syncthreads();
#pragma unroll
    for (int uu = 0; uu < 32; ++uu) w[uu] = xb[uu * 256 + rb + ((rl + uu) & 15)];
    __syncthreads();
  }
}

template <bool INV>
__global__ void __launch_bounds__(HxCfg::THREADS, MI355_HX_WAVES) fft_xcd_hx_kernel(const XcdFusedArgs f) {
  MI_SMEM_DECL(smem);
  cf* xb = reinterpret_cast<cf*>(smem);
  cf* tw1 = xb + HxCfg::HALF_ELEMS;
  unsigned* s_words = reinterpret_cast<unsigned*>(tw1 + HxCfg::TW1_ELEMS);
  const int t = threadIdx.x;
  for (int i = t; i < HxCfg::TW1_ELEMS; i += HxCfg::THREADS) tw1[i] = f.tw_a[i];
  if (!xcd_register(f.ctl, f.split, f.spin_limit, f.sticky_error, s_words)) return;
  const unsigned gslot = s_words[0], rank = s_words[1], gsize = s_words[2], gidx = s_words[4], groups = s_words[5];
  constexpr unsigned N1 = 1024, N2 = 1024, NT = 64;
  const bool two_slots = f.slots != 1u;
  cf* const W0 = f.wslots + (size_t)((two_slots ? 2u : 1u) * gslot) * (size_t)f.N;
  const int cl = t & 15, cu = t >> 4;       // column-side map (loads of pass A, stores of both passes)
  const int rl_ = t >> 5, ru = t & 31;      // row-side map (loads of pass B)
  const bool first = MI_UNIFORM_U32((unsigned)t >> 8) == 0u;   // consumers j2 = t div 16 < 16: waves 0..3
  const auto root = [&](unsigned m) { return cmul(f.tw_hi[m >> f.fs_shift], f.tw_lo[m & f.fs_lo_mask]); };
  unsigned k = 0;
  for (long long tr = gidx; tr < f.num_transforms; tr += groups, ++k) {
    cf* const W = W0 + (size_t)(two_slots ? (k & 1u) : 0u) * (size_t)f.N;
    const cf* const x = f.in + tr * f.in_pitch;
    // ---- phase A: 16 adjacent columns per tile, x -> W in place-layout ----
    for (unsigned tile = rank; tile < NT; tile += gsize) {
      cf v[32], w[32];
      {
        const cf* p = x + tile * 16u;
        const unsigned voff = (unsigned)cu * N2 + (unsigned)cl;
#pragma unroll
        for (int q = 0; q < 32; ++q) v[q] = cswap_if<INV>(ld_stream<RT_NT_IN>(sgpr_base(p + (unsigned)(32 * q) * N2) + voff));
      }
      fft_radix<32>(v);
      hx_exchange(v, w, xb, cl, cu, cl, cu, first);
      { int ti = cu; MI_OPAQUE_LANE_INT(ti);   // (not loop-invariant for the optimiser: hoisted, the 31 roots would pin 62 registers)
#pragma unroll
      for (int q = 1; q < 32; ++q) w[q] = cmul(w[q], tw1[(q - 1) * 32 + ti]); }
      fft_radix<32>(w);
      cf* po = W + tile * 16u;
      const unsigned so = (unsigned)cu * N2 + (unsigned)cl;
#pragma unroll
      for (int q = 0; q < 32; ++q) *(sgpr_base(po + (unsigned)(32 * q) * N2) + so) = w[q];
    }
    xcd_arrive(&f.ctl->bar[gslot][0]);
    if (!xcd_wait(&f.ctl->bar[gslot][0], (k + 1u) * gsize, f.spin_limit, f.sticky_error, &s_words[6])) return;
    // ---- phase B: 16 adjacent rows per tile, four-step roots on load, transposed store ----
    cf* const y = f.out + tr * f.out_pitch;
    for (unsigned tile = rank; tile < NT; tile += gsize) {
      cf v[32], w[32];
      {
        const unsigned k1 = tile * 16u + (unsigned)rl_;
        const cf* p = W + (size_t)(tile * 16u) * N2;
        const unsigned lo = (unsigned)rl_ * N2 + (unsigned)ru;
#pragma unroll
        for (int q = 0; q < 32; ++q) v[q] = *(sgpr_base(p + 32 * q) + lo);
        const cf step = root(k1 * 32u);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          cf r = root(k1 * (unsigned)(ru + 256 * g));
#pragma unroll
          for (int j = 0; j < 8; ++j) { v[8 * g + j] = cmul(v[8 * g + j], r); if (j < 7) r = cmul(r, step); }
        }
      }
      fft_radix<32>(v);
      hx_exchange(v, w, xb, rl_, ru, cl, cu, first);
      { int ti = cu; MI_OPAQUE_LANE_INT(ti);
#pragma unroll
      for (int q = 1; q < 32; ++q) w[q] = cmul(w[q], tw1[(q - 1) * 32 + ti]); }
      fft_radix<32>(w);
      cf* po = y + tile * 16u;
      const unsigned so = (unsigned)cu * N1 + (unsigned)cl;
#pragma unroll
      for (int q = 0; q < 32; ++q) {
        cf r = w[q];
        if (f.scale != 1.0f) r = r * f.scale;
        st_stream<XCD_NT>(sgpr_base(po + (unsigned)(32 * q) * N1) + so, cswap_if<INV>(r));
      }
    }
    if (!two_slots) {
      xcd_arrive(&f.ctl->bar[gslot][1]);
      if (!xcd_wait(&f.ctl->bar[gslot][1], (k + 1u) * gsize, f.spin_limit, f.sticky_error, &s_words[6])) return;
    }
  }
}

// ---- N = 1024 x 1024 on register tiles of 32 lines (256-byte segments) -----------------------------------------------------------
// Same scheme as the 16 x 2048 tiles above for the headline's size: a 32 x 1024 tile (2^15 points) in the registers of 512 threads,
// two radix-32 butterflies per thread and stage (1024 = 32 * 32), the one exchange through LDS in two halves of 128 KB (outputs
// p < 16 of every butterfly, then p >= 16).  Every global access is a run of 32 lanes x 8 bytes (or 16 lanes x 16 bytes) = 256 bytes
// where the LDS-resident 16 x 1024 tiles of kern_xcd.hpp move 128, and a workgroup has twice the bytes in flight.
//   column side:  line = t mod 32, h = t div 32 (0..15): producers u = h, h + 16; consumers j2 = h (first half), h + 16 (second)
//   rows as input: line = t div 16, hh = t mod 16: producers u = 2 hh, 2 hh + 1 — ONE 16-byte load per pair
// Exchange layout: slot(line, p', u) = u * 512 + p' * 32 + ((line + u div 2) mod 32): 16-lane writer groups (lines at fixed u, or hh
// at fixed line) hit 16 different 8-byte slots mod 16, 32-lane reader groups (32 lines) 32 different ones mod 32.
#ifndef MI355_RT1K_NT_OUT
#define MI355_RT1K_NT_OUT MI355_XCD_NT
#endif
#ifndef MI355_RT1K_NT_IN
#define MI355_RT1K_NT_IN MI355_RT_NT_IN
#endif
#ifndef MI355_RT1K_TW_FENCE
#define MI355_RT1K_TW_FENCE 8
#endif
#ifndef MI355_RT1K_PREFETCH
#define MI355_RT1K_PREFETCH 0    /* loads per thread (of 64) requested one tile ahead: 0, 8, 16, 32.  32: 276 B of scratch, 174.5 vs 196.4 GPoints/s (profiles/r03_headline_rt32_ab.log) */
#endif
#ifndef MI355_RT1K_W_NT
#define MI355_RT1K_W_NT 0      /* experiment: nontemporal accesses to the intermediate */
#endif
// T = 32: one 512-thread workgroup per CU.  T = 16 (r03 experiment, MI355FFT_XCD_HX=3): the same code on 16-line tiles with 256 threads — 72 KB of
// LDS and still 256 VGPRs per thread, so TWO independent workgroups share a CU and their load / compute / store phases interleave (the 512-thread
// two-per-CU form of fft_xcd_hx_kernel had to live in 128 VGPRs).
template <int T_ = 32> struct Rt1kCfgT {
  static constexpr int N = 1024, T = T_, THREADS = 16 * T_;
  static constexpr int HALF_ELEMS = T * 32 * 16;
  static constexpr int TW1_ELEMS = 31 * 32;
  static constexpr int LDS_BYTES = (HALF_ELEMS + TW1_ELEMS) * 8 + 64;
};
using Rt1kCfg = Rt1kCfgT<32>;
template <int T = 32> MI_DEV int rt1k_slot(int line, int pl, int u) { return u * (16 * T) + pl * T + ((line + (u >> 1)) & (T - 1)); }

// producers (line, ua), (line, ub) with outputs va[p], vb[p]; consumers (cl, h) in the first half and (cl, h + 16) in the second.
// `mid()` runs between the halves' reads, when consumer 0's inputs are in w and the second half is on its way through LDS.
template <int T = 32, class Mid>
MI_DEV void rt1k_exchange(const cf (&va)[32], const cf (&vb)[32], cf (&w)[32], cf* xb, int line, int ua, int ub, int cl, int h, Mid&& mid) {
  const int wa = rt1k_slot<T>(line, 0, ua), wb2 = rt1k_slot<T>(line, 0, ub);
#pragma unroll
  for (int p = 0; p < 16; ++p) { xb[wa + p * T] = va[p]; xb[wb2 + p * T] = vb[p]; }
  __syncthreads();
#pragma unroll
  for (int uu = 0; uu < 32; ++uu) w[uu] = xb[uu * (16 * T) + h * T + ((cl + (uu >> 1)) & (T - 1))];
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 16; ++p) { xb[wa + p * T] = va[16 + p]; xb[wb2 + p * T] = vb[16 + p]; }
  __syncthreads();
  MI_SCHED_FENCE();
  mid();
  MI_SCHED_FENCE();
#pragma unroll
  for (int uu = 0; uu < 32; ++uu) w[uu] = xb[uu * (16 * T) + h * T + ((cl + (uu >> 1)) & (T - 1))];
}

// VIEW (r03; src/kernels/ioview.js:56-380, zero_pad.js:21-80 for a 2^20-point line): pad-in-read / crop / zero ranges are predicates of pass A's
// loads and pass B's stores (XcdFusedArgs::v_*), so a view of a four-step line needs no embed / zero / extract launch either.
// N1_ = 2048 (r03, 2^21 as 2048 x 1024): pass A runs the 2048-point columns on the 16-line register tiles above (rt_finish_cols), pass B the
// 2048 rows of 1024 points on the 32-line tiles — 128- / 256-byte segments on every side (the 1024 x 2048 orientation: LDS-resident 16 x 1024 tiles + 16 x 2048 register tiles).
template <bool INV, int T = 32, bool VIEW = false, int N1_ = 1024>
__global__ void __launch_bounds__(Rt1kCfgT<T>::THREADS, 2) fft_xcd_rt1k_kernel(const XcdFusedArgs f) {   // (2 waves per SIMD: 256 registers per thread, VGPRs + AGPRs, for either tile width)
  using K = Rt1kCfgT<T>;
  static_assert(N1_ == 1024 || (N1_ == 2048 && T == 32 && !VIEW), "1024 x 1024, or 2048 x 1024 with pass A on 16 x 2048 register tiles");
  MI_SMEM_DECL(smem);
  cf* xb = reinterpret_cast<cf*>(smem);
  cf* tw1 = xb + K::HALF_ELEMS;
  cf* tw2 = tw1 + K::TW1_ELEMS;                                     // (N1_ = 2048: stage-2 roots of the 2048-point columns)
  unsigned* s_words = reinterpret_cast<unsigned*>(tw2 + (N1_ == 2048 ? RtCfg::TW2_ELEMS : 0));
  const int t = threadIdx.x;
  for (int i = t; i < K::TW1_ELEMS; i += K::THREADS) tw1[i] = f.tw_b[i];
  if constexpr (N1_ == 2048) { for (int i = t; i < RtCfg::TW2_ELEMS; i += K::THREADS) tw2[i] = f.tw_a[i]; }
  if (!xcd_register(f.ctl, f.split, f.spin_limit, f.sticky_error, s_words)) return;
  const unsigned gslot = s_words[0], rank = s_words[1], gsize = s_words[2], gidx = s_words[4], groups = s_words[5];
  constexpr unsigned N1 = N1_, N2 = 1024, NT = N1 / T, TU = (unsigned)T;
  const bool two_slots = f.slots != 1u;
  cf* const W0 = f.wslots + (size_t)((two_slots ? 2u : 1u) * gslot) * (size_t)f.N;
  const int cl = t & (T - 1), h = t / T;    // column-side map
  const int rl = t >> 4, hh = t & 15;       // row-side map
  const auto root = [&](unsigned m) { return cmul(f.tw_hi[m >> f.fs_shift], f.tw_lo[m & f.fs_lo_mask]); };
  const auto stage1 = [&](cf (&w)[32], int j2) {
    int ti = j2; MI_OPAQUE_LANE_INT(ti);     // (not loop-invariant for the optimiser: hoisted, the 31 roots would pin 62 registers)
#pragma unroll
    for (int q = 1; q < 32; ++q) {
      w[q] = cmul(w[q], tw1[(q - 1) * 32 + ti]);
      if (MI355_RT1K_TW_FENCE && q % MI355_RT1K_TW_FENCE == 0) MI_SCHED_FENCE();   // at most this many roots in flight (registers) at a time
    }
    fft_radix<32>(w);
  };
  unsigned k = 0;
  for (long long tr = gidx; tr < f.num_transforms; tr += groups, ++k) {
    cf* const W = W0 + (size_t)(two_slots ? (k & 1u) : 0u) * (size_t)f.N;
    const cf* const x = f.in + tr * f.in_pitch;
    // Software pipeline over a workgroup's tiles (MI355_RT1K_PREFETCH = P, experiment): P of the next tile's 64 loads per thread are requested
    // as soon as the second exchange half has left the registers that held this tile, i.e. they are in flight while both consumers'
    // stage 1 computes and stores; the rest follows at the top of the next iteration.  Off by default: the registers they pin cost more
    // (spills) than the overlap gains.
    // ---- phase A: 32 adjacent columns per tile ----
    if constexpr (N1_ == 2048) {
      for (unsigned tile = rank; tile < N2 / 16; tile += gsize) {
        cf v[64];
        rt_load_cols<INV, RT_NT_IN>(v, x, N2, tile * 16u, t);
        rt_finish_cols(v, W, N2, tile * 16u, t, xb, tw2, [](cf (&)[64]) {});
      }
    } else {
      cf va[32], vb[32], w[32];
      const unsigned voff = (unsigned)h * N2 + (unsigned)cl;
      constexpr int PF = MI355_RT1K_PREFETCH;     // how many of va's 32 loads are requested one tile ahead
      const auto load_x = [&](const cf* p, unsigned row_uniform, unsigned tile) {     // element (row_uniform + h) * N2 + tile * T + cl of the line
        if constexpr (VIEW) {
          const int i = (int)((row_uniform + (unsigned)h) * N2 + tile * TU + (unsigned)cl);
          const cf* const pe = p + row_uniform * N2 + voff;
          cf xv = {0.0f, 0.0f};
          if (i >= f.v_in_lo && i < f.v_in_hi) xv = *pe;
          return cswap_if<INV>(xv);
        } else return cswap_if<INV>(ld_stream<MI355_RT1K_NT_IN != 0>(sgpr_base(p + row_uniform * N2) + voff));
      };
      const auto load_a = [&](unsigned tile, int q0, int q1) {
        const cf* p = x + tile * TU;
#pragma unroll
        for (int q = 0; q < 32; ++q) if (q >= q0 && q < q1) va[q] = load_x(p, (unsigned)(32 * q), tile);
      };
      const auto load_b = [&](unsigned tile) {
        const cf* p = x + tile * TU;
#pragma unroll
        for (int q = 0; q < 32; ++q) vb[q] = load_x(p, (unsigned)(32 * q + 16), tile);
      };
      if (PF && rank < NT) load_a(rank, 0, PF);
      for (unsigned tile = rank; tile < NT; tile += gsize) {
        load_a(tile, PF, 32);
        load_b(tile);
        fft_radix<32>(va);
        fft_radix<32>(vb);
        cf* const po = W + tile * TU;
        const unsigned so = (unsigned)h * N2 + (unsigned)cl;
        rt1k_exchange<T>(va, vb, w, xb, cl, h, h + 16, cl, h, [&] {
          if (PF && tile + gsize < NT) load_a(tile + gsize, 0, PF);
          stage1(w, h);
#pragma unroll
          for (int q = 0; q < 32; ++q) st_stream<MI355_RT1K_W_NT != 0>(sgpr_base(po + (unsigned)(32 * q) * N2) + so, w[q]);
        });
        stage1(w, h + 16);
#pragma unroll
        for (int q = 0; q < 32; ++q) st_stream<MI355_RT1K_W_NT != 0>(sgpr_base(po + (unsigned)(32 * q + 16) * N2) + so, w[q]);
        __syncthreads();
      }
    }
    xcd_arrive(&f.ctl->bar[gslot][0]);
    if (!xcd_wait(&f.ctl->bar[gslot][0], (k + 1u) * gsize, f.spin_limit, f.sticky_error, &s_words[6])) return;
    // ---- phase B: 32 adjacent rows per tile, four-step roots on load, transposed store ----
    cf* const y = f.out + tr * f.out_pitch;
    {
      cf va[32], vb[32], w[32];
      const unsigned lo = (unsigned)rl * N2 + 2u * (unsigned)hh;
      constexpr int PF = MI355_RT1K_PREFETCH / 2;  // how many of the 32 16-byte loads are requested one tile ahead
      const auto load_rows = [&](unsigned tile, int q0, int q1) {      // elements 2hh + 32q and 2hh + 1 + 32q of row 32 tile + rl, q = q0 .. q1 - 1
        const cf* p = W + (size_t)(tile * TU) * N2;
#pragma unroll
        for (int q = 0; q < 32; ++q) if (q >= q0 && q < q1) {
          const cf4 pr = *reinterpret_cast<const cf4*>(sgpr_base(p + 32 * q) + lo);
          va[q] = cf{pr.x, pr.y}; vb[q] = cf{pr.z, pr.w};
        }
      };
      if (PF && rank < NT) load_rows(rank, 0, PF);
      for (unsigned tile = rank; tile < NT; tile += gsize) {
        load_rows(tile, PF, 32);
        {
          const unsigned k1 = tile * TU + (unsigned)rl;
          const cf step = root(k1 * 32u), one = root(k1);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            cf ra = root(k1 * (unsigned)(2 * hh + 256 * g));
            cf rb = cmul(ra, one);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              va[8 * g + j] = cmul(va[8 * g + j], ra); vb[8 * g + j] = cmul(vb[8 * g + j], rb);
              if (j < 7) { ra = cmul(ra, step); rb = cmul(rb, step); }
            }
          }
        }
        fft_radix<32>(va);
        fft_radix<32>(vb);
        cf* const po = y + tile * TU;
        const unsigned so = (unsigned)h * N1 + (unsigned)cl;
        const auto store = [&](cf (&ww)[32], unsigned off) {
#pragma unroll
          for (int q = 0; q < 32; ++q) {
            cf r = ww[q];
            if (f.scale != 1.0f) r = r * f.scale;
            if constexpr (VIEW) {
              const int kk = (int)(((unsigned)(32 * q) + off + (unsigned)h) * N1 + tile * TU + (unsigned)cl);      // k = k1 + N1 k2
              cf* const pe = po + ((unsigned)(32 * q) + off) * N1 + so;
              if (kk < f.v_zlo || kk >= f.v_zhi) r = cf{0.0f, 0.0f};
              if (kk >= f.v_out_lo && kk < f.v_out_hi) *pe = cswap_if<INV>(r);
            } else st_stream<MI355_RT1K_NT_OUT != 0>(sgpr_base(po + ((unsigned)(32 * q) + off) * N1) + so, cswap_if<INV>(r));
          }
        };
        rt1k_exchange<T>(va, vb, w, xb, rl, 2 * hh, 2 * hh + 1, cl, h, [&] {
          if (PF && tile + gsize < NT) load_rows(tile + gsize, 0, PF);
          stage1(w, h); store(w, 0u);
        });
        stage1(w, h + 16);
        store(w, 16u);
        __syncthreads();
      }
    }
    if (!two_slots) {
      xcd_arrive(&f.ctl->bar[gslot][1]);
      if (!xcd_wait(&f.ctl->bar[gslot][1], (k + 1u) * gsize, f.spin_limit, f.sticky_error, &s_words[6])) return;
    }
  }
}

// ---- fftconv of 2^20-point lines as ONE pipeline (r03; src/runtime/plans/fftconv.js:1659-1706: forward -> pointwise -> inverse) ---------
// y_k = IFFT(FFT(x) * H_k) / N, circular, dense lines, K kernel spectra H_k (transformed once per exec by the ordinary route).
// The forward four-step's pass B produces the spectrum tile of rows k1 = 32 tile .. +31 in the registers of the consumer threads
// (k2 = j2 + 32 q) — which IS the input of the inverse transform's pass A for the 32 columns n2 = k1 (input index k = k2 N1 + k1): same
// thread, same registers.  So the spectrum never travels: per data line
//   A    column FFTs of x -> W                                                        (16 B/point)
//   M_k  rows of W, four-step roots, row FFT = spectrum tile; times H_k (transposed read); inverse pass A -> W2   (24 B/point)
//   C_k  rows of W2, inverse four-step roots, row FFT, transposed store of y_k / N      (16 B/point)
// = 56 B/point for one kernel where forward, pointwise and inverse launches move 88 (and 40 instead of 56 for every further kernel).
// Inverse passes run the forward arithmetic on re/im-swapped data (radix.hpp): the product is swapped before M_k's second half, the
// result swapped back at C_k's store.  Group barriers: A | M_k, M_k | C_k, and C_k | M_k+1 when another kernel follows (W2 is reused).
template <int N1_>
__global__ void __launch_bounds__(Rt1kCfg::THREADS) fft_xcd_conv1m_kernel(const XcdFusedArgs f) {
  static_assert(N1_ == 1024, "1024 x 1024");
  MI_SMEM_DECL(smem);
  cf* xb = reinterpret_cast<cf*>(smem);
  cf* tw1 = xb + Rt1kCfg::HALF_ELEMS;
  unsigned* s_words = reinterpret_cast<unsigned*>(tw1 + Rt1kCfg::TW1_ELEMS);
  const int t = threadIdx.x;
  for (int i = t; i < Rt1kCfg::TW1_ELEMS; i += Rt1kCfg::THREADS) tw1[i] = f.tw_a[i];
  if (!xcd_register(f.ctl, f.split, f.spin_limit, f.sticky_error, s_words)) return;
  const unsigned gslot = s_words[0], rank = s_words[1], gsize = s_words[2], gidx = s_words[4], groups = s_words[5];
  constexpr unsigned N1 = 1024, N2 = 1024, NT = 32;
  cf* const Wg = f.wslots + (size_t)(2u * (f.slots ? f.slots : 1u) * gslot) * (size_t)f.N;     // per group and line of a round: W and W2
  const int cl = t & 31, h = t >> 5;        // column-side map
  const int rl = t >> 4, hh = t & 15;       // row-side map
  const bool first = MI_UNIFORM_U32((unsigned)t >> 8) == 0u;   // 16-row tiles of phase M: consumers ku = t div 16 < 16 are served in the first exchange half
  const auto root = [&](unsigned m) { return cmul(f.tw_hi[m >> f.fs_shift], f.tw_lo[m & f.fs_lo_mask]); };
  const auto stage1 = [&](cf (&w)[32], int j2) {
    int ti = j2; MI_OPAQUE_LANE_INT(ti);
#pragma unroll
    for (int q = 1; q < 32; ++q) w[q] = cmul(w[q], tw1[(q - 1) * 32 + ti]);
    fft_radix<32>(w);
  };
  // rows k1 = 32 tile + rl of a workspace slot, times the four-step roots e^{-2 pi i k1 n2/N}: this thread's elements n2 = 2hh (+1) + 32 q
  const auto load_rows = [&](cf (&va)[32], cf (&vb)[32], const cf* S, unsigned tile) {
    const unsigned k1 = tile * 32u + (unsigned)rl;
    const cf* p = S + (size_t)(tile * 32u) * N2;
    const unsigned lo = (unsigned)rl * N2 + 2u * (unsigned)hh;
#pragma unroll
    for (int q = 0; q < 32; ++q) {
      const cf4 pr = *reinterpret_cast<const cf4*>(sgpr_base(p + 32 * q) + lo);
      va[q] = cf{pr.x, pr.y}; vb[q] = cf{pr.z, pr.w};
    }
    const cf step = root(k1 * 32u), one = root(k1);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      cf ra = root(k1 * (unsigned)(2 * hh + 256 * g));
      cf rb = cmul(ra, one);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        va[8 * g + j] = cmul(va[8 * g + j], ra); vb[8 * g + j] = cmul(vb[8 * g + j], rb);
        if (j < 7) { ra = cmul(ra, step); rb = cmul(rb, step); }
      }
    }
  };
  unsigned bar = 0;                                  // barriers passed so far (the counter is monotonic)
  const auto group_barrier = [&]() -> bool {
    xcd_arrive(&f.ctl->bar[gslot][0]);
    ++bar;
    return xcd_wait(&f.ctl->bar[gslot][0], bar * gsize, f.spin_limit, f.sticky_error, &s_words[6]);
  };
  // A round = L = f.slots data lines per group (2 L workspace slots): with one line per round a workgroup of a 32-strong group has one
  // or two tiles per phase and the phases are pure latency; L lines give every phase L times the tiles between the same barriers.
  const unsigned L = f.slots ? f.slots : 1u;
  for (long long tr0 = (long long)gidx * L; tr0 < f.num_transforms; tr0 += (long long)groups * L) {
    const unsigned nl = (unsigned)(f.num_transforms - tr0 < (long long)L ? f.num_transforms - tr0 : (long long)L);
    // ---- A: 32 adjacent columns of x per tile -> W ----
    for (unsigned tt = rank; tt < NT * nl; tt += gsize) {
      const unsigned ll = tt / NT, tile = tt - ll * NT;
      const cf* const x = f.in + (tr0 + ll) * f.in_pitch;
      cf* const W = Wg + (size_t)(2u * ll) * (size_t)f.N;
      cf va[32], vb[32], w[32];
      {
        const cf* p = x + tile * 32u;
        const unsigned voff = (unsigned)h * N2 + (unsigned)cl;
#pragma unroll
        for (int q = 0; q < 32; ++q) {
          va[q] = ld_stream<MI355_RT1K_NT_IN != 0>(sgpr_base(p + (unsigned)(32 * q) * N2) + voff);
          vb[q] = ld_stream<MI355_RT1K_NT_IN != 0>(sgpr_base(p + (unsigned)(32 * q + 16) * N2) + voff);
        }
      }
      fft_radix<32>(va);
      fft_radix<32>(vb);
      cf* const po = W + tile * 32u;
      const unsigned so = (unsigned)h * N2 + (unsigned)cl;
      rt1k_exchange(va, vb, w, xb, cl, h, h + 16, cl, h, [&] {
        stage1(w, h);
#pragma unroll
        for (int q = 0; q < 32; ++q) *(sgpr_base(po + (unsigned)(32 * q) * N2) + so) = w[q];
      });
      stage1(w, h + 16);
#pragma unroll
      for (int q = 0; q < 32; ++q) *(sgpr_base(po + (unsigned)(32 * q + 16) * N2) + so) = w[q];
      __syncthreads();
    }
    if (!group_barrier()) return;
    for (unsigned kk = 0; kk < f.conv_k; ++kk) {
      const cf* const H = f.mul + (size_t)kk * (size_t)f.N;
      // ---- M_k: spectrum tile of 16 rows k1, product, inverse pass A -> W2 ----
      // (16-row tiles with ONE butterfly per thread — the exchange of fft_xcd_hx_kernel: a thread never holds more than its 32 values plus
      // the 32 arriving ones.  The 32-row form, two butterflies per thread, keeps consumer 0's 32 results alive across consumer 1's stage
      // and spilled 760 bytes per lane: 63 vs 70 GPoints/s for the three-launch route, profiles/r03_fftconv_pipeline_ab.log.)
      for (unsigned tt = rank; tt < 2 * NT * nl; tt += gsize) {
        const unsigned ll = tt / (2 * NT), tile = tt - ll * (2 * NT);
        const cf* const W = Wg + (size_t)(2u * ll) * (size_t)f.N;
        cf* const W2 = Wg + (size_t)(2u * ll + 1u) * (size_t)f.N;
        const int ml = t >> 5, mu = t & 31;          // row-side map of a 16-row tile: row ml, elements n2 = mu + 32 q
        const int kl = t & 15, ku = t >> 4;          // column-side map: row k1 = 16 tile + kl, butterfly ku
        cf v[32], w[32];
        {
          const unsigned k1 = tile * 16u + (unsigned)ml;
          const cf* p = W + (size_t)(tile * 16u) * N2;
          const unsigned lo = (unsigned)ml * N2 + (unsigned)mu;
#pragma unroll
          for (int q = 0; q < 32; ++q) v[q] = *(sgpr_base(p + 32 * q) + lo);
          const cf step = root(k1 * 32u);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            cf r = root(k1 * (unsigned)(mu + 256 * g));
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[8 * g + j] = cmul(v[8 * g + j], r); if (j < 7) r = cmul(r, step); }
          }
        }
        fft_radix<32>(v);
        hx_exchange(v, w, xb, ml, mu, kl, ku, first);
        stage1(w, ku);
        // w[q] = X[k1 + N1 (ku + 32 q)], k1 = 16 tile + kl: times H_k (conjugated for correlation), re/im swapped = the inverse transform's
        // forward-arithmetic input of column n2 = k1, butterfly u = ku
        {
          const cf* ph = H + tile * 16u;
          const unsigned ho = (unsigned)ku * N1 + (unsigned)kl;
#pragma unroll
          for (int q = 0; q < 32; ++q) {
            const cf hq = *(sgpr_base(ph + (unsigned)(32 * q) * N1) + ho);
            const cf pq = f.conv_conj ? cmul_conj(w[q], hq) : cmul(w[q], hq);
            v[q] = pq.yx;
          }
        }
        fft_radix<32>(v);
        hx_exchange(v, w, xb, kl, ku, kl, ku, first);
        stage1(w, ku);
        cf* const po = W2 + tile * 16u;
        const unsigned so = (unsigned)ku * N2 + (unsigned)kl;
#pragma unroll
        for (int q = 0; q < 32; ++q) *(sgpr_base(po + (unsigned)(32 * q) * N2) + so) = w[q];
      }
      if (!group_barrier()) return;
      // ---- C_k: rows of W2, roots, row FFT, transposed store of y_k (swapped back, times 1/N) ----
      for (unsigned tt = rank; tt < NT * nl; tt += gsize) {
        const unsigned ll = tt / NT, tile = tt - ll * NT;
        cf* const y = f.out + (tr0 + ll) * f.out_pitch + (long long)kk * f.out_kernel_pitch;
        const cf* const W2 = Wg + (size_t)(2u * ll + 1u) * (size_t)f.N;
        cf va[32], vb[32], w[32];
        load_rows(va, vb, W2, tile);
        fft_radix<32>(va);
        fft_radix<32>(vb);
        cf* const po = y + tile * 32u;
        const unsigned so = (unsigned)h * N1 + (unsigned)cl;
        const auto store = [&](cf (&ww)[32], unsigned off) {
#pragma unroll
          for (int q = 0; q < 32; ++q) st_stream<MI355_RT1K_NT_OUT != 0>(sgpr_base(po + ((unsigned)(32 * q) + off) * N1) + so, (ww[q] * f.scale).yx);
        };
        rt1k_exchange(va, vb, w, xb, rl, 2 * hh, 2 * hh + 1, cl, h, [&] { stage1(w, h); store(w, 0u); });
        stage1(w, h + 16);
        store(w, 16u);
        __syncthreads();
      }
      // W2 is rewritten by M_k+1 (a further kernel, which passes no other barrier first); the next data line's M_0 comes after its A | M barrier
      if (kk + 1 < f.conv_k) { if (!group_barrier()) return; }
    }
  }
}

}  // namespace mi355

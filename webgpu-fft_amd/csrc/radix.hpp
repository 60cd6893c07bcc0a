// radix.hpp — complex helpers and in-register radix-R DFTs for gfx950.
//
// Replaces the arithmetic core of the reference's generated WGSL (src/kernels/utils_wgsl.js:1-22 c_add /
// c_mul / cis, and the `for q < RADIX` accumulate of src/kernels/stockham_stage.js:93-99).  The
// reference evaluates an R-point DFT row per OUTPUT element (R complex MACs + an in-shader sincos per
// thread); here one lane owns all R inputs of a butterfly in VGPRs and produces all R outputs with a
// compile-time-unrolled FFT whose roots are literals, so a radix-32 step costs ~12 VALU ops per point
// and no transcendental.
//
// Everything computes the FORWARD transform (kernel e^{-2*pi*i*nk/R}).  Inverse transforms run the same
// code on re/im-swapped data: ifft(x) = swap(fft(swap(x))), swap(a+ib) = b+ia — a register rename at the
// first load and last store, never a second instruction stream.
#pragma once
#include "platform.hpp"

namespace mi355 {

typedef float cf __attribute__((ext_vector_type(2)));  // interleaved complex f32: x = re, y = im

#define MI_DEV __device__ __forceinline__

#ifndef MI355_EXP_NO_MATH
#define MI355_EXP_NO_MATH 0
#endif
MI_DEV cf cmul(cf a, cf b) {
  // (a.x*b.x - a.y*b.y, a.x*b.y + a.y*b.x) as 2 mul + 2 fma
  cf r;
#if MI355_EXP_NO_MATH   /* timing-only builds: the operands stay live, the product is not formed */
  r.x = __builtin_fmaf(0.0f, b.x, a.x); r.y = __builtin_fmaf(0.0f, b.y, a.y);
  return r;
#endif
  r.x = __builtin_fmaf(-a.y, b.y, a.x * b.x);
  r.y = __builtin_fmaf(a.y, b.x, a.x * b.y);
  return r;
}
MI_DEV cf cmul_conj(cf a, cf b) {  // a * conj(b)
  cf r;
  r.x = __builtin_fmaf(a.y, b.y, a.x * b.x);
  r.y = __builtin_fmaf(a.y, b.x, -(a.x * b.y));
  return r;
}
MI_DEV cf cswap(cf a) { return a.yx; }
template <bool S> MI_DEV cf cswap_if(cf a) { if constexpr (S) return a.yx; else return a; }
MI_DEV cf mul_neg_i(cf a) { cf r; r.x = a.y; r.y = -a.x; return r; }  // a * (-i)
MI_DEV cf mul_pos_i(cf a) { cf r; r.x = -a.y; r.y = a.x; return r; }  // a * (+i)

template <int R> struct RootTable;
#include "root_tables.inc"

constexpr int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
constexpr int bitrev(int x, int bits) { int y = 0; for (int i = 0; i < bits; ++i) { y = (y << 1) | (x & 1); x >>= 1; } return y; }

// One DIT butterfly with the root W32^K (K in [0,16)): (a, b) <- (a + w*b, a - w*b).
// The generic case is 6 FMAs (t = a + w*b in 4, then 2a - t in 2); K = 0, 8 are pure add/sub and K = 4, 12
// use the (1 -+ i)/sqrt(2) structure.
template <int K> MI_DEV void bfly32(cf& a, cf& b) {
  static_assert(K >= 0 && K < 16, "root index");
  if constexpr (K == 0) {
    cf t = a + b; b = a - b; a = t;
  } else if constexpr (K == 8) {                 // w = -i
    cf wb = mul_neg_i(b);
    cf t = a + wb; b = a - wb; a = t;
  } else if constexpr (K == 4) {                 // w = (1 - i)/sqrt2 : w*b = ((bx+by), (by-bx))/sqrt2
    constexpr float h = 0.70710678118654752440f;
    cf s; s.x = b.x + b.y; s.y = b.y - b.x;
    cf t; t.x = __builtin_fmaf(h, s.x, a.x); t.y = __builtin_fmaf(h, s.y, a.y);
    b.x = __builtin_fmaf(-h, s.x, a.x); b.y = __builtin_fmaf(-h, s.y, a.y);
    a = t;
  } else if constexpr (K == 12) {                // w = (-1 - i)/sqrt2 : w*b = ((by-bx), -(bx+by))/sqrt2
    constexpr float h = 0.70710678118654752440f;
    cf s; s.x = b.y - b.x; s.y = -(b.x + b.y);
    cf t; t.x = __builtin_fmaf(h, s.x, a.x); t.y = __builtin_fmaf(h, s.y, a.y);
    b.x = __builtin_fmaf(-h, s.x, a.x); b.y = __builtin_fmaf(-h, s.y, a.y);
    a = t;
  } else {
    constexpr float wr = (float)RootTable<32>::c[K];
    constexpr float wi = (float)(-RootTable<32>::s[K]);   // forward root e^{-2 pi i K/32}
    cf t;
    t.x = __builtin_fmaf(-wi, b.y, __builtin_fmaf(wr, b.x, a.x));
    t.y = __builtin_fmaf(wi, b.x, __builtin_fmaf(wr, b.y, a.y));
    b.x = __builtin_fmaf(2.0f, a.x, -t.x);
    b.y = __builtin_fmaf(2.0f, a.y, -t.y);
    a = t;
  }
}

// recursion helpers: fully unrolled (every index is a template constant, so v[] stays in VGPRs)
template <int R, int M, int G, int K> struct DitInner {
  static MI_DEV void run(cf* t) {
    bfly32<K*(32 / M)>(t[G + K], t[G + K + M / 2]);
    if constexpr (K + 1 < M / 2) DitInner<R, M, G, K + 1>::run(t);
  }
};
template <int R, int M, int G> struct DitGroup {
  static MI_DEV void run(cf* t) {
    DitInner<R, M, G, 0>::run(t);
    if constexpr (G + M < R) DitGroup<R, M, G + M>::run(t);
  }
};
template <int R, int M> struct DitStage {
  static MI_DEV void run(cf* t) {
    DitGroup<R, M, 0>::run(t);
    if constexpr (M < R) DitStage<R, M * 2>::run(t);
  }
};

// Forward R-point DFT, R in {1,2,4,8,16,32}, natural order in -> natural order out, in registers.
template <int R> MI_DEV void fft_pow2(cf (&v)[R]) {
  static_assert(R >= 1 && R <= 32 && (R & (R - 1)) == 0, "radix");
  if constexpr (R > 1) {
    cf t[R];
#pragma unroll
    for (int i = 0; i < R; ++i) t[i] = v[bitrev(i, ilog2(R))];
    DitStage<R, 2>::run(t);
#pragma unroll
    for (int i = 0; i < R; ++i) v[i] = t[i];
  }
}

// Forward R-point DFT for the odd radices of the reference's radix set {13,11,7,5,3}
// (src/plan.js:20-33): direct evaluation with literal roots, f32 FMA accumulation.
template <int R> MI_DEV void dft_odd(cf (&v)[R]) {
  cf o[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    cf acc = v[0];
#pragma unroll
    for (int n = 1; n < R; ++n) {
      const int m = (n * k) % R;
      const float wr = (float)RootTable<R>::c[m];
      const float wi = (float)(-RootTable<R>::s[m]);
      acc.x = __builtin_fmaf(-wi, v[n].y, __builtin_fmaf(wr, v[n].x, acc.x));
      acc.y = __builtin_fmaf(wi, v[n].x, __builtin_fmaf(wr, v[n].y, acc.y));
    }
    o[k] = acc;
  }
#pragma unroll
  for (int k = 0; k < R; ++k) v[k] = o[k];
}

template <int R> MI_DEV void fft_radix(cf (&v)[R]) {
#if MI355_EXP_NO_MATH
  return;
#endif
  if constexpr ((R & (R - 1)) == 0) fft_pow2<R>(v); else dft_odd<R>(v);
}

}  // namespace mi355

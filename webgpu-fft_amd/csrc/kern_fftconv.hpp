// kern_fftconv.hpp — single-launch fftconv for small 1-D circular problems (BASELINE config 4 is 4 x 256 points
// against 3 kernels: latency-bound, so launches are what matters).
//
// Replaces the whole exec loop of FftConvPlan (src/runtime/plans/fftconv.js:1415-1712) for this shape class:
// per kernel k the reference issues FFT(kernel_k) -> strided gather of the input lane as one 8-byte copy command
// per element (:841-857) -> forward c2c -> pointwise (src/kernels/fft_conv.js:33-66) -> inverse c2c -> per-element
// scatter (:860-899), i.e. ~K*(3 plans x 3 passes) dispatches + K*batch*N*2 copy commands.  Here ONE workgroup per
// batch entry holds the input line and all K kernel lines in LDS and does
//     X = FFT(x_b);  H_k = FFT(h_k);  y_k = IFFT(X .* (conj?)H_k) / N   for every k
// with the lane gather folded into the first load and the lane scatter into the last store.  Elements of the
// output outside the addressed lanes are never touched (sentinel tests).
//
// N = R0*R1 (two register stages, one LDS exchange per transform), ROW thread map of kern_lines.hpp:
// line = t / TPL, u = t % TPL; line 0 is the data, lines 1..K the kernels; TL = lines per workgroup >= K+1.
#pragma once
#include "platform.hpp"
#include "radix.hpp"

namespace mi355 {

struct FusedConvArgs {
  const cf* in;
  const cf* kern;    // K dense kernels of kern_len points (zero-extended to N)
  cf* out;
  const cf* tw;      // [R1-1][R0] roots of order N (rows q = 1..R1-1)
  long long batch;
  int K;
  int kern_len;
  int conj_kernel;   // correlation
  long long in_offset, in_batch_stride, in_stride;
  long long out_offset, out_kernel_stride, out_batch_stride, out_stride;
  float scale;       // 1/N
};

template <int N_, int R0_, int R1_, int TL_>
struct ConvCfg {
  static constexpr int N = N_, R0 = R0_, R1 = R1_, TL = TL_;
  static_assert(R0 * R1 == N && R0 >= R1, "two stages, R0 the larger radix");
  static constexpr int E = R0;                 // complex values per thread
  static constexpr int TPL = N / R0;           // threads per line
  static constexpr int NB1 = R0 / R1;          // stage-1 butterflies per thread
  static constexpr int THREADS = TL * TPL;
  static constexpr int PADSH = ilog2(R0);
  static constexpr int PITCH = N + (N >> PADSH) + 2;
  static constexpr int TW_ELEMS = (R1 - 1) * R0;
  static constexpr int LDS_BYTES = (TL * PITCH + TW_ELEMS) * 8;
  static_assert(THREADS <= 1024 && LDS_BYTES <= 160 * 1024, "workgroup limits");
};

template <class C> MI_DEV int conv_lds(int line, int idx) { return line * C::PITCH + idx + (idx >> C::PADSH); }

// stage 0 of a forward transform on values already in v[] (element idx = u + q*TPL for q < R0): FFT_R0, then
// Stockham write idx' = u*R0 + q
template <class C> MI_DEV void conv_stage0_write(cf (&v)[C::E], cf* lds, int line, int u) {
  fft_radix<C::R0>(v);
#pragma unroll
  for (int q = 0; q < C::R0; ++q) lds[conv_lds<C>(line, u * C::R0 + q)] = v[q];
}
// stage 1: read idx = j + q*(N/R1), twiddle by e^{-2 pi i q (j % R0)/N}, FFT_R1; results land in natural order
// at idx = j + q*(N/R1) (last Stockham stage), returned in v[b*R1 + q]
template <class C> MI_DEV void conv_stage1(cf (&v)[C::E], const cf* lds, const cf* tw, int line, int u) {
#pragma unroll
  for (int b = 0; b < C::NB1; ++b) {
    const int j = u + b * C::TPL;
    const int k = j % C::R0;
    cf w[C::R1];
#pragma unroll
    for (int q = 0; q < C::R1; ++q) w[q] = lds[conv_lds<C>(line, j + q * (C::N / C::R1))];
#pragma unroll
    for (int q = 1; q < C::R1; ++q) w[q] = cmul(w[q], tw[(q - 1) * C::R0 + k]);
    fft_radix<C::R1>(w);
#pragma unroll
    for (int q = 0; q < C::R1; ++q) v[b * C::R1 + q] = w[q];
  }
}

template <class C>
__global__ void __launch_bounds__(C::THREADS) fftconv_fused_kernel(const FusedConvArgs a) {
  MI_SMEM_DECL(smem);
  cf* lds = reinterpret_cast<cf*>(smem);
  cf* tw = lds + C::TL * C::PITCH;
  const int t = threadIdx.x;
  const int line = t / C::TPL, u = t % C::TPL;
  for (int i = t; i < C::TW_ELEMS; i += C::THREADS) tw[i] = a.tw[i];
  const bool live = line <= a.K;          // line 0 = data, 1..K = kernels; the rest of the workgroup idles

  for (long long b = blockIdx.x; b < a.batch; b += gridDim.x) {
    cf v[C::E];
    // ---- forward transforms of the data line and of every kernel line ----
#pragma unroll
    for (int q = 0; q < C::R0; ++q) {
      const int idx = u + q * C::TPL;
      cf x = {0.0f, 0.0f};
      if (line == 0) x = a.in[a.in_offset + b * a.in_batch_stride + (long long)idx * a.in_stride];
      else if (live && idx < a.kern_len) x = a.kern[(long long)(line - 1) * a.kern_len + idx];
      v[q] = x;
    }
    __syncthreads();                       // previous batch entry's readers are done with LDS (and tw is staged)
    conv_stage0_write<C>(v, lds, line, u);
    __syncthreads();
    conv_stage1<C>(v, lds, tw, line, u);
    __syncthreads();
#pragma unroll
    for (int bb = 0; bb < C::NB1; ++bb)
#pragma unroll
      for (int q = 0; q < C::R1; ++q) lds[conv_lds<C>(line, u + bb * C::TPL + q * (C::N / C::R1))] = v[bb * C::R1 + q];
    __syncthreads();
    // ---- per kernel line: product with the data spectrum, inverse transform (swap trick), scaled scatter ----
#pragma unroll
    for (int q = 0; q < C::R0; ++q) {
      const int idx = u + q * C::TPL;
      const cf xs = lds[conv_lds<C>(0, idx)];
      const cf hs = lds[conv_lds<C>(line, idx)];
      const cf y = a.conj_kernel ? cmul_conj(xs, hs) : cmul(xs, hs);
      v[q] = y.yx;                         // ifft(y) = swap(fft(swap(y)))
    }
    __syncthreads();
    conv_stage0_write<C>(v, lds, line, u);
    __syncthreads();
    conv_stage1<C>(v, lds, tw, line, u);
    if (live && line > 0) {
      cf* o = a.out + a.out_offset + (long long)(line - 1) * a.out_kernel_stride + b * a.out_batch_stride;
#pragma unroll
      for (int bb = 0; bb < C::NB1; ++bb)
#pragma unroll
        for (int q = 0; q < C::R1; ++q) {
          const int idx = u + bb * C::TPL + q * (C::N / C::R1);
          const cf r = v[bb * C::R1 + q] * a.scale;
          o[(long long)idx * a.out_stride] = r.yx;
        }
    }
  }
}

}  // namespace mi355

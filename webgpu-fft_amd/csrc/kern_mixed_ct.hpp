// kern_mixed_ct.hpp — mixed-radix line FFTs with COMPILE-TIME radix plans (r02).
//
// kern_mixed.hpp serves every 13-smooth length with one kernel whose radix plan, tile shape and index arithmetic are runtime
// arguments; it removed the per-stage HBM round trips of the stage route but is latency-bound (47-102 GPoints/s): per work item
// a reciprocal-multiply division chain, a switch on the radix, loops of unknown trip count.  Here the commonly used lengths
// (3*2^k, 5*2^k, 1000, the reference's test sizes: complete.suite.js:876-913) get one instantiation each: length, radices, lines
// per workgroup and threads are template constants, so every index is a shift / constant multiply, every loop is unrolled and the
// stage sequence is straight-line code.  Same data flow as kern_mixed.hpp (dense lines, S == 1): T lines per workgroup ping-pong
// between two padded LDS buffers, stage 0 reads HBM, the last stage writes it; stage tables [R][Ns_prev] ride in LDS.
// Replaces, for these lengths, the reference's one-dispatch-per-radix loop (plan.js:1250-1259, stockham_stage.js:17-106).
#pragma once
#include "kern_mixed.hpp"

namespace mi355 {

template <int N_, int T_, int THREADS_, int... RS>
struct MixedCt {
  static constexpr int N = N_, T = T_, THREADS = THREADS_, NST = (int)sizeof...(RS);
  static constexpr int R[NST] = {RS...};
  static constexpr int PITCH = N + (N >> 5) + 1;                  // kern_mixed.hpp mixed_pitch
  static constexpr int nsp(int s) { int p = 1; for (int i = 0; i < s; ++i) p *= R[i]; return p; }
  static constexpr int tw_off(int s) { int o = 0; for (int i = 0; i < s; ++i) o += R[i] * nsp(i); return o; }
  static constexpr int TW_TOTAL = tw_off(NST);
  static constexpr int LDS_BYTES = (2 * T * PITCH + TW_TOTAL) * 8;
  static_assert(nsp(NST) == N, "radix product");
  static_assert(LDS_BYTES <= 160 * 1024, "tile does not fit LDS");
};

template <class M, int S>
MI_DEV void mixedct_stage(const MixedArgs& a, const cf* tw, const cf* lin, cf* lout, long long L0, int tl) {
  constexpr int R = M::R[S], NB = M::N / R, NSP = M::nsp(S);
  constexpr bool FIRST = S == 0, LAST = S == M::NST - 1;
  constexpr int WORK = M::T * NB, ITERS = (WORK + M::THREADS - 1) / M::THREADS;
  const cf* twp = tw + M::tw_off(S);
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int w = it * M::THREADS + (int)threadIdx.x;
    const int line = w / NB, j = w - line * NB;
    if (line >= tl) continue;
    const int blk = j / NSP, k = j - blk * NSP;
    const long long base = (L0 + line) * (long long)M::N;
    cf v[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const int idx = j + q * NB;
      cf x;
      if constexpr (FIRST) { x = a.in[base + idx]; if (a.swap_in) x = x.yx; }
      else x = lin[line * M::PITCH + idx + (idx >> 5)];
      if constexpr (!FIRST) { if (q > 0) x = cmul(x, twp[q * NSP + k]); }
      v[q] = x;
    }
    fft_radix<R>(v);
    const int ob = blk * (NSP * R) + k;
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const int idx = ob + q * NSP;
      if constexpr (LAST) {
        cf y = v[q] * a.scale;
        if (a.swap_out) y = y.yx;
        a.out[base + idx] = y;
      } else {
        lout[line * M::PITCH + idx + (idx >> 5)] = v[q];
      }
    }
  }
}

template <class M, int S>
MI_DEV void mixedct_run(const MixedArgs& a, const cf* tw, cf* b0, cf* b1, long long L0, int tl) {
  mixedct_stage<M, S>(a, tw, (S & 1) ? b0 : b1, (S & 1) ? b1 : b0, L0, tl);     // stage s reads what stage s-1 wrote
  if constexpr (S + 1 < M::NST) {
    __syncthreads();
    mixedct_run<M, S + 1>(a, tw, b0, b1, L0, tl);
  }
}

template <class M>
__global__ void __launch_bounds__(M::THREADS) fft_lines_mixedct_kernel(const MixedArgs a) {
  MI_SMEM_DECL(smem);
  cf* b0 = reinterpret_cast<cf*>(smem);
  cf* b1 = b0 + M::T * M::PITCH;
  cf* tw = b1 + M::T * M::PITCH;
  for (int i = (int)threadIdx.x; i < M::TW_TOTAL; i += M::THREADS) tw[i] = a.tw[i];
  __syncthreads();
  const long long tiles = (a.lines + M::T - 1) / M::T;
  for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const long long L0 = tile * M::T;
    const int tl = (int)((a.lines - L0) < (long long)M::T ? (a.lines - L0) : (long long)M::T);
    mixedct_run<M, 0>(a, tw, b0, b1, L0, tl);
    __syncthreads();   // the buffers are re-used by the next tile
  }
}

}  // namespace mi355

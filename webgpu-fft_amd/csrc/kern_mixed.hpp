// kern_mixed.hpp — one-launch line FFT for mixed-radix lengths (factors 2,3,4,5,7,8,11,13,16,32; N <= 4096): T lines per
// workgroup live in LDS between the Stockham stages, so a line crosses HBM once in each direction whatever its number of
// stages (the global-memory stage route of kern_generic.hpp pays 16 B per point per stage).  The radix plan is a runtime
// argument — one kernel serves every length — and each stage dispatches to a compile-time butterfly.
// Replaces, for these lengths, the reference's one-dispatch-per-radix loop (`plan.js:1250-1259`, `stockham_stage.js:17-106`).
//
// Lines of an N-D array as in StageArgs: line L -> (o = L / S, inner = L % S), element p at o*S*N + inner + p*S.
// S == 1: lanes walk a line (LDS line-major);  S > 1: lanes walk T adjacent lines (LDS index-major) so that global
// accesses are runs of T consecutive elements.
#pragma once
#include "platform.hpp"
#include "radix.hpp"

namespace mi355 {

constexpr int MIXED_LDS_BYTES = 64 * 1024;   // two buffers of T*N points
constexpr int MIXED_MAX_STAGES = 12;

struct MixedArgs {
  const cf* in;
  cf* out;
  const cf* tw;                       // per stage [R][Ns_prev] (row 0 unused), concatenated
  long long lines;
  long long S;
  int N, T, nst;
  int radix[MIXED_MAX_STAGES];
  int tw_off[MIXED_MAX_STAGES];       // elements
  float scale;
  int swap_in, swap_out;
};

template <int R>
MI_DEV void mixed_stage(const MixedArgs& a, const cf* twp, int nsp, bool first, bool last, const cf* lin, cf* lout, long long L0, int tl) {
  const int nb = a.N / R;
  const int work = tl * nb;
  const bool row = a.S == 1;
  for (int w = (int)threadIdx.x; w < work; w += (int)blockDim.x) {
    int line, j;
    if (row) { line = w / nb; j = w - line * nb; } else { j = w / tl; line = w - j * tl; }
    const int k = j % nsp;
    long long base = 0;
    if (first || last) {
      const long long L = L0 + line, o = L / a.S;
      base = o * a.S * a.N + (L - o * a.S);
    }
    cf v[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const int idx = j + q * nb;
      cf x;
      if (first) { x = a.in[base + (long long)idx * a.S]; if (a.swap_in) x = x.yx; }
      else x = lin[row ? line * a.N + idx : idx * a.T + line];
      if (q > 0 && nsp > 1) x = cmul(x, twp[q * nsp + k]);
      v[q] = x;
    }
    fft_radix<R>(v);
    const int ob = (j / nsp) * (nsp * R) + k;
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const int idx = ob + q * nsp;
      if (last) {
        cf y = v[q] * a.scale;
        if (a.swap_out) y = y.yx;
        a.out[base + (long long)idx * a.S] = y;
      } else {
        lout[row ? line * a.N + idx : idx * a.T + line] = v[q];
      }
    }
  }
}

static __global__ void __launch_bounds__(256) fft_lines_mixed_kernel(const MixedArgs a) {
  MI_SMEM_DECL(smem);
  cf* buf[2] = {reinterpret_cast<cf*>(smem), reinterpret_cast<cf*>(smem) + (size_t)a.T * a.N};
  const long long tiles = (a.lines + a.T - 1) / a.T;
  for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const long long L0 = tile * a.T;
    const int tl = (int)((a.lines - L0) < (long long)a.T ? (a.lines - L0) : (long long)a.T);
    int nsp = 1;
    for (int s = 0; s < a.nst; ++s) {
      const bool first = s == 0, last = s == a.nst - 1;
      const cf* twp = a.tw + a.tw_off[s];
      const cf* lin = buf[(s + 1) & 1];
      cf* lout = buf[s & 1];
      switch (a.radix[s]) {
#define MI_MIXED_CASE(R) case R: mixed_stage<R>(a, twp, nsp, first, last, lin, lout, L0, tl); break;
        MI_MIXED_CASE(2) MI_MIXED_CASE(3) MI_MIXED_CASE(4) MI_MIXED_CASE(5) MI_MIXED_CASE(7) MI_MIXED_CASE(8)
        MI_MIXED_CASE(11) MI_MIXED_CASE(13) MI_MIXED_CASE(16) MI_MIXED_CASE(32)
#undef MI_MIXED_CASE
      }
      nsp *= a.radix[s];
      __syncthreads();
    }
  }
}

}  // namespace mi355

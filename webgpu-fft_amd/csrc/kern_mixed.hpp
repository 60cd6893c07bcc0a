// kern_mixed.hpp — one-launch line FFT for mixed-radix lengths (factors 2,3,4,5,7,8,11,13,16,32; N <= 4096): T lines per
// workgroup live in LDS between the Stockham stages, so a line crosses HBM once in each direction whatever its number of
// stages (the global-memory stage route of kern_generic.hpp pays 16 B per point per stage).  The radix plan is a runtime
// argument — one kernel serves every length — and each stage dispatches to a compile-time butterfly.
// Replaces, for these lengths, the reference's one-dispatch-per-radix loop (`plan.js:1250-1259`, `stockham_stage.js:17-106`).
// Measured (profiles/r01_mixed_radix.log): 47-102 GPoints/s, +10...+37 % over the stage route on most lengths; the
// runtime-plan stage loop is latency-bound, not bandwidth-bound, so compile-time radix plans for the common lengths (as
// the power-of-two line kernels have) remain the next step.  Radices are capped at 8 here: a stage's parallelism is N/R
// butterflies per line.
//
// Lines of an N-D array as in StageArgs: line L -> (o = L / S, inner = L % S), element p at o*S*N + inner + p*S.
// S == 1: lanes walk a line (LDS line-major);  S > 1: lanes walk T adjacent lines (LDS index-major) so that global
// accesses are runs of T consecutive elements.
#pragma once
#include "platform.hpp"
#include "radix.hpp"

namespace mi355 {

constexpr int MIXED_MAX_T = 64;
constexpr int MIXED_THREADS = 512;
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
  unsigned rcp_nb[MIXED_MAX_STAGES];  // ceil(2^32 / (N / R_s)): butterfly index -> line (0 when N/R_s == 1)
  unsigned rcp_nsp[MIXED_MAX_STAGES]; // ceil(2^32 / Ns_prev)                          (0 when Ns_prev == 1)
  float scale;
  int swap_in, swap_out;
  int lds_bytes;                      // bytes of the two line buffers; the per-line offsets and the stage tables sit behind them
  int tw_total;                       // elements of all stage tables if they are staged in LDS, else 0
};

MI_DEV unsigned mixed_pitch(int N) { return (unsigned)(N + (N >> 5) + 1); }   // elements per line buffer (S == 1 layout)

// floor(n / d) for n < 2^20 through the precomputed reciprocal r = ceil(2^32 / d) (r == 0 encodes d == 1)
MI_DEV unsigned mixed_div(unsigned n, unsigned r) { return r ? MI_UMULHI(n, r) : n; }

template <int R>
MI_DEV void mixed_stage(const MixedArgs& a, const cf* tw_lds, int s, int nsp, bool first, bool last, const cf* lin, cf* lout, const long long* s_base, int tl) {
  const unsigned nb = (unsigned)(a.N / R);
  const unsigned work = (unsigned)tl * nb;
  const bool row = a.S == 1;
  // contiguous lines are stored line-major with one pad element per 32 (a radix-32/16/8 first stage scatters its outputs
  // with a power-of-two stride: unpadded, a wave would hit one LDS bank)
  const unsigned pitch = mixed_pitch(a.N);
  const cf* twp = (a.tw_total ? tw_lds : a.tw) + a.tw_off[s];
  const unsigned rnb = a.rcp_nb[s], rnsp = a.rcp_nsp[s];
  const unsigned rtl = tl > 1 ? (unsigned)((0x100000000ull + (unsigned)tl - 1) / (unsigned)tl) : 0u;
  for (unsigned w = threadIdx.x; w < work; w += blockDim.x) {
    unsigned line, j;
    if (row) { line = mixed_div(w, rnb); j = w - line * nb; } else { j = mixed_div(w, rtl); line = w - j * (unsigned)tl; }
    const unsigned blk = mixed_div(j, rnsp);          // j / Ns_prev
    const unsigned k = j - blk * (unsigned)nsp;
    const long long base = (first || last) ? s_base[line] : 0;
    cf v[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const unsigned idx = j + q * nb;
      cf x;
      if (first) { x = a.in[base + (long long)idx * a.S]; if (a.swap_in) x = x.yx; }
      else x = lin[row ? line * pitch + idx + (idx >> 5) : idx * a.T + line];
      if (q > 0 && nsp > 1) x = cmul(x, twp[q * nsp + k]);
      v[q] = x;
    }
    fft_radix<R>(v);
    const unsigned ob = blk * (unsigned)(nsp * R) + k;
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const unsigned idx = ob + q * nsp;
      if (last) {
        cf y = v[q] * a.scale;
        if (a.swap_out) y = y.yx;
        a.out[base + (long long)idx * a.S] = y;
      } else {
        lout[row ? line * pitch + idx + (idx >> 5) : idx * a.T + line] = v[q];
      }
    }
  }
}

static __global__ void __launch_bounds__(MIXED_THREADS) fft_lines_mixed_kernel(const MixedArgs a) {
  MI_SMEM_DECL(smem);
  cf* buf[2] = {reinterpret_cast<cf*>(smem), reinterpret_cast<cf*>(smem) + (size_t)a.T * mixed_pitch(a.N)};
  long long* s_base = reinterpret_cast<long long*>(smem + a.lds_bytes);   // element offset of each line of the tile
  // stage tables -> LDS once per workgroup (the inner loops then touch global memory for the line data only)
  cf* tw_lds = reinterpret_cast<cf*>(smem + a.lds_bytes + MIXED_MAX_T * 8);
  for (int i = (int)threadIdx.x; i < a.tw_total; i += (int)blockDim.x) tw_lds[i] = a.tw[i];
  __syncthreads();
  const long long tiles = (a.lines + a.T - 1) / a.T;
  for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const long long L0 = tile * a.T;
    const int tl = (int)((a.lines - L0) < (long long)a.T ? (a.lines - L0) : (long long)a.T);
    if ((int)threadIdx.x < tl) {
      const long long L = L0 + threadIdx.x, o = L / a.S;
      s_base[threadIdx.x] = o * a.S * a.N + (L - o * a.S);
    }
    __syncthreads();
    int nsp = 1;
    for (int s = 0; s < a.nst; ++s) {
      const bool first = s == 0, last = s == a.nst - 1;
      const cf* lin = buf[(s + 1) & 1];
      cf* lout = buf[s & 1];
      switch (a.radix[s]) {
#define MI_MIXED_CASE(R) case R: mixed_stage<R>(a, tw_lds, s, nsp, first, last, lin, lout, s_base, tl); break;
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

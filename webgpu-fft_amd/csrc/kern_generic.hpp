// kern_generic.hpp — the route for everything the LDS line kernels do not cover (mixed radix 3/5/7/11/13,
// lines longer than the two-pass limit) plus the memory-bound glue kernels of the r2c / c2r / fftconv
// plans and the test/bench support kernels.
//
// Reference counterparts (paths into the reference repo):
//   stockham_stage_kernel   src/kernels/stockham_stage.js:17-106 — one Stockham stage, global -> global;
//                           one lane per BUTTERFLY (R loads, R stores) instead of one per output element
//                           (R loads, 1 store), roots from an f64-built table instead of in-shader cis().
//   scale_kernel            src/kernels/scale.js:3-30
//   r2c_post_kernel         src/kernels/real_complex.js:73-114 (pack) — here the half-length trick:
//                           X[k] from Z = FFT_{N/2}(x[2n] + i x[2n+1]), not a full-length complex FFT
//   c2r_pre_kernel          src/kernels/real_complex.js:116-201 (Hermitian unpack, self-conjugate bins'
//                           imaginary parts ignored) folded into the half-length pre-processing
//   real_to_complex / complex_to_real for odd N: src/kernels/real_complex.js:1-43
//   pointwise_mul_kernel    src/kernels/fft_conv.js:3-66
//   gather/scatter_strided  src/kernels/strided_complex.js:22-106
//   fill_random / sumsq     test + bench support: device twin of oracle.c's PRNG, f64 reductions
#pragma once
#include "platform.hpp"
#include "radix.hpp"

namespace mi355 {

// Lines of an N-D array, axis `a` of length N with element stride S = prod(shape[:a]):
//   line id L in [0, S*outer) -> inner = L % S, o = L / S; element p at o*S*N + inner + p*S.
struct StageArgs {
  const cf* in;
  cf* out;
  const cf* tw;        // this stage's table [R][Ns_prev]
  long long total;     // butterflies = lines * N / R
  long long N;         // line length
  long long S;         // element stride of the axis
  long long Nsp;       // Ns_prev (product of radices already applied)
  float scale;         // fused into the last stage
  int swap_in, swap_out;
};

template <int R>
static __global__ void __launch_bounds__(256) stockham_stage_kernel(const StageArgs a) {
  const long long nb = a.N / R;  // butterflies per line
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < a.total; g += (long long)gridDim.x * blockDim.x) {
    // consecutive threads walk the contiguous direction: inner index fastest when S > 1
    long long inner, j, o;
    if (a.S == 1) { o = g / nb; j = g - o * nb; inner = 0; }
    else { inner = g % a.S; const long long r = g / a.S; j = r % nb; o = r / nb; }
    const long long base = o * a.S * a.N + inner;
    const long long k = j % a.Nsp;
    cf w[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
      cf x = a.in[base + (j + q * nb) * a.S];
      if (a.swap_in) x = x.yx;
      if (q > 0 && a.Nsp > 1) x = cmul(x, a.tw[q * a.Nsp + k]);
      w[q] = x;
    }
    fft_radix<R>(w);
    const long long ob = (j / a.Nsp) * (a.Nsp * R) + k;
#pragma unroll
    for (int q = 0; q < R; ++q) {
      cf y = w[q] * a.scale;
      if (a.swap_out) y = y.yx;
      a.out[base + (ob + q * a.Nsp) * a.S] = y;
    }
  }
}

// copyBufferToBuffer for 16-byte-aligned ranges (WebGPU GPUCommandEncoder.copyBufferToBuffer, which the reference's downloadComplex
// and bench loops use): one short-lived workgroup per 16 KB slab, 16-byte nontemporal accesses — the form that reaches this chip's
// copy ceiling (6.4-6.5 TB/s of read + write traffic against 4.9-5.5 for hipMemcpy D2D, profiles/r02_copy_ceiling.log).  bench.py
// times it live as the fabric ceiling next to the transform (`roofline.attainable`).
typedef float f4v __attribute__((ext_vector_type(4)));
static __global__ void __launch_bounds__(256) stream_copy_kernel(const f4v* __restrict__ src, f4v* __restrict__ dst, unsigned long long n16) {
  for (unsigned long long slab = blockIdx.x; slab * 1024ull < n16; slab += gridDim.x) {
    const unsigned long long base = slab * 1024ull + threadIdx.x;
    f4v v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) if (base + 256ull * j < n16) v[j] = __builtin_nontemporal_load(src + base + 256ull * j);
#pragma unroll
    for (int j = 0; j < 4; ++j) if (base + 256ull * j < n16) __builtin_nontemporal_store(v[j], dst + base + 256ull * j);
  }
}

// data[i] *= s  (float granularity so the same kernel serves real and complex buffers)
static __global__ void __launch_bounds__(256) scale_kernel(float* data, long long count, float s) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x)
    data[i] *= s;
}

// r2c, even N, H = N/2.  Z[b][k] (k < H) = FFT_H of z[n] = x[2n] + i x[2n+1]; writes the packed spectrum
// X[b][k], k = 0..H (H+1 bins per line, reference packing docs/API.md "R2C/C2R packing"):
//   X[k] = (Z[k] + conj(Z[H-k]))/2 - (i/2) e^{-2 pi i k/N} (Z[k] - conj(Z[H-k])),  Z[H] := Z[0]
// The root e^{-2 pi i k/N} is formed as HI[k >> shift] * LO[k & mask] from two small cache-resident tables, not
// read from an (N/2+1)-entry table: at N = 2^22 that table would add 4 B per complex point of fabric traffic.
// the split passes stream every byte exactly once: nontemporal accesses (MI355_POST_NT, measured in profiles/r01_xcd_fused_ab.log)
#ifndef MI355_POST_NT
#define MI355_POST_NT 0
#endif
#if MI355_POST_NT
#define MI_POST_LD(p) __builtin_nontemporal_load(p)
#define MI_POST_ST(p, v) __builtin_nontemporal_store((v), (p))
#else
#define MI_POST_LD(p) (*(p))
#define MI_POST_ST(p, v) (*(p) = (v))
#endif
struct R2cPostArgs {
  const cf* z; cf* x; const cf* tw_lo; const cf* tw_hi;
  long long H, batch;
  long long x_line_stride;  // elements between packed lines (H+1 when dense)
  float scale;
  int shift; unsigned mask;
};
// Two adjacent bins per lane, moved as ONE 16-byte access on each of the four streams (Z ascending, Z mirrored, X ascending, X
// mirrored): k and H-k have the same parity, so one stream of every pair starts on an odd element — the vector type below is
// declared 8-byte aligned and the hardware splits nothing but the first and last line of a wave's 1 KiB run.  8-byte-per-lane
// accesses moved this pass at 4.1 TB/s (profiles/r01_rocprof_r2c_2p22_kernel_stats.csv).
typedef float f4u __attribute__((ext_vector_type(4), aligned(8)));
#if MI355_POST_NT
#define MI_POST_LD4(p) __builtin_nontemporal_load(reinterpret_cast<const f4u*>(p))
#define MI_POST_ST4(p, v) __builtin_nontemporal_store((v), reinterpret_cast<f4u*>(p))
#else
#define MI_POST_LD4(p) (*reinterpret_cast<const f4u*>(p))
#define MI_POST_ST4(p, v) (*reinterpret_cast<f4u*>(p) = (v))
#endif
// Work item = (line b, chunk of 256*U bin pairs): the divisions are wave-uniform (scalar), lanes walk k.
// One lane forms X[k], X[k+1] and their mirrors X[H-k], X[H-k-1] from (Z[k], Z[k+1], Z[H-k-1], Z[H-k]), k = 0, 2, .. <= H/2, so every Z
// is read once:   X[k] = E + w O,  X[H-k] = conj(E - w O),  E = (Z[k] + conj Z[H-k])/2,  O = -i (Z[k] - conj Z[H-k])/2,  w = e^{-2 pi i k/N}
static __global__ void __launch_bounds__(256) r2c_post_kernel(const R2cPostArgs a) {
  constexpr int U = 2;                       // bin pairs per lane per item: 4 independent 16-byte loads in flight
  const long long half = a.H / 2;            // bins 0..half are formed here (with their mirrors half..H)
  const long long pairs = half / 2 + 1;      // pairs (0,1), (2,3), .. up to the one that holds bin `half`
  const long long chunks = (pairs + 256 * U - 1) / (256 * U);
  // short lines (at most 256 pairs, N <= 2044): a 256-lane slice holds 256 >> lpsh whole lines of 1 << lpsh lanes each, so that
  // the lanes of an item are not mostly idle (N = 64: 9 of 512 lanes worked before — 48 vs 380 G real points/s for the c2r twin)
  const bool packed = pairs <= 256;
  int lpsh = 0;
  if (packed) while ((1ll << lpsh) < pairs) ++lpsh;
  const long long per_slice = packed ? (256 >> lpsh) : 0, per_item = per_slice * U;
  const long long items = packed ? (a.batch + per_item - 1) / per_item : a.batch * chunks;
  const cf w1 = a.tw_lo[1 & a.mask];         // e^{-2 pi i/N}: root of the odd bin of a pair from the even one
  // (line, first pair index) of slice j of item `it` for this lane; false: nothing to do
  const auto locate = [&](long long it, int j, long long& b, long long& p) {
    if (packed) {
      b = it * per_item + j * per_slice + ((long long)threadIdx.x >> lpsh);
      p = (long long)threadIdx.x & ((1ll << lpsh) - 1);
      return b < a.batch && p < pairs;
    }
    b = it / chunks;
    p = (it - b * chunks) * (256 * U) + threadIdx.x + j * 256;
    return true;
  };
  const auto split = [&](cf zk, cf zm, cf w, cf& xk, cf& xm) {
    const cf zmc = {zm.x, -zm.y};
    const cf e = (zk + zmc) * 0.5f;
    const cf od = mul_neg_i((zk - zmc) * 0.5f);
    const cf wo = cmul(w, od);
    xk = (e + wo) * a.scale;
    xm = (e - wo) * a.scale;
    xm.y = -xm.y;
  };
  for (long long it = blockIdx.x; it < items; it += gridDim.x) {
    f4u za[U], zb[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
      long long b, p;
      const bool live = locate(it, j, b, p);
      const cf* z = a.z + (live ? b : 0) * a.H;
      const long long k = 2 * p;
      // full pair: both bins <= half and the second one is not its own mirror; everything else (first pair's Z[H] := Z[0], the
      // last pair of a line, lanes beyond the line) goes element by element below
      const bool full = live && k > 0 && 2 * (k + 1) < a.H;
      const long long kc = full ? k : 2;                                  // clamped lanes read somewhere harmless and in range
      if (a.H >= 8) { za[j] = MI_POST_LD4(z + kc); zb[j] = MI_POST_LD4(z + (a.H - kc - 1)); }
    }
#pragma unroll
    for (int j = 0; j < U; ++j) {
      long long b, p;
      if (!locate(it, j, b, p)) continue;
      const cf* z = a.z + b * a.H;
      cf* x = a.x + b * a.x_line_stride;
      const long long k = 2 * p;
      if (k > half) continue;
      const bool full = k > 0 && 2 * (k + 1) < a.H && a.H >= 8;
      const cf w0 = cmul(a.tw_hi[(unsigned)k >> a.shift], a.tw_lo[(unsigned)k & a.mask]);
      if (full) {
        cf x0, m0, x1, m1;
        split(cf{za[j].x, za[j].y}, cf{zb[j].z, zb[j].w}, w0, x0, m0);                  // (Z[k],   Z[H-k])
        split(cf{za[j].z, za[j].w}, cf{zb[j].x, zb[j].y}, cmul(w0, w1), x1, m1);        // (Z[k+1], Z[H-k-1])
        const f4u up = {x0.x, x0.y, x1.x, x1.y}, dn = {m1.x, m1.y, m0.x, m0.y};
        MI_POST_ST4(x + k, up);
        MI_POST_ST4(x + (a.H - k - 1), dn);
      } else {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const long long kk = k + e;
          if (kk > half) continue;
          const long long km = kk == 0 ? 0 : a.H - kk;
          cf xk, xm;
          split(z[kk], z[km], e ? cmul(w0, w1) : w0, xk, xm);
          x[kk] = xk;
          if (kk == 0) x[a.H] = xm;                                       // X[H] = E[0] - O[0]
          else if (km != kk) x[km] = xm;
        }
      }
    }
  }
}

// c2r, even N, H = N/2.  Builds Z[b][k], k < H, such that IFFT_H(Z)[n] = x[2n] + i x[2n+1] (unnormalised
// by the full-length convention: the 1/2 of the split is absorbed):
//   E[k] = (X[k] + conj(X[H-k])),  O[k] = (X[k] - conj(X[H-k])) e^{+2 pi i k/N},  Z[k] = E[k] + i O[k]
// The imaginary parts of X[0] and X[H] are ignored (Hermitian unpack: real_complex.js:147-155,194-197).
// Pair form: Z[k] = E + i O and Z[H-k] = conj(E) + i conj(O) from (X[k], X[H-k]), k = 0..H/2.
struct C2rPreArgs {
  const cf* x; cf* z; const cf* tw_lo; const cf* tw_hi;
  long long H, batch;
  long long x_line_stride;
  int shift; unsigned mask;
};
static __global__ void __launch_bounds__(256) c2r_pre_kernel(const C2rPreArgs a) {
  constexpr int U = 2;                       // bin pairs per lane per item, 16-byte accesses as in r2c_post_kernel
  const long long half = a.H / 2;
  const long long pairs = half / 2 + 1;
  const long long chunks = (pairs + 256 * U - 1) / (256 * U);
  const bool packed = pairs <= 256;          // short lines: whole lines side by side in a 256-lane slice (see r2c_post_kernel)
  int lpsh = 0;
  if (packed) while ((1ll << lpsh) < pairs) ++lpsh;
  const long long per_slice = packed ? (256 >> lpsh) : 0, per_item = per_slice * U;
  const long long items = packed ? (a.batch + per_item - 1) / per_item : a.batch * chunks;
  const cf w1 = a.tw_lo[1 & a.mask];
  const auto locate = [&](long long it, int j, long long& b, long long& p) {
    if (packed) {
      b = it * per_item + j * per_slice + ((long long)threadIdx.x >> lpsh);
      p = (long long)threadIdx.x & ((1ll << lpsh) - 1);
      return b < a.batch && p < pairs;
    }
    b = it / chunks;
    p = (it - b * chunks) * (256 * U) + threadIdx.x + j * 256;
    return true;
  };
  const auto merge = [&](cf p, cf q, cf w, cf& zk, cf& zm) {
    const cf qc = {q.x, -q.y};
    const cf e = p + qc;
    const cf o = cmul_conj(p - qc, w);       // * e^{+2 pi i k/N}
    zk = e + mul_pos_i(o);
    const cf ec = {e.x, -e.y}, oc = {o.x, -o.y};
    zm = ec + mul_pos_i(oc);
  };
  for (long long it = blockIdx.x; it < items; it += gridDim.x) {
    f4u xa[U], xb[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
      long long b, p;
      const bool live = locate(it, j, b, p);
      const cf* x = a.x + (live ? b : 0) * a.x_line_stride;
      const long long k = 2 * p;
      const bool full = live && k > 0 && 2 * (k + 1) < a.H;
      const long long kc = full ? k : 2;
      if (a.H >= 8) { xa[j] = MI_POST_LD4(x + kc); xb[j] = MI_POST_LD4(x + (a.H - kc - 1)); }
    }
#pragma unroll
    for (int j = 0; j < U; ++j) {
      long long b, p;
      if (!locate(it, j, b, p)) continue;
      const cf* x = a.x + b * a.x_line_stride;
      cf* z = a.z + b * a.H;
      const long long k = 2 * p;
      if (k > half) continue;
      const bool full = k > 0 && 2 * (k + 1) < a.H && a.H >= 8;
      const cf w0 = cmul(a.tw_hi[(unsigned)k >> a.shift], a.tw_lo[(unsigned)k & a.mask]);
      if (full) {
        cf z0, m0, z1, m1;
        merge(cf{xa[j].x, xa[j].y}, cf{xb[j].z, xb[j].w}, w0, z0, m0);                  // (X[k],   X[H-k])
        merge(cf{xa[j].z, xa[j].w}, cf{xb[j].x, xb[j].y}, cmul(w0, w1), z1, m1);        // (X[k+1], X[H-k-1])
        const f4u up = {z0.x, z0.y, z1.x, z1.y}, dn = {m1.x, m1.y, m0.x, m0.y};
        MI_POST_ST4(z + k, up);
        MI_POST_ST4(z + (a.H - k - 1), dn);
      } else {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const long long kk = k + e;
          if (kk > half) continue;
          cf p = x[kk], q = x[a.H - kk];
          if (kk == 0) { p.y = 0.0f; q.y = 0.0f; }   // self-conjugate bins: imaginary parts ignored
          cf zk, zm;
          merge(p, q, e ? cmul(w0, w1) : w0, zk, zm);
          z[kk] = zk;
          if (kk != 0 && a.H - kk != kk) z[a.H - kk] = zm;
        }
      }
    }
  }
}

// odd-N real paths: expand real -> complex (imag 0) and take the real part
static __global__ void __launch_bounds__(256) real_to_complex_kernel(const float* x, cf* z, long long count) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x) {
    cf v = {x[i], 0.0f};
    z[i] = v;
  }
}
static __global__ void __launch_bounds__(256) complex_to_real_kernel(const cf* z, float* x, long long count, float s) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x)
    x[i] = z[i].x * s;
}
// full complex line (N) -> first N/2+1 bins, and the Hermitian expansion back (odd N route)
static __global__ void __launch_bounds__(256) pack_half_kernel(const cf* full, cf* packed, long long N, long long P, long long batch, long long packed_stride, float s) {
  const long long total = batch * P;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
    const long long b = g / P, k = g - b * P;
    packed[b * packed_stride + k] = full[b * N + k] * s;
  }
}
static __global__ void __launch_bounds__(256) unpack_hermitian_kernel(const cf* packed, cf* full, long long N, long long P, long long batch, long long packed_stride) {
  const long long total = batch * N;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
    const long long b = g / N, k = g - b * N;
    cf v;
    if (k < P) { v = packed[b * packed_stride + k]; if (k == 0 || 2 * k == N) v.y = 0.0f; }
    else { const cf m = packed[b * packed_stride + (N - k)]; v.x = m.x; v.y = -m.y; }
    full[g] = v;
  }
}

// Bluestein chirp-z (reference: src/runtime/algorithms/bluestein_axis.js:59-, src/kernels/bluestein.js) for
// lengths with a prime factor > 13:  X[k] = a[k] * sum_n (x[n] a[n]) conj(a)[k-n],  a[n] = e^{-i pi n^2/N},
// evaluated as a circular convolution of power-of-two length M >= 2N-1 on the fast routes.
//   pre : y[l][m] = x[l][m] * a[m] for m < N, 0 for N <= m < M      post: X[l][k] = z[l][k] * a[k] * scale
struct ChirpArgs {
  const cf* in; cf* out; const cf* chirp;
  long long N, M, lines;
  float scale;
  int swap_in, swap_out;
};
static __global__ void __launch_bounds__(256) bluestein_pre_kernel(const ChirpArgs a) {
  const long long total = a.lines * a.M;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
    const long long l = g / a.M, m = g - l * a.M;
    cf v = {0.0f, 0.0f};
    if (m < a.N) {
      cf x = a.in[l * a.N + m];
      if (a.swap_in) x = x.yx;
      v = cmul(x, a.chirp[m]);
    }
    a.out[g] = v;
  }
}
static __global__ void __launch_bounds__(256) bluestein_post_kernel(const ChirpArgs a) {
  const long long total = a.lines * a.N;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
    const long long l = g / a.N, k = g - l * a.N;
    cf v = cmul(a.in[l * a.M + k], a.chirp[k]) * a.scale;
    if (a.swap_out) v = v.yx;
    a.out[g] = v;
  }
}

// data[b][i] *= (conj?) kern[i]   — frequency-domain product of fftconv (fft_conv.js:3-31)
static __global__ void __launch_bounds__(256) pointwise_mul_kernel(const cf* data, cf* out, const cf* kern, long long L, long long total, int conj_kernel, float s) {
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
    const cf k = kern[g % L];
    const cf d = data[g];
    out[g] = (conj_kernel ? cmul_conj(d, k) : cmul(d, k)) * s;
  }
}

// Strided physical <-> dense logical, rank <= 8 (strided_complex.js:22-106).  Logical index is axis-0
// fastest within a batch entry; physical element = offset + b*batch_stride + sum coord_d*stride_d.
// `sub` selects a window of the dense side: dense coords = coord + sub_offset inside dense_shape
// (embedding a small kernel / input into a zero-padded FFT domain, or cropping a result).
struct StridedArgs {
  const void* src; void* dst;     // elements are complex (cf) or real (float): strided_copy_kernel<GATHER, ELEM>
  long long total;                 // batch * prod(shape)
  long long per;                   // prod(shape)  (shape = extent of the moved region)
  int rank;
  long long shape[8];
  long long phys_stride[8];
  long long phys_offset, phys_batch_stride;
  long long dense_stride[8];       // strides of the dense side's full domain
  long long dense_offset, dense_batch_stride;
};
template <bool GATHER, class ELEM = cf>
static __global__ void __launch_bounds__(256) strided_copy_kernel(const StridedArgs a) {
  const ELEM* src = static_cast<const ELEM*>(a.src);
  ELEM* dst = static_cast<ELEM*>(a.dst);
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < a.total; g += (long long)gridDim.x * blockDim.x) {
    const long long b = g / a.per;
    long long rem = g - b * a.per;
    long long p = a.phys_offset + b * a.phys_batch_stride;
    long long d = a.dense_offset + b * a.dense_batch_stride;
    for (int i = 0; i < a.rank; ++i) {
      const long long c = rem % a.shape[i];
      rem /= a.shape[i];
      p += c * a.phys_stride[i];
      d += c * a.dense_stride[i];
    }
    if constexpr (GATHER) dst[d] = src[p]; else dst[p] = src[d];
  }
}

// zeroPad stages (src/kernels/zero_pad.js:21-80): zero every logical element outside the box [start, end)
struct ZeroOutsideArgs {
  void* data;                      // complex (cf) or real (float) elements: zero_outside_kernel<ELEM>
  long long total, per;
  int rank;
  long long shape[8], start[8], end[8];
};
template <class ELEM = cf>
static __global__ void __launch_bounds__(256) zero_outside_kernel(const ZeroOutsideArgs a) {
  ELEM* data = static_cast<ELEM*>(a.data);
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < a.total; g += (long long)gridDim.x * blockDim.x) {
    long long rem = g % a.per;
    bool inside = true;
    for (int i = 0; i < a.rank; ++i) {
      const long long c = rem % a.shape[i];
      rem /= a.shape[i];
      inside = inside && c >= a.start[i] && c < a.end[i];
    }
    if (!inside) { ELEM z = {}; data[g] = z; }
  }
}

static __global__ void __launch_bounds__(256) zero_kernel(float* data, long long count) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x) data[i] = 0.0f;
}

// ---- test / bench support ---------------------------------------------------------------------------
MI_DEV unsigned mulberry32_at(unsigned seed, unsigned long long n) {
  unsigned t = seed + (unsigned)((n + 1ull) * 0x6D2B79F5ull);
  t = (t ^ (t >> 15)) * (t | 1u);
  t ^= t + (t ^ (t >> 7)) * (t | 61u);
  return t ^ (t >> 14);
}
MI_DEV unsigned stream_seed(unsigned seed0, unsigned long long b) {
  unsigned h = seed0 + (unsigned)((b + 1ull) * 0x9E3779B9ull);
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
static __global__ void __launch_bounds__(256) fill_random_kernel(float* out, unsigned long long row_floats, unsigned long long rows, unsigned seed0,
                                                         unsigned long long first_transform) {
  const unsigned long long total = row_floats * rows;
  for (unsigned long long g = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (unsigned long long)gridDim.x * blockDim.x) {
    const unsigned long long r = g / row_floats, i = g - r * row_floats;
    const unsigned u = mulberry32_at(stream_seed(seed0, first_transform + r), i);
    const double x = (double)u / 4294967296.0;
    out[g] = (float)((x * 2.0 - 1.0) * 0.5);
  }
}

// partial[block] = sum over the block's grid-stride slice of (a[i] - alpha*b[i])^2 in f64 (b may be null)
static __global__ void __launch_bounds__(256) diff_sumsq_kernel(const float* a, const float* b, double alpha, unsigned long long count, double* partial) {
  double acc = 0.0;
  for (unsigned long long g = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; g < count; g += (unsigned long long)gridDim.x * blockDim.x) {
    const double d = (double)a[g] - (b ? alpha * (double)b[g] : 0.0);
    acc += d * d;
  }
  MI_SMEM_DECL_STATIC(double, red, 256);
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

}  // namespace mi355

// kern_trig.hpp — DCT-I..IV / DST-I..IV as a complex FFT of length L with a pre- and a post-pass per axis (SURVEY.md 8f rank 4;
// replaces src/runtime/plans/dct_fft.js for the real, f32 case).  General route: three launches + one FFT of length L per axis;
// dct2/dst2/dct3/dst3 along a dense axis of even length take the real-FFT route at the end of this file.
//
// kind (the typeKind table of dct_fft.js:48-57; dct3 / dst3 are dct2 / dst2 with the directions exchanged), theta = pi/(2N):
//   0 dct1      L = 2(N-1)  z = even extension of x                         X[k] = Re Z[k]
//   1 dct2 fwd  L = 2N      z[n] = x[n] (n < N), 0 beyond                   X[k] = Re(e^{-i theta k} Z[k])
//   2 dct2 inv  L = 2N      z[k] = c_k X[k] e^{+i theta k}, c_0 = 1/2       x[n] = Re IFFT(z)[n]   (unnormalised)
//   3 dct4      L = 2N      z[n] = x[n] e^{-i theta n}                      X[k] = Re(e^{-i theta (k+1/2)} Z[k])
//   4 dst1      L = 2(N+1)  z = odd extension of x (z[0] = z[N+1] = 0)      X[k] = -Im Z[k+1] / 2
//   5 dst2 fwd  L = 2N      as kind 1                                       X[k] = -Im(e^{-i theta (k+1)} Z[k+1])
//   6 dst2 inv  L = 2N      z[m] = c_m X[m-1] e^{+i theta m}, m = 1..N, c_N = 1/2     x[n] = Im IFFT(z)[n]
//   7 dst4      L = 2N      as kind 3                                       X[k] = -Im(e^{-i theta (k+1/2)} Z[k])
// (each identity follows from writing the cosine / sine of the definitions in math.js:291-409 as the real / imaginary
// part of a complex exponential and splitting the exponent into the FFT kernel and per-index phase factors)
//
// Lines of an N-D real array as in StageArgs: line G -> (o = G / S, inner = G % S), element p at o*S*N + inner + p*S.
// The complex work array holds the lines back to back: [line][L].
#pragma once
#include "platform.hpp"
#include "radix.hpp"

namespace mi355 {

struct TrigArgs {
  const float* x;     // pre: real input array; post: unused
  cf* z;              // complex work lines
  float* y;           // post: real output array
  long long lines, N, L, S;
  int kind;
  float scale;        // post only
  long long stride;   // kinds >= 8: element stride of the axis (1: the dense kernels; > 1: the tiled ones)
};

// e^{i pi t}, |t| <= 2.  The argument arrives in f64 (m / 2N is formed exactly enough there) and is reduced to f32 for the
// evaluation: an f64 sincospi per element (~100 f64 operations) held the phase passes at 3.5 TB/s
// (profiles/r01_rocprof_widened_rows.log); the f32 evaluation of pi*t is good to a few 1e-7 absolute, two orders below the
// 1e-5 bar.  MI355_TRIG_F64_PHASE=1 restores the f64 form.
#ifndef MI355_TRIG_F64_PHASE
#define MI355_TRIG_F64_PHASE 0
#endif
MI_DEV cf trig_phase(double turns_half) {
  cf r;
#if defined(MI355_HOST_EMU) && MI355_TRIG_F64_PHASE
  r.x = (float)std::cos(3.14159265358979323846 * turns_half); r.y = (float)std::sin(3.14159265358979323846 * turns_half);
#elif defined(MI355_HOST_EMU)   /* as shipped: the argument rounded to f32 first (sincospif's input), evaluated exactly */
  r.x = (float)std::cos(3.14159265358979323846 * (double)(float)turns_half); r.y = (float)std::sin(3.14159265358979323846 * (double)(float)turns_half);
#elif MI355_TRIG_F64_PHASE
  double s, c;
  sincospi(turns_half, &s, &c);
  r.x = (float)c; r.y = (float)s;
#else
  float s, c;
  sincospif((float)turns_half, &s, &c);
  r.x = c; r.y = s;
#endif
  return r;
}

static __global__ void __launch_bounds__(256) trig_pre_kernel(const TrigArgs a) {
  const long long total = a.lines * a.L;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
    const long long G = g / a.L, m = g - G * a.L;
    const long long o = G / a.S, base = o * a.S * a.N + (G - o * a.S);
    const auto X = [&](long long p) { return a.x[base + p * a.S]; };
    const double inv2n = 1.0 / (2.0 * (double)a.N);
    cf z; z.x = 0.0f; z.y = 0.0f;
    switch (a.kind) {
      case 0: z.x = X(m < a.N ? m : a.L - m); break;
      case 1: case 5: if (m < a.N) z.x = X(m); break;
      case 2: if (m < a.N) { const cf w = trig_phase((double)m * inv2n); const float c = (m == 0 ? 0.5f : 1.0f) * X(m); z.x = c * w.x; z.y = c * w.y; } break;
      case 3: case 7: if (m < a.N) { const cf w = trig_phase(-(double)m * inv2n); const float v = X(m); z.x = v * w.x; z.y = v * w.y; } break;
      case 4: if (m >= 1 && m <= a.N) z.x = X(m - 1); else if (m > a.N + 1) z.x = -X(a.L - m - 1); break;
      case 6: if (m >= 1 && m <= a.N) { const cf w = trig_phase((double)m * inv2n); const float c = (m == a.N ? 0.5f : 1.0f) * X(m - 1); z.x = c * w.x; z.y = c * w.y; } break;
    }
    a.z[g] = z;
  }
}

static __global__ void __launch_bounds__(256) trig_post_kernel(const TrigArgs a) {
  const long long total = a.lines * a.N;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
    const long long G = g / a.N, k = g - G * a.N;
    const long long o = G / a.S, base = o * a.S * a.N + (G - o * a.S);
    const cf* Z = a.z + G * a.L;
    const double inv2n = 1.0 / (2.0 * (double)a.N);
    float r = 0.0f;
    switch (a.kind) {
      case 0: r = Z[k].x; break;
      case 1: { const cf w = trig_phase(-(double)k * inv2n); const cf v = Z[k]; r = v.x * w.x - v.y * w.y; } break;
      case 2: r = Z[k].x; break;
      case 3: { const cf w = trig_phase(-((double)k + 0.5) * inv2n); const cf v = Z[k]; r = v.x * w.x - v.y * w.y; } break;
      case 4: r = -0.5f * Z[k + 1].y; break;
      case 5: { const cf w = trig_phase(-(double)(k + 1) * inv2n); const cf v = Z[k + 1]; r = -(v.x * w.y + v.y * w.x); } break;
      case 6: r = Z[k].y; break;
      case 7: { const cf w = trig_phase(-((double)k + 0.5) * inv2n); const cf v = Z[k]; r = -(v.x * w.y + v.y * w.x); } break;
    }
    a.y[base + k * a.S] = r * a.scale;
  }
}

// ---- dct2 / dst2 / dct3 / dst3 over dense lines (S = 1) of even length: a REAL FFT of length N instead of a complex one of 2N ----
// Makhoul's permutation v[n] = x[2n], v[N-1-n] = x[2n+1] (n < N/2) turns the DCT-II sum into V = FFT_N(v) and one phase:
//   t_k = e^{-i theta k} V[k]:   X[k] = Re t_k,   X[N-k] = -Im t_k      (k = 0..N/2, V from the r2c route: N/2+1 bins)
// and back (dct3, unnormalised as kind 2):   V[k] = (X[k] - i X[N-k]) e^{+i theta k} / 2, X[N] := 0;  v = c2r_N(V), un-permuted.
// dst2(x)[k] = dct2((-1)^n x)[N-1-k] and dst3(X)[n] = (-1)^n dct3(reversed X)[n]: the sine kinds ride the same two passes
// with a sign on the odd samples and reversed bins.  Four times less FFT work and a quarter of the intermediate bytes.
//   kind 8 dct2 fwd, 9 dst2 fwd (pre: x -> y = v, real;  post: z = V packed [line][L = N/2+1] -> y)
//   kind 10 dct2 inv, 11 dst2 inv (pre: x -> z = V packed;  post: x = v, real -> y)
// dct4 / dst4 (kinds 12, 13; N even) through a COMPLEX FFT of length N/2:  t[m] = (x[2m] + i x[N-1-2m]) e^{-i pi (4m+1)/(4N)},
//   T = FFT_{N/2}(t), y = T[k] e^{-i pi k/N}:  X[2k] = Re y, X[N-1-2k] = -Im y;   dst4(x)[k] = (-1)^k dct4(reversed x)[k].
// dct1 / dst1 (kinds 14, 15): the even / odd extension of length M = 2(N-1) / 2(N+1) is REAL, so its spectrum comes from the
//   r2c route (M/2+1 bins, exactly the N bins needed) instead of a complex FFT of length M.  a.S carries M for these two.
static __global__ void __launch_bounds__(256) trig_real_pre_kernel(const TrigArgs a) {
  const int kind = a.kind;
  const bool sine = kind & 1;
  // elements written per line: v real [N] (8, 9) / [M] (14, 15); V packed [L] (10, 11); t complex [L = N/2] (12, 13)
  // work items per line: sample pairs (8, 9: N/2; 12, 13: pairs of t, ceil(N/4)) so that every load is a float2 of adjacent
  // samples and no line of x is fetched twice; one element of the extension / of V otherwise
  const long long H = a.N / 2;
  const long long per = kind < 10 ? H : (kind >= 14 ? a.S : (kind >= 12 ? (H + 1) / 2 : a.L)), total = a.lines * per;
  const double inv2n = 1.0 / (2.0 * (double)a.N);
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
    const long long G = g / per, m = g - G * per;
    const float* x = a.x + G * a.N;
    if (kind < 10) {
      const cf p = *reinterpret_cast<const cf*>(x + 2 * m);      // N even, buffer offsets multiples of 8 bytes
      float* v = a.y + G * a.N;
      v[m] = p.x;
      v[a.N - 1 - m] = sine ? -p.y : p.y;
    } else if (kind < 12) {
      const float re = sine ? x[a.N - 1 - m] : x[m];
      const float im = m == 0 ? 0.0f : (sine ? x[m - 1] : x[a.N - m]);
      const cf w = trig_phase((double)m * inv2n);
      cf v; v.x = 0.5f * (re * w.x + im * w.y); v.y = 0.5f * (re * w.y - im * w.x);
      a.z[g] = v;
    } else if (kind < 14) {
      // t[m] = (x[2m] + i x[N-1-2m]) e^{-i pi (4m+1)/(4N)}; dst4 reads the line reversed.  One item forms t[m] and t[H-1-m]
      // from the sample pairs (x[2m], x[2m+1]) and (x[N-2-2m], x[N-1-2m]).
      const long long m2 = H - 1 - m;
      const cf p = *reinterpret_cast<const cf*>(x + 2 * m), q = *reinterpret_cast<const cf*>(x + 2 * m2);
      cf* t = a.z + G * a.L;
      {
        const float re = sine ? q.y : p.x, im = sine ? p.x : q.y;
        const cf w = trig_phase(-(double)(4 * m + 1) * inv2n * 0.5);
        cf v; v.x = re * w.x - im * w.y; v.y = re * w.y + im * w.x;
        t[m] = v;
      }
      if (m2 != m) {
        const float re = sine ? p.y : q.x, im = sine ? q.x : p.y;
        const cf w = trig_phase(-(double)(4 * m2 + 1) * inv2n * 0.5);
        cf v; v.x = re * w.x - im * w.y; v.y = re * w.y + im * w.x;
        t[m2] = v;
      }
    } else if (kind == 14) {
      a.y[g] = x[m < a.N ? m : a.S - m];                      // even extension, M = 2(N-1)
    } else {
      a.y[g] = (m == 0 || m == a.N + 1) ? 0.0f : (m <= a.N ? x[m - 1] : -x[a.S - m - 1]);   // odd extension, M = 2(N+1)
    }
  }
}

static __global__ void __launch_bounds__(256) trig_real_post_kernel(const TrigArgs a) {
  const int kind = a.kind;
  const bool sine = kind & 1;
  const long long H = a.N / 2;
  const long long per = kind < 10 ? a.L : (kind < 12 ? H : (kind < 14 ? (H + 1) / 2 : a.N)), total = a.lines * per;   // 10..13: float2 stores
  const double inv2n = 1.0 / (2.0 * (double)a.N);
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
    const long long G = g / per, m = g - G * per;
    float* y = a.y + G * a.N;
    if (kind < 10) {
      const cf w = trig_phase(-(double)m * inv2n), v = a.z[g];
      const float re = (v.x * w.x - v.y * w.y) * a.scale, im = -(v.x * w.y + v.y * w.x) * a.scale;
      y[sine ? a.N - 1 - m : m] = re;
      if (m > 0 && 2 * m != a.N) y[sine ? m - 1 : a.N - m] = im;
    } else if (kind < 12) {
      const float* v = a.x + G * a.N;
      cf p; p.x = v[m] * a.scale; p.y = (sine ? -v[a.N - 1 - m] : v[a.N - 1 - m]) * a.scale;
      *reinterpret_cast<cf*>(y + 2 * m) = p;
    } else if (kind < 14) {
      // y = T[m] e^{-i pi m/N}:  X[2m] = Re y,  X[N-1-2m] = -Im y;  dst4: X[k] *= (-1)^k  (N even: N-1-2m is odd)
      // one item: bins m and H-1-m -> the sample pairs (X[2m], X[2m+1] = X[N-1-2(H-1-m)]) and (X[N-2-2m], X[N-1-2m])
      const long long m2 = H - 1 - m;
      const cf* T = a.z + G * a.L;
      const cf w = trig_phase(-(double)m * inv2n * 2.0), v = T[m];
      const cf w2 = trig_phase(-(double)m2 * inv2n * 2.0), v2 = T[m2];
      const float re = (v.x * w.x - v.y * w.y) * a.scale, im = -(v.x * w.y + v.y * w.x) * a.scale;
      const float re2 = (v2.x * w2.x - v2.y * w2.y) * a.scale, im2 = -(v2.x * w2.y + v2.y * w2.x) * a.scale;
      cf p, q;
      p.x = re;  p.y = sine ? -im2 : im2;
      q.x = re2; q.y = sine ? -im : im;
      *reinterpret_cast<cf*>(y + 2 * m) = p;
      if (m2 != m) *reinterpret_cast<cf*>(y + 2 * m2) = q;
    } else if (kind == 14) {
      y[m] = a.z[G * a.L + m].x * a.scale;
    } else {
      y[m] = -0.5f * a.z[G * a.L + m + 1].y * a.scale;
    }
  }
}

// ---- kinds 8..15 along a strided axis (S > 1: axes >= 1 of an N-D real array) ----
// Array side: sample n of line G = o*S + inner sits at o*S*N + inner + n*S, so adjacent LINES are adjacent floats; dense side:
// element e of line G at G*pitch + e.  32 x 32 tiles through LDS turn one walk into the other (array side along `inner`, dense
// side along e) so both sides move whole cache lines.  Same arithmetic as the dense kernels, one dense element per item.
MI_DEV cf trig_tile_pre_value(int kind, const float* xb, long long S, long long N, long long M, long long e) {
  const bool sine = kind & 1;
  const double inv2n = 1.0 / (2.0 * (double)N);
  const auto X = [&](long long n) { return xb[n * S]; };
  cf v; v.x = 0.0f; v.y = 0.0f;
  if (kind < 10) {
    const long long n = e < N / 2 ? 2 * e : 2 * (N - 1 - e) + 1;
    v.x = (sine && (n & 1)) ? -X(n) : X(n);
  } else if (kind < 12) {
    const float re = sine ? X(N - 1 - e) : X(e);
    const float im = e == 0 ? 0.0f : (sine ? X(e - 1) : X(N - e));
    const cf w = trig_phase((double)e * inv2n);
    v.x = 0.5f * (re * w.x + im * w.y); v.y = 0.5f * (re * w.y - im * w.x);
  } else if (kind < 14) {
    const float re = sine ? X(N - 1 - 2 * e) : X(2 * e), im = sine ? X(2 * e) : X(N - 1 - 2 * e);
    const cf w = trig_phase(-(double)(4 * e + 1) * inv2n * 0.5);
    v.x = re * w.x - im * w.y; v.y = re * w.y + im * w.x;
  } else if (kind == 14) {
    v.x = X(e < N ? e : M - e);
  } else {
    v.x = (e == 0 || e == N + 1) ? 0.0f : (e <= N ? X(e - 1) : -X(M - e - 1));
  }
  return v;
}

static __global__ void __launch_bounds__(256) trig_real_pre_tiled_kernel(const TrigArgs a) {
  MI_SMEM_DECL_STATIC(cf, tile, 32 * 33);
  const int kind = a.kind;
  const bool dense_real = kind < 10 || kind >= 14;
  const long long S = a.stride, M = a.S, outer = a.lines / S;
  const long long per = kind < 10 ? a.N : (kind >= 14 ? M : a.L);          // dense elements per line
  const long long pitch = dense_real ? per : a.L;
  const long long te_n = (per + 31) / 32, ti_n = (S + 31) / 32, tiles = outer * ti_n * te_n;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (long long t = blockIdx.x; t < tiles; t += gridDim.x) {
    const long long te = t % te_n, r0 = t / te_n, ti = r0 % ti_n, o = r0 / ti_n;
    {
      const long long inner = ti * 32 + tx;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long long e = te * 32 + ty + 8 * r;
        if (inner < S && e < per) tile[(ty + 8 * r) * 33 + tx] = trig_tile_pre_value(kind, a.x + o * S * a.N + inner, S, a.N, M, e);
      }
    }
    __syncthreads();
    {
      const long long e = te * 32 + tx;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long long inner = ti * 32 + ty + 8 * r;
        if (inner < S && e < per) {
          const cf v = tile[tx * 33 + ty + 8 * r];
          const long long G = o * S + inner;
          if (dense_real) a.y[G * pitch + e] = v.x; else a.z[G * pitch + e] = v;
        }
      }
    }
    __syncthreads();
  }
}

static __global__ void __launch_bounds__(256) trig_real_post_tiled_kernel(const TrigArgs a) {
  MI_SMEM_DECL_STATIC(cf, tile, 32 * 33);
  const int kind = a.kind;
  const bool sine = kind & 1;
  const long long S = a.stride, N = a.N, outer = a.lines / S;
  const long long per = kind < 10 ? a.L : (kind == 12 || kind == 13 ? a.L : N);   // dense elements walked per line
  const double inv2n = 1.0 / (2.0 * (double)N);
  const long long te_n = (per + 31) / 32, ti_n = (S + 31) / 32, tiles = outer * ti_n * te_n;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (long long t = blockIdx.x; t < tiles; t += gridDim.x) {
    const long long te = t % te_n, r0 = t / te_n, ti = r0 % ti_n, o = r0 / ti_n;
    {
      const long long e = te * 32 + tx;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long long inner = ti * 32 + ty + 8 * r;
        if (inner >= S || e >= per) continue;
        const long long G = o * S + inner;
        cf out; out.x = 0.0f; out.y = 0.0f;
        if (kind < 10) {
          const cf w = trig_phase(-(double)e * inv2n), v = a.z[G * a.L + e];
          out.x = (v.x * w.x - v.y * w.y) * a.scale; out.y = -(v.x * w.y + v.y * w.x) * a.scale;
        } else if (kind < 12) {
          const float v = a.x[G * N + ((e & 1) ? N - 1 - (e >> 1) : (e >> 1))];
          out.x = ((sine && (e & 1)) ? -v : v) * a.scale;
        } else if (kind < 14) {
          const cf w = trig_phase(-(double)e * inv2n * 2.0), v = a.z[G * a.L + e];
          out.x = (v.x * w.x - v.y * w.y) * a.scale;
          const float im = -(v.x * w.y + v.y * w.x) * a.scale;
          out.y = sine ? -im : im;
        } else if (kind == 14) {
          out.x = a.z[G * a.L + e].x * a.scale;
        } else {
          out.x = -0.5f * a.z[G * a.L + e + 1].y * a.scale;
        }
        tile[tx * 33 + ty + 8 * r] = out;
      }
    }
    __syncthreads();
    {
      const long long inner = ti * 32 + tx;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long long e = te * 32 + ty + 8 * r;
        if (inner >= S || e >= per) continue;
        const cf v = tile[(ty + 8 * r) * 33 + tx];
        float* y = a.y + o * S * N + inner;
        if (kind < 10) {
          y[(sine ? N - 1 - e : e) * S] = v.x;
          if (e > 0 && 2 * e != N) y[(sine ? e - 1 : N - e) * S] = v.y;
        } else if (kind == 12 || kind == 13) {
          y[2 * e * S] = v.x;
          y[(N - 1 - 2 * e) * S] = v.y;
        } else {
          y[e * S] = v.x;
        }
      }
    }
    __syncthreads();
  }
}

}  // namespace mi355

/*
 * oracle.c — CPU restatement of the reference's JavaScript correctness oracle.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product path (libmi355fft.so, the N-API
 * addon, the JS/Python hosts) never links, loads or calls anything in this directory.
 *
 * What it restates (all citations are into /root/reference):
 *   src/utils/math.js:14-19    normalizeScaleFactor
 *   src/utils/math.js:25-88    fft1dRefInterleaved      radix-2 DIT, f32 storage, f64 twiddle recurrence
 *   src/utils/math.js:90-148   fftNdRefInterleaved      per-axis pow-2 reference + scale
 *   src/utils/math.js:150-158  randomComplexInterleaved (rng()*2-1)*0.5 in f64, stored f32
 *   src/utils/math.js:160-184  dft1dRefInterleaved      naive O(N^2), f64 accumulate, f32 output
 *   src/utils/math.js:186-236  fftNdRefAnySizeInterleaved
 *   src/utils/math.js:238-258  r2cRefPackedInterleaved
 *   src/utils/math.js:260-289  c2rRefFromPackedInterleaved
 *   src/utils/math.js:469-603  fftConvRef
 *
 * Parity pin: tests/test_oracle_golden.py checks every function here against fixtures that
 * oracle/gen_fixtures.mjs produced by importing the reference's own math.js under Node in the
 * build container (tests/golden/manifest.json).  The radix-2 path is pinned BIT-EXACT (the
 * arithmetic is IEEE f64 ops + f32 stores in the reference's operation order; the only libm
 * calls are cos/sin of +-2*pi/len for the 22 power-of-two lengths, which are checked too).
 * The O(N^2) DFT path calls cos/sin on large arguments where V8's fdlibm port and glibc may
 * differ in the last f64 bit, so it is pinned to 2e-6 relative instead.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: no FMA contraction, or results would
 * differ from JavaScript's separately rounded multiply and add).
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

/* ---- seeded PRNG shared by fixtures, tests, bench and the device-side generator ------------
 * mulberry32 with the state kept in 32-bit wraparound arithmetic, so that draw number n of a
 * stream is a pure function of (seed, n): state_n = seed + (n+1)*0x6D2B79F5 (mod 2^32).
 * The JS twin is in oracle/gen_fixtures.mjs, the numpy twin in oracle/oracle.py and the device
 * twin in webgpu-fft_amd/csrc (mi355fft_fill_random). */
static inline uint32_t mulberry32_at(uint32_t seed, uint64_t n) {
  uint32_t t = seed + (uint32_t)((n + 1u) * 0x6D2B79F5u);
  t = (t ^ (t >> 15)) * (t | 1u);
  t ^= t + (t ^ (t >> 7)) * (t | 61u);
  return t ^ (t >> 14);
}

ORACLE_API double oracle_rng_at(uint32_t seed, uint64_t n) {
  return (double)mulberry32_at(seed, n) / 4294967296.0;
}

/* seed of the independent stream used for transform `b` of a batch (keeps every stream far
 * shorter than the generator's 2^32 period even for 2^33-float workloads) */
ORACLE_API uint32_t oracle_stream_seed(uint32_t seed0, uint64_t b) {
  uint32_t h = seed0 + (uint32_t)((b + 1u) * 0x9E3779B9u);
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}

/* math.js:150-158 — out has 2*lengthComplex floats; draw order re, im, re, im ... */
ORACLE_API void oracle_random_complex_interleaved(float* out, uint64_t lengthComplex, uint32_t seed) {
  for (uint64_t i = 0; i < 2 * lengthComplex; i++) {
    double r = oracle_rng_at(seed, i);
    out[i] = (float)((r * 2.0 - 1.0) * 0.5);
  }
}

/* real-valued analogue used for r2c inputs: same distribution, one draw per sample */
ORACLE_API void oracle_random_real(float* out, uint64_t length, uint32_t seed) {
  for (uint64_t i = 0; i < length; i++) {
    double r = oracle_rng_at(seed, i);
    out[i] = (float)((r * 2.0 - 1.0) * 0.5);
  }
}

/* math.js:14-19.  normalize: 0 none, 1 backward, 2 unitary */
ORACLE_API double oracle_normalize_scale_factor(int normalize, int inverse, double nTotal) {
  if (normalize == 0) return 1.0;
  if (normalize == 2) return 1.0 / sqrt(nTotal);
  return inverse ? 1.0 / nTotal : 1.0;
}

static unsigned reverse_bits(unsigned x, int bits) {
  unsigned y = 0;
  for (int i = 0; i < bits; i++) { y = (y << 1) | (x & 1u); x >>= 1; }
  return y;
}

/* math.js:25-88.  in/out: 2*N floats (may not alias).  N power of two >= 2.  Returns 0 on ok. */
ORACLE_API int oracle_fft1d_ref(const float* in, float* out, int64_t N, int inverse) {
  if (N < 2 || (N & (N - 1)) != 0) return -1;
  memcpy(out, in, (size_t)(2 * N) * sizeof(float));
  int bits = 0;
  while (((int64_t)1 << bits) < N) bits++;

  for (int64_t i = 0; i < N; i++) {           /* :35-47 bit-reversal swap */
    int64_t j = (int64_t)reverse_bits((unsigned)i, bits);
    if (j > i) {
      float tr = out[2 * i], ti = out[2 * i + 1];
      out[2 * i] = out[2 * j]; out[2 * i + 1] = out[2 * j + 1];
      out[2 * j] = tr; out[2 * j + 1] = ti;
    }
  }
  const double sign = inverse ? 1.0 : -1.0;   /* :49 */
  for (int64_t len = 2; len <= N; len <<= 1) { /* :51-85 */
    const int64_t half = len >> 1;
    const double ang = (sign * 2.0 * M_PI) / (double)len;
    const double wlenRe = cos(ang), wlenIm = sin(ang);
    for (int64_t i = 0; i < N; i += len) {
      double wRe = 1.0, wIm = 0.0;
      for (int64_t j = 0; j < half; j++) {
        const int64_t a = 2 * (i + j), b = 2 * (i + j + half);
        const double uRe = out[a], uIm = out[a + 1];
        const double vRe0 = out[b], vIm0 = out[b + 1];
        const double vRe = vRe0 * wRe - vIm0 * wIm;
        const double vIm = vRe0 * wIm + vIm0 * wRe;
        out[a] = (float)(uRe + vRe);       /* f32 store after every butterfly */
        out[a + 1] = (float)(uIm + vIm);
        out[b] = (float)(uRe - vRe);
        out[b + 1] = (float)(uIm - vIm);
        const double nRe = wRe * wlenRe - wIm * wlenIm;
        const double nIm = wRe * wlenIm + wIm * wlenRe;
        wRe = nRe; wIm = nIm;
      }
    }
  }
  return 0;
}

/* cos/sin of the 22+ stage angles, exposed so the golden test can pin libm against V8 */
ORACLE_API void oracle_stage_twiddle(int64_t len, int inverse, double* re, double* im) {
  const double sign = inverse ? 1.0 : -1.0;
  const double ang = (sign * 2.0 * M_PI) / (double)len;
  *re = cos(ang); *im = sin(ang);
}

/* math.js:160-184 */
ORACLE_API int oracle_dft1d_ref(const float* in, float* out, int64_t N, int inverse) {
  if (N <= 0) return -1;
  const double sign = inverse ? 1.0 : -1.0;
  for (int64_t k = 0; k < N; k++) {
    double re = 0, im = 0;
    for (int64_t n = 0; n < N; n++) {
      const double ang = (sign * 2.0 * M_PI * (double)n * (double)k) / (double)N;
      const double c = cos(ang), s = sin(ang);
      const double xr = in[2 * n], xi = in[2 * n + 1];
      re += xr * c - xi * s;
      im += xr * s + xi * c;
    }
    out[2 * k] = (float)re; out[2 * k + 1] = (float)im;
  }
  return 0;
}

typedef int (*line_fn)(const float*, float*, int64_t, int);

/* shared body of math.js:90-148 and :186-236 — axis 0 fastest, one line at a time, then scale */
static int fftnd_generic(const float* in, float* out, const int64_t* shape, int rank, int inverse,
                         int normalize, line_fn fn) {
  if (rank < 1 || rank > 8) return -1;
  int64_t nTotal = 1, strides[8];
  for (int d = 0; d < rank; d++) { if (shape[d] <= 0) return -1; strides[d] = nTotal; nTotal *= shape[d]; }
  if (out != in) memcpy(out, in, (size_t)(2 * nTotal) * sizeof(float));
  for (int axis = 0; axis < rank; axis++) {
    const int64_t N = shape[axis], stride = strides[axis], lineCount = nTotal / N;
    float* line = (float*)malloc((size_t)(2 * N) * sizeof(float));
    float* res = (float*)malloc((size_t)(2 * N) * sizeof(float));
    if (!line || !res) { free(line); free(res); return -2; }
    for (int64_t l = 0; l < lineCount; l++) {
      int64_t rem = l, base = 0;
      for (int d = 0; d < rank; d++) {
        if (d == axis) continue;
        const int64_t c = rem % shape[d]; rem /= shape[d]; base += c * strides[d];
      }
      for (int64_t p = 0; p < N; p++) {
        const int64_t idx = 2 * (base + p * stride);
        line[2 * p] = out[idx]; line[2 * p + 1] = out[idx + 1];
      }
      int rc = fn(line, res, N, inverse);
      if (rc) { free(line); free(res); return rc; }
      for (int64_t p = 0; p < N; p++) {
        const int64_t idx = 2 * (base + p * stride);
        out[idx] = res[2 * p]; out[idx + 1] = res[2 * p + 1];
      }
    }
    free(line); free(res);
  }
  const double scale = oracle_normalize_scale_factor(normalize, inverse, (double)nTotal);
  if (scale != 1.0)
    for (int64_t i = 0; i < 2 * nTotal; i++) out[i] = (float)((double)out[i] * scale);
  return 0;
}

/* math.js:90-148 (power-of-two dims only) */
ORACLE_API int oracle_fftnd_ref(const float* in, float* out, const int64_t* shape, int rank, int inverse,
                                int normalize) {
  for (int d = 0; d < rank; d++)
    if (shape[d] < 1 || (shape[d] & (shape[d] - 1)) != 0) return -1;
  /* the reference's 1-D routine rejects N<2; a length-1 axis never reaches a caller in scope */
  return fftnd_generic(in, out, shape, rank, inverse, normalize, oracle_fft1d_ref);
}

/* math.js:186-236 */
ORACLE_API int oracle_fftnd_anysize_ref(const float* in, float* out, const int64_t* shape, int rank,
                                        int inverse, int normalize) {
  return fftnd_generic(in, out, shape, rank, inverse, normalize, oracle_dft1d_ref);
}

/* math.js:238-258.  inReal: N floats; out: 2*(N/2+1) floats.  use_pow2=1 swaps the O(N^2) DFT for
 * the radix-2 reference (the only feasible oracle at N=2^20..2^22). */
ORACLE_API int oracle_r2c_ref_packed(const float* inReal, float* out, int64_t N, int normalize, int use_pow2) {
  if (N < 2) return -1;
  float* cplx = (float*)calloc((size_t)(2 * N), sizeof(float));
  float* full = (float*)malloc((size_t)(2 * N) * sizeof(float));
  if (!cplx || !full) { free(cplx); free(full); return -2; }
  for (int64_t i = 0; i < N; i++) cplx[2 * i] = inReal[i];
  int64_t shape[1] = {N};
  int rc = use_pow2 ? oracle_fftnd_ref(cplx, full, shape, 1, 0, 0) : oracle_fftnd_anysize_ref(cplx, full, shape, 1, 0, 0);
  if (!rc) {
    const int64_t outLen = N / 2 + 1;
    memcpy(out, full, (size_t)(2 * outLen) * sizeof(float));
    const double scale = oracle_normalize_scale_factor(normalize, 0, (double)N);
    if (scale != 1.0) for (int64_t i = 0; i < 2 * outLen; i++) out[i] = (float)((double)out[i] * scale);
  }
  free(cplx); free(full);
  return rc;
}

/* math.js:260-289.  inPacked: 2*(N/2+1) floats; out: N floats */
ORACLE_API int oracle_c2r_ref_from_packed(const float* inPacked, float* out, int64_t N, int normalize, int use_pow2) {
  if (N < 2) return -1;
  const int64_t outLen = N / 2 + 1;
  float* full = (float*)calloc((size_t)(2 * N), sizeof(float));
  float* time = (float*)malloc((size_t)(2 * N) * sizeof(float));
  if (!full || !time) { free(full); free(time); return -2; }
  memcpy(full, inPacked, (size_t)(2 * outLen) * sizeof(float));
  const int64_t kMaxMirror = (N % 2 == 0) ? (N / 2) - 1 : N / 2;
  for (int64_t k = 1; k <= kMaxMirror; k++) {       /* :271-275 Hermitian mirror */
    full[2 * (N - k)] = full[2 * k];
    full[2 * (N - k) + 1] = -full[2 * k + 1];
  }
  int64_t shape[1] = {N};
  int rc = use_pow2 ? oracle_fftnd_ref(full, time, shape, 1, 1, 0) : oracle_fftnd_anysize_ref(full, time, shape, 1, 1, 0);
  if (!rc) {
    for (int64_t n = 0; n < N; n++) out[n] = time[2 * n];
    const double scale = oracle_normalize_scale_factor(normalize, 1, (double)N);
    if (scale != 1.0) for (int64_t i = 0; i < N; i++) out[i] = (float)((double)out[i] * scale);
  }
  free(full); free(time);
  return rc;
}

static int64_t prod_shape(const int64_t* s, int rank) { int64_t p = 1; for (int d = 0; d < rank; d++) p *= s[d]; return p; }

/* math.js:504-537 embedAtOffset / extractAtOffset, axis 0 fastest */
static void embed_at_offset(float* dst, const int64_t* dstShape, const float* src, const int64_t* srcShape,
                            const int64_t* offset, int rank) {
  int64_t dstStr[8], acc = 1;
  for (int d = 0; d < rank; d++) { dstStr[d] = acc; acc *= dstShape[d]; }
  const int64_t srcN = prod_shape(srcShape, rank);
  for (int64_t i = 0; i < srcN; i++) {
    int64_t rem = i, di = 0;
    for (int d = 0; d < rank; d++) { const int64_t c = rem % srcShape[d]; rem /= srcShape[d]; di += (offset[d] + c) * dstStr[d]; }
    dst[2 * di] = src[2 * i]; dst[2 * di + 1] = src[2 * i + 1];
  }
}
static void extract_at_offset(float* dst, const int64_t* dstShape, const float* src, const int64_t* srcShape,
                              const int64_t* offset, int rank) {
  int64_t srcStr[8], acc = 1;
  for (int d = 0; d < rank; d++) { srcStr[d] = acc; acc *= srcShape[d]; }
  const int64_t dstN = prod_shape(dstShape, rank);
  for (int64_t i = 0; i < dstN; i++) {
    int64_t rem = i, si = 0;
    for (int d = 0; d < rank; d++) { const int64_t c = rem % dstShape[d]; rem /= dstShape[d]; si += (offset[d] + c) * srcStr[d]; }
    dst[2 * i] = src[2 * si]; dst[2 * i + 1] = src[2 * si + 1];
  }
}

/* math.js:469-603.  mode: 0 convolution, 1 correlation.  boundary: 0 circular, 1 linear-full,
 * 2 linear-same, 3 linear-valid.  kernelShape NULL => shape.  use_pow2: radix-2 reference instead of
 * the O(N^2) DFT (only valid when every fftShape dim is a power of two).
 * out must hold 2*batch*prod(outShape) floats; outShapeOut (rank entries) receives outShape. */
ORACLE_API int oracle_fftconv_ref(const float* input, const float* kernel, float* out, const int64_t* shape, int rank,
                                  int64_t batch, int mode, int boundary, const int64_t* kernelShape, int use_pow2,
                                  int64_t* outShapeOut) {
  if (rank < 1 || rank > 8 || batch <= 0) return -1;
  int64_t kShape[8], fftShape[8], outShape[8], outOffset[8], zeros[8] = {0};
  for (int d = 0; d < rank; d++) {
    kShape[d] = kernelShape ? kernelShape[d] : shape[d];
    if (shape[d] <= 0 || kShape[d] <= 0) return -1;
    if (boundary == 0 && kShape[d] > shape[d]) return -3;
    fftShape[d] = boundary == 0 ? shape[d] : shape[d] + kShape[d] - 1;
    if (boundary == 0) { outShape[d] = shape[d]; outOffset[d] = 0; }
    else if (boundary == 1) { outShape[d] = fftShape[d]; outOffset[d] = 0; }
    else if (boundary == 2) { outShape[d] = shape[d]; outOffset[d] = (kShape[d] - 1) / 2; }
    else { outShape[d] = shape[d] - kShape[d] + 1; if (outShape[d] <= 0) return -4; outOffset[d] = kShape[d] - 1; }
    if (outShapeOut) outShapeOut[d] = outShape[d];
  }
  const int64_t inputN = prod_shape(shape, rank), fftN = prod_shape(fftShape, rank), outN = prod_shape(outShape, rank);
  float* kPad = (float*)calloc((size_t)(2 * fftN), sizeof(float));
  float* kf = (float*)malloc((size_t)(2 * fftN) * sizeof(float));
  float* xPad = (float*)malloc((size_t)(2 * fftN) * sizeof(float));
  float* xf = (float*)malloc((size_t)(2 * fftN) * sizeof(float));
  float* yf = (float*)malloc((size_t)(2 * fftN) * sizeof(float));
  float* yFull = (float*)malloc((size_t)(2 * fftN) * sizeof(float));
  int rc = (!kPad || !kf || !xPad || !xf || !yf || !yFull) ? -2 : 0;
  int (*nd)(const float*, float*, const int64_t*, int, int, int) = use_pow2 ? oracle_fftnd_ref : oracle_fftnd_anysize_ref;
  if (!rc) {
    embed_at_offset(kPad, fftShape, kernel, kShape, zeros, rank);
    rc = nd(kPad, kf, fftShape, rank, 0, 0);
  }
  for (int64_t b = 0; b < batch && !rc; b++) {
    memset(xPad, 0, (size_t)(2 * fftN) * sizeof(float));
    embed_at_offset(xPad, fftShape, input + 2 * b * inputN, shape, zeros, rank);
    rc = nd(xPad, xf, fftShape, rank, 0, 0);
    if (rc) break;
    for (int64_t i = 0; i < fftN; i++) {             /* :589-596 — JS evaluates in f64, stores f32 */
      const double ar = xf[2 * i], ai = xf[2 * i + 1];
      const double br = kf[2 * i];
      const double bi = mode == 1 ? -(double)kf[2 * i + 1] : (double)kf[2 * i + 1];
      yf[2 * i] = (float)(ar * br - ai * bi);
      yf[2 * i + 1] = (float)(ar * bi + ai * br);
    }
    rc = nd(yf, yFull, fftShape, rank, 1, 1 /* backward */);
    if (rc) break;
    extract_at_offset(out + 2 * b * outN, outShape, yFull, fftShape, outOffset, rank);
  }
  free(kPad); free(kf); free(xPad); free(xf); free(yf); free(yFull);
  return rc;
}

/* ---- batched driver for bench.py's cpu_baseline leg (kind "port") ---------------------------
 * `batch` independent length-N transforms, axis 0 fastest / batch outermost, split over
 * `nthreads` POSIX threads.  Same arithmetic as oracle_fft1d_ref, one call per transform. */
typedef struct { const float* in; float* out; int64_t N; int64_t b0, b1; int inverse; int rc; } batch_job;
static void* batch_worker(void* p) {
  batch_job* j = (batch_job*)p;
  for (int64_t b = j->b0; b < j->b1; b++) {
    int rc = oracle_fft1d_ref(j->in + 2 * b * j->N, j->out + 2 * b * j->N, j->N, j->inverse);
    if (rc) { j->rc = rc; break; }
  }
  return NULL;
}
ORACLE_API int oracle_fft1d_ref_batch(const float* in, float* out, int64_t N, int64_t batch, int inverse, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  if (nthreads > batch) nthreads = (int)batch;
  pthread_t th[256]; batch_job jobs[256];
  for (int t = 0; t < nthreads; t++) {
    jobs[t] = (batch_job){in, out, N, batch * t / nthreads, batch * (t + 1) / nthreads, inverse, 0};
    if (pthread_create(&th[t], NULL, batch_worker, &jobs[t]) != 0) {
      jobs[t].rc = -5; batch_worker(&jobs[t]); th[t] = 0;
    }
  }
  int rc = 0;
  for (int t = 0; t < nthreads; t++) { if (th[t]) pthread_join(th[t], NULL); if (jobs[t].rc) rc = jobs[t].rc; }
  return rc;
}

/* FNV-1a 64 over raw bytes — how the big fixtures are pinned without committing megabytes */
ORACLE_API uint64_t oracle_fnv1a64(const void* data, uint64_t nbytes) {
  const unsigned char* p = (const unsigned char*)data;
  uint64_t h = 0xcbf29ce484222325ull;
  for (uint64_t i = 0; i < nbytes; i++) { h ^= p[i]; h *= 0x100000001b3ull; }
  return h;
}

/* ---- DCT / DST references (math.js:291-409): O(N^2) sums, f64 accumulation, f32 output ----
 * kind: 0 dct1, 1 dct2 forward (= dct3 inverse), 2 dct2 inverse (= dct3 forward), 3 dct4, 4 dst1, 5 dst2 forward (= dst3
 * inverse), 6 dst2 inverse (= dst3 forward), 7 dst4 — the typeKind table of runtime/plans/dct_fft.js:48-57 */
ORACLE_API int oracle_trig1d_ref(const float* in, float* out, int64_t N, int kind) {
  if (N < 1 || kind < 0 || kind > 7) return -1;
  if ((kind == 0 || kind == 4) && N < 2) return -1;
  const double PI = 3.141592653589793;          /* Math.PI */
  for (int64_t k = 0; k < N; k++) {
    double sum = 0.0;
    switch (kind) {
      case 0:   /* dct1Ref :395-409 */
        sum = (double)in[0] + ((k % 2 == 0) ? 1.0 : -1.0) * (double)in[N - 1];
        for (int64_t n = 1; n <= N - 2; n++) sum += 2.0 * (double)in[n] * cos((PI * (double)n * (double)k) / (double)(N - 1));
        break;
      case 1:   /* dct2Ref forward :295-303 */
        for (int64_t n = 0; n < N; n++) sum += (double)in[n] * cos((PI / (double)N) * ((double)n + 0.5) * (double)k);
        break;
      case 2:   /* dct2Ref inverse :306-313 (k is the output sample index n) */
        sum = (double)in[0] * 0.5;
        for (int64_t q = 1; q < N; q++) sum += (double)in[q] * cos((PI / (double)N) * ((double)k + 0.5) * (double)q);
        break;
      case 3:   /* dct4Ref :321-333 */
        for (int64_t n = 0; n < N; n++) sum += (double)in[n] * cos((PI / (double)N) * ((double)n + 0.5) * ((double)k + 0.5));
        break;
      case 4:   /* dst1Ref :335-349 */
        for (int64_t n = 0; n < N; n++) sum += (double)in[n] * sin((PI * (double)(n + 1) * (double)(k + 1)) / (double)(N + 1));
        break;
      case 5:   /* dst2Ref forward :355-363 */
        for (int64_t n = 0; n < N; n++) sum += (double)in[n] * sin((PI / (double)N) * ((double)n + 0.5) * (double)(k + 1));
        break;
      case 6:   /* dst2Ref inverse :366-373 (k is the output sample index n) */
        sum = ((k % 2 == 0) ? 0.5 : -0.5) * (double)in[N - 1];
        for (int64_t q = 0; q < N - 1; q++) sum += (double)in[q] * sin((PI / (double)N) * ((double)k + 0.5) * (double)(q + 1));
        break;
      default:  /* dst4Ref :381-393 */
        for (int64_t n = 0; n < N; n++) sum += (double)in[n] * sin((PI / (double)N) * ((double)n + 0.5) * ((double)k + 0.5));
        break;
    }
    out[k] = (float)sum;
  }
  return 0;
}

/* N-D: the 1-D reference along every axis in turn (axis 0 fastest, f32 between axes as the reference's plan stores its
 * intermediates, dct_fft.js:331-333), then ONE scale by normalizeScaleFactor(prod(shape)) (:882).  This composition is
 * this build's restatement — the reference has no N-D CPU oracle for these transforms; its suites compare N-D plans to
 * compositions of 1-D ones. */
ORACLE_API int oracle_trig_nd_ref(const float* in, float* out, const int64_t* shape, int rank, int64_t batch, int kind, int inverse, int normalize) {
  const int64_t total = prod_shape(shape, rank);
  if (total <= 0) return -1;
  memcpy(out, in, (size_t)(total * batch) * sizeof(float));
  int64_t S = 1;
  for (int a = 0; a < rank; a++) {
    const int64_t N = shape[a];
    float* line = (float*)malloc((size_t)N * sizeof(float));
    float* res = (float*)malloc((size_t)N * sizeof(float));
    if (!line || !res) { free(line); free(res); return -2; }
    const int64_t lines = batch * (total / N);
    for (int64_t L = 0; L < lines; L++) {
      const int64_t o = L / S, inner = L - o * S;
      float* base = out + o * S * N + inner;
      for (int64_t n = 0; n < N; n++) line[n] = base[n * S];
      const int rc = oracle_trig1d_ref(line, res, N, kind);
      if (rc) { free(line); free(res); return rc; }
      for (int64_t n = 0; n < N; n++) base[n * S] = res[n];
    }
    free(line); free(res);
    S *= N;
  }
  const double scale = oracle_normalize_scale_factor(normalize, inverse, (double)total);
  if (scale != 1.0) for (int64_t i = 0; i < total * batch; i++) out[i] = (float)((double)out[i] * scale);
  return 0;
}

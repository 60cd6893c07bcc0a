// oracle.mjs — JavaScript restatement of the reference's CPU oracle, for the Node-side parity tests.
//
// TEST INFRASTRUCTURE ONLY: imported by webgpu-fft_amd/js/test/*.mjs, never by the product (js/*.js).
// Same algorithms as oracle.c (each function cites the reference lines it follows); pinned against
// tests/golden by js/test/host_logic.test.mjs: the radix-2 path bit-exact, the O(N^2) DFT path to 2e-6.
// Node 12 subset.

// seeded PRNG twin of oracle.c:mulberry32_at (32-bit wraparound state)
export function mulberry32(seed) {
  let a = seed | 0;
  return function () {
    a = (a + 0x6d2b79f5) | 0;
    let t = a;
    t = Math.imul(t ^ (t >>> 15), t | 1);
    t ^= t + Math.imul(t ^ (t >>> 7), t | 61);
    return ((t ^ (t >>> 14)) >>> 0) / 4294967296;
  };
}
// math.js:150-158
export function randomComplexInterleaved(lengthComplex, rng) {
  const out = new Float32Array(2 * lengthComplex);
  for (let i = 0; i < 2 * lengthComplex; i++) out[i] = (rng() * 2 - 1) * 0.5;
  return out;
}
export function randomReal(n, rng) {
  const out = new Float32Array(n);
  for (let i = 0; i < n; i++) out[i] = (rng() * 2 - 1) * 0.5;
  return out;
}
// math.js:14-19
export function normalizeScaleFactor(normalize, direction, nTotal) {
  if (normalize === "none") return 1.0;
  if (normalize === "unitary") return 1.0 / Math.sqrt(nTotal);
  if (normalize === "backward") return direction === "inverse" ? 1.0 / nTotal : 1.0;
  throw new Error("Unknown normalize mode: " + normalize);
}
// math.js:25-88 — radix-2 DIT, Float32Array storage (f32 rounding after every butterfly), f64 twiddle recurrence
export function fft1dRef(input, N, direction) {
  if (N < 2 || (N & (N - 1)) !== 0) throw new Error("N must be a power of two >= 2");
  const out = new Float32Array(input);
  let bits = 0;
  while ((1 << bits) < N) bits++;
  for (let i = 0; i < N; i++) {
    let j = 0, x = i;
    for (let b = 0; b < bits; b++) { j = (j << 1) | (x & 1); x >>>= 1; }
    if (j > i) {
      const tr = out[2 * i], ti = out[2 * i + 1];
      out[2 * i] = out[2 * j]; out[2 * i + 1] = out[2 * j + 1];
      out[2 * j] = tr; out[2 * j + 1] = ti;
    }
  }
  const sign = direction === "forward" ? -1.0 : 1.0;
  for (let len = 2; len <= N; len <<= 1) {
    const half = len >>> 1;
    const ang = (sign * 2.0 * Math.PI) / len;
    const cRe = Math.cos(ang), cIm = Math.sin(ang);
    for (let i = 0; i < N; i += len) {
      let wRe = 1.0, wIm = 0.0;
      for (let j = 0; j < half; j++) {
        const a = 2 * (i + j), b = 2 * (i + j + half);
        const uRe = out[a], uIm = out[a + 1], v0r = out[b], v0i = out[b + 1];
        const vRe = v0r * wRe - v0i * wIm, vIm = v0r * wIm + v0i * wRe;
        out[a] = uRe + vRe; out[a + 1] = uIm + vIm;
        out[b] = uRe - vRe; out[b + 1] = uIm - vIm;
        const nRe = wRe * cRe - wIm * cIm, nIm = wRe * cIm + wIm * cRe;
        wRe = nRe; wIm = nIm;
      }
    }
  }
  return out;
}
// math.js:160-184
export function dft1dRef(input, N, direction) {
  const out = new Float32Array(2 * N);
  const sign = direction === "forward" ? -1.0 : 1.0;
  for (let k = 0; k < N; k++) {
    let re = 0, im = 0;
    for (let n = 0; n < N; n++) {
      const ang = (sign * 2.0 * Math.PI * n * k) / N;
      const c = Math.cos(ang), s = Math.sin(ang);
      re += input[2 * n] * c - input[2 * n + 1] * s;
      im += input[2 * n] * s + input[2 * n + 1] * c;
    }
    out[2 * k] = re; out[2 * k + 1] = im;
  }
  return out;
}
function isPow2(n) { return n >= 2 && (n & (n - 1)) === 0; }
// math.js:90-148 / 186-236 — axis 0 fastest, line by line, then one scale
export function fftNdRef(input, shape, direction, normalize, anySize) {
  const rank = shape.length;
  const nTotal = shape.reduce((a, b) => a * b, 1);
  const strides = [];
  let acc = 1;
  for (let d = 0; d < rank; d++) { strides.push(acc); acc *= shape[d]; }
  const useDft = anySize === undefined ? !shape.every(isPow2) : anySize;
  const data = new Float32Array(input);
  for (let axis = 0; axis < rank; axis++) {
    const N = shape[axis], stride = strides[axis], lines = nTotal / N;
    const line = new Float32Array(2 * N);
    for (let l = 0; l < lines; l++) {
      let rem = l, base = 0;
      for (let d = 0; d < rank; d++) {
        if (d === axis) continue;
        const c = rem % shape[d];
        rem = (rem - c) / shape[d];
        base += c * strides[d];
      }
      for (let p = 0; p < N; p++) { line[2 * p] = data[2 * (base + p * stride)]; line[2 * p + 1] = data[2 * (base + p * stride) + 1]; }
      const res = N === 1 ? line : (useDft ? dft1dRef(line, N, direction) : fft1dRef(line, N, direction));
      for (let p = 0; p < N; p++) { data[2 * (base + p * stride)] = res[2 * p]; data[2 * (base + p * stride) + 1] = res[2 * p + 1]; }
    }
  }
  const scale = normalizeScaleFactor(normalize || "none", direction, nTotal);
  if (scale !== 1.0) for (let i = 0; i < data.length; i++) data[i] = data[i] * scale;
  return data;
}
export function c2cRefBatch(input, shape, batch, direction, normalize) {
  const n = shape.reduce((a, b) => a * b, 1);
  const out = new Float32Array(2 * n * batch);
  for (let b = 0; b < batch; b++) out.set(fftNdRef(input.subarray(2 * b * n, 2 * (b + 1) * n), shape, direction, normalize), 2 * b * n);
  return out;
}
// math.js:238-258 (pow-2 N uses the radix-2 route: SURVEY.md 8c F4)
export function r2cRefPacked(x, N, normalize) {
  const c = new Float32Array(2 * N);
  for (let i = 0; i < N; i++) c[2 * i] = x[i];
  const full = fftNdRef(c, [N], "forward", "none");
  const P = Math.floor(N / 2) + 1;
  const out = new Float32Array(full.subarray(0, 2 * P));
  const scale = normalizeScaleFactor(normalize || "none", "forward", N);
  if (scale !== 1.0) for (let i = 0; i < out.length; i++) out[i] *= scale;
  return out;
}
// math.js:260-289
export function c2rRefFromPacked(packed, N, normalize) {
  const P = Math.floor(N / 2) + 1;
  const full = new Float32Array(2 * N);
  full.set(packed.subarray(0, 2 * P));
  const kMax = N % 2 === 0 ? N / 2 - 1 : Math.floor(N / 2);
  for (let k = 1; k <= kMax; k++) { full[2 * (N - k)] = full[2 * k]; full[2 * (N - k) + 1] = -full[2 * k + 1]; }
  const time = fftNdRef(full, [N], "inverse", "none");
  const out = new Float32Array(N);
  for (let n = 0; n < N; n++) out[n] = time[2 * n];
  const scale = normalizeScaleFactor(normalize || "none", "inverse", N);
  if (scale !== 1.0) for (let i = 0; i < N; i++) out[i] *= scale;
  return out;
}
// math.js:469-603
export function fftConvRef({ input, kernel, shape, batch, mode, boundary, kernelShape }) {
  batch = batch || 1; mode = mode || "convolution"; boundary = boundary || "circular";
  const rank = shape.length;
  const kShape = kernelShape || shape.slice();
  const fftShape = boundary === "circular" ? shape.slice() : shape.map((n, d) => n + kShape[d] - 1);
  let outShape, outOffset;
  if (boundary === "circular" || boundary === "linear-same") outShape = shape.slice(); else if (boundary === "linear-full") outShape = fftShape.slice(); else outShape = shape.map((n, d) => n - kShape[d] + 1);
  if (boundary === "linear-same") outOffset = kShape.map((n) => Math.floor((n - 1) / 2)); else if (boundary === "linear-valid") outOffset = kShape.map((n) => n - 1); else outOffset = new Array(rank).fill(0);
  const prodS = (s) => s.reduce((a, b) => a * b, 1);
  const stridesOf = (s) => { const o = []; let a = 1; for (const v of s) { o.push(a); a *= v; } return o; };
  const move = (dst, dstShape, dstOff, src, srcShape, srcOff, extent) => {
    const ds = stridesOf(dstShape), ss = stridesOf(srcShape), n = prodS(extent);
    for (let i = 0; i < n; i++) {
      let rem = i, di = 0, si = 0;
      for (let d = 0; d < rank; d++) { const c = rem % extent[d]; rem = (rem - c) / extent[d]; di += (dstOff[d] + c) * ds[d]; si += (srcOff[d] + c) * ss[d]; }
      dst[2 * di] = src[2 * si]; dst[2 * di + 1] = src[2 * si + 1];
    }
  };
  const zero = new Array(rank).fill(0);
  const inN = prodS(shape), fN = prodS(fftShape), oN = prodS(outShape);
  const kPad = new Float32Array(2 * fN);
  move(kPad, fftShape, zero, kernel, kShape, zero, kShape);
  const kf = fftNdRef(kPad, fftShape, "forward", "none", true);
  const out = new Float32Array(2 * oN * batch);
  for (let b = 0; b < batch; b++) {
    const xPad = new Float32Array(2 * fN);
    move(xPad, fftShape, zero, input.subarray(2 * b * inN, 2 * (b + 1) * inN), shape, zero, shape);
    const xf = fftNdRef(xPad, fftShape, "forward", "none", true);
    const yf = new Float32Array(2 * fN);
    for (let i = 0; i < fN; i++) {
      const ar = xf[2 * i], ai = xf[2 * i + 1], br = kf[2 * i], bi = mode === "correlation" ? -kf[2 * i + 1] : kf[2 * i + 1];
      yf[2 * i] = ar * br - ai * bi; yf[2 * i + 1] = ar * bi + ai * br;
    }
    const y = fftNdRef(yf, fftShape, "inverse", "backward", true);
    move(out.subarray(2 * b * oN, 2 * (b + 1) * oN), outShape, zero, y, fftShape, outOffset, outShape);
  }
  return out;
}
// parity metrics (BASELINE.md section 4)
export function relL2(a, e) { let d = 0, n = 0; for (let i = 0; i < e.length; i++) { d += (a[i] - e[i]) * (a[i] - e[i]); n += e[i] * e[i]; } return n > 0 ? Math.sqrt(d / n) : Math.sqrt(d); }
export function relMax(a, e) { let d = 0, m = 0; for (let i = 0; i < e.length; i++) { d = Math.max(d, Math.abs(a[i] - e[i])); m = Math.max(m, Math.abs(e[i])); } return m > 0 ? d / m : d; }
// the reference's per-element form (test/complete.node.test.js:14-25)
export function assertCloseArray(a, e, atol, rtol, what) {
  if (a.length !== e.length) throw new Error(what + ": length " + a.length + " != " + e.length);
  for (let i = 0; i < e.length; i++) {
    if (!(Math.abs(a[i] - e[i]) <= atol + rtol * Math.abs(e[i]))) throw new Error(what + ": element " + i + ": got " + a[i] + " expected " + e[i]);
  }
}
export function fnv1a64(typed) {
  const bytes = new Uint8Array(typed.buffer, typed.byteOffset, typed.byteLength);
  let hi = 0xcbf29ce4 | 0, lo = 0x84222325 | 0;
  for (let i = 0; i < bytes.length; i++) {
    lo ^= bytes[i];
    const loU = lo >>> 0, hiU = hi >>> 0;
    const p0 = (loU & 0xffff) * 0x1b3;
    const p1 = (loU >>> 16) * 0x1b3 + (p0 >>> 16);
    const newLo = ((p1 & 0xffff) << 16) | (p0 & 0xffff);
    const carry = Math.floor(p1 / 65536);
    hi = (Math.imul(hiU, 0x1b3) + (loU << 8) + carry) | 0;
    lo = newLo | 0;
  }
  return (hi >>> 0).toString(16).padStart(8, "0") + (lo >>> 0).toString(16).padStart(8, "0");
}

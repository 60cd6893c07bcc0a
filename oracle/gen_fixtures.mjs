// gen_fixtures.mjs — generates tests/golden/* by IMPORTING the reference's own CPU oracle
// (/root/reference/src/utils/math.js) under Node in the build container.
//
// Test infrastructure only.  The reference file is imported from where it lies, never copied; the
// outputs are data (inputs are regenerated from the seeded PRNG below, expected outputs are stored
// dense for small cases and as FNV-1a-64 hash + sampled bins for large ones).
//
//   node oracle/gen_fixtures.mjs [/root/reference] [tests/golden]
//
// Written to the Node-12 subset (no ?., ??, top-level await).
import fs from "fs";
import path from "path";
import { fileURLToPath, pathToFileURL } from "url";

const here = path.dirname(fileURLToPath(import.meta.url));
const refRoot = process.argv[2] || "/root/reference";
const outDir = process.argv[3] || path.join(here, "..", "tests", "golden");

// ---- seeded PRNG: mulberry32 with 32-bit wraparound state (twin of oracle.c:mulberry32_at) ----
function mulberry32(seed) {
  let a = seed | 0;
  return function () {
    a = (a + 0x6d2b79f5) | 0;
    let t = a;
    t = Math.imul(t ^ (t >>> 15), t | 1);
    t ^= t + Math.imul(t ^ (t >>> 7), t | 61);
    return ((t ^ (t >>> 14)) >>> 0) / 4294967296;
  };
}
function randomReal(n, rng) {
  const out = new Float32Array(n);
  for (let i = 0; i < n; i++) out[i] = (rng() * 2 - 1) * 0.5;
  return out;
}

function fnv1a64(typed) {
  const bytes = new Uint8Array(typed.buffer, typed.byteOffset, typed.byteLength);
  let h = 0xcbf29ce484222325n;
  const prime = 0x100000001b3n;
  const mask = 0xffffffffffffffffn;
  for (let i = 0; i < bytes.length; i++) {
    h ^= BigInt(bytes[i]);
    h = (h * prime) & mask;
  }
  return h.toString(16).padStart(16, "0");
}
// BigInt per byte is slow for MiB-sized buffers: do the 64-bit multiply with two 32-bit halves.
function fnv1a64Fast(typed) {
  const bytes = new Uint8Array(typed.buffer, typed.byteOffset, typed.byteLength);
  let hi = 0xcbf29ce4 | 0, lo = 0x84222325 | 0;
  for (let i = 0; i < bytes.length; i++) {
    lo ^= bytes[i];
    // (hi:lo) * 0x00000100000001b3  mod 2^64
    const loU = lo >>> 0, hiU = hi >>> 0;
    const lo16a = loU & 0xffff, lo16b = loU >>> 16;
    // low * 0x1b3
    let p0 = lo16a * 0x1b3;
    let p1 = lo16b * 0x1b3 + (p0 >>> 16);
    const newLo = ((p1 & 0xffff) << 16) | (p0 & 0xffff);
    const carry = Math.floor(p1 / 65536);
    // high word: hi*0x1b3 + lo*0x100 (from the 2^40 term: 0x100 << 32) + carry
    const newHi = (Math.imul(hiU, 0x1b3) + (loU << 8) + carry) | 0;
    lo = newLo | 0;
    hi = newHi | 0;
  }
  return (hi >>> 0).toString(16).padStart(8, "0") + (lo >>> 0).toString(16).padStart(8, "0");
}

function f64hex(x) {
  const b = Buffer.alloc(8);
  b.writeDoubleLE(x, 0);
  return b.toString("hex");
}
function writeF32(name, arr) {
  fs.writeFileSync(path.join(outDir, name), Buffer.from(arr.buffer, arr.byteOffset, arr.byteLength));
  return name;
}
function head(arr, n) {
  return Array.from(arr.subarray(0, Math.min(n, arr.length)));
}
function sampleIdx(total, count, seed) {
  const rng = mulberry32(seed);
  const idx = [];
  for (let i = 0; i < count; i++) idx.push(Math.floor(rng() * total));
  return idx;
}

import(pathToFileURL(path.join(refRoot, "src/utils/math.js")).href).then((ref) => {
  fs.mkdirSync(outDir, { recursive: true });
  const manifest = {
    schema: "mi355fft-golden",
    version: 1,
    generator: "oracle/gen_fixtures.mjs",
    reference: "MaximEremenko/WebGPU-FFT src/utils/math.js (imported, not copied)",
    node: process.version,
    prng: "mulberry32, 32-bit wraparound state; draw n = f(seed + (n+1)*0x6D2B79F5)",
    cases: [],
  };
  const add = (c) => manifest.cases.push(c);

  // sanity: the fast hash equals the BigInt hash
  {
    const probe = randomReal(257, mulberry32(1));
    if (fnv1a64(probe) !== fnv1a64Fast(probe)) throw new Error("fnv1a64Fast mismatch");
  }

  // ---- libm pin: the only transcendental calls on the radix-2 path (math.js:53-55) ----
  {
    const tw = [];
    for (let len = 2; len <= 1 << 24; len <<= 1) {
      for (const inverse of [0, 1]) {
        const ang = ((inverse ? 1.0 : -1.0) * 2.0 * Math.PI) / len;
        tw.push({ len, inverse, cos: f64hex(Math.cos(ang)), sin: f64hex(Math.sin(ang)) });
      }
    }
    add({ kind: "stage_twiddles", name: "stage_twiddles", entries: tw });
  }

  // ---- PRNG pin ----
  {
    const rng = mulberry32(0x5eed0000);
    const first = [];
    for (let i = 0; i < 8; i++) first.push(f64hex(rng()));
    const rc = ref.randomComplexInterleaved(16, mulberry32(0x5eed0001));
    add({ kind: "rng", name: "rng", seed: 0x5eed0000, first_f64: first, seed_complex: 0x5eed0001, complex16: head(rc, 32) });
  }

  // ---- normalizeScaleFactor (math.js:14-19) ----
  {
    const rows = [];
    for (const normalize of ["none", "backward", "unitary"])
      for (const direction of ["forward", "inverse"])
        for (const nTotal of [1, 8, 12, 1024, 1048576])
          rows.push({ normalize, direction, nTotal, value: f64hex(ref.normalizeScaleFactor({ normalize, direction, nTotal })) });
    add({ kind: "normalize_scale", name: "normalize_scale", rows });
  }

  // ---- F1/F2: c2c pow-2 via fftNdRefInterleaved, batch outermost ----
  let seedCounter = 0x5eed1000;
  for (const N of [2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096]) {
    for (const batch of [1, 4]) {
      for (const direction of ["forward", "inverse"]) {
        for (const normalize of ["none", "backward", "unitary"]) {
          const seed = seedCounter++;
          const input = ref.randomComplexInterleaved(N * batch, mulberry32(seed));
          const out = new Float32Array(2 * N * batch);
          for (let b = 0; b < batch; b++)
            out.set(ref.fftNdRefInterleaved(input.subarray(2 * b * N, 2 * (b + 1) * N), [N], direction, normalize), 2 * b * N);
          const c = {
            kind: "c2c_pow2", name: `c2c_N${N}_b${batch}_${direction}_${normalize}`, shape: [N], batch, direction, normalize,
            seed, out_fnv1a64: fnv1a64Fast(out), out_head: head(out, 8),
          };
          if ((N === 1024 && batch === 1 && direction === "forward" && normalize === "none") || (N === 8 && batch === 4)) {
            c.in_file = writeF32(c.name + ".in.f32", input);
            c.out_file = writeF32(c.name + ".out.f32", out);
          }
          add(c);
        }
      }
    }
  }

  // ---- N-D pow-2 (axis 0 fastest) ----
  for (const shape of [[8, 4], [16, 16], [4, 8, 2]]) {
    const n = shape.reduce((a, b) => a * b, 1);
    for (const direction of ["forward", "inverse"]) {
      const seed = seedCounter++;
      const input = ref.randomComplexInterleaved(n, mulberry32(seed));
      const out = ref.fftNdRefInterleaved(input, shape, direction, "unitary");
      add({ kind: "c2c_pow2", name: `c2c_nd_${shape.join("x")}_${direction}`, shape, batch: 1, direction, normalize: "unitary", seed,
            out_fnv1a64: fnv1a64Fast(out), out_head: head(out, 8) });
    }
  }

  // ---- F3: large pow-2 single transforms: hash + sampled bins ----
  for (const lg of [16, 20, 21]) {
    const N = 1 << lg;
    for (const direction of lg === 20 ? ["forward", "inverse"] : ["forward"]) {
      const seed = seedCounter++;
      const input = ref.randomComplexInterleaved(N, mulberry32(seed));
      const out = ref.fft1dRefInterleaved(input, N, direction);
      const idx = sampleIdx(N, 1024, seed ^ 0x1234);
      const samples = [];
      for (const k of idx) samples.push(out[2 * k], out[2 * k + 1]);
      let s = 0;
      for (let i = 0; i < out.length; i++) s += out[i] * out[i];
      add({ kind: "c2c_large", name: `c2c_N2p${lg}_${direction}`, shape: [N], batch: 1, direction, normalize: "none", seed,
            out_fnv1a64: fnv1a64Fast(out), sample_idx: idx, sample_vals: samples, out_sumsq: s });
    }
  }

  // ---- F4/F5: r2c / c2r ----
  // (a) the reference's own definition (O(N^2) DFT): small N, dense
  for (const N of [8, 16, 17, 64, 256, 1024]) {
    for (const normalize of ["none", "unitary"]) {
      const seed = seedCounter++;
      const x = randomReal(N, mulberry32(seed));
      const X = ref.r2cRefPackedInterleaved(x, N, "forward", normalize);
      const name = `r2c_dft_N${N}_${normalize}`;
      add({ kind: "r2c_dft", name, N, normalize, seed, out_file: writeF32(name + ".out.f32", X) });
    }
    for (const normalize of ["none", "backward"]) {
      const seed = seedCounter++;
      // a Hermitian-consistent packed spectrum: r2c of a random real signal
      const x = randomReal(N, mulberry32(seed));
      const X = ref.r2cRefPackedInterleaved(x, N, "forward", "none");
      const y = ref.c2rRefFromPackedInterleaved(X, N, normalize);
      const name = `c2r_dft_N${N}_${normalize}`;
      add({ kind: "c2r_dft", name, N, normalize, seed, in_file: writeF32(name + ".in.f32", X), out_file: writeF32(name + ".out.f32", y) });
    }
  }
  // (b) pow-2 route (radix-2 oracle on zero-imag input, first N/2+1 bins): SURVEY.md 8c F4
  for (const lg of [4, 10, 12, 20, 22]) {
    const N = 1 << lg;
    const seed = seedCounter++;
    const x = randomReal(N, mulberry32(seed));
    const cplx = new Float32Array(2 * N);
    for (let i = 0; i < N; i++) cplx[2 * i] = x[i];
    const full = ref.fft1dRefInterleaved(cplx, N, "forward");
    const packed = full.slice(0, 2 * (N / 2 + 1));
    const idx = sampleIdx(N / 2 + 1, 512, seed ^ 0x4321);
    const samples = [];
    for (const k of idx) samples.push(packed[2 * k], packed[2 * k + 1]);
    add({ kind: "r2c_pow2", name: `r2c_pow2_N2p${lg}`, N, normalize: "none", seed, out_fnv1a64: fnv1a64Fast(packed),
          sample_idx: idx, sample_vals: samples });
  }

  // ---- dft1dRefInterleaved, non-pow2 ----
  for (const N of [12, 17, 21, 34, 210]) {
    for (const direction of ["forward", "inverse"]) {
      const seed = seedCounter++;
      const input = ref.randomComplexInterleaved(N, mulberry32(seed));
      const out = ref.dft1dRefInterleaved(input, N, direction);
      const name = `dft_N${N}_${direction}`;
      add({ kind: "dft", name, N, direction, seed, out_file: writeF32(name + ".out.f32", out) });
    }
  }

  // ---- F6: fftconv ----
  const convCases = [
    { name: "fftconv_N12_b2_conv", shape: [12], batch: 2, kernels: 2, mode: "convolution", boundary: "circular", kernelShape: null },
    { name: "fftconv_N12_b2_corr", shape: [12], batch: 2, kernels: 1, mode: "correlation", boundary: "circular", kernelShape: null },
    { name: "fftconv_N21_b2_k3", shape: [21], batch: 2, kernels: 3, mode: "convolution", boundary: "circular", kernelShape: null },
    { name: "fftconv_N64_b3_corr", shape: [64], batch: 3, kernels: 2, mode: "correlation", boundary: "circular", kernelShape: null },
    { name: "fftconv_cfg4_N256_b4_k3", shape: [256], batch: 4, kernels: 3, mode: "convolution", boundary: "circular", kernelShape: null },
    { name: "fftconv_N17_k7_full", shape: [17], batch: 1, kernels: 1, mode: "convolution", boundary: "linear-full", kernelShape: [7] },
    { name: "fftconv_N17_k7_same", shape: [17], batch: 2, kernels: 1, mode: "convolution", boundary: "linear-same", kernelShape: [7] },
    { name: "fftconv_N17_k7_valid", shape: [17], batch: 1, kernels: 1, mode: "correlation", boundary: "linear-valid", kernelShape: [7] },
    { name: "fftconv_N26_k7_full_pow2", shape: [26], batch: 2, kernels: 2, mode: "convolution", boundary: "linear-full", kernelShape: [7] },
    { name: "fftconv_2d_8x4", shape: [8, 4], batch: 2, kernels: 2, mode: "convolution", boundary: "circular", kernelShape: null },
  ];
  for (const cc of convCases) {
    const seed = seedCounter++;
    const n = cc.shape.reduce((a, b) => a * b, 1);
    const kn = (cc.kernelShape || cc.shape).reduce((a, b) => a * b, 1);
    const input = ref.randomComplexInterleaved(n * cc.batch, mulberry32(seed));
    const kernels = ref.randomComplexInterleaved(kn * cc.kernels, mulberry32(seed ^ 0x00c0ffee));
    const outs = [];
    for (let k = 0; k < cc.kernels; k++) {
      outs.push(ref.fftConvRef({ input, kernel: kernels.subarray(2 * k * kn, 2 * (k + 1) * kn), shape: cc.shape, batch: cc.batch,
                                 mode: cc.mode, boundary: cc.boundary, kernelShape: cc.kernelShape }));
    }
    const per = outs[0].length;
    const all = new Float32Array(per * cc.kernels);        // kernel-major: [kernel][batch][logical]
    for (let k = 0; k < cc.kernels; k++) all.set(outs[k], k * per);
    add({ kind: "fftconv", name: cc.name, shape: cc.shape, batch: cc.batch, kernelCount: cc.kernels, mode: cc.mode,
          boundary: cc.boundary, kernelShape: cc.kernelShape, seed, kernel_seed: (seed ^ 0x00c0ffee) >>> 0,
          out_layout: "kernel-major", out_file: writeF32(cc.name + ".out.f32", all) });
  }

  // ---- F7: channel-lane preset known answers.  The reference's preset module does not parse under
  // Node 12 (?? operator), so these are the literal expectations its own unit tests assert
  // (test/c2c_large_batch.unit.test.js:5470-5563) plus the README example (README.md:55-63 / SURVEY 8a10).
  add({
    kind: "preset_known_answers", name: "preset_known_answers",
    source: "test/c2c_large_batch.unit.test.js:5470-5563; SURVEY.md 8(a) row a10",
    ok: [
      { fn: "createFftConvChannelLanePreset",
        opts: { shape: [8, 4], batch: 2, kernelCount: 3, input: { channels: 6 }, output: { channels: 12, kernelStepChannels: 2 } },
        expect: { shape: [8, 4], batch: 2, layout: { interleavedComplex: true },
                  fftConv: { mode: "convolution", boundary: "circular", outputLayout: "kernel-major", kernelCount: 3,
                             channelPolicy: { input: { channels: 6, channelIndex: 0, channelStrideElements: 32, batchStrideElements: 192, offsetElements: 0 },
                                              output: { channels: 12, channelIndex: 0, channelStrideElements: 32, batchStrideElements: 384, offsetElements: 0, kernelStepChannels: 2 } } } } },
      { fn: "createFftConvKernelMajorChannelLanePreset",
        opts: { shape: [256], batch: 4, kernelCount: 3, input: { channels: 64 }, output: { channels: 128, kernelStepChannels: 16 } },
        expect: { shape: [256], batch: 4, layout: { interleavedComplex: true },
                  fftConv: { mode: "convolution", boundary: "circular", outputLayout: "kernel-major", kernelCount: 3,
                             channelPolicy: { input: { channels: 64, channelIndex: 0, channelStrideElements: 256, batchStrideElements: 16384, offsetElements: 0 },
                                              output: { channels: 128, channelIndex: 0, channelStrideElements: 256, batchStrideElements: 32768, offsetElements: 0, kernelStepChannels: 16 } } } } },
    ],
    layout_forced: [
      { fn: "createFftConvKernelMajorChannelLanePreset", outputLayoutIn: "batch-major", expect: "kernel-major" },
      { fn: "createFftConvBatchMajorChannelLanePreset", outputLayoutIn: "kernel-major", expect: "batch-major" },
    ],
    layout_forced_base: { shape: [16], batch: 1, input: { channels: 8, channelIndex: 1 }, output: { channels: 8, channelIndex: 2, kernelStepChannels: 1 } },
    throws: [
      { opts: { shape: [32], batch: 1, kernelCount: 3, input: { channels: 8 }, output: { channels: 4, channelIndex: 1, kernelStepChannels: 2 } },
        regex: "does not fit kernelCount=3" },
      { opts: { shape: [32], batch: 1, input: { channels: 4 }, output: { channels: 4 }, layout: { whdcn: { channels: 4 } } },
        regex: "layout\\.whdcn cannot be combined" },
      { opts: { shape: [32], batch: 1, input: { channels: 4 }, output: { channels: 4 }, layout: { inputStrides: [1] } },
        regex: "layout\\.inputStrides cannot be combined" },
    ],
  });

  // ---- DCT / DST (math.js:291-409) — appended last so that the seeds of everything above stay what they were ----
  {
    let trigSeed = 0x7a160000;
    const fns = { dct1: (x, N) => ref.dct1Ref(x, N), dct2: (x, N, d) => ref.dct2Ref(x, N, d), dct3: (x, N, d) => ref.dct3Ref(x, N, d), dct4: (x, N) => ref.dct4Ref(x, N),
                  dst1: (x, N) => ref.dst1Ref(x, N), dst2: (x, N, d) => ref.dst2Ref(x, N, d), dst3: (x, N, d) => ref.dst3Ref(x, N, d), dst4: (x, N) => ref.dst4Ref(x, N) };
    for (const type of Object.keys(fns)) {
      for (const N of [2, 3, 8, 16, 17, 30]) {
        for (const direction of ["forward", "inverse"]) {
          const seed = trigSeed++;
          const input = randomReal(N, mulberry32(seed));
          const out = fns[type](input, N, direction);
          const name = `trig_${type}_N${N}_${direction}`;
          add({ kind: "trig", name, type, N, direction, seed, out_file: writeF32(name + ".out.f32", out) });
        }
      }
    }
  }

  fs.writeFileSync(path.join(outDir, "manifest.json"), JSON.stringify(manifest, null, 1));
  console.log(`wrote ${manifest.cases.length} cases to ${outDir}`);
}).catch((e) => { console.error(e); process.exit(1); });

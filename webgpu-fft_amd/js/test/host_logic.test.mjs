// CPU tier (no GPU): host option resolution against the reference's own unit-test expectations, the JS oracle
// twin against the golden fixtures, and that the addon loads, exports everything and fails loudly without a GPU.
import fs from "fs";
import path from "path";
import { fileURLToPath } from "url";
import { createRequire } from "module";
import { test, assert, assertThrows, deepEqual, run } from "./harness.mjs";
import * as fft from "../index.js";
import * as orc from "../../../oracle/oracle.mjs";

const here = path.dirname(fileURLToPath(import.meta.url));
const golden = path.join(here, "..", "..", "..", "tests", "golden");
const manifest = JSON.parse(fs.readFileSync(path.join(golden, "manifest.json"), "utf8"));
const cases = {};
for (const c of manifest.cases) cases[c.name] = c;
const loadF32 = (name) => { const b = fs.readFileSync(path.join(golden, name)); return new Float32Array(b.buffer.slice(b.byteOffset, b.byteOffset + b.byteLength)); };

test("preset known answers (test/c2c_large_batch.unit.test.js:5470-5563)", () => {
  const ka = cases.preset_known_answers;
  for (const ok of ka.ok) assert(deepEqual(fft[ok.fn](ok.opts), ok.expect), ok.fn);
  for (const lf of ka.layout_forced) {
    const opts = Object.assign({}, ka.layout_forced_base, { outputLayout: lf.outputLayoutIn });
    assert(fft[lf.fn](opts).fftConv.outputLayout === lf.expect, lf.fn);
  }
  for (const th of ka.throws) assertThrows(() => fft.createFftConvChannelLanePreset(th.opts), new RegExp(th.regex), th.regex);
});

test("cfg4 README example resolves to the lane descriptors of SURVEY 8(a) a10", () => {
  const p = fft.createFftConvKernelMajorChannelLanePreset({ shape: [256], batch: 4, kernelCount: 3, input: { channels: 64 }, output: { channels: 128, kernelStepChannels: 16 } });
  const { desc } = fft.resolvePlanOptions(Object.assign({ type: "fftconv" }, p));
  assert(deepEqual(desc.input, { strides: [1], offset: 0, batchStride: 16384 }));
  assert(deepEqual(desc.output, { strides: [1], offset: 0, batchStride: 32768 }));
  assert(desc.convOutputKernelStrideElements === 4096 && desc.convOutputLayout === 0);
});

test("option validation messages", () => {
  const bad = [
    [{ type: "c2c", shape: [], direction: "forward" }, /shape must be an array/],
    [{ type: "c2c", shape: [8, 0], direction: "forward" }, /positive ints/],
    [{ type: "c2c", shape: [8], direction: "sideways" }, /direction must be one of/],
    [{ type: "c2c", shape: [8], direction: "forward", normalize: "ortho" }, /normalize must be one of/],
    [{ type: "c2c", shape: [8], direction: "forward", batch: 0 }, /batch must be positive int/],
    [{ type: "r2c", shape: [8], direction: "inverse" }, /r2c supports direction:"forward" only/],
    [{ type: "c2r", shape: [8], direction: "forward" }, /c2r supports direction:"inverse" only/],
    [{ type: "fftconv", shape: [8], fftConv: { kernelShape: [9] } }, /must be <= shape/],
    [{ type: "dct2", shape: [8], direction: "forward", layout: { interleavedComplex: true } }, /DCT\/DST uses real buffers; set layout.interleavedComplex=false/],
    [{ type: "dst1", shape: [8, 1], layout: { interleavedComplex: false } }, /All DCT\/DST dimensions must be >= 2/],
    [{ type: "conv2d", shape: [8] }, /outside the MI355X hot path/],
    [{ type: "bogus", shape: [8] }, /type must be one of/],
  ];
  for (const [opts, re] of bad) assertThrows(() => fft.resolvePlanOptions(opts), re, JSON.stringify(opts));
});

test("ioView / zeroPad on r2c and c2r resolve against the real and the packed domain", () => {
  // r2c: the input view lives on the real domain (16), the output view and zeroPad.write on the packed one (9 bins)
  const a = fft.resolvePlanOptions({ type: "r2c", shape: [16], batch: 2, direction: "forward",
                                 ioView: { input: { shape: [10], placement: "center" }, output: { shape: [5], offset: [2] } },
                                 zeroPad: { write: { start: [0], end: [6] } } });
  assert(deepEqual(a.meta.ioView.input, { shape: [10], offset: [3], clearOutside: false }));
  assert(deepEqual(a.meta.ioView.output, { shape: [5], offset: [2], clearOutside: false }));
  assert(deepEqual(a.meta.zeroPad, { read: null, write: { start: [0], end: [6] } }));
  assertThrows(() => fft.resolvePlanOptions({ type: "r2c", shape: [16], direction: "forward", zeroPad: { write: { start: [0], end: [10] } } }),
               /zeroPad\.write\.end\[0\] must be <= shape\[0\] \(9\); got 10/, "r2c zeroPad.write");
  // c2r: mirrored — a full-domain packed view is a no-op, the real side takes the window
  const c = fft.resolvePlanOptions({ type: "c2r", shape: [16], direction: "inverse", ioView: { input: { shape: [9] }, output: { shape: [24], placement: "center", clearOutside: true } } });
  assert(c.meta.ioView.input === null);
  assert(deepEqual(c.meta.ioView.output, { shape: [24], offset: [-4], clearOutside: true }));
  assertThrows(() => fft.resolvePlanOptions({ type: "fftconv", shape: [16], fftConv: { kernelCount: 1 }, ioView: { input: { shape: [4] } } }), /ioView is not an fftconv option/, "fftconv views");
  // fftconv zeroPad ranges live on the FFT domain: 16 + 5 - 1 = 20 for a linear mode
  const f = fft.resolvePlanOptions({ type: "fftconv", shape: [16], fftConv: { kernelCount: 1, kernelShape: [5], boundary: "linear-full" }, zeroPad: { write: { start: [2], end: [19] } } });
  assert(deepEqual(f.desc.zeroWrite, { start: [2], end: [19] }));
});

test("DCT / DST options resolve (dct_fft.js:66-101): real buffers, direction defaults to forward", () => {
  const r = fft.resolvePlanOptions({ type: "dct3", shape: [16, 4], batch: 2, normalize: "unitary", layout: { interleavedComplex: false } });
  assert(r.desc.type === 6 && r.desc.direction === 0 && r.desc.normalize === 2 && r.meta.direction === "forward");
  assert(fft.resolvePlanOptions({ type: "dst4", shape: [8], direction: "inverse", layout: { interleavedComplex: false } }).desc.type === 11);
});

test("normalizeScaleFactor matches the reference bit for bit", () => {
  for (const r of cases.normalize_scale.rows) {
    const want = Buffer.from(r.value, "hex").readDoubleLE(0);
    assert(fft.normalizeScaleFactor(r) === want && orc.normalizeScaleFactor(r.normalize, r.direction, r.nTotal) === want, JSON.stringify(r));
  }
});

test("JS oracle twin: radix-2 path bit-exact against the reference fixtures", () => {
  let n = 0;
  for (const c of manifest.cases) {
    if (c.kind !== "c2c_pow2" || c.shape.length !== 1 || c.shape[0] > 1024) continue;
    const N = c.shape[0];
    const x = orc.randomComplexInterleaved(N * c.batch, orc.mulberry32(c.seed));
    const out = orc.c2cRefBatch(x, c.shape, c.batch, c.direction, c.normalize);
    assert(orc.fnv1a64(out) === c.out_fnv1a64, c.name);
    n++;
  }
  assert(n >= 100, "cases checked: " + n);
});

test("JS oracle twin: DFT path (r2c / c2r / fftconv) within 2e-6 of the reference fixtures", () => {
  const close = (a, e, what) => { let m = 1; for (const v of e) m = Math.max(m, Math.abs(v)); for (let i = 0; i < e.length; i++) assert(Math.abs(a[i] - e[i]) <= 4e-6 * m, what + " @" + i); };
  for (const c of manifest.cases) {
    if (c.kind === "r2c_dft" && c.N <= 256) close(orc.r2cRefPacked(orc.randomReal(c.N, orc.mulberry32(c.seed)), c.N, c.normalize), loadF32(c.out_file), c.name);
    if (c.kind === "c2r_dft" && c.N <= 256) close(orc.c2rRefFromPacked(loadF32(c.in_file), c.N, c.normalize), loadF32(c.out_file), c.name);
    if (c.kind === "fftconv" && c.shape.reduce((a, b) => a * b, 1) <= 64) {
      const n = c.shape.reduce((a, b) => a * b, 1), ks = c.kernelShape || c.shape, kn = ks.reduce((a, b) => a * b, 1);
      const x = orc.randomComplexInterleaved(n * c.batch, orc.mulberry32(c.seed));
      const k = orc.randomComplexInterleaved(kn * c.kernelCount, orc.mulberry32(c.kernel_seed));
      const want = loadF32(c.out_file), per = want.length / c.kernelCount;
      for (let i = 0; i < c.kernelCount; i++) {
        const got = orc.fftConvRef({ input: x, kernel: k.subarray(2 * i * kn, 2 * (i + 1) * kn), shape: c.shape, batch: c.batch, mode: c.mode, boundary: c.boundary, kernelShape: ks });
        close(got, want.subarray(i * per, (i + 1) * per), c.name);
      }
    }
  }
});

test("addon exports every entry point and has no CPU fallback", () => {
  const require = createRequire(import.meta.url);
  const native = require(path.join(here, "..", "..", "lib", "mi355fft.node"));
  for (const fn of ["abiVersion", "deviceCount", "deviceOpen", "deviceClose", "deviceInfo", "bufferAlloc", "bufferFree", "bufferWrite", "bufferRead",
    "bufferReadAsync", "planCreate", "planWorkspaceBytes", "planDescribe", "planExec", "planDestroy", "planRelease", "encoderBegin", "encoderCopyBuffer",
    "encoderFinish", "encoderDiscard", "queueSubmit", "commandsRelease", "queueWait", "queueWaitAsync", "fillRandom", "sumsq"]) {
    assert(typeof native[fn] === "function", fn);
  }
  assert(native.abiVersion() === 1);
  if (!fs.existsSync("/dev/kfd")) assertThrows(() => fft.openDevice(), /HIP|ROCm|device/, "openDevice without a GPU");
});

test("BufferView has the reference's constructor, fromBuffer and validation (src/utils/buffer_view.js:18-41)", () => {
  const fake = { size: 64, destroy() {} };
  const v = new fft.BufferView({ segments: [{ buffer: fake, offsetBytes: 8, sizeBytes: 32 }], logicalByteOffset: 8, lengthBytes: 16 });
  assert(v.segments.length === 1 && v.logicalByteOffset === 8 && v.lengthBytes === 16);
  const w = fft.BufferView.fromBuffer(fake, 16);
  assert(w.lengthBytes === 48 && w.logicalByteOffset === 0 && w.segments[0].offsetBytes === 16 && w.segments[0].sizeBytes === 48);
  assert(fft.BufferView.fromBuffer(fake).lengthBytes === 64);
  assertThrows(() => new fft.BufferView({ segments: [], lengthBytes: 8 }), /segments must be a non-empty array/);
  assertThrows(() => new fft.BufferView({ segments: [{ buffer: fake, offsetBytes: 0, sizeBytes: 8 }], logicalByteOffset: -1, lengthBytes: 8 }), /logicalByteOffset must be a non-negative integer/);
  assertThrows(() => new fft.BufferView({ segments: [{ buffer: fake, offsetBytes: 0, sizeBytes: 8 }] }), /lengthBytes must be a positive integer/);
  assertThrows(() => new fft.BufferView({ segments: [{ offsetBytes: 0, sizeBytes: 8 }], lengthBytes: 8 }), /segment missing buffer/);
  assertThrows(() => new fft.BufferView({ segments: [{ buffer: fake, offsetBytes: 60, sizeBytes: 16 }], lengthBytes: 16 }), /segment out of bounds: offsetBytes\+sizeBytes=76 > buffer.size=64/);
  // round-1 spellings stay as aliases
  const a = fft.BufferView.from(fake, 8, 32);
  assert(a.size === 32 && a.lengthBytes === 32 && a.segments.length === 1);
  assert(new fft.BufferView([{ buffer: fake, offsetBytes: 8 }]).lengthBytes === 56);
});

run();

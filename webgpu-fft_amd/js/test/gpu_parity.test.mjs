// GPU tier: the JavaScript host -> N-API addon -> C ABI -> HIP kernels, written the way the reference's own
// suite is (test/complete.suite.js: randomComplexInterleaved -> uploadComplex -> createPlan -> exec -> submit ->
// downloadComplex -> CPU oracle -> assertCloseArray), with seeded inputs and the norm-relative 1e-5 bar on top.
import fs from "fs";
import path from "path";
import { fileURLToPath } from "url";
import { test, assert, assertThrows, run } from "./harness.mjs";
import * as fft from "../index.js";
import * as orc from "../../../oracle/oracle.mjs";

const here = path.dirname(fileURLToPath(import.meta.url));
const golden = path.join(here, "..", "..", "..", "tests", "golden");
const manifest = JSON.parse(fs.readFileSync(path.join(golden, "manifest.json"), "utf8"));
const cases = {};
for (const c of manifest.cases) cases[c.name] = c;
const loadF32 = (name) => { const b = fs.readFileSync(path.join(golden, name)); return new Float32Array(b.buffer.slice(b.byteOffset, b.byteOffset + b.byteLength)); };
const TOL = 1e-5;
let device = null;
async function ensureDevice() { if (!device) device = await fft.requestDevice(); return device; }
function check(got, want, atol, rtol, what) {
  const l2 = orc.relL2(got, want), mx = orc.relMax(got, want);
  assert(l2 <= TOL && mx <= TOL, what + ": rel_l2=" + l2.toExponential(3) + " rel_max=" + mx.toExponential(3));
  orc.assertCloseArray(got, want, atol, rtol, what);
}

for (const N of [8, 16, 34, 128, 1024, 4096, 210, 17, 29, 97, 2039]) {   // complete.suite.js:664-676 sizes (+ pow-2)
  test("c2c 1D forward/inverse N=" + N + " matches the CPU oracle", async () => {
    const dev = await ensureDevice();
    for (const direction of ["forward", "inverse"]) {
      const input = orc.randomComplexInterleaved(N, orc.mulberry32(1000 + N));
      const inBuf = fft.uploadComplex(dev, input);
      const outBuf = dev.createBuffer({ size: input.byteLength, usage: GPUBufferUsage.STORAGE | GPUBufferUsage.COPY_SRC | GPUBufferUsage.COPY_DST });
      const plan = fft.createPlan(dev, { type: "c2c", shape: [N], direction, normalize: "none" });
      const enc = dev.createCommandEncoder();
      plan.exec(enc, { input: inBuf, output: outBuf });
      dev.queue.submit([enc.finish()]);
      await dev.queue.onSubmittedWorkDone();
      const gpu = await fft.downloadComplex(dev, outBuf, N);
      check(gpu, orc.fftNdRef(input, [N], direction, "none"), 3e-4, 3e-4, "c2c N=" + N + " " + direction);
      plan.destroy(); inBuf.destroy(); outBuf.destroy();
    }
  });
}

test("c2c batch (N=32, batch=4) + golden fixture (N=8, batch=4)", async () => {
  const dev = await ensureDevice();
  const c = cases.c2c_N8_b4_forward_none;
  const input = loadF32(c.in_file), want = loadF32(c.out_file);
  const inBuf = fft.uploadComplex(dev, input);
  const outBuf = dev.createBuffer({ size: input.byteLength, usage: 0 });
  const plan = fft.createPlan(dev, { type: "c2c", shape: [8], batch: 4, direction: "forward", normalize: "none" });
  const enc = dev.createCommandEncoder();
  plan.exec(enc, { input: inBuf, output: outBuf });
  dev.queue.submit([enc.finish()]);
  check(await fft.downloadComplex(dev, outBuf, 32), want, 5e-4, 5e-4, "golden N=8 b=4");
  plan.destroy(); inBuf.destroy(); outBuf.destroy();
});

test("round trip in one encoder (forward then inverse backward) N=2^16, in order (complete.suite.js:619-662)", async () => {
  const dev = await ensureDevice();
  const N = 1 << 16;
  const input = orc.randomComplexInterleaved(N, orc.mulberry32(77));
  const a = fft.uploadComplex(dev, input);
  const b = dev.createBuffer({ size: input.byteLength, usage: 0 });
  const fwd = fft.createPlan(dev, { type: "c2c", shape: [N], direction: "forward", normalize: "none" });
  const inv = fft.createPlan(dev, { type: "c2c", shape: [N], direction: "inverse", normalize: "backward", inPlace: true });
  const enc = dev.createCommandEncoder();
  fwd.exec(enc, { input: a, output: b });
  inv.exec(enc, { input: b });
  dev.queue.submit([enc.finish({ useGraph: true })]);
  await dev.queue.onSubmittedWorkDone();
  check(await fft.downloadComplex(dev, b, N), input, 3e-3, 3e-3, "round trip");
  fwd.destroy(); inv.destroy(); a.destroy(); b.destroy();
});

test("createFftPlan low-level API: batch at exec, in-place vs out-of-place (test/fft_correctness.test.js:86-205)", async () => {
  const dev = await ensureDevice();
  const N = 64, batch = 4;
  const input = orc.randomComplexInterleaved(N * batch, orc.mulberry32(5));
  const want = orc.c2cRefBatch(input, [N], batch, "forward", "unitary");
  const a = fft.uploadComplex(dev, input);
  const b = dev.createBuffer({ size: input.byteLength, usage: 0 });
  const p = fft.createFftPlan(dev, { shape: [N], direction: "forward", normalize: "unitary", inPlace: false });
  let enc = dev.createCommandEncoder();
  p.exec(enc, { input: a, output: b, batch });
  dev.queue.submit([enc.finish()]);
  check(await fft.downloadComplex(dev, b, N * batch), want, 1e-4, 1e-4, "FftPlan out-of-place");
  assertThrows(() => p.exec(dev.createCommandEncoder(), { input: a, output: a, batch }), /input !== output/);
  const q = fft.createFftPlan(dev, { shape: [N], direction: "forward", normalize: "unitary", inPlace: true });
  enc = dev.createCommandEncoder();
  q.exec(enc, { input: a, batch });
  dev.queue.submit([enc.finish()]);
  check(await fft.downloadComplex(dev, a, N * batch), want, 1e-4, 1e-4, "FftPlan in-place");
  p.destroy(); q.destroy();
  assertThrows(() => p.exec(dev.createCommandEncoder(), { input: a, output: b }), /FftPlan is destroyed/);
  a.destroy(); b.destroy();
});

test("createFftPlan axes subsets (plan.js:1335-1339): axis 1 of a 16x8 array only", async () => {
  const dev = await ensureDevice();
  const N0 = 16, N1 = 8, batch = 3;
  const input = orc.randomComplexInterleaved(N0 * N1 * batch, orc.mulberry32(77));
  // reference: for every (batch, i0) the line over i1 (stride N0) through the 1-D oracle
  const want = new Float32Array(input.length);
  for (let b = 0; b < batch; ++b) {
    for (let i0 = 0; i0 < N0; ++i0) {
      const line = new Float32Array(2 * N1);
      for (let i1 = 0; i1 < N1; ++i1) { const p = 2 * (b * N0 * N1 + i1 * N0 + i0); line[2 * i1] = input[p]; line[2 * i1 + 1] = input[p + 1]; }
      const y = orc.c2cRefBatch(line, [N1], 1, "forward", "none");
      for (let i1 = 0; i1 < N1; ++i1) { const p = 2 * (b * N0 * N1 + i1 * N0 + i0); want[p] = y[2 * i1]; want[p + 1] = y[2 * i1 + 1]; }
    }
  }
  const a = fft.uploadComplex(dev, input);
  const o = dev.createBuffer({ size: input.byteLength, usage: 0 });
  const p = fft.createFftPlan(dev, { shape: [N0, N1], direction: "forward", axes: [1] });
  const enc = dev.createCommandEncoder();
  p.exec(enc, { input: a, output: o, batch });
  dev.queue.submit([enc.finish()]);
  check(await fft.downloadComplex(dev, o, N0 * N1 * batch), want, 1e-4, 1e-4, "FftPlan axes=[1]");
  assertThrows(() => fft.createFftPlan(dev, { shape: [N0, N1], direction: "forward", axes: [2] }), /Invalid axis 2 for rank 2/);
  assertThrows(() => fft.createFftPlan(dev, { shape: [N0, N1], direction: "forward", axes: [] }), /axes must be null or a non-empty array/);
  p.destroy(); a.destroy(); o.destroy();
});

test("r2c / c2r N=16, 1024 (complete.suite.js:1776-1813 shape) + golden fixtures", async () => {
  const dev = await ensureDevice();
  for (const N of [16, 1024]) {
    const x = orc.randomReal(N, orc.mulberry32(300 + N));
    const P = N / 2 + 1;
    const xb = dev.createBuffer({ size: N * 4, usage: 0 });
    dev.queue.writeBuffer(xb, 0, x);
    const sb = dev.createBuffer({ size: P * 8, usage: 0 });
    const r2c = fft.createPlan(dev, { type: "r2c", shape: [N], direction: "forward", normalize: "none" });
    let enc = dev.createCommandEncoder();
    r2c.exec(enc, { input: xb, output: sb });
    dev.queue.submit([enc.finish()]);
    const spec = await fft.downloadComplex(dev, sb, P);
    check(spec, orc.r2cRefPacked(x, N, "none"), 8e-4, 8e-4, "r2c N=" + N);
    const yb = dev.createBuffer({ size: N * 4, usage: 0 });
    const c2r = fft.createPlan(dev, { type: "c2r", shape: [N], direction: "inverse", normalize: "backward" });
    enc = dev.createCommandEncoder();
    c2r.exec(enc, { input: sb, output: yb });
    dev.queue.submit([enc.finish()]);
    check(await fft.downloadF32(dev, yb, N), x, 2e-3, 2e-3, "c2r(r2c) N=" + N);
    r2c.destroy(); c2r.destroy(); xb.destroy(); sb.destroy(); yb.destroy();
  }
  const c = cases.r2c_dft_N64_none;
  const x = orc.randomReal(64, orc.mulberry32(c.seed));
  const xb = dev.createBuffer({ size: 256, usage: 0 });
  dev.queue.writeBuffer(xb, 0, x);
  const sb = dev.createBuffer({ size: 33 * 8, usage: 0 });
  const plan = fft.createPlan(dev, { type: "r2c", shape: [64], direction: "forward" });
  const enc = dev.createCommandEncoder();
  plan.exec(enc, { input: xb, output: sb });
  dev.queue.submit([enc.finish()]);
  check(await fft.downloadComplex(dev, sb, 33), loadF32(c.out_file), 8e-4, 8e-4, "r2c golden N=64");
  plan.destroy(); xb.destroy(); sb.destroy();
});

test("fftconv channelPolicy maps multi-kernel outputs into channel lanes (N=12,batch=2,kernels=2) — complete.suite.js:4715-4836", async () => {
  const dev = await ensureDevice();
  const shape = [12], batch = 2, kernelCount = 2, n = 12;
  const inputChannels = 3, inputChannelIndex = 1, inputChannelStride = 16, inputBatchStride = 80;
  const outputChannels = 5, outputChannelIndex = 1, outputChannelStride = 20, outputBatchStride = 160, kernelStepChannels = 1;
  const outputLast = (outputChannelIndex + (kernelCount - 1) * kernelStepChannels) * outputChannelStride;
  const outputElems = outputLast + (batch - 1) * outputBatchStride + n;
  const logicalInput = orc.randomComplexInterleaved(n * batch, orc.mulberry32(4242));
  const inElems = inputChannelIndex * inputChannelStride + (batch - 1) * inputBatchStride + n;
  const inputPhys = new Float32Array(2 * inElems).fill(9.0);
  for (let b = 0; b < batch; b++) inputPhys.set(logicalInput.subarray(2 * b * n, 2 * (b + 1) * n), 2 * (inputChannelIndex * inputChannelStride + b * inputBatchStride));
  const kernels = [orc.randomComplexInterleaved(n, orc.mulberry32(1)), orc.randomComplexInterleaved(n, orc.mulberry32(2))];
  const inBuf = fft.uploadComplex(dev, inputPhys);
  const outBuf = dev.createBuffer({ size: outputElems * 8, usage: GPUBufferUsage.STORAGE | GPUBufferUsage.COPY_SRC | GPUBufferUsage.COPY_DST });
  const outSentinel = new Float32Array(2 * outputElems);
  for (let i = 0; i < outputElems; i++) { outSentinel[2 * i] = 77.0; outSentinel[2 * i + 1] = -55.0; }
  dev.queue.writeBuffer(outBuf, 0, outSentinel);
  const plan = fft.createPlan(dev, {
    type: "fftconv", shape, batch, layout: { interleavedComplex: true }, precision: "f32",
    fftConv: { mode: "convolution", kernelCount, outputLayout: "batch-major",
      channelPolicy: { input: { channels: inputChannels, channelIndex: inputChannelIndex, channelStrideElements: inputChannelStride, batchStrideElements: inputBatchStride },
        output: { channels: outputChannels, channelIndex: outputChannelIndex, channelStrideElements: outputChannelStride, batchStrideElements: outputBatchStride, kernelStepChannels } } },
  });
  assert(plan._usesStridedInput === true && plan._usesStridedOutput === true, "strided policy routing");
  assert(plan._stridedOutputKernelStrideElements === outputChannelStride * kernelStepChannels, "kernel stride");
  assert(plan._outputOffsetElements === outputChannelIndex * outputChannelStride && plan._outputBatchStrideElements === outputBatchStride && plan._outputStrides[0] === 1);
  const enc = dev.createCommandEncoder();
  plan.exec(enc, { input: inBuf, output: outBuf, kernel: kernels });
  dev.queue.submit([enc.finish()]);
  await dev.queue.onSubmittedWorkDone();
  const gpuPhys = await fft.downloadF32(dev, outBuf, 2 * outputElems);
  const expected = new Float32Array(outSentinel);
  for (let k = 0; k < kernelCount; k++) {
    const cpu = orc.fftConvRef({ input: logicalInput, kernel: kernels[k], shape, batch, mode: "convolution" });
    for (let b = 0; b < batch; b++) {
      const lane = (outputChannelIndex + k * kernelStepChannels) * outputChannelStride + b * outputBatchStride;
      expected.set(cpu.subarray(2 * b * n, 2 * (b + 1) * n), 2 * lane);
    }
  }
  orc.assertCloseArray(gpuPhys, expected, 5e-3, 5e-3, "fftconv channelPolicy multi-kernel lanes");
  plan.destroy(); inBuf.destroy(); outBuf.destroy();
});

test("fftconv kernels as an array of GPU buffers / BufferViews (array-sources, fftconv.js:920-941) and BufferView input", async () => {
  const dev = await ensureDevice();
  const n = 64, batch = 3, K = 3;
  const x = orc.randomComplexInterleaved(n * batch, orc.mulberry32(777));
  const kern = [0, 1, 2].map((k) => orc.randomComplexInterleaved(n, orc.mulberry32(900 + k)));
  // kernel 0: its own buffer; kernel 1: a BufferView into the middle of a larger buffer; kernel 2: a view with a logical offset
  const k0 = fft.uploadComplex(dev, kern[0]);
  const big = dev.createBuffer({ size: 4 * n * 8, usage: 0 });
  dev.queue.writeBuffer(big, n * 8, kern[1]);
  dev.queue.writeBuffer(big, 3 * n * 8, kern[2]);
  const k1 = fft.BufferView.fromBuffer(big, n * 8, n * 8);
  const k2 = new fft.BufferView({ segments: [{ buffer: big, offsetBytes: 2 * n * 8, sizeBytes: 2 * n * 8 }], logicalByteOffset: n * 8, lengthBytes: n * 8 });
  // the data sits behind a 128-byte prefix of a larger buffer and is passed as a view
  const pre = 16;
  const inPhys = new Float32Array(2 * (pre + n * batch)).fill(3.0);
  inPhys.set(x, 2 * pre);
  const inBuf = fft.uploadComplex(dev, inPhys);
  const outBuf = dev.createBuffer({ size: K * batch * n * 8, usage: 0 });
  const plan = fft.createPlan(dev, { type: "fftconv", shape: [n], batch, fftConv: { mode: "convolution", boundary: "circular", kernelCount: K } });
  const enc = dev.createCommandEncoder();
  plan.exec(enc, { input: fft.BufferView.fromBuffer(inBuf, pre * 8), output: outBuf, kernel: [k0, k1, k2] });
  dev.queue.submit([enc.finish()]);
  const got = await fft.downloadComplex(dev, outBuf, K * batch * n);
  const want = new Float32Array(2 * K * batch * n);
  for (let k = 0; k < K; k++) want.set(orc.fftConvRef({ input: x, kernel: kern[k], shape: [n], batch, mode: "convolution" }), 2 * k * batch * n);
  orc.assertCloseArray(got, want, 5e-3, 5e-3, "fftconv array-sources");
  assert(orc.relL2(got, want) < TOL, "fftconv array-sources rel_l2");
  assertThrows(() => plan.exec(dev.createCommandEncoder(), { input: inBuf, output: outBuf, kernel: [k0, kern[1], k2] }), /all Float32Array or all GPUBuffer/);
  assertThrows(() => plan.exec(dev.createCommandEncoder(), { input: inBuf, output: outBuf, kernel: [k0, fft.BufferView.fromBuffer(big, 0, 8), k2] }), /too small/);
  plan.destroy(); inBuf.destroy(); outBuf.destroy(); k0.destroy(); big.destroy();
});

test("fftconv BASELINE config 4 (README preset: shape=[256] batch=4, 64ch -> 128ch, 3 kernels) vs reference fixture", async () => {
  const dev = await ensureDevice();
  const c = cases.fftconv_cfg4_N256_b4_k3;
  const preset = fft.createFftConvKernelMajorChannelLanePreset({ shape: [256], batch: 4, kernelCount: 3, input: { channels: 64 }, output: { channels: 128, kernelStepChannels: 16 } });
  const n = 256, batch = 4, K = 3;
  const logical = orc.randomComplexInterleaved(n * batch, orc.mulberry32(c.seed));
  const kern = orc.randomComplexInterleaved(n * K, orc.mulberry32(c.kernel_seed));
  const phys = new Float32Array(2 * 4 * 64 * 256).fill(5.0);
  for (let b = 0; b < batch; b++) phys.set(logical.subarray(2 * b * n, 2 * (b + 1) * n), 2 * b * 16384);
  const outElems = 4 * 128 * 256;
  const sentinel = new Float32Array(2 * outElems);
  for (let i = 0; i < outElems; i++) { sentinel[2 * i] = 77.0; sentinel[2 * i + 1] = -55.0; }
  const inBuf = fft.uploadComplex(dev, phys);
  const outBuf = fft.uploadComplex(dev, sentinel);
  const plan = fft.createPlan(dev, Object.assign({ type: "fftconv" }, preset));
  const enc = dev.createCommandEncoder();
  plan.exec(enc, { input: inBuf, output: outBuf, kernel: kern });
  dev.queue.submit([enc.finish()]);
  const got = await fft.downloadF32(dev, outBuf, 2 * outElems);
  const gold = loadF32(c.out_file);
  const want = new Float32Array(sentinel);
  for (let k = 0; k < K; k++) for (let b = 0; b < batch; b++) want.set(gold.subarray((k * batch + b) * 2 * n, (k * batch + b + 1) * 2 * n), 2 * (k * 16 * 256 + b * 32768));
  orc.assertCloseArray(got, want, 5e-3, 5e-3, "cfg4");
  assert(orc.relL2(got, want) < TOL, "cfg4 rel_l2");
  plan.destroy(); inBuf.destroy(); outBuf.destroy();
});

test("c2c ioView pad-in-read + zeroPad.write (docs/API.md examples)", async () => {
  const dev = await ensureDevice();
  const N = 16, V = 10;
  const view = orc.randomComplexInterleaved(V, orc.mulberry32(99));
  const logical = new Float32Array(2 * N);
  logical.set(view, 2 * 3);                                   // placement "center": offset floor((16-10)/2) = 3
  const want = orc.fftNdRef(logical, [N], "forward", "none");
  for (let k = 0; k < N; k++) if (k < 2 || k >= 14) { want[2 * k] = 0; want[2 * k + 1] = 0; }   // zeroPad.write [2,14)
  const inBuf = fft.uploadComplex(dev, view);
  const outBuf = dev.createBuffer({ size: N * 8, usage: 0 });
  const plan = fft.createPlan(dev, { type: "c2c", shape: [N], direction: "forward", ioView: { input: { shape: [V], placement: "center" } },
    zeroPad: { write: { start: [2], end: [14] } } });
  const enc = dev.createCommandEncoder();
  plan.exec(enc, { input: inBuf, output: outBuf });
  dev.queue.submit([enc.finish()]);
  orc.assertCloseArray(await fft.downloadComplex(dev, outBuf, N), want, 3e-4, 3e-4, "ioView + zeroPad");
  plan.destroy(); inBuf.destroy(); outBuf.destroy();
});

test("error behaviour: destroyed plan, missing output, in-place aliasing, offsets", async () => {
  const dev = await ensureDevice();
  const a = dev.createBuffer({ size: 1024, usage: 0 }), b = dev.createBuffer({ size: 1024, usage: 0 });
  const p = fft.createPlan(dev, { type: "c2c", shape: [64], direction: "forward" });
  assertThrows(() => p.exec(dev.createCommandEncoder(), { input: a }), /exec requires output when inPlace=false/);
  assertThrows(() => p.exec(dev.createCommandEncoder(), {}), /exec requires input/);
  assertThrows(() => p.exec(dev.createCommandEncoder(), { input: a, output: b, inputOffsetBytes: 4 }), /multiples of 8/);
  assertThrows(() => p.exec(dev.createCommandEncoder(), { input: a, output: b, inputOffsetBytes: 1016 }), /too small/);
  p.destroy(); p.destroy();
  assertThrows(() => p.exec(dev.createCommandEncoder(), { input: a, output: b }), /plan destroyed/);
  const q = fft.createPlan(dev, { type: "c2c", shape: [64], direction: "forward", inPlace: true });
  assertThrows(() => q.exec(dev.createCommandEncoder(), { input: a, output: b }), /inPlace=true requires output omitted or equal to input/);
  assert(typeof q.getWorkspaceSizeBytes() === "number" && q.getPipelineCacheSnapshot().schema === "webgpufft.pipeline-cache");
  q.destroy(); a.destroy(); b.destroy();
  assertThrows(() => fft.createPlan(dev, { type: "conv2d", shape: [16] }), /Unsupported/);
});

test("dct2 / dst2 over real buffers (complete.suite.js:3885-3932 shape): N=16 against the defining sums", async () => {
  const dev = await ensureDevice();
  const N = 16;
  const x = orc.randomReal(N, orc.mulberry32(901));
  const xb = dev.createBuffer({ size: N * 4, usage: 0 });
  dev.queue.writeBuffer(xb, 0, x);
  for (const type of ["dct2", "dst2"]) {
    const want = new Float32Array(N);
    for (let k = 0; k < N; ++k) {
      let sum = 0;
      for (let n = 0; n < N; ++n) sum += x[n] * (type === "dct2" ? Math.cos((Math.PI / N) * (n + 0.5) * k) : Math.sin((Math.PI / N) * (n + 0.5) * (k + 1)));
      want[k] = sum;
    }
    const yb = dev.createBuffer({ size: N * 4, usage: 0 });
    const p = fft.createPlan(dev, { type, shape: [N], direction: "forward", normalize: "none", layout: { interleavedComplex: false }, precision: "f32" });
    const enc = dev.createCommandEncoder();
    p.exec(enc, { input: xb, output: yb });
    dev.queue.submit([enc.finish()]);
    const got = await fft.downloadComplex(dev, yb, N / 2);          // N real floats
    check(got, want, 2e-3, 2e-3, type + " N=16");
    p.destroy(); yb.destroy();
  }
  xb.destroy();
});

run();

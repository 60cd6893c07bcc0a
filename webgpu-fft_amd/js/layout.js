// layout.js — host-side option resolution: createPlan opts -> the resolved form the C ABI takes.
//
// Mirrors (same names, argument meaning and error text where the reference's tests match on it):
//   runtime/fftconv_channel_lane_presets.js:139-206   preset builders
//   runtime/layout_semantics.js:120-232               layout.{strides,...} and layout.whdcn resolution
//   runtime/plans/fftconv.js:144-281                  fftConv.channelPolicy -> whdcn side descriptors
//   runtime/plans/c2c.js:547-558, fftconv.js:320-338  option validation
// Python twin: webgpu-fft_amd/python/mi355fft/layout.py.  Written to the Node 12 subset (no ?. / ??).

const FFTCONV_MODES = ["convolution", "correlation"];
const FFTCONV_BOUNDARIES = ["circular", "linear-full", "linear-same", "linear-valid"];
const FFTCONV_OUTPUT_LAYOUTS = ["kernel-major", "batch-major"];
const CONFLICTING_LAYOUT_KEYS = ["whdcn", "strides", "inputStrides", "outputStrides", "offsetElements", "inputOffsetElements",
  "outputOffsetElements", "batchStrideElements", "inputBatchStrideElements", "outputBatchStrideElements"];
const HOT_PATH_TYPES = ["c2c", "r2c", "c2r", "fftconv"];
const ALL_TYPES = ["c2c", "r2c", "c2r", "dct1", "dct2", "dct3", "dct4", "dst1", "dst2", "dst3", "dst4", "fftconv", "conv2d"];

export const TYPE_CODE = { c2c: 0, r2c: 1, c2r: 2, fftconv: 3, dct1: 4, dct2: 5, dct3: 6, dct4: 7, dst1: 8, dst2: 9, dst3: 10, dst4: 11 };
const TRIG_TYPES = ["dct1", "dct2", "dct3", "dct4", "dst1", "dst2", "dst3", "dst4"];   // real-to-real, real f32 buffers (dct_fft.js)
export const DIRECTION_CODE = { forward: 0, inverse: 1 };
export const NORMALIZE_CODE = { none: 0, backward: 1, unitary: 2 };
export const CONV_MODE_CODE = { convolution: 0, correlation: 1 };
export const CONV_BOUNDARY_CODE = { circular: 0, "linear-full": 1, "linear-same": 2, "linear-valid": 3 };
export const CONV_LAYOUT_CODE = { "kernel-major": 0, "batch-major": 1 };

function hasOwn(o, k) { return Object.prototype.hasOwnProperty.call(o, k); }
function dflt(v, d) { return v === undefined || v === null ? d : v; }
function isPlainObject(v) { return v != null && typeof v === "object" && !Array.isArray(v); }
function assertOneOf(value, allowed, name) {
  if (!allowed.includes(value)) {
    throw new Error(name + " must be one of " + allowed.map((v) => JSON.stringify(v)).join(", ") + "; got " + JSON.stringify(value));
  }
}
function assertPositiveSafeInt(v, name) {
  if (!Number.isInteger(v) || v <= 0 || !Number.isSafeInteger(v)) throw new Error(name + " must be a positive safe integer");
}
function assertNonNegativeSafeInt(v, name) {
  if (!Number.isInteger(v) || v < 0 || !Number.isSafeInteger(v)) throw new Error(name + " must be a non-negative safe integer");
}
export function prod(arr) { let p = 1; for (const v of arr) p *= v; return p; }

// ---- preset builders (fftconv_channel_lane_presets.js) ------------------------------------------------
function normalizeSideDescriptor(side, sideName, logicalSpan, kernelCount, allowKernelStep) {
  if (!isPlainObject(side)) throw new Error(sideName + " must be an object");
  assertPositiveSafeInt(side.channels, sideName + ".channels");
  const channels = side.channels;
  const channelIndex = dflt(side.channelIndex, 0);
  assertNonNegativeSafeInt(channelIndex, sideName + ".channelIndex");
  if (channelIndex >= channels) throw new Error(sideName + ".channelIndex (" + channelIndex + ") must be < " + sideName + ".channels (" + channels + ")");
  const channelStrideElements = dflt(side.channelStrideElements, logicalSpan);
  assertPositiveSafeInt(channelStrideElements, sideName + ".channelStrideElements");
  if (channelStrideElements < logicalSpan) throw new Error(sideName + ".channelStrideElements must be >= logical span (" + logicalSpan + ")");
  const defaultBatchStride = channels * channelStrideElements;
  if (!Number.isSafeInteger(defaultBatchStride)) throw new Error(sideName + ".batchStrideElements exceeds safe integer range");
  const batchStrideElements = dflt(side.batchStrideElements, defaultBatchStride);
  assertPositiveSafeInt(batchStrideElements, sideName + ".batchStrideElements");
  if (batchStrideElements < defaultBatchStride) {
    throw new Error(sideName + ".batchStrideElements must be >= channels*channelStrideElements (" + defaultBatchStride + ")");
  }
  const offsetElements = dflt(side.offsetElements, 0);
  assertNonNegativeSafeInt(offsetElements, sideName + ".offsetElements");
  const desc = { channels, channelIndex, channelStrideElements, batchStrideElements, offsetElements };
  if (allowKernelStep) {
    const kernelStepChannels = dflt(side.kernelStepChannels, 1);
    assertPositiveSafeInt(kernelStepChannels, sideName + ".kernelStepChannels");
    if (kernelCount > 1) {
      const maxChannelIndex = channelIndex + (kernelCount - 1) * kernelStepChannels;
      if (!Number.isSafeInteger(maxChannelIndex)) throw new Error(sideName + ".kernelStepChannels mapping exceeds safe integer range");
      if (maxChannelIndex >= channels) {
        throw new Error(sideName + " does not fit kernelCount=" + kernelCount + ": max channel index " + maxChannelIndex + " exceeds channels=" + channels +
          " (channelIndex=" + channelIndex + ", kernelStepChannels=" + kernelStepChannels + ")");
      }
    }
    desc.kernelStepChannels = kernelStepChannels;
  }
  return desc;
}

function buildPreset(opts, forcedOutputLayout) {
  if (!isPlainObject(opts)) throw new Error("opts must be an object");
  const shape = opts.shape;
  if (!Array.isArray(shape) || shape.length === 0) throw new Error("shape must be a non-empty array");
  let span = 1;
  for (let i = 0; i < shape.length; i++) {
    assertPositiveSafeInt(shape[i], "shape[" + i + "]");
    span *= shape[i];
    if (!Number.isSafeInteger(span)) throw new Error("shape product exceeds safe integer range");
  }
  const batch = opts.batch;
  const kernelCount = dflt(opts.kernelCount, 1);
  const mode = dflt(opts.mode, "convolution");
  const boundary = dflt(opts.boundary, "circular");
  const outputLayout = dflt(opts.outputLayout, "kernel-major");
  const layout = dflt(opts.layout, {});
  assertPositiveSafeInt(batch, "batch");
  assertPositiveSafeInt(kernelCount, "kernelCount");
  assertOneOf(mode, FFTCONV_MODES, "mode");
  assertOneOf(boundary, FFTCONV_BOUNDARIES, "boundary");
  assertOneOf(outputLayout, FFTCONV_OUTPUT_LAYOUTS, "outputLayout");
  if (!isPlainObject(layout)) throw new Error("layout must be an object");
  if (hasOwn(layout, "interleavedComplex") && layout.interleavedComplex !== true) {
    throw new Error("layout.interleavedComplex must be true for fftconv channel-lane presets");
  }
  for (const key of CONFLICTING_LAYOUT_KEYS) {
    if (hasOwn(layout, key)) throw new Error("layout." + key + " cannot be combined with fftConv.channelPolicy presets");
  }
  const finalOutputLayout = forcedOutputLayout != null ? forcedOutputLayout : outputLayout;
  const inputDesc = normalizeSideDescriptor(opts.input, "input", span, kernelCount, false);
  const outputDesc = normalizeSideDescriptor(opts.output, "output", span, kernelCount, true);
  return {
    shape: shape.slice(),
    batch,
    layout: Object.assign({ interleavedComplex: true }, layout),
    fftConv: { mode, boundary, kernelCount, outputLayout: finalOutputLayout, channelPolicy: { input: inputDesc, output: outputDesc } },
  };
}
export function createFftConvChannelLanePreset(opts) { return buildPreset(opts, null); }
export function createFftConvKernelMajorChannelLanePreset(opts) { return buildPreset(opts, "kernel-major"); }
export function createFftConvBatchMajorChannelLanePreset(opts) { return buildPreset(opts, "batch-major"); }

// ---- layout resolution (layout_semantics.js) ----------------------------------------------------------
function contiguousStrides(shape) { const out = []; let acc = 1; for (const s of shape) { out.push(acc); acc *= s; } return out; }
function stridedSpan(shape, strides) { let s = 1; for (let d = 0; d < shape.length; d++) s += (shape[d] - 1) * strides[d]; return s; }
function sideField(side, suffix) { return side + suffix; }
function hasExplicitSideLayout(layout, side) {
  return hasOwn(layout, sideField(side, "Strides")) || hasOwn(layout, sideField(side, "OffsetElements")) ||
    hasOwn(layout, sideField(side, "BatchStrideElements")) || hasOwn(layout, "strides") || hasOwn(layout, "offsetElements") ||
    hasOwn(layout, "batchStrideElements");
}
function optPosIntArray(v, rank, name) {
  if (v == null) return null;
  if (!Array.isArray(v) || v.length !== rank || !v.every((x) => Number.isInteger(x) && x > 0)) throw new Error(name + " must be an array of " + rank + " positive ints");
  return v.slice();
}
function optNonNeg(v, name) {
  if (v == null) return null;
  if (!Number.isInteger(v) || v < 0) throw new Error(name + " must be a non-negative integer");
  return v;
}
function optPos(v, name) {
  if (v == null) return null;
  if (!Number.isInteger(v) || v <= 0) throw new Error(name + " must be a positive integer");
  return v;
}
function firstDefined(a, b) { return a !== undefined ? a : b; }
function arraysEqual(a, b) { return a.length === b.length && a.every((v, i) => v === b[i]); }

function resolveExplicitSide(layout, side, rank, shape) {
  let strides = optPosIntArray(firstDefined(layout[sideField(side, "Strides")], layout.strides), rank, "layout." + side + "Strides");
  const offset = optNonNeg(firstDefined(layout[sideField(side, "OffsetElements")], layout.offsetElements), "layout." + side + "OffsetElements");
  const bstride = optNonNeg(firstDefined(layout[sideField(side, "BatchStrideElements")], layout.batchStrideElements), "layout." + side + "BatchStrideElements");
  if (strides == null && offset == null && bstride == null) return null;
  strides = strides || contiguousStrides(shape);
  const span = stridedSpan(shape, strides);
  return { strides, offset: offset || 0, batchStride: bstride ? bstride : Math.max(span, prod(shape)) };
}

function resolveWhdcnSide(desc, side, rank, shape) {
  if (!desc || desc.enabled === false) return null;
  const keys = ["strides", "offsetElements", "batchStrideElements", "channels", "channelIndex", "channelStrideElements"];
  if (!keys.some((k) => hasOwn(desc, k))) return null;
  const path = "layout.whdcn." + side;
  const strides = optPosIntArray(desc.strides, rank, path + ".strides") || contiguousStrides(shape);
  const span = stridedSpan(shape, strides);
  const channels = optPos(desc.channels, path + ".channels") || 1;
  const cidx = optNonNeg(desc.channelIndex, path + ".channelIndex") || 0;
  if (cidx >= channels) throw new Error(path + ".channelIndex (" + cidx + ") must be < " + path + ".channels (" + channels + ")");
  const cstride = optPos(desc.channelStrideElements, path + ".channelStrideElements") || span;
  if (cstride < span) throw new Error(path + ".channelStrideElements must be >= addressed span (" + span + ")");
  const baseOff = optNonNeg(desc.offsetElements, path + ".offsetElements") || 0;
  const offset = baseOff + cidx * cstride;
  const defaultB = cstride * channels;
  let bstride = optNonNeg(desc.batchStrideElements, path + ".batchStrideElements");
  if (bstride == null) bstride = defaultB;
  if (bstride < defaultB) throw new Error(path + ".batchStrideElements must be >= channels*channelStrideElements (" + defaultB + ")");
  if (arraysEqual(strides, contiguousStrides(shape)) && offset === 0 && bstride === prod(shape) && channels === 1 && cidx === 0 && cstride === span) return null;
  return { strides, offset, batchStride: bstride };
}

export function resolveLayoutSemantics(layout, rank, inputShape, outputShape) {
  const l = layout || {};
  if (!isPlainObject(l)) throw new Error("layout must be an object");
  let input = resolveExplicitSide(l, "input", rank, inputShape);
  let output = resolveExplicitSide(l, "output", rank, outputShape);
  if (l.whdcn != null) {
    if (!isPlainObject(l.whdcn)) throw new Error("layout.whdcn must be an object");
    const glob = Object.assign({}, l.whdcn);
    delete glob.input;
    delete glob.output;
    for (const [side, shape] of [["input", inputShape], ["output", outputShape]]) {
      if (hasExplicitSideLayout(l, side)) continue;
      const sd = l.whdcn[side];
      if (sd != null && !isPlainObject(sd)) throw new Error("layout.whdcn." + side + " must be an object");
      const r = resolveWhdcnSide(Object.assign({}, glob, sd || {}), side, rank, shape);
      if (r) { if (side === "input") input = r; else output = r; }
    }
  }
  return { input, output };
}

function resolveChannelPolicy(layout, policy, kernelCount, inTotal, outTotal) {
  if (policy == null) return { layout: layout || {}, kernelStride: 0 };
  if (!isPlainObject(policy)) throw new Error("fftConv.channelPolicy must be an object");
  const hasIn = policy.input != null, hasOut = policy.output != null;
  if (!hasIn && !hasOut) throw new Error("fftConv.channelPolicy must provide input and/or output descriptors");
  const lay = layout || {};
  if (lay.whdcn != null) throw new Error("fftConv.channelPolicy cannot be combined with layout.whdcn");
  if (hasIn && hasExplicitSideLayout(lay, "input")) throw new Error("fftConv.channelPolicy.input cannot be combined with explicit input stride fields");
  if (hasOut && hasExplicitSideLayout(lay, "output")) throw new Error("fftConv.channelPolicy.output cannot be combined with explicit output stride fields");
  const side = (desc, path, span, allowStep) => {
    if (!isPlainObject(desc)) throw new Error(path + " must be an object");
    const channels = optPos(desc.channels, path + ".channels");
    if (channels == null) throw new Error(path + ".channels is required");
    const cidx = optNonNeg(desc.channelIndex, path + ".channelIndex") || 0;
    if (cidx >= channels) throw new Error(path + ".channelIndex (" + cidx + ") must be < " + path + ".channels (" + channels + ")");
    const cstride = optPos(desc.channelStrideElements, path + ".channelStrideElements") || span;
    if (cstride < span) throw new Error(path + ".channelStrideElements must be >= logical span (" + span + ")");
    const off = optNonNeg(desc.offsetElements, path + ".offsetElements") || 0;
    const defaultB = channels * cstride;
    let bstride = optNonNeg(desc.batchStrideElements, path + ".batchStrideElements");
    if (bstride == null) bstride = defaultB;
    if (bstride < defaultB) throw new Error(path + ".batchStrideElements must be >= channels*channelStrideElements (" + defaultB + ")");
    const step = allowStep ? (optPos(desc.kernelStepChannels, path + ".kernelStepChannels") || 1) : 1;
    if (allowStep && kernelCount > 1) {
      const maxIdx = cidx + (kernelCount - 1) * step;
      if (maxIdx >= channels) {
        throw new Error(path + " does not fit kernelCount=" + kernelCount + ": max channel index " + maxIdx + " exceeds channels=" + channels +
          " (channelIndex=" + cidx + ", kernelStepChannels=" + step + ")");
      }
    }
    return { desc: { channels, channelIndex: cidx, channelStrideElements: cstride, batchStrideElements: bstride, offsetElements: off }, kstride: cstride * step };
  };
  const wh = {};
  let kernelStride = 0;
  if (hasIn) wh.input = side(policy.input, "fftConv.channelPolicy.input", inTotal, false).desc;
  if (hasOut) {
    const r = side(policy.output, "fftConv.channelPolicy.output", outTotal, true);
    wh.output = r.desc;
    if (kernelCount > 1) kernelStride = r.kstride;
  }
  return { layout: Object.assign({}, lay, { whdcn: wh }), kernelStride };
}

const isPosInt = (x) => Number.isInteger(x) && x > 0;

// runtime/ioview.js:7-37; a side that maps 1:1 onto the logical domain resolves to null
export function normalizeIoView(rank, logicalShape, ioView) {
  const one = (v, kind) => {
    if (!v) return null;
    const shape = v.shape;
    if (!Array.isArray(shape) || shape.length !== rank || !shape.every(isPosInt)) throw new Error("ioView." + kind + ".shape must be an array of " + rank + " positive ints");
    const placement = dflt(v.placement, "start");
    if (placement !== "start" && placement !== "center") throw new Error("ioView." + kind + '.placement must be "start"|"center"');
    let offset = v.offset;
    if (offset != null) {
      if (!Array.isArray(offset) || offset.length !== rank || !offset.every((x) => Number.isInteger(x))) throw new Error("ioView." + kind + ".offset must be an array of " + rank + " integers");
      offset = offset.slice();
    } else if (placement === "center") offset = shape.map((s, d) => Math.floor((logicalShape[d] - s) / 2));
    else offset = new Array(rank).fill(0);
    if (arraysEqual(shape, logicalShape) && offset.every((o) => o === 0)) return null;
    return { shape: shape.slice(), offset, clearOutside: kind === "output" ? !!v.clearOutside : false };
  };
  const iv = ioView || {};
  return { input: one(iv.input, "input"), output: one(iv.output, "output") };
}

// runtime/zero_pad.js:11-45; a stage covering the whole domain resolves to null
export function normalizeZeroPad(rank, shape, zeroPad, name) {
  const nm0 = name || "zeroPad";
  if (!zeroPad) return { read: null, write: null };
  if (typeof zeroPad !== "object") throw new Error(nm0 + " must be an object with optional read/write stage configs");
  const stage = (st, nm) => {
    if (!st) return null;
    if (typeof st !== "object") throw new Error(nm + " must be an object with optional start/end arrays");
    const src = st.range && typeof st.range === "object" ? st.range : st;
    const bound = (v, which, d) => {
      if (v == null) return d.slice();
      if (!Array.isArray(v) || v.length !== rank || !v.every((x) => Number.isInteger(x))) throw new Error(nm + "." + which + " must be an array of " + rank + " integers");
      return v.slice();
    };
    const start = bound(src.start, "start", new Array(rank).fill(0)), end = bound(src.end, "end", shape);
    for (let d = 0; d < rank; d++) {
      if (start[d] < 0) throw new Error(nm + ".start[" + d + "] must be >= 0; got " + start[d]);
      if (end[d] < 0) throw new Error(nm + ".end[" + d + "] must be >= 0; got " + end[d]);
      if (start[d] > end[d]) throw new Error(nm + ": start[" + d + "] must be <= end[" + d + "]");
      if (end[d] > shape[d]) throw new Error(nm + ".end[" + d + "] must be <= shape[" + d + "] (" + shape[d] + "); got " + end[d]);
    }
    if (start.every((s0) => s0 === 0) && end.every((e, d) => e === shape[d])) return null;
    return { start, end };
  };
  return { read: stage(zeroPad.read, nm0 + ".read"), write: stage(zeroPad.write, nm0 + ".write") };
}

// Validates createPlan opts; returns { desc (native planCreate argument), meta }
export function resolvePlanOptions(opts) {
  if (!isPlainObject(opts)) throw new Error("createPlan expects an options object");
  const type = opts.type;
  assertOneOf(type, ALL_TYPES, "type");
  const trig = TRIG_TYPES.includes(type);
  if (!HOT_PATH_TYPES.includes(type) && !trig) {
    throw new Error('Unsupported: type "' + type + '" is outside the MI355X hot path (c2c/r2c/c2r/fftconv); see DESIGN.md "out of scope"');
  }
  const shapeIn = opts.shape;
  if (!Array.isArray(shapeIn) || shapeIn.length < 1) {
    throw new Error("shape must be an array of one or more positive dimensions; got " + JSON.stringify(shapeIn));
  }
  if (!shapeIn.every(isPosInt)) throw new Error("shape elements must be positive ints; got " + JSON.stringify(shapeIn));
  const shape = shapeIn.slice();
  const rank = shape.length;
  const batch = dflt(opts.batch, 1);
  if (!Number.isInteger(batch) || batch <= 0) throw new Error("batch must be positive int; got " + batch);
  const layout = dflt(opts.layout, { interleavedComplex: !trig });
  if (trig) {
    if (!isPlainObject(layout) || layout.interleavedComplex !== false) throw new Error("DCT/DST uses real buffers; set layout.interleavedComplex=false");
    if (shape.some((n) => n < 2)) throw new Error("All DCT/DST dimensions must be >= 2; got shape=" + JSON.stringify(shape));
    if (opts.inPlace) throw new Error("DCT/DST inPlace is not supported in current implementation");
  } else if (!isPlainObject(layout) || layout.interleavedComplex !== true) throw new Error(type + " requires layout.interleavedComplex=true");
  const precision = dflt(opts.precision, "f32");
  assertOneOf(precision, ["f32", "f16-storage"], "precision");
  if (precision !== "f32") throw new Error('Unsupported: precision "f16-storage" is outside the MI355X hot path (f32 only)');
  // logical domains of the two sides: r2c writes / c2r reads the PACKED spectrum (r2c.js:72-123, c2r.js:168-220)
  const packedShape = [Math.floor(shape[0] / 2) + 1].concat(shape.slice(1));
  const inLogical = type === "c2r" ? packedShape : shape;
  const outLogical = type === "r2c" ? packedShape : shape;
  const ivIn = opts.ioView || {};
  const zpIn = opts.zeroPad;
  if (zpIn !== undefined && zpIn !== null && typeof zpIn !== "object") throw new Error("zeroPad must be an object with optional read/write stage configs");
  const ioView = { input: normalizeIoView(rank, inLogical, { input: ivIn.input }).input,
                   output: normalizeIoView(rank, outLogical, { output: ivIn.output }).output };
  if (type === "fftconv" && (ioView.input || ioView.output)) throw new Error("ioView is not an fftconv option (fftconv.js:308-320)");
  // fftconv: resolved below against the FFT domain (fftconv.js:353,386)
  const zeroPad = type === "fftconv" ? { read: null, write: null }
    : { read: normalizeZeroPad(rank, inLogical, { read: (zpIn || {}).read }).read,
        write: normalizeZeroPad(rank, outLogical, { write: (zpIn || {}).write }).write };
  const inPlace = !!opts.inPlace;
  const meta = { type, shape, rank, batch, inPlace, ioView, zeroPad };
  const desc = { type: TYPE_CODE[type], shape, batch, inPlace: inPlace ? 1 : 0, direction: 0, normalize: 0 };
  if (ioView.input) desc.ioInput = ioView.input;
  if (ioView.output) desc.ioOutput = Object.assign({}, ioView.output, { clearOutside: ioView.output.clearOutside ? 1 : 0 });
  if (zeroPad.read) desc.zeroRead = zeroPad.read;
  if (zeroPad.write) desc.zeroWrite = zeroPad.write;
  if (opts.axes !== undefined && opts.axes !== null) {      // createFftPlan({axes}) forwards its subset here (c2c only)
    if (type !== "c2c") throw new Error("axes is a createFftPlan (c2c) option");
    let mask = 0;
    for (const axis of opts.axes) {
      if (!Number.isInteger(axis) || axis < 0 || axis >= rank) throw new Error("Invalid axis " + axis + " for rank " + rank);
      mask |= 1 << axis;
    }
    desc.axesMask = mask;
    meta.axes = opts.axes.slice();
  }

  if (type === "fftconv") {
    const fc = opts.fftConv || {};
    const mode = dflt(fc.mode, "convolution");
    assertOneOf(mode, FFTCONV_MODES, "fftConv.mode");
    const boundary = dflt(fc.boundary, "circular");
    assertOneOf(boundary, FFTCONV_BOUNDARIES, "fftConv.boundary");
    const kernelCount = dflt(fc.kernelCount, 1);
    if (!Number.isInteger(kernelCount) || kernelCount <= 0) throw new Error("fftConv.kernelCount must be a positive integer; got " + kernelCount);
    const outputLayout = dflt(fc.outputLayout, "kernel-major");
    assertOneOf(outputLayout, FFTCONV_OUTPUT_LAYOUTS, "fftConv.outputLayout");
    const kernelShape = dflt(fc.kernelShape, shape);
    if (!Array.isArray(kernelShape) || kernelShape.length !== rank || !kernelShape.every(isPosInt)) {
      throw new Error("fftConv.kernelShape must be an array of " + rank + " positive ints");
    }
    let outputShape;
    if (boundary === "circular") {
      for (let d = 0; d < rank; d++) {
        if (kernelShape[d] > shape[d]) throw new Error("fftConv.kernelShape[" + d + "] must be <= shape[" + d + '] when fftConv.boundary="circular"');
      }
      outputShape = shape.slice();
    } else if (boundary === "linear-full") outputShape = shape.map((n, d) => n + kernelShape[d] - 1);
    else if (boundary === "linear-same") outputShape = shape.slice();
    else {
      outputShape = shape.map((n, d) => n - kernelShape[d] + 1);
      for (let d = 0; d < rank; d++) {
        if (outputShape[d] <= 0) throw new Error('fftConv.boundary="linear-valid" requires kernelShape[' + d + "] <= shape[" + d + "]");
      }
    }
    const fftShape = boundary === "circular" ? shape.slice() : shape.map((n, d) => n + kernelShape[d] - 1);
    const zpConv = normalizeZeroPad(rank, fftShape, zpIn);
    if (zpConv.read) desc.zeroRead = zpConv.read;
    if (zpConv.write) desc.zeroWrite = zpConv.write;
    meta.zeroPad = zpConv;
    if (inPlace) throw new Error("fftconv inPlace=true is not supported in current implementation");
    const explicitKStride = optPos(fc.outputKernelStrideElements, "fftConv.outputKernelStrideElements") || 0;
    const pol = resolveChannelPolicy(layout, fc.channelPolicy, kernelCount, prod(shape), prod(outputShape));
    if (explicitKStride && pol.kernelStride && explicitKStride !== pol.kernelStride) {
      throw new Error("fftConv.outputKernelStrideElements conflicts with fftConv.channelPolicy.output kernel step mapping");
    }
    const sides = resolveLayoutSemantics(pol.layout, rank, shape, outputShape);
    Object.assign(desc, {
      convMode: CONV_MODE_CODE[mode], convBoundary: CONV_BOUNDARY_CODE[boundary], convKernelCount: kernelCount,
      convOutputLayout: CONV_LAYOUT_CODE[outputLayout], convKernelShape: kernelShape.slice(),
      convOutputKernelStrideElements: explicitKStride || pol.kernelStride || 0,
    });
    if (sides.input) desc.input = sides.input;
    if (sides.output) desc.output = sides.output;
    Object.assign(meta, { mode, boundary, kernelCount, outputLayout, kernelShape: kernelShape.slice(), outputShape,
      inputLayout: sides.input, outputLayoutResolved: sides.output, outputKernelStrideElements: desc.convOutputKernelStrideElements });
    return { desc, meta };
  }

  const direction = opts.direction === undefined && trig ? "forward" : opts.direction;   // dct_fft.js:70: DCT/DST default to forward
  assertOneOf(direction, ["forward", "inverse"], "direction");
  const normalize = dflt(opts.normalize, "none");
  assertOneOf(normalize, ["none", "backward", "unitary"], "normalize");
  if (type === "r2c" && direction !== "forward") throw new Error('r2c supports direction:"forward" only');
  if (type === "c2r" && direction !== "inverse") throw new Error('c2r supports direction:"inverse" only');
  if (inPlace && type !== "c2c") throw new Error("inPlace=true is supported only on c2c");
  const packed = [Math.floor(shape[0] / 2) + 1].concat(shape.slice(1));
  const inShape = ioView.input ? ioView.input.shape : (type === "c2r" ? packed : shape);
  const outShape = ioView.output ? ioView.output.shape : (type === "r2c" ? packed : shape);
  const sides = resolveLayoutSemantics(layout, rank, inShape, outShape);
  desc.direction = DIRECTION_CODE[direction];
  desc.normalize = NORMALIZE_CODE[normalize];
  if (sides.input) desc.input = sides.input;
  if (sides.output) desc.output = sides.output;
  Object.assign(meta, { direction, normalize, inputLayout: sides.input, outputLayoutResolved: sides.output });
  return { desc, meta };
}

// runtime/common.js:35-40
export function normalizeScaleFactor({ normalize, direction, nTotal }) {
  if (normalize === "none") return 1.0;
  if (normalize === "unitary") return 1.0 / Math.sqrt(nTotal);
  if (normalize === "backward") return direction === "inverse" ? 1.0 / nTotal : 1.0;
  throw new Error("Unknown normalize mode: " + normalize);
}

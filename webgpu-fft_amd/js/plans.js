// plans.js — createPlan / createFftPlan over the native planner + HIP kernels.
//
// Replaces src/plan.js (N-API dispatch instead of WGSL pipeline creation) and the exec half of
// src/runtime/plans/{c2c,r2c,c2r,fftconv}.js.  Everything numeric happens in libmi355fft.so; this file only
// validates options the way the reference does, resolves layouts (layout.js) and forwards handles.
import native from "./native.js";
import { HipBuffer } from "./device.js";
import { BufferView } from "./buffer_view.js";
import { resolvePlanOptions, prod } from "./layout.js";

function unwrapBuffer(x, what) {
  // GPUBuffer, or a single-segment BufferView (multi-segment views exist in the reference only because WebGPU
  // buffers are small; one hipMalloc spans 288 GB — SURVEY.md section 2)
  if (x instanceof HipBuffer) return { buf: x, offset: 0 };
  if (x instanceof BufferView) {
    if (x.segments.length !== 1) throw new Error("Unsupported: " + what + " BufferView with more than one segment");
    if (!(x.segments[0].buffer instanceof HipBuffer)) throw new Error(what + " BufferView segment must be a buffer created by this device");
    if (x.logicalByteOffset + x.lengthBytes > x.segments[0].sizeBytes) throw new Error(what + " BufferView range exceeds its segment");
    return { buf: x.segments[0].buffer, offset: x.segments[0].offsetBytes + x.logicalByteOffset };
  }
  throw new Error(what + " must be a buffer created by this device (or a single-segment BufferView)");
}

class NativePlanBase {
  constructor(device) {
    if (!device || !device._h) throw new Error("Expected a WebGPU device");
    this.device = device;
    this._destroyed = false;
  }
  getPipelineCacheSnapshot() {
    // kernels are ahead-of-time gfx950 code objects: there is no shader cache to export (pipeline_cache.js)
    return { schema: "webgpufft.pipeline-cache", version: 2, createdAtMs: Date.now(), metadata: { backend: "mi355-hip" }, shaderCodes: [], pipelineKeys: [] };
  }
}

export class Plan extends NativePlanBase {
  constructor(device, opts) {
    super(device);
    const { desc, meta } = resolvePlanOptions(opts);
    Object.assign(this, meta);
    // route metadata the reference exposes on its plans and its tests read (c2c.js:661-666, complete.suite.js:4790-4808)
    this._usesStridedInput = !!meta.inputLayout;
    this._usesStridedOutput = !!meta.outputLayoutResolved;
    this._inputStrides = meta.inputLayout ? meta.inputLayout.strides : null;
    this._outputStrides = meta.outputLayoutResolved ? meta.outputLayoutResolved.strides : null;
    this._inputOffsetElements = meta.inputLayout ? meta.inputLayout.offset : 0;
    this._outputOffsetElements = meta.outputLayoutResolved ? meta.outputLayoutResolved.offset : 0;
    this._inputBatchStrideElements = meta.inputLayout ? meta.inputLayout.batchStride : prod(meta.shape);
    this._outputBatchStrideElements = meta.outputLayoutResolved ? meta.outputLayoutResolved.batchStride : prod(meta.outputShape || meta.shape);
    this._stridedOutputKernelStrideElements = meta.outputKernelStrideElements || 0;
    this._h = native.planCreate(device._h, desc);
    const d = native.planDescribe(this._h);
    this._route = d.route;
    this._launchesPerExec = d.launches;
    this._kernelUpload = null;
  }

  getWorkspaceSizeBytes() {
    return native.planWorkspaceBytes(this._h);
  }

  // commandEncoder: kernels given as an array of GPU buffers / BufferViews ("array-sources", fftconv.js:920-941) are packed by
  // one copyBufferToBuffer per kernel recorded ahead of the plan's own launches
  _prepareKernel(kernel, commandEncoder) {
    const single = 2 * prod(this.kernelShape);
    const packed = single * this.kernelCount;
    let payload = null;
    if (kernel instanceof Float32Array) {
      if (this.kernelCount === 1) {
        if (kernel.length !== single && kernel.length !== packed) throw new Error("kernel Float32Array length must be " + single + "; got " + kernel.length);
      } else if (kernel.length !== packed) {
        throw new Error("kernel Float32Array length must be " + packed + " for kernelCount=" + this.kernelCount + "; got " + kernel.length);
      }
      payload = kernel;
    } else if (Array.isArray(kernel)) {
      if (kernel.length !== this.kernelCount) throw new Error("kernel array length must equal fftConv.kernelCount=" + this.kernelCount + "; got " + kernel.length);
      if (kernel.every((k) => k instanceof Float32Array)) {
        payload = new Float32Array(packed);
        for (let i = 0; i < this.kernelCount; i++) {
          if (kernel[i].length !== single) throw new Error("kernel[" + i + "] Float32Array length must be " + single + "; got " + kernel[i].length);
          payload.set(kernel[i], i * single);
        }
      } else if (kernel.some((k) => k instanceof Float32Array)) {
        throw new Error("kernel array items must be all Float32Array or all GPUBuffer/BufferView values");
      } else {
        const bytesEach = single * 4;
        const srcs = kernel.map((k, i) => {
          const u = unwrapBuffer(k, "kernel[" + i + "]");
          const have = (k instanceof BufferView ? k.lengthBytes : u.buf.size - u.offset);
          if (have < bytesEach) throw new Error("kernel[" + i + "] is too small: need " + bytesEach + " bytes, have " + have);
          return u;
        });
        this._ensureKernelUpload(packed * 4);
        for (let i = 0; i < this.kernelCount; i++) {
          native.encoderCopyBuffer(commandEncoder._h, srcs[i].buf._h, srcs[i].offset, this._kernelUpload._h, i * bytesEach, bytesEach);
        }
        return { buf: this._kernelUpload, offset: 0 };
      }
    } else {
      return unwrapBuffer(kernel, "kernel");
    }
    this._ensureKernelUpload(packed * 4);
    this.device.queue.writeBuffer(this._kernelUpload, 0, payload);
    return { buf: this._kernelUpload, offset: 0 };
  }

  _ensureKernelUpload(bytes) {
    if (!this._kernelUpload || this._kernelUpload.size < bytes) {
      if (this._kernelUpload) this._kernelUpload.destroy();
      this._kernelUpload = this.device.createBuffer({ size: bytes, usage: 0 });
    }
  }

  exec(commandEncoder, execOpts) {
    if (this._destroyed) throw new Error("plan destroyed");
    const o = execOpts || {};
    if (!o.input) throw new Error("exec requires input");
    if (!this.inPlace && !o.output) throw new Error("exec requires output when inPlace=false");
    if (this.inPlace && o.output && o.output !== o.input) throw new Error("inPlace=true requires output omitted or equal to input");
    if (!commandEncoder || !commandEncoder._h) throw new Error("exec requires a command encoder from this device");
    const inp = unwrapBuffer(o.input, "input");
    const args = { input: inp.buf._h, inputOffsetBytes: inp.offset + (o.inputOffsetBytes || 0) };
    if (o.output && !(this.inPlace && o.output === o.input)) {
      const out = unwrapBuffer(o.output, "output");
      args.output = out.buf._h;
      args.outputOffsetBytes = out.offset + (o.outputOffsetBytes || 0);
    }
    if (o.temp) {
      const t = unwrapBuffer(o.temp, "temp");
      if (t.offset === 0) args.temp = t.buf._h;
    }
    if (this.type === "fftconv") {
      if (o.kernel == null) throw new Error("fftconv exec requires kernel");
      const k = this._prepareKernel(o.kernel, commandEncoder);
      args.kernel = k.buf._h;
      args.kernelOffsetBytes = k.offset;
    }
    native.planExec(this._h, commandEncoder._h, args);
  }

  destroy() {
    if (this._destroyed) return;       // idempotent (base_plan.js:49-53)
    this._destroyed = true;
    native.planDestroy(this._h);
    if (this._kernelUpload) { this._kernelUpload.destroy(); this._kernelUpload = null; }
  }
}

// runtime/create_plan.js:12-23
export function createPlan(device, opts) {
  return new Plan(device, opts);
}

// Low-level FFT core (src/plan.js:1298-1512): batch is an exec argument here, a ctor argument on createPlan.
export class FftPlan extends NativePlanBase {
  constructor(device, opts) {
    super(device);
    const o = opts || {};
    const { shape, direction } = o;
    if (!Array.isArray(shape) || shape.length < 1) throw new Error("shape must be an array of one or more dimensions; got " + JSON.stringify(shape));
    if (!shape.every((n) => Number.isInteger(n) && n >= 2)) throw new Error("every shape dimension must be an integer >= 2; got " + JSON.stringify(shape));
    if (direction !== "forward" && direction !== "inverse") throw new Error('direction must be "forward" or "inverse"; got ' + JSON.stringify(direction));
    const normalize = o.normalize === undefined ? "none" : o.normalize;
    if (!["none", "backward", "unitary"].includes(normalize)) throw new Error("normalize must be none|backward|unitary; got " + JSON.stringify(normalize));
    // axes: null => every axis; otherwise the listed axes only (plan.js:1335-1339; the scale factor keeps prod(shape))
    this.axes = null;
    if (o.axes !== undefined && o.axes !== null) {
      if (!Array.isArray(o.axes) || o.axes.length === 0) throw new Error("axes must be null or a non-empty array");
      for (const axis of o.axes) if (!Number.isInteger(axis) || axis < 0 || axis >= shape.length) throw new Error("Invalid axis " + axis + " for rank " + shape.length);
      this.axes = o.axes.slice();
    }
    this.shape = shape.slice();
    this.direction = direction;
    this.normalize = normalize;
    this.inPlace = !!o.inPlace;
    this._plans = new Map();   // batch -> native plan
  }
  _planFor(batch) {
    let p = this._plans.get(batch);
    if (!p) {
      p = new Plan(this.device, { type: "c2c", shape: this.shape, batch, direction: this.direction, normalize: this.normalize, inPlace: this.inPlace, axes: this.axes });
      this._plans.set(batch, p);
    }
    return p;
  }
  getWorkspaceSizeBytes(batch) {
    return this._planFor(batch === undefined ? 1 : batch).getWorkspaceSizeBytes();
  }
  exec(commandEncoder, execOpts) {
    if (this._destroyed) throw new Error("FftPlan is destroyed");
    const o = execOpts || {};
    const batch = o.batch === undefined ? 1 : o.batch;
    if (!Number.isInteger(batch) || batch <= 0) throw new Error("batch must be a positive integer; got " + batch);
    const io = (o.inputOffsetBytes || 0), oo = (o.outputOffsetBytes || 0);
    if (io % 8 !== 0 || oo % 8 !== 0) throw new Error("inputOffsetBytes/outputOffsetBytes must be multiples of 8");
    if (!o.input) throw new Error("exec requires input");
    if (this.inPlace) {
      if (o.output && o.output !== o.input) throw new Error("inPlace=true requires output omitted or equal to input");
    } else {
      if (!o.output) throw new Error("exec requires output when inPlace=false");
      if (o.output === o.input) throw new Error("inPlace=false requires input !== output");
    }
    this._planFor(batch).exec(commandEncoder, { input: o.input, output: this.inPlace ? undefined : o.output, temp: o.temp, inputOffsetBytes: io, outputOffsetBytes: oo });
  }
  destroy() {
    if (this._destroyed) return;
    this._destroyed = true;
    for (const p of this._plans.values()) p.destroy();
    this._plans.clear();
  }
}

export function createFftPlan(device, opts) {
  return new FftPlan(device, opts);
}

// pipeline_cache.js:201-221 — API-surface shims: nothing to cache for AOT kernels
export function exportPipelineCacheSnapshot(device) {
  void device;
  return { schema: "webgpufft.pipeline-cache", version: 2, createdAtMs: Date.now(), metadata: { backend: "mi355-hip" }, shaderCodes: [], pipelineKeys: [] };
}
export function importPipelineCacheSnapshot(device, snapshot) {
  void device;
  if (!snapshot || typeof snapshot !== "object") throw new Error("importPipelineCacheSnapshot expects a snapshot object");
  return exportPipelineCacheSnapshot(device);
}

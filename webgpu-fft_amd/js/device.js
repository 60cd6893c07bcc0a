// device.js — the HIP-backed stand-in for the WebGPU objects the reference's API takes and returns:
// GPUDevice (createBuffer / createCommandEncoder / queue / limits / features), GPUBuffer ({size, destroy}),
// GPUCommandEncoder (copyBufferToBuffer / finish) and GPUQueue (writeBuffer / submit / onSubmittedWorkDone).
// Only the caller-visible subset exists (SURVEY.md 8b "The device argument"); pipelines, bind groups and
// shader modules have no counterpart — kernels are ahead-of-time gfx950 code objects inside libmi355fft.so.
import native from "./native.js";

export const GPUBufferUsage = Object.freeze({
  MAP_READ: 0x0001, MAP_WRITE: 0x0002, COPY_SRC: 0x0004, COPY_DST: 0x0008, INDEX: 0x0010, VERTEX: 0x0020,
  UNIFORM: 0x0040, STORAGE: 0x0080, INDIRECT: 0x0100, QUERY_RESOLVE: 0x0200,
});
export const GPUMapMode = Object.freeze({ READ: 0x0001, WRITE: 0x0002 });

// the reference's tests define these globals when the runtime lacks them (c2c_large_batch.unit.test.js:12-25)
if (typeof globalThis.GPUBufferUsage === "undefined") globalThis.GPUBufferUsage = GPUBufferUsage;
if (typeof globalThis.GPUMapMode === "undefined") globalThis.GPUMapMode = GPUMapMode;

export class HipBuffer {
  constructor(device, handle, size, usage) {
    this.device = device;
    this._h = handle;
    this.size = size;          // numeric .size + .destroy(): how the reference recognises a buffer (common.js:55-57)
    this.usage = usage | 0;
    this._mapped = null;
  }
  destroy() {
    if (this._h) {
      native.bufferFree(this._h);
      this._h = null;
    }
  }
  // MAP_READ staging-buffer protocol of downloadComplex (utils/webgpu.js:43-52)
  mapAsync(mode, offset, size) {
    const off = offset === undefined ? 0 : offset;
    const len = size === undefined ? this.size - off : size;
    this._assertAlive();
    return native.bufferReadAsync(this._h, off, len).then((ab) => {
      this._mapped = { offset: off, data: ab };
    });
  }
  getMappedRange(offset, size) {
    if (!this._mapped) throw new Error("getMappedRange: buffer is not mapped");
    const rel = (offset === undefined ? this._mapped.offset : offset) - this._mapped.offset;
    const len = size === undefined ? this._mapped.data.byteLength - rel : size;
    return this._mapped.data.slice(rel, rel + len);
  }
  unmap() {
    this._mapped = null;
  }
  _assertAlive() {
    if (!this._h) throw new Error("buffer destroyed");
  }
}

export class HipCommandBuffer {
  constructor(device, handle) {
    this.device = device;
    this._h = handle;
  }
  // command lists may be re-submitted (bench loops); release() frees the hipGraph / op list
  release() {
    if (this._h) {
      native.commandsRelease(this._h);
      this._h = null;
    }
  }
}

export class HipCommandEncoder {
  constructor(device) {
    this.device = device;
    this._h = native.encoderBegin(device._h);
  }
  copyBufferToBuffer(src, srcOffset, dst, dstOffset, size) {
    if (!this._h) throw new Error("command encoder already finished");
    native.encoderCopyBuffer(this._h, src._h, srcOffset, dst._h, dstOffset, size);
  }
  finish(desc) {
    if (!this._h) throw new Error("command encoder already finished");
    // useGraph: true -> hipGraph, false -> op list, "auto" (default) -> graph for lists of >= 8 launches
    const useGraph = desc && desc.useGraph !== undefined ? desc.useGraph : this.device.useGraph;
    const mode = useGraph === "auto" ? 2 : (useGraph ? 1 : 0);
    const h = this._h;
    this._h = null;
    return new HipCommandBuffer(this.device, native.encoderFinish(h, mode));
  }
}

class HipQueue {
  constructor(device) {
    this.device = device;
  }
  writeBuffer(buffer, bufferOffset, data, dataOffset, size) {
    buffer._assertAlive();
    let view = data;
    if (dataOffset !== undefined || size !== undefined) {
      // WebGPU: dataOffset/size are in elements for typed arrays, bytes for ArrayBuffers
      const isView = ArrayBuffer.isView(data);
      const bpe = isView && data.BYTES_PER_ELEMENT ? data.BYTES_PER_ELEMENT : 1;
      const start = (dataOffset || 0) * bpe;
      const base = isView ? data.byteOffset : 0;
      const buf = isView ? data.buffer : data;
      const total = isView ? data.byteLength : data.byteLength;
      const len = size === undefined ? total - start : size * bpe;
      view = new Uint8Array(buf, base + start, len);
    }
    native.bufferWrite(buffer._h, bufferOffset, view);
  }
  submit(commandBuffers) {
    for (const cb of commandBuffers) {
      if (!cb || !cb._h) throw new Error("submit: invalid or released command buffer");
      native.queueSubmit(this.device._h, cb._h);
    }
  }
  onSubmittedWorkDone() {
    return native.queueWaitAsync(this.device._h);
  }
  // synchronous variant for scripts that cannot await
  waitIdle() {
    native.queueWait(this.device._h);
  }
}

export class HipDevice {
  constructor(ordinal, opts) {
    this.ordinal = ordinal | 0;
    this._h = native.deviceOpen(this.ordinal);
    this.useGraph = opts && opts.useGraph !== undefined ? opts.useGraph : "auto";
    this.queue = new HipQueue(this);
    const info = native.deviceInfo(this._h);
    this.info = info;
    this.features = new Set();      // no shader-f16 / subgroups: precision "f16-storage" is out of scope
    this.limits = Object.freeze({
      maxBufferSize: info.hbmTotal,
      maxStorageBufferBindingSize: info.hbmTotal,   // flat 64-bit device pointers: no binding windows
      maxComputeWorkgroupSizeX: 1024,
      maxComputeInvocationsPerWorkgroup: 1024,
      maxComputeWorkgroupStorageSize: 160 * 1024,
      minStorageBufferOffsetAlignment: 8,
    });
  }
  createBuffer(desc) {
    if (!desc || typeof desc.size !== "number") throw new Error("createBuffer expects {size, usage}");
    const size = Math.max(desc.size, 4);
    return new HipBuffer(this, native.bufferAlloc(this._h, size), desc.size, desc.usage);
  }
  createCommandEncoder() {
    return new HipCommandEncoder(this);
  }
  destroy() {
    if (this._h) {
      native.deviceClose(this._h);
      this._h = null;
    }
  }
}

// navigator.gpu.requestAdapter() / adapter.requestDevice() in one step
export function deviceCount() {
  return native.deviceCount();
}
export async function requestDevice(opts) {
  const o = opts || {};
  return new HipDevice(o.ordinal === undefined ? 0 : o.ordinal, o);
}
export function openDevice(opts) {
  const o = opts || {};
  return new HipDevice(o.ordinal === undefined ? 0 : o.ordinal, o);
}

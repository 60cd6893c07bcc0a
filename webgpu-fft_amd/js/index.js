// index.js — package entry with the reference's export list (src/index.js:3-13, src/public_api.js:3-9)
// plus the HIP device shim that stands in for navigator.gpu.
export { createPlan, createFftPlan, Plan, FftPlan, exportPipelineCacheSnapshot, importPipelineCacheSnapshot } from "./plans.js";
export { uploadComplex, downloadComplex, downloadF32 } from "./webgpu.js";
export { BufferView } from "./buffer_view.js";
export { createFftConvChannelLanePreset, createFftConvKernelMajorChannelLanePreset, createFftConvBatchMajorChannelLanePreset,
  normalizeScaleFactor, resolvePlanOptions } from "./layout.js";
export { requestDevice, openDevice, deviceCount, HipDevice, HipBuffer, GPUBufferUsage, GPUMapMode } from "./device.js";

// webgpu.js — uploadComplex / downloadComplex with the reference's signatures (src/utils/webgpu.js:9-23, 29-55).
import { GPUBufferUsage, GPUMapMode } from "./device.js";

function assertDevice(device) {
  if (!device) throw new Error("Expected a WebGPU device");
}

export function uploadComplex(device, data) {
  assertDevice(device);
  if (!(data instanceof Float32Array)) throw new Error("uploadComplex expects a Float32Array");
  const buffer = device.createBuffer({ size: data.byteLength, usage: GPUBufferUsage.STORAGE | GPUBufferUsage.COPY_DST | GPUBufferUsage.COPY_SRC });
  device.queue.writeBuffer(buffer, 0, data);
  return buffer;
}

export async function downloadComplex(device, buffer, lengthComplex, offsetBytes) {
  assertDevice(device);
  const off = offsetBytes === undefined ? 0 : offsetBytes;
  if (!buffer) throw new Error("downloadComplex expects a GPUBuffer");
  if (!Number.isInteger(lengthComplex) || lengthComplex <= 0) throw new Error("lengthComplex must be a positive integer; got " + lengthComplex);
  if (!Number.isInteger(off) || off < 0 || off % 8 !== 0) throw new Error("offsetBytes must be a non-negative multiple of 8; got " + off);
  const byteLength = lengthComplex * 8;
  // same shape as the reference: the readback waits for submitted work, off the JS thread
  await buffer.mapAsync(GPUMapMode.READ, off, byteLength);
  const out = new Float32Array(buffer.getMappedRange(off, byteLength));
  buffer.unmap();
  return out;
}

export async function downloadF32(device, buffer, count, offsetBytes) {
  assertDevice(device);
  const off = offsetBytes === undefined ? 0 : offsetBytes;
  await buffer.mapAsync(GPUMapMode.READ, off, count * 4);
  const out = new Float32Array(buffer.getMappedRange(off, count * 4));
  buffer.unmap();
  return out;
}

// buffer_view.js — BufferView (src/utils/buffer_view.js:11-43): a logical byte range over GPU buffers.
// The reference needs multi-segment views because WebGPU maxBufferSize is small; here a view is accepted as an
// input shape only and must have exactly one segment when handed to plan.exec (plans.js).
export class BufferView {
  constructor(segments) {
    if (!Array.isArray(segments) || segments.length === 0) throw new Error("BufferView expects a non-empty segment array");
    let total = 0;
    this.segments = segments.map((s, i) => {
      if (!s || !s.buffer || typeof s.buffer.size !== "number") throw new Error("BufferView segment " + i + " needs a buffer");
      const offsetBytes = s.offsetBytes === undefined ? 0 : s.offsetBytes;
      const sizeBytes = s.sizeBytes === undefined ? s.buffer.size - offsetBytes : s.sizeBytes;
      if (!Number.isInteger(offsetBytes) || offsetBytes < 0 || !Number.isInteger(sizeBytes) || sizeBytes <= 0 || offsetBytes + sizeBytes > s.buffer.size) {
        throw new Error("BufferView segment " + i + " range is outside its buffer");
      }
      total += sizeBytes;
      return { buffer: s.buffer, offsetBytes, sizeBytes };
    });
    this.size = total;
  }
  static from(buffer, offsetBytes, sizeBytes) {
    return new BufferView([{ buffer, offsetBytes, sizeBytes }]);
  }
}

// buffer_view.js — BufferView (src/utils/buffer_view.js:11-43): a logical byte range over GPU buffers, with the reference's
// constructor ({segments, logicalByteOffset, lengthBytes}), its validation messages and `static fromBuffer`.
// The reference needs multi-segment views because WebGPU maxBufferSize is small; one hipMalloc spans 288 GB, so a view is
// accepted as an input shape and must have exactly one segment when handed to plan.exec (plans.js unwrapBuffer).
export class BufferView {
  constructor(opts) {
    // round-1 form kept as an alias: new BufferView([{buffer, offsetBytes?, sizeBytes?}, ...])
    if (Array.isArray(opts)) {
      if (opts.length === 0) throw new Error("BufferView.segments must be a non-empty array");
      let total = 0;
      const segs = opts.map((s) => {
        if (!s || !s.buffer) throw new Error("BufferView segment missing buffer");
        const offsetBytes = s.offsetBytes === undefined ? 0 : s.offsetBytes;
        const sizeBytes = s.sizeBytes === undefined ? s.buffer.size - offsetBytes : s.sizeBytes;
        total += sizeBytes;
        return { buffer: s.buffer, offsetBytes, sizeBytes };
      });
      opts = { segments: segs, logicalByteOffset: 0, lengthBytes: total };
    }
    const o = opts || {};
    const segments = o.segments;
    const logicalByteOffset = o.logicalByteOffset === undefined ? 0 : o.logicalByteOffset;
    const lengthBytes = o.lengthBytes;
    if (!Array.isArray(segments) || segments.length === 0) throw new Error("BufferView.segments must be a non-empty array");
    if (!Number.isInteger(logicalByteOffset) || logicalByteOffset < 0) throw new Error("BufferView.logicalByteOffset must be a non-negative integer");
    if (!Number.isInteger(lengthBytes) || lengthBytes <= 0) throw new Error("BufferView.lengthBytes must be a positive integer");
    for (const s of segments) {
      if (!s || !s.buffer) throw new Error("BufferView segment missing buffer");
      if (!Number.isInteger(s.offsetBytes) || s.offsetBytes < 0) throw new Error("BufferView segment offsetBytes must be non-negative integer");
      if (!Number.isInteger(s.sizeBytes) || s.sizeBytes <= 0) throw new Error("BufferView segment sizeBytes must be positive integer");
      if (s.offsetBytes + s.sizeBytes > s.buffer.size) {
        throw new Error("BufferView segment out of bounds: offsetBytes+sizeBytes=" + (s.offsetBytes + s.sizeBytes) + " > buffer.size=" + s.buffer.size);
      }
    }
    this.segments = segments;
    this.logicalByteOffset = logicalByteOffset;
    this.lengthBytes = lengthBytes;
  }

  get size() { return this.lengthBytes; }   // round-1 name

  static fromBuffer(buffer, offsetBytes, lengthBytes) {
    const off = offsetBytes === undefined ? 0 : offsetBytes;
    const len = lengthBytes === undefined ? buffer.size - off : lengthBytes;
    return new BufferView({ segments: [{ buffer, offsetBytes: off, sizeBytes: len }], logicalByteOffset: 0, lengthBytes: len });
  }
  static from(buffer, offsetBytes, sizeBytes) { return BufferView.fromBuffer(buffer, offsetBytes, sizeBytes); }
}

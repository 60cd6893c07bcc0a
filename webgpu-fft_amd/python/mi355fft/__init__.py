"""Python harness over the libmi355fft.so C ABI, mirroring the reference's public API names
(src/public_api.js:3-9: createPlan, uploadComplex, downloadComplex, the fftconv preset builders) so the
parity tests read like the reference's own (test/complete.suite.js).

The product host is the JavaScript package in webgpu-fft_amd/js (N-API addon over the same C ABI); this
module exists for tests/ and bench.py.  There is no CPU path: every call goes to the HIP library and
raises Mi355Error when it is missing or reports an error.
"""
import ctypes
import os

import numpy as np

from . import _abi
from .layout import (createFftConvBatchMajorChannelLanePreset, createFftConvChannelLanePreset,  # noqa: F401
                     createFftConvKernelMajorChannelLanePreset, resolve_plan_options)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI355FFT_LIB") or os.path.normpath(os.path.join(_HERE, "..", "..", "lib", "libmi355fft.so"))   # env: A/B builds
_LIB = None


class Mi355Error(RuntimeError):
    def __init__(self, code, message):
        super().__init__(message)
        self.code = code


def lib():
    """Loads the C-ABI library (fails loudly when it has not been built)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise Mi355Error(-1, f"{LIB_PATH} is missing: run __graft_entry__.build() (make -C webgpu-fft_amd/csrc)")
        L = ctypes.CDLL(LIB_PATH)
        _abi.declare(L)
        if L.mi355fft_abi_version() != 1:
            raise Mi355Error(-1, "libmi355fft ABI version mismatch")
        _LIB = L
    return _LIB


def _chk(rc):
    if rc != 0:
        raise Mi355Error(rc, lib().mi355fft_last_error().decode())


class Buffer:
    """GPUBuffer analogue: numeric .size and .destroy() (runtime/common.js:55-57)."""

    def __init__(self, device, handle, size):
        self.device, self._h, self.size = device, handle, int(size)

    @property
    def device_ptr(self):
        return lib().mi355fft_buffer_device_ptr(self._h)

    def destroy(self):
        if self._h:
            lib().mi355fft_buffer_free(self._h)
            self._h = None


class CommandBuffer:
    def __init__(self, device, handle):
        self.device, self._h = device, handle

    def release(self):
        if self._h:
            lib().mi355fft_commands_release(self._h)
            self._h = None


class CommandEncoder:
    def __init__(self, device):
        self.device = device
        h = ctypes.c_void_p()
        _chk(lib().mi355fft_encoder_begin(device._h, ctypes.byref(h)))
        self._h = h

    def copyBufferToBuffer(self, src, srcOffset, dst, dstOffset, size):
        _chk(lib().mi355fft_encoder_copy_buffer(self._h, src._h, srcOffset, dst._h, dstOffset, size))

    def finish(self, use_graph=None):
        if use_graph is None:
            use_graph = self.device.use_graph
        h = ctypes.c_void_p()
        enc, self._h = self._h, None
        mode = 2 if use_graph == "auto" else (1 if use_graph else 0)
        _chk(lib().mi355fft_encoder_finish(enc, mode, ctypes.byref(h)))
        return CommandBuffer(self.device, h)


class Queue:
    def __init__(self, device):
        self.device = device

    def writeBuffer(self, buffer, offset, data):
        a = np.ascontiguousarray(data)
        _chk(lib().mi355fft_buffer_write(buffer._h, offset, a.ctypes.data, a.nbytes))

    def submit(self, command_buffers):
        for cb in command_buffers:
            _chk(lib().mi355fft_queue_submit(self.device._h, cb._h))

    def onSubmittedWorkDone(self):
        _chk(lib().mi355fft_queue_wait(self.device._h))


class Device:
    """The `device` argument of createPlan: a HIP device + its queue (stream)."""

    def __init__(self, ordinal=0, use_graph="auto"):
        h = ctypes.c_void_p()
        _chk(lib().mi355fft_device_open(ordinal, ctypes.byref(h)))
        self._h = h
        self.ordinal = ordinal
        self.use_graph = use_graph
        self.queue = Queue(self)

    def info(self):
        tot, fr, cus = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_int()
        arch = (ctypes.c_char * 64)()
        _chk(lib().mi355fft_device_info(self._h, ctypes.byref(tot), ctypes.byref(fr), ctypes.byref(cus), arch))
        return {"hbm_total": tot.value, "hbm_free": fr.value, "compute_units": cus.value, "arch": arch.value.decode()}

    @property
    def stream(self):
        return lib().mi355fft_device_stream(self._h)

    def createBuffer(self, desc):
        size = desc["size"] if isinstance(desc, dict) else int(desc)
        h = ctypes.c_void_p()
        _chk(lib().mi355fft_buffer_alloc(self._h, size, ctypes.byref(h)))
        return Buffer(self, h, size)

    def wrapBuffer(self, device_ptr, size):
        h = ctypes.c_void_p()
        _chk(lib().mi355fft_buffer_wrap(self._h, device_ptr, size, ctypes.byref(h)))
        return Buffer(self, h, size)

    def createCommandEncoder(self):
        return CommandEncoder(self)

    def close(self):
        if self._h:
            lib().mi355fft_device_close(self._h)
            self._h = None

    # test / bench support
    def fillRandom(self, buffer, offset_bytes, row_floats, rows, seed0, first_transform=0):
        _chk(lib().mi355fft_fill_random(self._h, buffer._h, offset_bytes, row_floats, rows, seed0 & 0xFFFFFFFF, first_transform))

    def sumsq(self, buffer, offset_bytes, count):
        out = ctypes.c_double()
        _chk(lib().mi355fft_sumsq(self._h, buffer._h, offset_bytes, count, ctypes.byref(out)))
        return out.value

    def diffSumsq(self, a, a_off, b, b_off, alpha, count):
        out = ctypes.c_double()
        _chk(lib().mi355fft_diff_sumsq(self._h, a._h, a_off, b._h, b_off, alpha, count, ctypes.byref(out)))
        return out.value


class Plan:
    """createPlan(...) result: exec / getWorkspaceSizeBytes / getPipelineCacheSnapshot / destroy
    (runtime/base_plan.js:31-54)."""

    def __init__(self, device, opts):
        self.device = device
        self.opts = dict(opts)
        resolved = resolve_plan_options(opts)
        self._resolved = resolved
        self.type = resolved["type"]
        self._desc = _abi.make_desc(resolved["type"], resolved["shape"], resolved["batch"], resolved["direction"], resolved["normalize"],
                                    resolved["inPlace"], resolved["input_layout"], resolved["output_layout"], resolved.get("conv"),
                                    resolved.get("io_view"), resolved.get("zero_pad"), resolved.get("axes"))
        h = ctypes.c_void_p()
        _chk(lib().mi355fft_plan_create(device._h, ctypes.byref(self._desc), ctypes.byref(h)))
        self._h = h
        self._destroyed = False
        self._kernel_upload = None

    def describe(self):
        text = ctypes.create_string_buffer(1024)
        n = ctypes.c_int()
        _chk(lib().mi355fft_plan_describe(self._h, text, 1024, ctypes.byref(n)))
        return text.value.decode(), n.value

    def getWorkspaceSizeBytes(self):
        out = ctypes.c_uint64()
        _chk(lib().mi355fft_plan_workspace_bytes(self._h, ctypes.byref(out)))
        return out.value

    def getPipelineCacheSnapshot(self):
        # kernels are ahead-of-time gfx950 code objects: nothing to cache (pipeline_cache.js)
        return {"schema": "webgpufft.pipeline-cache", "version": 2, "shaderCodes": [], "pipelineKeys": []}

    def _kernel_buffer(self, kernel):
        if isinstance(kernel, Buffer):
            return kernel
        conv = self._resolved["conv"]
        kn = int(np.prod(conv.get("kernelShape") or self._resolved["shape"]))
        single, packed = 2 * kn, 2 * kn * conv["kernelCount"]
        if isinstance(kernel, (list, tuple)):
            if len(kernel) != conv["kernelCount"]:
                raise Mi355Error(1, f"kernel array length must equal fftConv.kernelCount={conv['kernelCount']}; got {len(kernel)}")
            for i, k in enumerate(kernel):
                if np.asarray(k).size != single:
                    raise Mi355Error(1, f"kernel[{i}] Float32Array length must be {single}; got {np.asarray(k).size}")
            kernel = np.concatenate([np.asarray(k, dtype=np.float32).reshape(-1) for k in kernel])
        kernel = np.ascontiguousarray(kernel, dtype=np.float32).reshape(-1)
        if kernel.size != packed:
            raise Mi355Error(1, f"kernel Float32Array length must be {packed} for kernelCount={conv['kernelCount']}; got {kernel.size}")
        if self._kernel_upload is None or self._kernel_upload.size < kernel.nbytes:
            if self._kernel_upload is not None:
                self._kernel_upload.destroy()
            self._kernel_upload = self.device.createBuffer({"size": kernel.nbytes})
        self.device.queue.writeBuffer(self._kernel_upload, 0, kernel)
        return self._kernel_upload

    def exec(self, commandEncoder, execOpts):
        if self._destroyed:
            raise Mi355Error(_abi.ERR_DESTROYED, "plan destroyed")
        o = execOpts or {}
        a = _abi.ExecArgs()
        a.struct_size = ctypes.sizeof(_abi.ExecArgs)
        inp, out, temp, kern = o.get("input"), o.get("output"), o.get("temp"), o.get("kernel")
        a.input = inp._h if inp is not None else None
        a.output = out._h if out is not None else None
        a.temp = temp._h if temp is not None else None
        if kern is not None:
            a.kernel = self._kernel_buffer(kern)._h
        a.input_offset_bytes = int(o.get("inputOffsetBytes", 0))
        a.output_offset_bytes = int(o.get("outputOffsetBytes", 0))
        _chk(lib().mi355fft_plan_exec(self._h, commandEncoder._h, ctypes.byref(a)))

    def destroy(self):
        if self._destroyed:
            return
        self._destroyed = True
        _chk(lib().mi355fft_plan_destroy(self._h))
        if self._kernel_upload is not None:
            self._kernel_upload.destroy()
            self._kernel_upload = None

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().mi355fft_plan_release(self._h)
                self._h = None
        except Exception:
            pass


def createPlan(device, opts):
    """runtime/create_plan.js:12-23"""
    if device is None:
        raise Mi355Error(1, "Expected a device")
    return Plan(device, opts)


def uploadComplex(device, data):
    """utils/webgpu.js:9-23"""
    if device is None:
        raise Mi355Error(1, "Expected a device")
    if not isinstance(data, np.ndarray) or data.dtype != np.float32:
        raise Mi355Error(1, "uploadComplex expects a Float32Array")
    buf = device.createBuffer({"size": max(int(data.nbytes), 4)})
    device.queue.writeBuffer(buf, 0, data)
    return buf


def downloadComplex(device, buffer, lengthComplex, offsetBytes=0):
    """utils/webgpu.js:29-55 (synchronous here: waits for submitted work, then reads back)"""
    if buffer is None:
        raise Mi355Error(1, "downloadComplex expects a buffer")
    if not isinstance(lengthComplex, int) or lengthComplex <= 0:
        raise Mi355Error(1, f"lengthComplex must be a positive integer; got {lengthComplex}")
    if not isinstance(offsetBytes, int) or offsetBytes < 0 or offsetBytes % 8 != 0:
        raise Mi355Error(1, f"offsetBytes must be a non-negative multiple of 8; got {offsetBytes}")
    out = np.empty(2 * lengthComplex, dtype=np.float32)
    _chk(lib().mi355fft_buffer_read(buffer._h, offsetBytes, out.ctypes.data, out.nbytes))
    return out


def downloadF32(device, buffer, count, offsetBytes=0):
    out = np.empty(count, dtype=np.float32)
    _chk(lib().mi355fft_buffer_read(buffer._h, offsetBytes, out.ctypes.data, out.nbytes))
    return out

"""ctypes mirror of include/mi355fft.h (structs, enums, prototypes).

Used by the Python harness (tests/, bench.py) to call libmi355fft.so through its C ABI exactly as the
N-API addon does for the JavaScript host; also by the host-emulation tests for the shared plan_desc.
"""
import ctypes

MAX_RANK = 8

C2C, R2C, C2R, FFTCONV = 0, 1, 2, 3
FORWARD, INVERSE = 0, 1
NORM = {"none": 0, "backward": 1, "unitary": 2}
TYPE = {"c2c": C2C, "r2c": R2C, "c2r": C2R, "fftconv": FFTCONV,
        "dct1": 4, "dct2": 5, "dct3": 6, "dct4": 7, "dst1": 8, "dst2": 9, "dst3": 10, "dst4": 11}
DIRECTION = {"forward": FORWARD, "inverse": INVERSE}
CONV_MODE = {"convolution": 0, "correlation": 1}
CONV_BOUNDARY = {"circular": 0, "linear-full": 1, "linear-same": 2, "linear-valid": 3}
CONV_LAYOUT = {"kernel-major": 0, "batch-major": 1}

OK, ERR_INVALID, ERR_UNSUPPORTED, ERR_HIP, ERR_DESTROYED, ERR_NOMEM = 0, 1, 2, 3, 4, 5


class SideLayout(ctypes.Structure):
    _fields_ = [
        ("strided", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
        ("strides", ctypes.c_int64 * MAX_RANK),
        ("offset_elements", ctypes.c_int64),
        ("batch_stride_elements", ctypes.c_int64),
    ]


class IoView(ctypes.Structure):
    _fields_ = [("enabled", ctypes.c_int32), ("clear_outside", ctypes.c_int32), ("shape", ctypes.c_int64 * MAX_RANK),
                ("offset", ctypes.c_int64 * MAX_RANK)]


class ZeroRange(ctypes.Structure):
    _fields_ = [("enabled", ctypes.c_int32), ("reserved", ctypes.c_int32), ("start", ctypes.c_int64 * MAX_RANK),
                ("end", ctypes.c_int64 * MAX_RANK)]


class PlanDesc(ctypes.Structure):
    _fields_ = [
        ("struct_size", ctypes.c_uint32),
        ("type", ctypes.c_int32),
        ("rank", ctypes.c_int32),
        ("direction", ctypes.c_int32),
        ("normalize", ctypes.c_int32),
        ("in_place", ctypes.c_int32),
        ("shape", ctypes.c_int64 * MAX_RANK),
        ("batch", ctypes.c_int64),
        ("input", SideLayout),
        ("output", SideLayout),
        ("conv_mode", ctypes.c_int32),
        ("conv_boundary", ctypes.c_int32),
        ("conv_kernel_count", ctypes.c_int32),
        ("conv_output_layout", ctypes.c_int32),
        ("conv_kernel_shape", ctypes.c_int64 * MAX_RANK),
        ("conv_output_kernel_stride_elements", ctypes.c_int64),
        ("io_input", IoView),
        ("io_output", IoView),
        ("zero_read", ZeroRange),
        ("zero_write", ZeroRange),
        ("axes_mask", ctypes.c_uint32),
        ("reserved2", ctypes.c_uint32),
    ]


class ExecArgs(ctypes.Structure):
    _fields_ = [
        ("struct_size", ctypes.c_uint32),
        ("reserved", ctypes.c_uint32),
        ("input", ctypes.c_void_p),
        ("output", ctypes.c_void_p),
        ("temp", ctypes.c_void_p),
        ("kernel", ctypes.c_void_p),
        ("input_offset_bytes", ctypes.c_uint64),
        ("output_offset_bytes", ctypes.c_uint64),
        ("kernel_offset_bytes", ctypes.c_uint64),
    ]


def _fill_side(side, spec, rank):
    """spec: None (dense) or dict(strides=[...], offset=int, batch_stride=int)."""
    if not spec:
        return
    side.strided = 1
    strides = spec.get("strides")
    if strides is None:
        raise ValueError("resolved side layout needs strides")
    for i in range(rank):
        side.strides[i] = int(strides[i])
    side.offset_elements = int(spec.get("offset", 0))
    side.batch_stride_elements = int(spec.get("batch_stride", 0))


def _fill_view(dst, spec, rank):
    if not spec:
        return
    dst.enabled = 1
    dst.clear_outside = 1 if spec.get("clearOutside") else 0
    for i in range(rank):
        dst.shape[i] = int(spec["shape"][i])
        dst.offset[i] = int(spec["offset"][i])


def _fill_range(dst, spec, rank):
    if not spec:
        return
    dst.enabled = 1
    for i in range(rank):
        dst.start[i] = int(spec["start"][i])
        dst.end[i] = int(spec["end"][i])


def make_desc(type, shape, batch=1, direction="forward", normalize="none", in_place=False, input_layout=None, output_layout=None,
              conv=None, io_view=None, zero_pad=None, axes=None):
    """Build a PlanDesc from already-RESOLVED options (layout resolution lives in mi355fft.plans)."""
    d = PlanDesc()
    d.struct_size = ctypes.sizeof(PlanDesc)
    d.type = TYPE[type] if isinstance(type, str) else int(type)
    d.rank = len(shape)
    for i, s in enumerate(shape):
        d.shape[i] = int(s)
    d.batch = int(batch)
    d.direction = DIRECTION[direction] if isinstance(direction, str) else int(direction)
    d.normalize = NORM[normalize] if isinstance(normalize, str) else int(normalize)
    d.in_place = 1 if in_place else 0
    _fill_side(d.input, input_layout, d.rank)
    _fill_side(d.output, output_layout, d.rank)
    if conv:
        d.conv_mode = CONV_MODE[conv.get("mode", "convolution")]
        d.conv_boundary = CONV_BOUNDARY[conv.get("boundary", "circular")]
        d.conv_kernel_count = int(conv.get("kernelCount", 1))
        d.conv_output_layout = CONV_LAYOUT[conv.get("outputLayout", "kernel-major")]
        ks = conv.get("kernelShape")
        if ks:
            for i, s in enumerate(ks):
                d.conv_kernel_shape[i] = int(s)
        d.conv_output_kernel_stride_elements = int(conv.get("outputKernelStrideElements", 0))
    else:
        d.conv_kernel_count = 1
    if io_view:
        _fill_view(d.io_input, io_view.get("input"), d.rank)
        _fill_view(d.io_output, io_view.get("output"), d.rank)
    if zero_pad:
        _fill_range(d.zero_read, zero_pad.get("read"), d.rank)
        _fill_range(d.zero_write, zero_pad.get("write"), d.rank)
    if axes is not None:
        for a in axes:
            d.axes_mask |= 1 << int(a)
    return d


def declare(lib):
    """Attach prototypes for every symbol include/mi355fft.h declares."""
    vp, u64, i32 = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int
    P = ctypes.POINTER
    sigs = {
        "mi355fft_abi_version": (i32, []),
        "mi355fft_last_error": (ctypes.c_char_p, []),
        "mi355fft_device_count": (i32, [P(i32)]),
        "mi355fft_device_open": (i32, [i32, P(vp)]),
        "mi355fft_device_close": (i32, [vp]),
        "mi355fft_device_info": (i32, [vp, P(u64), P(u64), P(i32), ctypes.c_char * 64]),
        "mi355fft_device_stream": (vp, [vp]),
        "mi355fft_buffer_alloc": (i32, [vp, u64, P(vp)]),
        "mi355fft_buffer_wrap": (i32, [vp, vp, u64, P(vp)]),
        "mi355fft_buffer_free": (i32, [vp]),
        "mi355fft_buffer_size": (u64, [vp]),
        "mi355fft_buffer_device_ptr": (vp, [vp]),
        "mi355fft_buffer_write": (i32, [vp, u64, vp, u64]),
        "mi355fft_buffer_read": (i32, [vp, u64, vp, u64]),
        "mi355fft_plan_create": (i32, [vp, P(PlanDesc), P(vp)]),
        "mi355fft_plan_workspace_bytes": (i32, [vp, P(u64)]),
        "mi355fft_plan_exec": (i32, [vp, vp, P(ExecArgs)]),
        "mi355fft_plan_destroy": (i32, [vp]),
        "mi355fft_plan_release": (i32, [vp]),
        "mi355fft_plan_describe": (i32, [vp, ctypes.c_char_p, ctypes.c_size_t, P(i32)]),
        "mi355fft_encoder_begin": (i32, [vp, P(vp)]),
        "mi355fft_encoder_copy_buffer": (i32, [vp, vp, u64, vp, u64, u64]),
        "mi355fft_encoder_finish": (i32, [vp, i32, P(vp)]),
        "mi355fft_encoder_discard": (i32, [vp]),
        "mi355fft_queue_submit": (i32, [vp, vp]),
        "mi355fft_commands_release": (i32, [vp]),
        "mi355fft_queue_wait": (i32, [vp]),
        "mi355fft_fill_random": (i32, [vp, vp, u64, u64, u64, ctypes.c_uint32, u64]),
        "mi355fft_sumsq": (i32, [vp, vp, u64, u64, P(ctypes.c_double)]),
        "mi355fft_diff_sumsq": (i32, [vp, vp, u64, vp, u64, ctypes.c_double, u64, P(ctypes.c_double)]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    return sorted(sigs)


ABI_SYMBOLS = [
    "mi355fft_abi_version", "mi355fft_last_error", "mi355fft_device_count", "mi355fft_device_open", "mi355fft_device_close",
    "mi355fft_device_info", "mi355fft_device_stream", "mi355fft_buffer_alloc", "mi355fft_buffer_wrap", "mi355fft_buffer_free",
    "mi355fft_buffer_size", "mi355fft_buffer_device_ptr", "mi355fft_buffer_write", "mi355fft_buffer_read", "mi355fft_plan_create",
    "mi355fft_plan_workspace_bytes", "mi355fft_plan_exec", "mi355fft_plan_destroy", "mi355fft_plan_release", "mi355fft_plan_describe",
    "mi355fft_encoder_begin", "mi355fft_encoder_copy_buffer", "mi355fft_encoder_finish", "mi355fft_encoder_discard",
    "mi355fft_queue_submit", "mi355fft_commands_release", "mi355fft_queue_wait", "mi355fft_fill_random", "mi355fft_sumsq",
    "mi355fft_diff_sumsq",
]

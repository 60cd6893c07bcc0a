"""Host-side option resolution: createPlan opts -> the resolved form the C ABI takes.

Mirrors (same names, argument meaning and error text where the reference's tests match on it):
  runtime/fftconv_channel_lane_presets.js:139-206   preset builders
  runtime/layout_semantics.js:120-232               layout.{strides,...} and layout.whdcn resolution
  runtime/plans/fftconv.js:144-281                  fftConv.channelPolicy -> whdcn side descriptors
  runtime/plans/c2c.js:547-558, fftconv.js:320-338  option validation
The JavaScript twin is webgpu-fft_amd/js/layout.js.
"""
import math

MAX_SAFE = 2 ** 53 - 1

FFTCONV_MODES = ("convolution", "correlation")
FFTCONV_BOUNDARIES = ("circular", "linear-full", "linear-same", "linear-valid")
FFTCONV_OUTPUT_LAYOUTS = ("kernel-major", "batch-major")
CONFLICTING_LAYOUT_KEYS = ("whdcn", "strides", "inputStrides", "outputStrides", "offsetElements", "inputOffsetElements",
                           "outputOffsetElements", "batchStrideElements", "inputBatchStrideElements", "outputBatchStrideElements")
HOT_PATH_TYPES = ("c2c", "r2c", "c2r", "fftconv")
TRIG_TYPES = ("dct1", "dct2", "dct3", "dct4", "dst1", "dst2", "dst3", "dst4")   # real-to-real, real f32 buffers (dct_fft.js)
ALL_TYPES = ("c2c", "r2c", "c2r", "dct1", "dct2", "dct3", "dct4", "dst1", "dst2", "dst3", "dst4", "fftconv", "conv2d")


def _is_int(v):
    return isinstance(v, int) and not isinstance(v, bool)


def _quote_list(values):
    return ", ".join('"%s"' % v for v in values)


def _assert_one_of(value, allowed, name):
    if value not in allowed:
        raise ValueError(f"{name} must be one of {_quote_list(allowed)}; got {value!r}")


def _pos_safe(v, name):
    if not _is_int(v) or v <= 0 or v > MAX_SAFE:
        raise ValueError(f"{name} must be a positive safe integer")


def _nonneg_safe(v, name):
    if not _is_int(v) or v < 0 or v > MAX_SAFE:
        raise ValueError(f"{name} must be a non-negative safe integer")


def _prod(shape):
    p = 1
    for s in shape:
        p *= s
    return p


# ---- preset builders ------------------------------------------------------------------------------------
def _normalize_side(side, side_name, logical_span, kernel_count, allow_kernel_step):
    if not isinstance(side, dict):
        raise ValueError(f"{side_name} must be an object")
    _pos_safe(side.get("channels"), f"{side_name}.channels")
    channels = side["channels"]
    channel_index = side.get("channelIndex", 0) if side.get("channelIndex") is not None else 0
    _nonneg_safe(channel_index, f"{side_name}.channelIndex")
    if channel_index >= channels:
        raise ValueError(f"{side_name}.channelIndex ({channel_index}) must be < {side_name}.channels ({channels})")
    cstride = side.get("channelStrideElements") if side.get("channelStrideElements") is not None else logical_span
    _pos_safe(cstride, f"{side_name}.channelStrideElements")
    if cstride < logical_span:
        raise ValueError(f"{side_name}.channelStrideElements must be >= logical span ({logical_span})")
    default_bstride = channels * cstride
    if default_bstride > MAX_SAFE:
        raise ValueError(f"{side_name}.batchStrideElements exceeds safe integer range")
    bstride = side.get("batchStrideElements") if side.get("batchStrideElements") is not None else default_bstride
    _pos_safe(bstride, f"{side_name}.batchStrideElements")
    if bstride < default_bstride:
        raise ValueError(f"{side_name}.batchStrideElements must be >= channels*channelStrideElements ({default_bstride})")
    offset = side.get("offsetElements") if side.get("offsetElements") is not None else 0
    _nonneg_safe(offset, f"{side_name}.offsetElements")
    desc = {"channels": channels, "channelIndex": channel_index, "channelStrideElements": cstride, "batchStrideElements": bstride,
            "offsetElements": offset}
    if allow_kernel_step:
        step = side.get("kernelStepChannels") if side.get("kernelStepChannels") is not None else 1
        _pos_safe(step, f"{side_name}.kernelStepChannels")
        if kernel_count > 1:
            max_idx = channel_index + (kernel_count - 1) * step
            if max_idx >= channels:
                raise ValueError(f"{side_name} does not fit kernelCount={kernel_count}: max channel index {max_idx} exceeds channels={channels} "
                                 f"(channelIndex={channel_index}, kernelStepChannels={step})")
        desc["kernelStepChannels"] = step
    return desc


def _build_preset(opts, forced_output_layout=None):
    if not isinstance(opts, dict):
        raise ValueError("opts must be an object")
    shape = opts.get("shape")
    if not isinstance(shape, (list, tuple)) or len(shape) == 0:
        raise ValueError("shape must be a non-empty array")
    span = 1
    for i, dim in enumerate(shape):
        _pos_safe(dim, f"shape[{i}]")
        span *= dim
        if span > MAX_SAFE:
            raise ValueError("shape product exceeds safe integer range")
    batch = opts.get("batch")
    kernel_count = opts.get("kernelCount", 1)
    mode = opts.get("mode", "convolution")
    boundary = opts.get("boundary", "circular")
    output_layout = opts.get("outputLayout", "kernel-major")
    layout = opts.get("layout", {})
    _pos_safe(batch, "batch")
    _pos_safe(kernel_count, "kernelCount")
    _assert_one_of(mode, FFTCONV_MODES, "mode")
    _assert_one_of(boundary, FFTCONV_BOUNDARIES, "boundary")
    _assert_one_of(output_layout, FFTCONV_OUTPUT_LAYOUTS, "outputLayout")
    if not isinstance(layout, dict):
        raise ValueError("layout must be an object")
    if "interleavedComplex" in layout and layout["interleavedComplex"] is not True:
        raise ValueError("layout.interleavedComplex must be true for fftconv channel-lane presets")
    for key in CONFLICTING_LAYOUT_KEYS:
        if key in layout:
            raise ValueError(f"layout.{key} cannot be combined with fftConv.channelPolicy presets")
    final_layout = forced_output_layout if forced_output_layout is not None else output_layout
    in_desc = _normalize_side(opts.get("input"), "input", span, kernel_count, False)
    out_desc = _normalize_side(opts.get("output"), "output", span, kernel_count, True)
    merged_layout = {"interleavedComplex": True}
    merged_layout.update(layout)
    return {
        "shape": list(shape),
        "batch": batch,
        "layout": merged_layout,
        "fftConv": {"mode": mode, "boundary": boundary, "kernelCount": kernel_count, "outputLayout": final_layout,
                    "channelPolicy": {"input": in_desc, "output": out_desc}},
    }


def createFftConvChannelLanePreset(opts):
    return _build_preset(opts, None)


def createFftConvKernelMajorChannelLanePreset(opts):
    return _build_preset(opts, "kernel-major")


def createFftConvBatchMajorChannelLanePreset(opts):
    return _build_preset(opts, "batch-major")


# ---- layout resolution ------------------------------------------------------------------------------------
def _contiguous_strides(shape):
    out, acc = [], 1
    for s in shape:
        out.append(acc)
        acc *= s
    return out


def _strided_span(shape, strides):
    return sum((s - 1) * st for s, st in zip(shape, strides)) + 1


def _side_field(side, suffix):
    return f"{side}{suffix}"


def _has_explicit_side(layout, side):
    return any(k in layout for k in (_side_field(side, "Strides"), _side_field(side, "OffsetElements"), _side_field(side, "BatchStrideElements"),
                                     "strides", "offsetElements", "batchStrideElements"))


def _opt_pos_int_array(v, rank, name):
    if v is None:
        return None
    if not isinstance(v, (list, tuple)) or len(v) != rank or not all(_is_int(x) and x > 0 for x in v):
        raise ValueError(f"{name} must be an array of {rank} positive ints")
    return list(v)


def _opt_nonneg(v, name):
    if v is None:
        return None
    if not _is_int(v) or v < 0:
        raise ValueError(f"{name} must be a non-negative integer")
    return v


def _opt_pos(v, name):
    if v is None:
        return None
    if not _is_int(v) or v <= 0:
        raise ValueError(f"{name} must be a positive integer")
    return v


def _resolve_explicit_side(layout, side, rank, shape):
    strides = _opt_pos_int_array(layout.get(_side_field(side, "Strides"), layout.get("strides")), rank, f"layout.{side}Strides")
    offset = _opt_nonneg(layout.get(_side_field(side, "OffsetElements"), layout.get("offsetElements")), f"layout.{side}OffsetElements")
    bstride = _opt_nonneg(layout.get(_side_field(side, "BatchStrideElements"), layout.get("batchStrideElements")),
                          f"layout.{side}BatchStrideElements")
    if strides is None and offset is None and bstride is None:
        return None
    strides = strides or _contiguous_strides(shape)
    span = _strided_span(shape, strides)
    return {"strides": strides, "offset": offset or 0, "batch_stride": bstride if bstride else max(span, _prod(shape))}


def _resolve_whdcn_side(desc, side, rank, shape):
    if not desc or desc.get("enabled") is False:
        return None
    if not any(k in desc for k in ("strides", "offsetElements", "batchStrideElements", "channels", "channelIndex", "channelStrideElements")):
        return None
    path = f"layout.whdcn.{side}"
    strides = _opt_pos_int_array(desc.get("strides"), rank, f"{path}.strides") or _contiguous_strides(shape)
    span = _strided_span(shape, strides)
    channels = _opt_pos(desc.get("channels"), f"{path}.channels") or 1
    cidx = _opt_nonneg(desc.get("channelIndex"), f"{path}.channelIndex") or 0
    if cidx >= channels:
        raise ValueError(f"{path}.channelIndex ({cidx}) must be < {path}.channels ({channels})")
    cstride = _opt_pos(desc.get("channelStrideElements"), f"{path}.channelStrideElements") or span
    if cstride < span:
        raise ValueError(f"{path}.channelStrideElements must be >= addressed span ({span})")
    base_off = _opt_nonneg(desc.get("offsetElements"), f"{path}.offsetElements") or 0
    offset = base_off + cidx * cstride
    default_b = cstride * channels
    bstride = _opt_nonneg(desc.get("batchStrideElements"), f"{path}.batchStrideElements")
    if bstride is None:
        bstride = default_b
    if bstride < default_b:
        raise ValueError(f"{path}.batchStrideElements must be >= channels*channelStrideElements ({default_b})")
    if strides == _contiguous_strides(shape) and offset == 0 and bstride == _prod(shape) and channels == 1 and cidx == 0 and cstride == span:
        return None
    return {"strides": strides, "offset": offset, "batch_stride": bstride}


def resolve_layout_semantics(layout, rank, input_shape, output_shape):
    l = layout or {}
    if not isinstance(l, dict):
        raise ValueError("layout must be an object")
    inp = _resolve_explicit_side(l, "input", rank, input_shape)
    out = _resolve_explicit_side(l, "output", rank, output_shape)
    wh = l.get("whdcn")
    if wh is not None:
        if not isinstance(wh, dict):
            raise ValueError("layout.whdcn must be an object")
        glob = {k: v for k, v in wh.items() if k not in ("input", "output")}
        for side, shape in (("input", input_shape), ("output", output_shape)):
            if _has_explicit_side(l, side):
                continue
            sd = wh.get(side)
            if sd is not None and not isinstance(sd, dict):
                raise ValueError(f"layout.whdcn.{side} must be an object")
            merged = dict(glob)
            merged.update(sd or {})
            r = _resolve_whdcn_side(merged, side, rank, shape)
            if r:
                if side == "input":
                    inp = r
                else:
                    out = r
    return inp, out


def _resolve_channel_policy(layout, policy, kernel_count, in_total, out_total):
    if policy is None:
        return layout or {}, 0
    if not isinstance(policy, dict):
        raise ValueError("fftConv.channelPolicy must be an object")
    has_in = policy.get("input") is not None
    has_out = policy.get("output") is not None
    if not has_in and not has_out:
        raise ValueError("fftConv.channelPolicy must provide input and/or output descriptors")
    lay = layout or {}
    if lay.get("whdcn") is not None:
        raise ValueError("fftConv.channelPolicy cannot be combined with layout.whdcn")
    if has_in and _has_explicit_side(lay, "input"):
        raise ValueError("fftConv.channelPolicy.input cannot be combined with explicit input stride fields")
    if has_out and _has_explicit_side(lay, "output"):
        raise ValueError("fftConv.channelPolicy.output cannot be combined with explicit output stride fields")

    def side(desc, path, span, allow_step):
        if desc is None:
            return None
        if not isinstance(desc, dict):
            raise ValueError(f"{path} must be an object")
        channels = _opt_pos(desc.get("channels"), f"{path}.channels")
        if channels is None:
            raise ValueError(f"{path}.channels is required")
        cidx = _opt_nonneg(desc.get("channelIndex"), f"{path}.channelIndex") or 0
        if cidx >= channels:
            raise ValueError(f"{path}.channelIndex ({cidx}) must be < {path}.channels ({channels})")
        cstride = _opt_pos(desc.get("channelStrideElements"), f"{path}.channelStrideElements") or span
        if cstride < span:
            raise ValueError(f"{path}.channelStrideElements must be >= logical span ({span})")
        off = _opt_nonneg(desc.get("offsetElements"), f"{path}.offsetElements") or 0
        default_b = channels * cstride
        bstride = _opt_nonneg(desc.get("batchStrideElements"), f"{path}.batchStrideElements")
        if bstride is None:
            bstride = default_b
        if bstride < default_b:
            raise ValueError(f"{path}.batchStrideElements must be >= channels*channelStrideElements ({default_b})")
        step = (_opt_pos(desc.get("kernelStepChannels"), f"{path}.kernelStepChannels") or 1) if allow_step else 1
        if allow_step and kernel_count > 1:
            max_idx = cidx + (kernel_count - 1) * step
            if max_idx >= channels:
                raise ValueError(f"{path} does not fit kernelCount={kernel_count}: max channel index {max_idx} exceeds channels={channels} "
                                 f"(channelIndex={cidx}, kernelStepChannels={step})")
        return {"channels": channels, "channelIndex": cidx, "channelStrideElements": cstride, "batchStrideElements": bstride,
                "offsetElements": off}, cstride * step

    wh = {}
    kstride = 0
    if has_in:
        wh["input"], _ = side(policy["input"], "fftConv.channelPolicy.input", in_total, False)
    if has_out:
        wh["output"], ks = side(policy["output"], "fftConv.channelPolicy.output", out_total, True)
        if kernel_count > 1:
            kstride = ks
    merged = dict(lay)
    merged["whdcn"] = wh
    return merged, kstride


def normalize_io_view(rank, logical_shape, io_view):
    """runtime/ioview.js:7-37; a side that maps 1:1 onto the logical domain resolves to None"""
    def one(v, kind):
        if not v:
            return None
        shape = v.get("shape")
        if not isinstance(shape, (list, tuple)) or len(shape) != rank or not all(_is_int(x) and x > 0 for x in shape):
            raise ValueError(f"ioView.{kind}.shape must be an array of {rank} positive ints")
        placement = v.get("placement", "start")
        if placement not in ("start", "center"):
            raise ValueError(f'ioView.{kind}.placement must be "start"|"center"')
        offset = v.get("offset")
        if offset is not None:
            if not isinstance(offset, (list, tuple)) or len(offset) != rank or not all(_is_int(x) for x in offset):
                raise ValueError(f"ioView.{kind}.offset must be an array of {rank} integers")
            offset = list(offset)
        elif placement == "center":
            offset = [(logical_shape[d] - shape[d]) // 2 for d in range(rank)]
        else:
            offset = [0] * rank
        if list(shape) == list(logical_shape) and all(o == 0 for o in offset):
            return None
        return {"shape": list(shape), "offset": offset, "clearOutside": bool(v.get("clearOutside")) if kind == "output" else False}
    iv = io_view or {}
    return {"input": one(iv.get("input"), "input"), "output": one(iv.get("output"), "output")}


def normalize_zero_pad(rank, shape, zero_pad, name="zeroPad"):
    """runtime/zero_pad.js:11-45; a stage covering the whole domain resolves to None"""
    if not zero_pad:
        return {"read": None, "write": None}
    if not isinstance(zero_pad, dict):
        raise ValueError(f"{name} must be an object with optional read/write stage configs")

    def stage(st, nm):
        if not st:
            return None
        if not isinstance(st, dict):
            raise ValueError(f"{nm} must be an object with optional start/end arrays")
        src = st["range"] if isinstance(st.get("range"), dict) else st

        def bound(v, which, dflt):
            if v is None:
                return list(dflt)
            if not isinstance(v, (list, tuple)) or len(v) != rank or not all(_is_int(x) for x in v):
                raise ValueError(f"{nm}.{which} must be an array of {rank} integers")
            return list(v)
        start, end = bound(src.get("start"), "start", [0] * rank), bound(src.get("end"), "end", shape)
        for d in range(rank):
            if start[d] < 0:
                raise ValueError(f"{nm}.start[{d}] must be >= 0; got {start[d]}")
            if end[d] < 0:
                raise ValueError(f"{nm}.end[{d}] must be >= 0; got {end[d]}")
            if start[d] > end[d]:
                raise ValueError(f"{nm}: start[{d}] must be <= end[{d}]")
            if end[d] > shape[d]:
                raise ValueError(f"{nm}.end[{d}] must be <= shape[{d}] ({shape[d]}); got {end[d]}")
        if all(s0 == 0 for s0 in start) and all(e == shape[d] for d, e in enumerate(end)):
            return None
        return {"start": start, "end": end}
    return {"read": stage(zero_pad.get("read"), f"{name}.read"), "write": stage(zero_pad.get("write"), f"{name}.write")}


def resolve_plan_options(opts):
    """Validates createPlan opts and returns the resolved dict Plan/_abi.make_desc consume."""
    if not isinstance(opts, dict):
        raise ValueError("createPlan expects an options object")
    typ = opts.get("type")
    _assert_one_of(typ, ALL_TYPES, "type")
    if typ not in HOT_PATH_TYPES and typ not in TRIG_TYPES:
        raise NotImplementedError(f'type "{typ}" is outside the MI355X hot path (c2c/r2c/c2r/fftconv); see DESIGN.md "out of scope"')
    shape = opts.get("shape")
    if not isinstance(shape, (list, tuple)) or len(shape) < 1:
        raise ValueError(f"shape must be an array of one or more positive dimensions; got {shape!r}")
    if not all(_is_int(s) and s > 0 for s in shape):
        raise ValueError(f"shape elements must be positive ints; got {list(shape)!r}")
    shape = list(shape)
    rank = len(shape)
    batch = opts.get("batch", 1)
    if not _is_int(batch) or batch <= 0:
        raise ValueError(f"batch must be positive int; got {batch}")
    trig = typ in TRIG_TYPES
    layout = opts.get("layout", {"interleavedComplex": not trig})
    if trig:
        if not isinstance(layout, dict) or layout.get("interleavedComplex") is not False:
            raise ValueError("DCT/DST uses real buffers; set layout.interleavedComplex=false")
        if any(s < 2 for s in shape):
            raise ValueError(f"All DCT/DST dimensions must be >= 2; got shape={shape!r}")
        if opts.get("inPlace", False):
            raise ValueError("DCT/DST inPlace is not supported in current implementation")
    elif not isinstance(layout, dict) or layout.get("interleavedComplex") is not True:
        raise ValueError(f"{typ} requires layout.interleavedComplex=true")
    precision = opts.get("precision", "f32")
    _assert_one_of(precision, ("f32", "f16-storage"), "precision")
    if precision != "f32":
        raise NotImplementedError('precision "f16-storage" is outside the MI355X hot path (f32 only)')
    # logical domains of the two sides: r2c writes / c2r reads the PACKED spectrum (r2c.js:72-123, c2r.js:168-220)
    packed_shape = [shape[0] // 2 + 1] + shape[1:]
    in_logical = packed_shape if typ == "c2r" else shape
    out_logical = packed_shape if typ == "r2c" else shape
    iv = opts.get("ioView") or {}
    zp = opts.get("zeroPad")
    io_view = {"input": normalize_io_view(rank, in_logical, {"input": iv.get("input")})["input"],
               "output": normalize_io_view(rank, out_logical, {"output": iv.get("output")})["output"]}
    if zp is not None and not isinstance(zp, dict):
        raise ValueError("zeroPad must be an object with optional read/write stage configs")
    if typ == "fftconv":
        if io_view["input"] or io_view["output"]:
            raise ValueError("ioView is not an fftconv option (fftconv.js:308-320)")
        zero_pad = {"read": None, "write": None}      # resolved below against the FFT domain (fftconv.js:353,386)
    else:
        zero_pad = {"read": normalize_zero_pad(rank, in_logical, {"read": (zp or {}).get("read")})["read"],
                    "write": normalize_zero_pad(rank, out_logical, {"write": (zp or {}).get("write")})["write"]}
    in_place = bool(opts.get("inPlace", False))
    normalize = opts.get("normalize", "none")
    axes = opts.get("axes")            # createFftPlan({axes}) (plan.js:1335-1339): c2c only; the scale factor keeps prod(shape)
    if axes is not None:
        if typ != "c2c":
            raise ValueError("axes is a createFftPlan (c2c) option")
        if not isinstance(axes, (list, tuple)) or len(axes) == 0:
            raise ValueError("axes must be null or a non-empty array")
        for a in axes:
            if not _is_int(a) or a < 0 or a >= rank:
                raise ValueError(f"Invalid axis {a} for rank {rank}")
        axes = list(axes)
    out = {"type": typ, "shape": shape, "batch": batch, "inPlace": in_place, "normalize": normalize, "conv": None,
           "io_view": io_view, "zero_pad": zero_pad, "axes": axes}

    if typ == "fftconv":
        fc = opts.get("fftConv") or {}
        mode = fc.get("mode", "convolution")
        _assert_one_of(mode, FFTCONV_MODES, "fftConv.mode")
        boundary = fc.get("boundary", "circular")
        _assert_one_of(boundary, FFTCONV_BOUNDARIES, "fftConv.boundary")
        kernel_count = fc.get("kernelCount", 1)
        if not _is_int(kernel_count) or kernel_count <= 0:
            raise ValueError(f"fftConv.kernelCount must be a positive integer; got {kernel_count}")
        output_layout = fc.get("outputLayout", "kernel-major")
        _assert_one_of(output_layout, FFTCONV_OUTPUT_LAYOUTS, "fftConv.outputLayout")
        kshape = fc.get("kernelShape") or shape
        if not isinstance(kshape, (list, tuple)) or len(kshape) != rank or not all(_is_int(k) and k > 0 for k in kshape):
            raise ValueError(f"fftConv.kernelShape must be an array of {rank} positive ints")
        kshape = list(kshape)
        if boundary == "circular":
            for d in range(rank):
                if kshape[d] > shape[d]:
                    raise ValueError(f'fftConv.kernelShape[{d}] must be <= shape[{d}] when fftConv.boundary="circular"')
            out_shape = list(shape)
        elif boundary == "linear-full":
            out_shape = [s + k - 1 for s, k in zip(shape, kshape)]
        elif boundary == "linear-same":
            out_shape = list(shape)
        else:
            out_shape = [s - k + 1 for s, k in zip(shape, kshape)]
            for d in range(rank):
                if out_shape[d] <= 0:
                    raise ValueError(f'fftConv.boundary="linear-valid" requires kernelShape[{d}] <= shape[{d}]')
        fft_shape = list(shape) if boundary == "circular" else [s + k - 1 for s, k in zip(shape, kshape)]
        out["zero_pad"] = normalize_zero_pad(rank, fft_shape, zp)
        explicit_kstride = _opt_pos(fc.get("outputKernelStrideElements"), "fftConv.outputKernelStrideElements") or 0
        merged_layout, policy_kstride = _resolve_channel_policy(layout, fc.get("channelPolicy"), kernel_count, _prod(shape), _prod(out_shape))
        if explicit_kstride and policy_kstride and explicit_kstride != policy_kstride:
            raise ValueError("fftConv.outputKernelStrideElements conflicts with fftConv.channelPolicy.output kernel step mapping")
        inp, outl = resolve_layout_semantics(merged_layout, rank, shape, out_shape)
        out.update({"direction": "forward", "normalize": "none", "input_layout": inp, "output_layout": outl,
                    "conv": {"mode": mode, "boundary": boundary, "kernelCount": kernel_count, "outputLayout": output_layout,
                             "kernelShape": kshape, "outputKernelStrideElements": explicit_kstride or policy_kstride or 0},
                    "outputShape": out_shape})
        if in_place:
            raise ValueError("fftconv inPlace=true is not supported in current implementation")
        return out

    direction = opts.get("direction", "forward" if trig else None)     # dct_fft.js:70: DCT/DST default to forward
    _assert_one_of(direction, ("forward", "inverse"), "direction")
    _assert_one_of(normalize, ("none", "backward", "unitary"), "normalize")
    if typ == "r2c" and direction != "forward":
        raise ValueError('r2c supports direction:"forward" only')
    if typ == "c2r" and direction != "inverse":
        raise ValueError('c2r supports direction:"inverse" only')
    if in_place and typ != "c2c":
        raise ValueError("inPlace=true is supported only on c2c")
    packed = [shape[0] // 2 + 1] + shape[1:]
    in_shape = io_view["input"]["shape"] if io_view["input"] else (packed if typ == "c2r" else shape)
    out_shape = io_view["output"]["shape"] if io_view["output"] else (packed if typ == "r2c" else shape)
    inp, outl = resolve_layout_semantics(layout, rank, in_shape, out_shape)
    out.update({"direction": direction, "input_layout": inp, "output_layout": outl})
    return out


def normalizeScaleFactor(normalize, direction, nTotal):
    """runtime/common.js:35-40"""
    if normalize == "none":
        return 1.0
    if normalize == "unitary":
        return 1.0 / math.sqrt(nTotal)
    if normalize == "backward":
        return 1.0 / nTotal if direction == "inverse" else 1.0
    raise ValueError(f"Unknown normalize mode: {normalize}")

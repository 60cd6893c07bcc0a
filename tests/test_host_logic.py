"""CPU tier: host-side option resolution (preset known answers from the reference's own unit tests),
planner routing, and that the C-ABI library loads and exports every symbol include/mi355fft.h declares."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from mi355fft import _abi, layout


def test_preset_known_answers(manifest):
    cases, _ = manifest
    ka = cases["preset_known_answers"]
    for ok in ka["ok"]:
        got = getattr(layout, ok["fn"])(ok["opts"])
        assert got == ok["expect"], ok["fn"]
    for lf in ka["layout_forced"]:
        opts = dict(ka["layout_forced_base"], outputLayout=lf["outputLayoutIn"])
        assert getattr(layout, lf["fn"])(opts)["fftConv"]["outputLayout"] == lf["expect"]
    for th in ka["throws"]:
        with pytest.raises(ValueError, match=th["regex"]):
            layout.createFftConvChannelLanePreset(th["opts"])


def test_cfg4_descriptor_resolution():
    """the README example's resolved strides/offsets"""
    p = layout.createFftConvKernelMajorChannelLanePreset({"shape": [256], "batch": 4, "kernelCount": 3, "input": {"channels": 64},
                                                          "output": {"channels": 128, "kernelStepChannels": 16}})
    r = layout.resolve_plan_options(dict(p, type="fftconv"))
    assert r["input_layout"] == {"strides": [1], "offset": 0, "batch_stride": 16384}
    assert r["output_layout"] == {"strides": [1], "offset": 0, "batch_stride": 32768}
    assert r["conv"]["outputKernelStrideElements"] == 4096 and r["conv"]["outputLayout"] == "kernel-major"


def test_option_validation_messages():
    bad = [
        ({"type": "c2c", "shape": [], "direction": "forward"}, "shape must be an array"),
        ({"type": "c2c", "shape": [8, 0], "direction": "forward"}, "positive ints"),
        ({"type": "c2c", "shape": [8], "direction": "sideways"}, "direction must be one of"),
        ({"type": "c2c", "shape": [8], "direction": "forward", "normalize": "ortho"}, "normalize must be one of"),
        ({"type": "c2c", "shape": [8], "direction": "forward", "batch": 0}, "batch must be positive int"),
        ({"type": "c2c", "shape": [8], "direction": "forward", "layout": {"interleavedComplex": False}}, "interleavedComplex"),
        ({"type": "r2c", "shape": [8], "direction": "inverse"}, 'r2c supports direction:"forward" only'),
        ({"type": "c2r", "shape": [8], "direction": "forward"}, 'c2r supports direction:"inverse" only'),
        ({"type": "r2c", "shape": [8], "direction": "forward", "inPlace": True}, "only on c2c"),
        ({"type": "fftconv", "shape": [8], "fftConv": {"kernelShape": [9]}}, "must be <= shape"),
        ({"type": "fftconv", "shape": [8], "fftConv": {"kernelCount": 0}}, "kernelCount must be a positive integer"),
        ({"type": "fftconv", "shape": [8], "fftConv": {"channelPolicy": {"input": {"channels": 2}}}, "layout": {"interleavedComplex": True, "whdcn": {"channels": 2}}},
         "cannot be combined with layout.whdcn"),
        ({"type": "bogus", "shape": [8]}, "type must be one of"),
    ]
    for opts, frag in bad:
        with pytest.raises(ValueError, match=re.escape(frag)):
            layout.resolve_plan_options(opts)
    with pytest.raises(NotImplementedError):
        layout.resolve_plan_options({"type": "conv2d", "shape": [8]})
    assert layout.resolve_plan_options({"type": "dct2", "shape": [8], "direction": "forward"})["type"] == "dct2"   # real buffers by default


def test_whdcn_resolution():
    r = layout.resolve_plan_options({"type": "c2c", "shape": [8, 4], "batch": 2, "direction": "forward",
                                     "layout": {"interleavedComplex": True, "whdcn": {"channels": 3, "channelIndex": 2}}})
    assert r["input_layout"] == {"strides": [1, 8], "offset": 64, "batch_stride": 96} == r["output_layout"]
    r = layout.resolve_plan_options({"type": "c2c", "shape": [8], "direction": "forward",
                                     "layout": {"interleavedComplex": True, "whdcn": {"channels": 1}}})
    assert r["input_layout"] is None  # no-op descriptor resolves to dense


def test_normalize_scale_factor(manifest):
    import struct
    cases, _ = manifest
    for row in cases["normalize_scale"]["rows"]:
        want = struct.unpack("<d", bytes.fromhex(row["value"]))[0]
        assert layout.normalizeScaleFactor(row["normalize"], row["direction"], row["nTotal"]) == want


def test_library_exports_every_declared_symbol():
    """builds nothing: the .so travels with the snapshot (see __graft_entry__.build)"""
    import mi355fft
    if not os.path.exists(mi355fft.LIB_PATH):
        pytest.skip("libmi355fft.so not built in this checkout (run __graft_entry__.build())")
    header = open(os.path.join(ROOT, "include", "mi355fft.h")).read()
    declared = set(re.findall(r"\b(mi355fft_[a-z0-9_]+)\s*\(", header))
    declared -= {"mi355fft_plan_desc", "mi355fft_exec_args", "mi355fft_side_layout"}
    assert declared == set(_abi.ABI_SYMBOLS), declared ^ set(_abi.ABI_SYMBOLS)
    L = ctypes.CDLL(mi355fft.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(L, name), name
    _abi.declare(L)
    assert L.mi355fft_abi_version() == 1
    assert ctypes.sizeof(_abi.PlanDesc) == 912 and ctypes.sizeof(_abi.ExecArgs) == 64


def test_no_cpu_fallback_without_a_gpu():
    """the product path must fail loudly when there is no HIP device"""
    import mi355fft
    if os.path.exists("/dev/kfd") or not os.path.exists(mi355fft.LIB_PATH):
        pytest.skip("needs a GPU-less host with the library built")
    with pytest.raises(mi355fft.Mi355Error) as e:
        mi355fft.Device(0)
    assert e.value.code == _abi.ERR_HIP


def test_strided_axis_beyond_32_bit_span_stays_off_the_column_line_kernels():
    """kern_lines.hpp addresses a column tile with 32-bit element offsets (idx * S): an axis whose plane spans >= 2^32 elements
    must not take that route (it would wrap silently); the stage route indexes with 64 bits.  Planner only — nothing is run."""
    import emu_harness as emu
    from mi355fft import _abi
    small = _abi.make_desc("c2c", [1 << 16, 1024], 1, "forward", "none")          # axis 1: N = 1024 over S = 2^16 -> 2^26 elements
    route, _, _ = emu.plan_only(small)
    assert "columns[N=1024,S=65536]" in route, route
    big = _abi.make_desc("c2c", [1 << 22, 1024], 1, "forward", "none")            # axis 1: N = 1024 over S = 2^22 -> 2^32 elements (32 GiB)
    route, _, _ = emu.plan_only(big)
    assert "columns[" not in route and "stages[" in route, route


def test_xcd_resident_route_is_opt_in_and_needs_whole_xcds(monkeypatch):
    import emu_harness as emu
    from mi355fft import _abi
    d = _abi.make_desc("c2c", [1 << 20], 64, "forward", "none")
    assert emu.plan_only(d)[0].startswith("xcd-fused-rt32[N=1024x1024]")
    monkeypatch.setenv("MI355FFT_XCD_RES", "1")
    route, launches, work = emu.plan_only(d)
    assert route.startswith("xcd-resident[N=1024x1024,depth=4]") and launches == 2 and work >= 16 * 4 * (1 << 20)
    assert emu.plan_only(d, compute_units=250)[0].startswith("xcd-fused")      # not a multiple of 32 CUs: no resident groups

"""CPU tier: fftconv plans (planner + kernels under host emulation) against the reference's fftConvRef
fixtures (tests/golden, generated from src/utils/math.js:469-603) and the oracle, including the
channel-lane sentinel test that mirrors test/complete.suite.js:4715-4836."""
import os

import numpy as np
import pytest

import emu_harness as emu
from conftest import GOLDEN
from mi355fft import _abi
from mi355fft.layout import createFftConvKernelMajorChannelLanePreset, resolve_plan_options


def _desc(opts):
    r = resolve_plan_options(opts)
    return _abi.make_desc(r["type"], r["shape"], r["batch"], r["direction"], r["normalize"], r["inPlace"], r["input_layout"],
                          r["output_layout"], r["conv"], None, r["zero_pad"]), r


def _close(got, want, atol, rtol, what):
    from oracle import oracle as orc
    orc.assert_close_elementwise(got, want, atol, rtol, what)


@pytest.mark.parametrize("layout", ["kernel-major", "batch-major"])
def test_fftconv_golden_fixtures(manifest, oracle, layout):
    cases, _ = manifest
    ran = 0
    for c in cases.values():
        if c["kind"] != "fftconv":
            continue
        shape, batch, K = c["shape"], c["batch"], c["kernelCount"]
        ks = c["kernelShape"] or shape
        fft_shape = shape if c["boundary"] == "circular" else [s + k - 1 for s, k in zip(shape, ks)]
        n, kn = int(np.prod(shape)), int(np.prod(ks))
        x = oracle.random_complex_interleaved(n * batch, c["seed"])
        kern = oracle.random_complex_interleaved(kn * K, c["kernel_seed"])
        want = np.fromfile(os.path.join(GOLDEN, c["out_file"]), dtype=np.float32).reshape(K, batch, -1)
        desc, r = _desc({"type": "fftconv", "shape": shape, "batch": batch,
                         "fftConv": {"mode": c["mode"], "boundary": c["boundary"], "kernelCount": K, "kernelShape": c["kernelShape"],
                                     "outputLayout": layout}})
        got, route, _ = emu.run_plan(desc, x, want.size, kernel=kern)
        if layout == "batch-major":
            want = want.transpose(1, 0, 2)
        _close(got, want.reshape(-1), 4e-3, 4e-3, c["name"])   # the reference's own tolerance (complete.suite.js:4663)
        assert oracle.rel_l2(got, want.reshape(-1)) < 1e-5, c["name"]
        ran += 1
    assert ran >= 10


def _make_strided_physical(shape, batch, offset, batch_stride, logical, total_elems, fill=(0.0, 0.0)):
    n = int(np.prod(shape))
    phys = np.empty(2 * total_elems, np.float32)
    phys[0::2], phys[1::2] = fill
    for b in range(batch):
        base = offset + b * batch_stride
        phys[2 * base:2 * (base + n)] = logical[2 * b * n:2 * (b + 1) * n]
    return phys


def test_fftconv_channel_policy_lanes_preserve_sentinels(oracle):
    """mirror of complete.suite.js:4715-4836 (N=12, batch=2, kernels=2, batch-major, lanes + sentinels)"""
    shape, batch, K, n = [12], 2, 2, 12
    in_ch, in_idx, in_cs, in_bs = 3, 1, 16, 80
    out_ch, out_idx, out_cs, out_bs, step = 5, 1, 20, 160, 1
    out_last = (out_idx + (K - 1) * step) * out_cs
    out_elems = out_last + (batch - 1) * out_bs + n
    logical = oracle.random_complex_interleaved(n * batch, 4242)
    in_elems = in_idx * in_cs + (batch - 1) * in_bs + n
    phys_in = _make_strided_physical(shape, batch, in_idx * in_cs, in_bs, logical, in_elems, fill=(9.0, 9.0))
    kernels = oracle.random_complex_interleaved(n * K, 4343)
    sentinel = np.empty(2 * out_elems, np.float32)
    sentinel[0::2], sentinel[1::2] = 77.0, -55.0
    desc, r = _desc({"type": "fftconv", "shape": shape, "batch": batch, "layout": {"interleavedComplex": True}, "precision": "f32",
                     "fftConv": {"mode": "convolution", "kernelCount": K, "outputLayout": "batch-major",
                                 "channelPolicy": {"input": {"channels": in_ch, "channelIndex": in_idx, "channelStrideElements": in_cs,
                                                             "batchStrideElements": in_bs},
                                                   "output": {"channels": out_ch, "channelIndex": out_idx, "channelStrideElements": out_cs,
                                                              "batchStrideElements": out_bs, "kernelStepChannels": step}}}})
    assert r["input_layout"] == {"strides": [1], "offset": in_idx * in_cs, "batch_stride": in_bs}
    assert r["output_layout"] == {"strides": [1], "offset": out_idx * out_cs, "batch_stride": out_bs}
    assert r["conv"]["outputKernelStrideElements"] == out_cs * step
    got, _, _ = emu.run_plan(desc, phys_in, sentinel.size, kernel=kernels, out_init=sentinel)
    want = sentinel.copy()
    for k in range(K):
        cpu, _ = oracle.fftconv_ref(logical, kernels[2 * k * n:2 * (k + 1) * n], shape, batch, "convolution")
        for b in range(batch):
            lane = (out_idx + k * step) * out_cs + b * out_bs
            want[2 * lane:2 * (lane + n)] = cpu[2 * b * n:2 * (b + 1) * n]
    _close(got, want, 5e-3, 5e-3, "channelPolicy lanes")
    untouched = np.ones(out_elems, bool)
    for k in range(K):
        for b in range(batch):
            lane = (out_idx + k * step) * out_cs + b * out_bs
            untouched[lane:lane + n] = False
    assert np.array_equal(got.reshape(-1, 2)[untouched], sentinel.reshape(-1, 2)[untouched]), "elements outside the lanes were written"


def test_fftconv_cfg4_preset_against_golden(manifest, oracle):
    """BASELINE config 4: shape=[256] batch=4, in=64ch out=128ch, 3 kernels, kernel-major channel-lane preset"""
    cases, _ = manifest
    c = cases["fftconv_cfg4_N256_b4_k3"]
    preset = createFftConvKernelMajorChannelLanePreset({"shape": [256], "batch": 4, "kernelCount": 3, "input": {"channels": 64},
                                                        "output": {"channels": 128, "kernelStepChannels": 16}})
    opts = dict(preset, type="fftconv")
    desc, r = _desc(opts)
    n, batch, K = 256, 4, 3
    logical = oracle.random_complex_interleaved(n * batch, c["seed"])
    kern = oracle.random_complex_interleaved(n * K, c["kernel_seed"])
    phys_in = _make_strided_physical([n], batch, 0, 64 * 256, logical, 4 * 64 * 256, fill=(5.0, -5.0))
    out_elems = 4 * 128 * 256
    sentinel = np.empty(2 * out_elems, np.float32)
    sentinel[0::2], sentinel[1::2] = 77.0, -55.0
    got, route, launches = emu.run_plan(desc, phys_in, sentinel.size, kernel=kern, out_init=sentinel)
    assert route.startswith("fftconv-fused[N=256,K=3") and launches == 1      # the reference: ~K*(3 plans x 3 passes) + per-element copies
    gold = np.fromfile(os.path.join(GOLDEN, c["out_file"]), dtype=np.float32).reshape(K, batch, 2 * n)
    want = sentinel.copy()
    for k in range(K):
        for b in range(batch):
            lane = (0 + k * 16) * 256 + b * 32768
            want[2 * lane:2 * (lane + n)] = gold[k, b]
    _close(got, want, 4e-3, 4e-3, "cfg4 lanes")
    lanes = np.zeros(out_elems, bool)
    for k in range(K):
        for b in range(batch):
            lane = k * 16 * 256 + b * 32768
            lanes[lane:lane + n] = True
    assert oracle.rel_l2(got.reshape(-1, 2)[lanes], want.reshape(-1, 2)[lanes]) < 1e-5
    assert np.array_equal(got.reshape(-1, 2)[~lanes], sentinel.reshape(-1, 2)[~lanes])


@pytest.mark.parametrize("n,K,batch,mode,layout,klen", [(64, 2, 3, "correlation", "batch-major", 64), (128, 5, 2, "convolution", "kernel-major", 128),
                                                         (256, 15, 2, "convolution", "batch-major", 256), (512, 1, 5, "correlation", "kernel-major", 7),
                                                         (1024, 3, 2, "convolution", "kernel-major", 100)])
def test_fftconv_fused_matches_composed_route_and_oracle(oracle, n, K, batch, mode, layout, klen):
    x = oracle.random_complex_interleaved(n * batch, 900 + n)
    kern = oracle.random_complex_interleaved(klen * K, 901 + n)
    opts = {"type": "fftconv", "shape": [n], "batch": batch, "fftConv": {"mode": mode, "kernelCount": K, "outputLayout": layout, "kernelShape": [klen]}}
    desc, _ = _desc(opts)
    fused, route, launches = emu.run_plan(desc, x, 2 * n * batch * K, kernel=kern)
    assert route.startswith("fftconv-fused[") and launches == 1
    composed, route2, launches2 = emu.run_plan(desc, x, 2 * n * batch * K, kernel=kern, force_generic=True)
    assert "fftconv[K=" in route2 and launches2 > 1
    assert oracle.rel_l2(fused, composed) < 2e-6
    want = np.empty((K, batch, 2 * n), np.float32)
    for k in range(K):
        ref, _ = oracle.fftconv_ref(x, kern[2 * k * klen:2 * (k + 1) * klen], [n], batch, mode, "circular", [klen], use_pow2=True)
        want[k] = ref.reshape(batch, 2 * n)
    if layout == "batch-major":
        want = want.transpose(1, 0, 2)
    _close(fused, want.reshape(-1), 4e-3, 4e-3, f"fused N={n} K={K}")
    assert oracle.rel_l2(fused, want.reshape(-1)) < 1e-5


@pytest.mark.parametrize("shape,kshape,boundary,zero_pad", [
    ([16], None, "circular", {"read": {"start": [2], "end": [12]}, "write": {"start": [4], "end": [16]}}),
    ([64], [5], "circular", {"read": {"start": [8], "end": [60]}}),                      # would take the fused kernel without zeroPad
    ([12, 5], [3, 2], "linear-same", {"read": {"start": [1, 0], "end": [10, 5]}, "write": {"start": [2, 1], "end": [13, 5]}}),
    ([10], [4], "linear-valid", {"write": {"start": [0], "end": [8]}}),
    ([10], [4], "linear-full", {"write": {"start": [3], "end": [11]}}),
    ([100], [29], "linear-full", {"read": {"start": [3], "end": [90]}, "write": {"start": [3], "end": [111]}}),       # FFT domain 128
    ([30, 6], [3, 3], "linear-same", {"read": {"start": [1, 0], "end": [30, 5]}, "write": {"start": [2, 1], "end": [31, 8]}}),   # 32 x 8
])
@pytest.mark.parametrize("fuse", [1, 0])
def test_fftconv_zero_pad(oracle, monkeypatch, shape, kshape, boundary, zero_pad, fuse):
    """zeroPad ranges live on the FFT domain (fftconv.js:353,386): read zeroes the embedded data before the forward
    transform, write zeroes the inverse transform before the crop.  fuse=1 (default): with a power-of-two FFT domain the embed, the
    zero ranges and the crop are the address maps of the first forward / last inverse line-kernel launch"""
    monkeypatch.setenv("MI355_EMU_FUSE_VIEWS", str(fuse))
    batch, K, rank = 2, 2, len(shape)
    ks = kshape or shape
    n, kn = int(np.prod(shape)), int(np.prod(ks))
    x = oracle.random_complex_interleaved(n * batch, 5150 + n)
    kern = oracle.random_complex_interleaved(kn * K, 5151 + n)
    desc, r = _desc({"type": "fftconv", "shape": shape, "batch": batch, "zeroPad": zero_pad,
                     "fftConv": {"boundary": boundary, "kernelCount": K, "kernelShape": kshape}})
    zr, zw = r["zero_pad"]["read"], r["zero_pad"]["write"]
    if boundary == "circular":
        oshape, ooff = list(shape), [0] * rank
    elif boundary == "linear-full":
        oshape, ooff = [s + k - 1 for s, k in zip(shape, ks)], [0] * rank
    elif boundary == "linear-same":
        oshape, ooff = list(shape), [(k - 1) // 2 for k in ks]
    else:
        oshape, ooff = [s - k + 1 for s, k in zip(shape, ks)], [k - 1 for k in ks]
    xin = x.reshape(batch, *reversed(shape), 2).copy()
    if zr:      # the data sits at the origin of the domain: domain coordinates = input coordinates
        keep = np.zeros(tuple(reversed(shape)), bool)
        keep[tuple(slice(zr["start"][d], min(zr["end"][d], shape[d])) for d in reversed(range(rank)))] = True
        xin[:, ~keep] = 0
    on = int(np.prod(oshape))
    want = np.zeros((K, batch, *reversed(oshape), 2), np.float32)
    for k in range(K):
        y, _ = oracle.fftconv_ref(xin.reshape(-1), kern[2 * k * kn:2 * (k + 1) * kn], shape, batch, "convolution", boundary, kshape)
        want[k] = np.asarray(y, np.float32).reshape(batch, *reversed(oshape), 2)
    if zw:      # output coordinate o is domain coordinate o + ooff
        keep = np.zeros(tuple(reversed(oshape)), bool)
        sl = tuple(slice(max(0, zw["start"][d] - ooff[d]), max(0, min(oshape[d], zw["end"][d] - ooff[d]))) for d in reversed(range(rank)))
        keep[sl] = True
        want[:, :, ~keep] = 0
    got, route, _ = emu.run_plan(desc, x, 2 * on * K * batch, kernel=kern)
    assert not route.startswith("fftconv-fused")
    fs = list(shape) if boundary == "circular" else [s_ + k_ - 1 for s_, k_ in zip(shape, ks)]
    if fuse and all(f & (f - 1) == 0 for f in fs):
        assert "mapped[" in route and not [w for w in ("gather", "scatter", "zero-read", "zero-write") if w in route], route
    else:
        assert ("zero-read" in route) == bool(zr) and ("zero-write" in route) == bool(zw), route
    _close(got, want.reshape(-1), 4e-3, 4e-3, f"fftconv zeroPad {shape} {boundary}")
    assert oracle.rel_l2(got, want.reshape(-1)) < 1e-5


@pytest.mark.parametrize("shape,ks,boundary,mode,K", [([8192], None, "circular", "convolution", 2), ([8192], [100], "circular", "correlation", 1),
                                                       ([100], [29], "linear-full", "convolution", 3), ([1000], [25], "linear-same", "correlation", 2),
                                                       ([4000], [97], "linear-valid", "convolution", 1)])
def test_fftconv_product_fused_into_forward_lines(oracle, monkeypatch, shape, ks, boundary, mode, K):
    """1-D, power-of-two FFT length: kernel-spectrum product behind the forward line FFT's last stage (fft_lines_mul_kernel), one
    launch per kernel; against the oracle's fftConvRef restatement and against the forward + pointwise route"""
    batch = 5 if shape[0] < 8192 else 2
    n, kn = shape[0], (ks or shape)[0]
    x = oracle.random_complex_interleaved(n * batch, 0xC0DE + n)
    kern = oracle.random_complex_interleaved(kn * K, 0xC1DE + kn)
    desc, _ = _desc({"type": "fftconv", "shape": shape, "batch": batch,
                     "fftConv": {"mode": mode, "boundary": boundary, "kernelCount": K, "kernelShape": ks}})
    want = np.concatenate([oracle.fftconv_ref(x, kern[2 * k * kn:2 * (k + 1) * kn], shape, batch, mode, boundary, ks)[0] for k in range(K)])
    got, route, launches = emu.run_plan(desc, x, want.size, kernel=kern)
    # linear modes: the zero-padded embed of kernels and data and the crop ride the launches too (kernel FFT, K x (forward-mul, inverse))
    assert ("lines-mul-mapped[" if boundary != "circular" else "lines-mul[") in route, route
    assert launches == 1 + 2 * K, route
    _close(got, want, 4e-3, 4e-3, route)
    assert oracle.rel_l2(got, want) < 1e-5, route
    monkeypatch.setenv("MI355_EMU_CONV_LINES", "0")
    old, route0, _ = emu.run_plan(desc, x, want.size, kernel=kern)
    assert "lines-mul" not in route0, route0
    assert oracle.rel_l2(got, old) < 1e-6


@pytest.mark.parametrize("ks,mode,K,layout,cus", [(None, "convolution", 1, "kernel-major", 2), ([1000], "correlation", 2, "batch-major", 3)])
def test_fftconv_pipeline_2p20(oracle, monkeypatch, ks, mode, K, layout, cus):
    """2^20-point circular lines: forward transform, product with the kernel spectra and inverse transforms in ONE persistent launch
    (kern_regtile.hpp fft_xcd_conv1m_kernel: the spectrum tile of the forward pass B is the input tile of the inverse pass A and never
    leaves the registers); K = 2 re-runs the middle and last phase per kernel.  Against the oracle's fftConvRef restatement."""
    monkeypatch.setenv("MI355_EMU_XCD_FUSED", "1")
    monkeypatch.setenv("MI355_EMU_CUS", str(cus))
    monkeypatch.setenv("MI355_EMU_XCDS", "1")
    n, batch = 1 << 20, 2
    kn = (ks or [n])[0]
    x = oracle.random_complex_interleaved(n * batch, 0xC4DE)
    kern = oracle.random_complex_interleaved(kn * K, 0xC5DE)
    desc, _ = _desc({"type": "fftconv", "shape": [n], "batch": batch,
                     "fftConv": {"mode": mode, "boundary": "circular", "kernelCount": K, "kernelShape": ks, "outputLayout": layout}})
    per = [oracle.fftconv_ref(x, kern[2 * k * kn:2 * (k + 1) * kn], [n], batch, mode, "circular", ks, use_pow2=True)[0].reshape(batch, 2 * n) for k in range(K)]
    want = (np.concatenate(per) if layout == "kernel-major" else np.stack(per, axis=1)).reshape(-1)
    got, route, launches = emu.run_plan(desc, x, want.size, kernel=kern)
    assert "fftconv-pipeline[N=1024x1024,K=%d]" % K in route, route
    _close(got, want, 4e-3, 4e-3, route)
    assert oracle.rel_l2(got, want) < 1e-5, route


def test_fftconv_fused_route_gives_way_to_line_kernels_at_throughput_sizes(oracle, monkeypatch):
    """the one-launch fftconv kernel is the latency route; above conv_fused_max_points the forward-mul + inverse line launches take
    over (same results)"""
    shape, batch, K = [256], 6, 2
    x = oracle.random_complex_interleaved(256 * batch, 0xC2DE)
    kern = oracle.random_complex_interleaved(256 * K, 0xC3DE)
    desc, _ = _desc({"type": "fftconv", "shape": shape, "batch": batch, "fftConv": {"mode": "convolution", "boundary": "circular", "kernelCount": K}})
    a, route_a, launches_a = emu.run_plan(desc, x, 2 * 256 * batch * K, kernel=kern)
    assert route_a.startswith("fftconv-fused[") and launches_a == 1, route_a
    monkeypatch.setenv("MI355_EMU_CONV_FUSED_MAX_POINTS", "1024")
    b, route_b, _ = emu.run_plan(desc, x, 2 * 256 * batch * K, kernel=kern)
    assert "lines-mul[N=256]" in route_b and "fftconv-fused" not in route_b, route_b
    assert oracle.rel_l2(a, b) < 1e-6

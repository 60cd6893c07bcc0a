"""CPU tier: the HIP kernel sources compiled for the host (tests/emu) + the planner, checked against the
oracle on seeded inputs.  Validates index math, LDS layouts, barrier placement, twiddle tables and
routing without a GPU; the -m gpu tests repeat the same comparisons on the real kernels."""
import numpy as np
import pytest

import emu_harness as emu
from mi355fft import _abi

TOL = 1e-5  # north_star: 1e-5 relative (norm-relative)


def check(got, want, what, tol=TOL):
    from oracle import oracle as orc
    l2, mx = orc.rel_l2(got, want), orc.rel_max(got, want)
    assert l2 <= tol and mx <= tol, f"{what}: rel_l2={l2:.3e} rel_max={mx:.3e}"


def test_registry_matches_device_constants():
    import ctypes
    msg = ctypes.create_string_buffer(256)
    assert emu.lib().emu_check_registry(msg, 256) == 0, msg.value


@pytest.mark.parametrize("n", [2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096])
@pytest.mark.parametrize("direction", ["forward", "inverse"])
def test_c2c_row_lines(oracle, n, direction):
    batch = 5 if n <= 1024 else 3          # not a multiple of the tile: exercises the masked tail lines
    x = oracle.random_complex_batch(n, batch, 0xA000 + n).reshape(-1)
    norm = "unitary" if direction == "forward" else "backward"
    desc = _abi.make_desc("c2c", [n], batch, direction, norm)
    got, route, launches = emu.run_plan(desc, x, x.size)
    assert route.startswith("lines[") and launches == 1
    check(got, oracle.c2c_ref_batch(x, [n], batch, direction, norm), f"c2c N={n} {direction}")


def test_c2c_in_place(oracle):
    x = oracle.random_complex_batch(256, 3, 7).reshape(-1)
    desc = _abi.make_desc("c2c", [256], 3, "forward", "none", in_place=True)
    got, _, _ = emu.run_plan(desc, x, x.size)
    check(got, oracle.c2c_ref_batch(x, [256], 3, "forward"), "in-place")


@pytest.mark.parametrize("lg,direction", [(13, "forward"), (13, "inverse"), (14, "forward"), (15, "inverse"), (16, "forward"), (17, "forward")])
def test_c2c_two_pass(oracle, monkeypatch, lg, direction):
    monkeypatch.setenv("MI355_EMU_MAX_LINE", "4096")      # 2^13 and 2^14 would otherwise run as single-workgroup lines
    n, batch = 1 << lg, 3
    x = oracle.random_complex_batch(n, batch, 0xB000 + lg).reshape(-1)
    desc = _abi.make_desc("c2c", [n], batch, direction, "backward")
    got, route, launches = emu.run_plan(desc, x, x.size, chunk_bytes=2 * n * 8)   # 2 transforms per chunk -> 2 chunks
    assert route.startswith("two-pass[") and launches == 4
    check(got, oracle.c2c_ref_batch(x, [n], batch, direction, "backward"), f"two-pass 2^{lg} {direction}")


def test_c2c_line32k(oracle, monkeypatch):
    """N = 2^15 in one workgroup (kern_line32k.hpp): the line in the registers of 512 threads, both Stockham exchanges through LDS
    in two halves; more lines than (emulated) workgroups; MI355_EMU_LINE32K=0 is the four-step route it replaces"""
    n, batch = 1 << 15, 5
    x = oracle.random_complex_batch(n, batch, 0xB320).reshape(-1)
    for direction, norm in (("forward", "none"), ("inverse", "backward"), ("forward", "unitary")):
        desc = _abi.make_desc("c2c", [n], batch, direction, norm)
        got, route, launches = emu.run_plan(desc, x, x.size)
        assert route.split() == ["line32k[N=32768]"] and launches == 1, route
        check(got, oracle.c2c_ref_batch(x, [n], batch, direction, norm), f"line32k {direction} {norm}")
    monkeypatch.setenv("MI355FFT_LINE32K", "0")          # (plan_only reads the product planner's environment)
    assert emu.plan_only(_abi.make_desc("c2c", [n], batch, "forward", "none"))[0].startswith("xcd-solo[N=128x256]")
    monkeypatch.delenv("MI355FFT_LINE32K")
    # the real transforms of 2^16 points ride it through the half-length route
    m = 1 << 16
    xr = oracle.random_real_batch(m, 2, 0xB321).reshape(-1)
    want = np.concatenate([oracle.r2c_ref_packed(xr[b * m:(b + 1) * m], m, "none") for b in range(2)])
    got, route, _ = emu.run_plan(_abi.make_desc("r2c", [m], 2, "forward", "none"), xr, want.size)
    assert route.split() == ["line32k[N=32768]", "r2c-split"], route
    check(got, want, "r2c 2^16 over line32k", 1e-5)
    back, route, _ = emu.run_plan(_abi.make_desc("c2r", [m], 2, "inverse", "backward"), want, m * 2)
    assert "line32k[N=32768]" in route and "c2r-split" in route, route
    check(back, xr, "c2r 2^16 over line32k", 1e-5)


@pytest.mark.parametrize("lg,grid", [(13, 4), (14, 8), (15, 8)])
def test_c2c_two_pass_hoisted_fourstep_roots(oracle, lg, grid, monkeypatch):
    """grid*T a multiple of N1: pass B computes its four-step roots once per launch (loop-invariant registers)"""
    monkeypatch.setenv("MI355_EMU_MAX_GRID", str(grid))
    monkeypatch.setenv("MI355_EMU_MAX_LINE", "4096")
    n, batch = 1 << lg, 5
    x = oracle.random_complex_batch(n, batch, 0xB100 + lg).reshape(-1)
    for direction in ("forward", "inverse"):
        desc = _abi.make_desc("c2c", [n], batch, direction, "none")
        got, route, _ = emu.run_plan(desc, x, x.size)
        assert route.startswith("two-pass[")
        check(got, oracle.c2c_ref_batch(x, [n], batch, direction, "none"), f"hoisted 2^{lg} {direction}")


def test_c2c_two_pass_2p22(oracle):
    """the largest power of two of the route: 2048 x 2048, three-stage column tiles of 8 in pass A"""
    n = 1 << 22
    x = oracle.random_complex_batch(n, 1, 0xB222).reshape(-1)
    for direction in ("forward", "inverse"):
        desc = _abi.make_desc("c2c", [n], 1, direction, "none")
        got, route, launches = emu.run_plan(desc, x, x.size)
        assert route.startswith("two-pass[N=2048x2048") and launches == 2, route
        check(got, oracle.c2c_ref_batch(x, [n], 1, direction, "none"), f"two-pass 2^22 {direction}")


@pytest.mark.parametrize("mixed", [0, 1])
@pytest.mark.parametrize("n", [3, 5, 6, 7, 11, 12, 13, 15, 21, 24, 96, 105, 210, 1001, 8 * 13 * 11, 2187, 3 * 1024, 4095])
def test_c2c_generic_mixed_radix(oracle, monkeypatch, n, mixed):
    """mixed-radix lengths on the one-launch LDS line kernel (kern_mixed.hpp) and on the global-memory stage route"""
    monkeypatch.setenv("MI355_EMU_MIXED_LINES", str(2 * mixed))   # 2: also where the planner would keep the stage route
    monkeypatch.setenv("MI355_EMU_MIXED_CT", "0")                 # the compile-time-plan instances have their own test below
    batch = 5 if n < 200 else 2
    x = oracle.random_complex_batch(n, batch, 0xC000 + n).reshape(-1)
    for direction, norm in (("forward", "none"), ("inverse", "backward")):
        desc = _abi.make_desc("c2c", [n], batch, direction, norm)
        got, route, launches = emu.run_plan(desc, x, x.size)
        if mixed and n not in (3, 5, 7, 11, 13):
            assert route.startswith("mixed-lines[") and launches == 1, route
        else:
            assert route.startswith("stages[")
        check(got, oracle.c2c_ref_batch(x, [n], batch, direction, norm), f"N={n} {direction} {route}", 2e-6 if n < 1000 else 1e-5)


MIXED_CT = [96, 192, 384, 768, 1536, 3072, 160, 320, 640, 1280, 2560, 1000, 2000, 3000, 105, 1001, 360, 1920, 2187, 500, 1500, 120, 240, 480, 720, 1440]


@pytest.mark.parametrize("n", MIXED_CT)
def test_c2c_mixed_radix_compile_time_plans(oracle, n):
    """every instance of kern_mixed_ct.hpp (MI355_MIXEDCT_LIST in plan.hpp): lengths 3*2^k, 5*2^k, 1000 and the reference's
    mixed-radix test sizes with radices, tile shape and thread count as template constants; batches that leave a ragged last tile"""
    from mi355fft import _abi as abi
    batch = {True: 37, False: 3}[n < 400]
    x = oracle.random_complex_batch(n, batch, 0xC700 + n).reshape(-1)
    for direction, norm in (("forward", "unitary"), ("inverse", "backward")):
        desc = abi.make_desc("c2c", [n], batch, direction, norm)
        got, route, launches = emu.run_plan(desc, x, x.size)
        assert route.startswith("mixed-ct[N=%d," % n) and launches == 1, route
        check(got, oracle.c2c_ref_batch(x, [n], batch, direction, norm), f"N={n} {direction} {route}", 2e-6 if n < 1000 else 1e-5)


@pytest.mark.parametrize("shape,expect", [([96, 105], 1), ([24, 25, 27], 3), ([6, 10, 4], 2), ([1000, 3], 0)])
def test_c2c_nd_mixed_radix_lines(oracle, shape, expect):
    """N-D shapes of the reference's suites (complete.suite.js:876-913): every axis, contiguous or strided, in one launch"""
    batch = 2
    n = int(np.prod(shape))
    x = oracle.random_complex_batch(n, batch, 0xC500 + n).reshape(-1)
    for direction, norm in (("forward", "none"), ("inverse", "backward")):
        desc = _abi.make_desc("c2c", shape, batch, direction, norm)
        got, route, launches = emu.run_plan(desc, x, x.size)
        assert route.count("mixed-lines[") == expect, route   # power-of-two and single-radix axes keep their routes
        check(got, oracle.c2c_ref_batch(x, shape, batch, direction, norm), f"{shape} {direction} {route}", 1e-5)


def _fft_axes_ref(oracle, x, shape, batch, axes, direction, scale):
    """oracle 1-D transforms along the chosen axes only (axis 0 fastest = last numpy axis before re/im)"""
    rank = len(shape)
    a = x.reshape(batch, *reversed(shape), 2).copy()
    for ax in sorted(set(axes)):
        npax = rank - ax                           # numpy axis of logical axis ax (batch is numpy axis 0)
        m = np.moveaxis(a, npax, -2)
        lines = np.ascontiguousarray(m).reshape(-1, shape[ax], 2)
        out = oracle.c2c_ref_batch(lines.reshape(-1), [shape[ax]], lines.shape[0], direction, "none").reshape(m.shape)
        a = np.moveaxis(out, -2, npax)
    return (np.ascontiguousarray(a) * np.float32(scale)).reshape(-1)


@pytest.mark.parametrize("shape,axes", [([16, 8], [0]), ([16, 8], [1]), ([8, 4, 6], [0, 2]), ([8, 4, 6], [1]), ([32, 3], [1, 0])])
def test_c2c_axes_subset(oracle, shape, axes):
    """createFftPlan({axes}) (plan.js:1335-1339): only the listed axes are transformed; the unitary factor keeps prod(shape)"""
    batch = 2
    n = int(np.prod(shape))
    x = oracle.random_complex_batch(n, batch, 0xA7E5 + n).reshape(-1)
    for direction, norm, scale in (("forward", "none", 1.0), ("inverse", "backward", 1.0 / n), ("forward", "unitary", 1.0 / np.sqrt(n))):
        desc = _abi.make_desc("c2c", shape, batch, direction, norm, axes=axes)
        got, route, _ = emu.run_plan(desc, x, x.size)
        check(got, _fft_axes_ref(oracle, x, shape, batch, axes, direction, scale), f"axes={axes} {direction} {route}", 2e-6)
    desc = _abi.make_desc("c2c", shape, batch, "forward", "none", in_place=True, axes=axes)
    got, _, _ = emu.run_plan(desc, x, x.size)
    check(got, _fft_axes_ref(oracle, x, shape, batch, axes, "forward", 1.0), f"axes={axes} in place", 2e-6)


@pytest.mark.parametrize("n,reg", [(8192, 1), (8192, 0), (16384, 0), (16384, 2)])
def test_c2c_single_workgroup_long_lines(oracle, monkeypatch, n, reg):
    """N = 8192 and 16384 still fit one workgroup's LDS (three stages, the last table read from global memory): one launch,
    one HBM round trip; r2c / c2r of twice the length ride on them.  reg: the register-resident form of the same launch
    (kern_line_reg.hpp: N/64 threads, exchanges through LDS in halves) — the default at 8192, opt-in (2) at 16384"""
    monkeypatch.setenv("MI355_EMU_LINE32K", str(reg))
    batch = 5
    x = oracle.random_complex_batch(n, batch, 0xB16 + n).reshape(-1)
    for direction, norm in (("forward", "none"), ("inverse", "backward")):
        desc = _abi.make_desc("c2c", [n], batch, direction, norm)
        got, route, launches = emu.run_plan(desc, x, x.size)
        assert route.startswith(f"line-reg[N={n}]" if reg else f"lines[N={n}]") and launches == 1, route
        check(got, oracle.c2c_ref_batch(x, [n], batch, direction, norm), f"lines {n} {direction}", 2e-6)
    xr = oracle.random_real_batch(2 * n, 2, 0xB17 + n).reshape(-1)
    want = np.concatenate([oracle.r2c_ref_packed(xr[b * 2 * n:(b + 1) * 2 * n], 2 * n, "none") for b in range(2)])
    got, route, _ = emu.run_plan(_abi.make_desc("r2c", [2 * n], 2, "forward", "none"), xr, want.size)
    assert route.startswith(f"lines-r2c[N={2 * n}]"), route
    check(got, want, f"r2c {2 * n} over lines {n}", 1e-5)


@pytest.mark.parametrize("shape", [[40, 128], [17, 64], [24, 64, 3]])
def test_c2c_ragged_column_groups(oracle, shape):
    """a power-of-two axis whose stride is not a multiple of the 16-line tile: per-group tiles with a ragged last one"""
    batch = 2
    n = int(np.prod(shape))
    x = oracle.random_complex_batch(n, batch, 0x4A66 + n).reshape(-1)
    for direction, norm in (("forward", "none"), ("inverse", "backward")):
        desc = _abi.make_desc("c2c", shape, batch, direction, norm)
        got, route, _ = emu.run_plan(desc, x, x.size)
        assert f"columns-ragged[N={shape[1]},S={shape[0]}]" in route, route
        check(got, oracle.c2c_ref_batch(x, shape, batch, direction, norm), f"ragged {shape} {direction}", 3e-6)


def test_r2c_c2r_2d_packed_columns(oracle):
    """2-D r2c: the second axis runs over the 33 packed bins of a 64-point first axis (ragged column groups); c2r back"""
    shape, batch = [64, 128], 2
    n, p = 64 * 128, 33
    x = oracle.random_real_batch(n, batch, 0x4A77).reshape(-1)
    want = []
    for b in range(batch):
        cplx = np.zeros(2 * n, np.float32)
        cplx[0::2] = x[b * n:(b + 1) * n]
        full = oracle.fftnd_ref(cplx, shape, "forward", "none").reshape(shape[1], shape[0], 2)
        want.append(full[:, :p, :].reshape(-1))
    want = np.concatenate(want)
    got, route, _ = emu.run_plan(_abi.make_desc("r2c", shape, batch, "forward", "none"), x, want.size)
    assert "columns-ragged[N=128,S=33]" in route, route
    check(got, want, "r2c 2-D ragged columns", 3e-6)
    back, route, _ = emu.run_plan(_abi.make_desc("c2r", shape, batch, "inverse", "backward"), want, n * batch)
    assert "columns-ragged[N=128,S=33]" in route, route
    check(back, x, "c2r 2-D ragged columns", 3e-6)


def test_c2c_lane_layout_needs_no_staging(oracle):
    """rank-1 lane layouts (unit stride along the line, arbitrary offset / batch pitch on both sides): one line-kernel launch,
    elements between the lanes keep their sentinel"""
    n, batch = 256, 5
    x = oracle.random_complex_batch(n, batch, 0x1A9E).reshape(-1)
    in_off, in_pitch, out_off, out_pitch = 7, 300, 3, 512
    phys = np.full(2 * (in_off + (batch - 1) * in_pitch + n), 9.0, np.float32)
    for b in range(batch):
        phys[2 * (in_off + b * in_pitch):2 * (in_off + b * in_pitch + n)] = x[2 * b * n:2 * (b + 1) * n]
    out_elems = out_off + (batch - 1) * out_pitch + n
    sentinel = np.tile(np.array([77.0, -55.0], np.float32), out_elems)
    for direction, norm in (("forward", "none"), ("inverse", "backward")):
        desc = _abi.make_desc("c2c", [n], batch, direction, norm,
                              input_layout={"strides": [1], "offset": in_off, "batch_stride": in_pitch},
                              output_layout={"strides": [1], "offset": out_off, "batch_stride": out_pitch})
        got, route, launches = emu.run_plan(desc, phys, 2 * out_elems, out_init=sentinel)
        assert route.startswith("lines[N=256,pitch=300/512]") and launches == 1, route
        want = sentinel.copy()
        ref = oracle.c2c_ref_batch(x, [n], batch, direction, norm)
        for b in range(batch):
            want[2 * (out_off + b * out_pitch):2 * (out_off + b * out_pitch + n)] = ref[2 * b * n:2 * (b + 1) * n]
        check(got, want, f"lane layout {direction}", 2e-6)
        assert np.array_equal(got[:2 * out_off], sentinel[:2 * out_off])


@pytest.mark.parametrize("lg,label", [(17, "xcd-fused[N=256x512]"), (20, "xcd-fused-rt32[N=1024x1024]")])
def test_c2c_lane_layout_on_four_step_sizes(oracle, monkeypatch, lg, label):
    """rank-1 lane layouts of four-step sizes (r03): the fused kernel takes the two pitches and the base offsets as they are — one
    launch (+ its control-block reset), no gather / scatter; elements between the lanes keep their sentinel"""
    monkeypatch.setenv("MI355_EMU_XCD_FUSED", "1")
    monkeypatch.setenv("MI355_EMU_CUS", "3")
    monkeypatch.setenv("MI355_EMU_XCDS", "1")
    n, batch = 1 << lg, 3
    x = oracle.random_complex_batch(n, batch, 0x1A9F + lg).reshape(-1)
    in_off, in_pitch, out_off, out_pitch = 7, n + 300, 3, 2 * n
    phys = np.full(2 * (in_off + (batch - 1) * in_pitch + n), 9.0, np.float32)
    for b in range(batch):
        phys[2 * (in_off + b * in_pitch):2 * (in_off + b * in_pitch + n)] = x[2 * b * n:2 * (b + 1) * n]
    out_elems = out_off + (batch - 1) * out_pitch + n
    sentinel = np.tile(np.array([77.0, -55.0], np.float32), out_elems)
    for direction, norm in (("forward", "none"), ("inverse", "backward")):
        desc = _abi.make_desc("c2c", [n], batch, direction, norm,
                              input_layout={"strides": [1], "offset": in_off, "batch_stride": in_pitch},
                              output_layout={"strides": [1], "offset": out_off, "batch_stride": out_pitch})
        got, route, launches = emu.run_plan(desc, phys, 2 * out_elems, out_init=sentinel)
        assert route.startswith(label) and "lanes[pitch=%d/%d]" % (in_pitch, out_pitch) in route and launches == 2, route
        want = sentinel.copy()
        ref = oracle.c2c_ref_batch(x, [n], batch, direction, norm)
        for b in range(batch):
            want[2 * (out_off + b * out_pitch):2 * (out_off + b * out_pitch + n)] = ref[2 * b * n:2 * (b + 1) * n]
        check(got, want, f"lane layout 2^{lg} {direction}", 2e-6)
        assert np.array_equal(got[:2 * out_off], sentinel[:2 * out_off])
        gap = got[2 * (out_off + n):2 * (out_off + out_pitch)]
        assert np.array_equal(gap, sentinel[2 * (out_off + n):2 * (out_off + out_pitch)])


def test_c2c_generic_route_matches_lines_route(oracle):
    x = oracle.random_complex_batch(1024, 2, 3).reshape(-1)
    desc = _abi.make_desc("c2c", [1024], 2, "forward", "none")
    a, _, _ = emu.run_plan(desc, x, x.size)
    b, route, _ = emu.run_plan(desc, x, x.size, force_generic=True)
    assert route.startswith("stages[")
    check(a, b, "routes agree", 2e-6)


@pytest.mark.parametrize("shape", [[16, 64], [32, 128, 2], [64, 64]])
def test_c2c_nd_column_line_kernels(oracle, shape):
    """axes with stride S > 1 and S % 16 == 0 run the column line kernels instead of global stages"""
    n, batch = int(np.prod(shape)), 2
    x = oracle.random_complex_batch(n, batch, 0xD100 + n).reshape(-1)
    for direction in ("forward", "inverse"):
        desc = _abi.make_desc("c2c", shape, batch, direction, "backward")
        got, route, _ = emu.run_plan(desc, x, x.size)
        assert "columns[" in route
        check(got, oracle.c2c_ref_batch(x, shape, batch, direction, "backward"), f"nd columns {shape} {direction}")


@pytest.mark.parametrize("shape", [[8, 4], [16, 16], [4, 8, 2], [12, 5], [64, 3, 2]])
def test_c2c_nd(oracle, shape):
    n, batch = int(np.prod(shape)), 2
    x = oracle.random_complex_batch(n, batch, 0xD000 + n).reshape(-1)
    for direction in ("forward", "inverse"):
        desc = _abi.make_desc("c2c", shape, batch, direction, "unitary")
        got, _, _ = emu.run_plan(desc, x, x.size)
        check(got, oracle.c2c_ref_batch(x, shape, batch, direction, "unitary"), f"nd {shape} {direction}")


@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.parametrize("n", [17, 29, 34, 97, 2039])
def test_c2c_bluestein_lengths(oracle, monkeypatch, n, fused):
    """lengths with a prime factor > 13 (the reference's own test sizes: complete.suite.js:664-676) run the chirp-z route.
    fused=1 (default): two line-kernel launches — chirp + zero-padded embed on the forward launch's loads, the product with the
    chirp's spectrum on its last-stage store, chirp + crop on the inverse launch's store pass; fused=0: the five-launch form"""
    monkeypatch.setenv("MI355_EMU_FUSE_VIEWS", str(fused))
    batch = 5
    x = oracle.random_complex_batch(n, batch, 0xC100 + n).reshape(-1)
    for direction, norm in (("forward", "none"), ("inverse", "backward")):
        desc = _abi.make_desc("c2c", [n], batch, direction, norm)
        got, route, launches = emu.run_plan(desc, x, x.size)
        assert (route.startswith("bluestein-lines[") and launches == 2) if fused else ("bluestein[" in route and launches >= 5), route
        want = oracle.c2c_ref_batch(x, [n], batch, direction, norm)
        l2, mx = oracle.rel_l2(got, want), oracle.rel_max(got, want)
        assert l2 <= 1e-5 and mx <= 2e-5, f"bluestein N={n} {direction}: {l2:.2e} {mx:.2e}"
        oracle.assert_close_elementwise(got, want, 3e-4, 3e-4, f"bluestein N={n}")


@pytest.mark.parametrize("shape", [[17, 4], [8, 19], [5, 23, 2]])
def test_c2c_nd_with_bluestein_axes(oracle, shape):
    n, batch = int(np.prod(shape)), 2
    x = oracle.random_complex_batch(n, batch, 0xD200 + n).reshape(-1)
    desc = _abi.make_desc("c2c", shape, batch, "forward", "unitary")
    got, route, _ = emu.run_plan(desc, x, x.size)
    assert "bluestein[" in route or "bluestein-lines[" in route
    check(got, oracle.c2c_ref_batch(x, shape, batch, "forward", "unitary"), f"nd bluestein {shape}", 2e-5)


def test_too_long_bluestein_axis_is_a_clean_error():
    n = (1 << 22) + 1          # prime factors > 13 and beyond the chirp-z limit
    desc = _abi.make_desc("c2c", [n], 1, "forward", "none")
    with pytest.raises(emu.EmuError) as e:
        emu.run_plan(desc, np.zeros(2 * n, np.float32), 2 * n)
    assert e.value.code == _abi.ERR_UNSUPPORTED


@pytest.mark.parametrize("n", [2, 4, 8, 16, 64, 1024, 4096, 16384, 6, 10, 12, 30, 9, 15, 21, 17, 34])
def test_r2c_and_c2r(oracle, n):
    batch = 3
    x = oracle.random_real_batch(n, batch, 0xE000 + n).reshape(-1)
    p = n // 2 + 1
    want = np.concatenate([oracle.r2c_ref_packed(x[b * n:(b + 1) * n], n, "none") for b in range(batch)])
    desc = _abi.make_desc("r2c", [n], batch, "forward", "none")
    got, route, _ = emu.run_plan(desc, x, 2 * p * batch)
    check(got, want, f"r2c N={n} ({route})", 1e-5)
    # c2r of the oracle's packed spectrum, backward-normalised, returns the signal
    desc = _abi.make_desc("c2r", [n], batch, "inverse", "backward")
    back, route, _ = emu.run_plan(desc, want, n * batch)
    want_back = np.concatenate([oracle.c2r_ref_from_packed(want[2 * b * p:2 * (b + 1) * p], n, "backward") for b in range(batch)])
    check(back, want_back, f"c2r N={n} ({route})", 1e-5)
    check(back, x, f"c2r(r2c) round trip N={n}", 1e-5)


@pytest.mark.parametrize("n", [128, 256, 2048, 4096, 8192, 16384, 32768])     # 4096 ... 16384 (4096: two lines per tile, the odd batch leaves a ragged one), 32768 opt-in: the c2r twin copies the packed line raw into LDS and pre-splits in place (r03)
def test_r2c_split_fused_into_the_line_kernel(oracle, monkeypatch, n):
    """even N with a power-of-two half length >= 64: ONE launch (line FFT of the packed pairs + split from LDS); same numbers as
    the two-launch route"""
    batch = 5
    x = oracle.random_real_batch(n, batch, 0xE300 + n).reshape(-1)
    want = np.concatenate([oracle.r2c_ref_packed(x[b * n:(b + 1) * n], n, "unitary") for b in range(batch)])
    desc = _abi.make_desc("r2c", [n], batch, "forward", "unitary")
    got, route, launches = emu.run_plan(desc, x, want.size)
    assert route.startswith(f"lines-r2c[N={n}]") and launches == 1, route
    check(got, want, f"lines-r2c {n}", 1e-5)
    monkeypatch.setenv("MI355_EMU_LINES_C2R", "2")       # the c2r twin is used up to N = 2^14 by default; 2 forces it for every length
    back, route, launches = emu.run_plan(_abi.make_desc("c2r", [n], batch, "inverse", "unitary"), want, n * batch)
    assert route.startswith(f"lines-c2r[N={n}]") and launches == 1, route
    check(back, x, f"lines-c2r {n}", 1e-5)
    monkeypatch.setenv("MI355_EMU_LINES_C2R", "0")
    monkeypatch.setenv("MI355_EMU_LINES_R2C", "0")
    got2, route2, launches2 = emu.run_plan(desc, x, want.size)
    assert "r2c-split" in route2 and launches2 == 2, route2
    check(got2, want, f"two-launch r2c {n}", 1e-5)
    back2, route2, launches2 = emu.run_plan(_abi.make_desc("c2r", [n], batch, "inverse", "unitary"), want, n * batch)
    assert "c2r-split" in route2 and launches2 == 2, route2
    check(back2, x, f"two-launch c2r {n}", 1e-5)


def test_c2r_ignores_imag_of_self_conjugate_bins(oracle):
    n = 64
    x = oracle.random_real(n, 5)
    spec = oracle.r2c_ref_packed(x, n, "none").copy()
    spec[1] = 3.25        # imag of bin 0
    spec[2 * (n // 2) + 1] = -7.5  # imag of bin N/2
    desc = _abi.make_desc("c2r", [n], 1, "inverse", "backward")
    got, _, _ = emu.run_plan(desc, spec, n)
    check(got, oracle.c2r_ref_from_packed(spec, n, "backward"), "self-conjugate bins")
    check(got, x, "self-conjugate bins vs signal")


@pytest.mark.parametrize("cus,xcds,split,slots", [(2, 2, 1, 2), (6, 3, 1, 1), (8, 2, 2, 2), (6, 1, 4, 1), (9, 1, 8, 2)])
def test_r2c_xcd_fused_route(oracle, monkeypatch, cus, xcds, split, slots):
    """real four-step in one persistent launch (kern_xcd_real.hpp), test instance 64 x 64; unitary scale included"""
    monkeypatch.setenv("MI355_EMU_XCD_FUSED", "2")
    monkeypatch.setenv("MI355_EMU_CUS", str(cus))
    monkeypatch.setenv("MI355_EMU_XCDS", str(xcds))
    monkeypatch.setenv("MI355_EMU_XCD_SPLIT", str(split))
    monkeypatch.setenv("MI355_EMU_XCD_SLOTS", str(slots))
    n, batch = 4096, 7
    x = oracle.random_real_batch(n, batch, 0xD00D + cus).reshape(-1)
    for norm in ("none", "unitary"):
        want = np.concatenate([oracle.r2c_ref_packed(x[b * n:(b + 1) * n], n, norm) for b in range(batch)])
        desc = _abi.make_desc("r2c", [n], batch, "forward", norm)
        got, route, launches = emu.run_plan(desc, x, batch * (n // 2 + 1) * 2)
        assert route.startswith("xcd-r2c[N=64x64]") and launches == 2, route
        check(got, want, f"xcd-r2c {norm} cus={cus} xcds={xcds} split={split}", 1e-5)


@pytest.mark.parametrize("lg,label", [(18, "512x512"), (19, "512x1024"), (20, "1024x1024"), (21, "1024x2048")])
def test_r2c_xcd_fused_product_sizes(oracle, monkeypatch, lg, label):
    monkeypatch.setenv("MI355_EMU_XCD_FUSED", "1")
    monkeypatch.setenv("MI355_EMU_XCD_RT", "0")          # 2^21: the LDS-resident instance (register tiles: test_r2c_xcd_regtile)
    monkeypatch.setenv("MI355_EMU_CUS", "4")
    monkeypatch.setenv("MI355_EMU_XCDS", "1")
    monkeypatch.setenv("MI355_EMU_XCD_SPLIT", "2")
    n, batch = 1 << lg, 3
    x = oracle.random_real_batch(n, batch, 0xD100 + lg).reshape(-1)
    want = np.concatenate([oracle.r2c_ref_packed(x[b * n:(b + 1) * n], n, "none") for b in range(batch)])
    desc = _abi.make_desc("r2c", [n], batch, "forward", "none")
    got, route, launches = emu.run_plan(desc, x, batch * (n // 2 + 1) * 2)
    assert route.startswith(f"xcd-r2c[N={label}]") and launches == 2, route
    check(got, want, f"xcd-r2c {label}", 1e-5)


@pytest.mark.parametrize("lg,label,cus,xcds,split,slots,norm", [(22, "2048x2048", 2, 2, 1, 2, "none"), (22, "2048x2048", 3, 1, 1, 1, "unitary"), (22, "2048x2048", 4, 1, 2, 2, "backward"),
                                                                (21, "1024x2048", 3, 1, 0, 0, "none")])
def test_r2c_xcd_regtile(oracle, monkeypatch, lg, label, cus, xcds, split, slots, norm):
    """config 5's line (r2c N = 2^22) as a real four-step on register tiles (kern_regtile.hpp fft_xcd_rt_r2c_kernel): the
    separation of the two real columns of a complex column in the registers of the thread that owns both mirror consumers,
    65 row tiles (the last with one live row), Hermitian-mirrored stores incl. the Nyquist bin"""
    monkeypatch.setenv("MI355_EMU_XCD_FUSED", "1")
    monkeypatch.setenv("MI355_EMU_CUS", str(cus))
    monkeypatch.setenv("MI355_EMU_XCDS", str(xcds))
    monkeypatch.setenv("MI355_EMU_XCD_SPLIT", str(split))
    monkeypatch.setenv("MI355_EMU_XCD_SLOTS", str(slots))
    n, batch = 1 << lg, 3
    x = oracle.random_real_batch(n, batch, 0xD122 + cus).reshape(-1)
    want = np.concatenate([oracle.r2c_ref_packed(x[b * n:(b + 1) * n], n, norm) for b in range(batch)])
    desc = _abi.make_desc("r2c", [n], batch, "forward", norm)
    got, route, launches = emu.run_plan(desc, x, batch * (n // 2 + 1) * 2)
    assert route.startswith(f"xcd-r2c-rt[N={label}]") and launches == 2, route
    check(got, want, f"xcd-r2c-rt {label} {norm}", 1e-5)


@pytest.mark.parametrize("cus,xcds,split,slots", [(2, 2, 1, 2), (6, 3, 1, 1), (8, 2, 2, 2), (9, 1, 8, 2)])
def test_c2r_xcd_fused_route(oracle, monkeypatch, cus, xcds, split, slots):
    """Hermitian four-step c2r in one persistent launch (kern_xcd_real.hpp), test instance 64 x 64"""
    monkeypatch.setenv("MI355_EMU_XCD_FUSED", "2")
    monkeypatch.setenv("MI355_EMU_CUS", str(cus))
    monkeypatch.setenv("MI355_EMU_XCDS", str(xcds))
    monkeypatch.setenv("MI355_EMU_XCD_SPLIT", str(split))
    monkeypatch.setenv("MI355_EMU_XCD_SLOTS", str(slots))
    n, batch = 4096, 7
    p = n // 2 + 1
    x = oracle.random_real_batch(n, batch, 0xD20D + cus).reshape(-1)
    spec = np.concatenate([oracle.r2c_ref_packed(x[b * n:(b + 1) * n], n, "none") for b in range(batch)])
    for norm in ("backward", "none"):
        desc = _abi.make_desc("c2r", [n], batch, "inverse", norm)
        got, route, launches = emu.run_plan(desc, spec, n * batch)
        assert route.startswith("xcd-c2r[N=64x64]") and launches == 2, route
        want = np.concatenate([oracle.c2r_ref_from_packed(spec[2 * b * p:2 * (b + 1) * p], n, norm) for b in range(batch)])
        check(got, want, f"xcd-c2r {norm} cus={cus}", 1e-5)
        if norm == "backward":
            check(got, x, "c2r(r2c(x)) = x", 1e-5)


@pytest.mark.parametrize("lg,label,cus,xcds,split,slots", [(22, "2048x2048", 2, 2, 1, 2), (22, "2048x2048", 3, 1, 1, 1), (21, "1024x2048", 3, 1, 0, 0)])
def test_c2r_xcd_regtile(oracle, monkeypatch, lg, label, cus, xcds, split, slots):
    """c2r N = 2^22 as a Hermitian four-step on register tiles (kern_regtile.hpp fft_xcd_rt_c2r_kernel): 65 column tiles (the last
    with one live column), mirrored loads of the lower half columns, packed row pairs; against the signal and the oracle's c2r"""
    monkeypatch.setenv("MI355_EMU_XCD_FUSED", "1")
    monkeypatch.setenv("MI355_EMU_CUS", str(cus))
    monkeypatch.setenv("MI355_EMU_XCDS", str(xcds))
    monkeypatch.setenv("MI355_EMU_XCD_SPLIT", str(split))
    monkeypatch.setenv("MI355_EMU_XCD_SLOTS", str(slots))
    n, batch = 1 << lg, 3
    x = oracle.random_real_batch(n, batch, 0xD322 + cus).reshape(-1)
    spec = np.concatenate([oracle.r2c_ref_packed(x[b * n:(b + 1) * n], n, "none") for b in range(batch)])
    desc = _abi.make_desc("c2r", [n], batch, "inverse", "backward")
    got, route, launches = emu.run_plan(desc, spec, n * batch)
    assert route.startswith(f"xcd-c2r-rt[N={label}]") and launches == 2, route
    check(got, x, f"xcd-c2r-rt {label}", 1e-5)


@pytest.mark.parametrize("lg,label", [(18, "512x512"), (19, "512x1024"), (20, "1024x1024")])
def test_c2r_xcd_fused_product_sizes(oracle, monkeypatch, lg, label):
    monkeypatch.setenv("MI355_EMU_XCD_FUSED", "1")
    monkeypatch.setenv("MI355_EMU_CUS", "4")
    monkeypatch.setenv("MI355_EMU_XCDS", "1")
    monkeypatch.setenv("MI355_EMU_XCD_SPLIT", "2")
    n, batch = 1 << lg, 3
    p = n // 2 + 1
    x = oracle.random_real_batch(n, batch, 0xD300 + lg).reshape(-1)
    spec = np.concatenate([oracle.r2c_ref_packed(x[b * n:(b + 1) * n], n, "none") for b in range(batch)])
    desc = _abi.make_desc("c2r", [n], batch, "inverse", "backward")
    got, route, launches = emu.run_plan(desc, spec, n * batch)
    assert route.startswith(f"xcd-c2r[N={label}]") and launches == 2, route
    check(got, x, f"xcd-c2r {label}", 1e-5)


@pytest.mark.parametrize("lg,label", [(15, "128x256"), (16, "256x256"), (17, "256x512")])
def test_real_four_step_solo_sizes(oracle, monkeypatch, lg, label):
    """real / Hermitian four-step with one workgroup per transform (real lines of at most 512 KB): r2c against the oracle, c2r back"""
    monkeypatch.setenv("MI355_EMU_XCD_FUSED", "1")
    monkeypatch.setenv("MI355_EMU_LINES_R2C", "0")       # 2^15 would otherwise be a single line kernel launch
    monkeypatch.setenv("MI355_EMU_LINES_C2R", "3")       # ... on the c2r side too since r02 (3: line kernel up to N = 2^14 only)
    monkeypatch.setenv("MI355_EMU_LINE32K", "0")         # 2^16 real would otherwise take the half-length route over the 2^15 line kernel
    monkeypatch.setenv("MI355_EMU_SOLO_MAX_KB", "1024")  # r2c 2^17 runs shared (XCD groups) by default since r02
    monkeypatch.setenv("MI355_EMU_CUS", "3")
    monkeypatch.setenv("MI355_EMU_MAX_GRID", "3")
    n, batch = 1 << lg, 7
    x = oracle.random_real_batch(n, batch, 0xD400 + lg).reshape(-1)
    want = np.concatenate([oracle.r2c_ref_packed(x[b * n:(b + 1) * n], n, "none") for b in range(batch)])
    got, route, launches = emu.run_plan(_abi.make_desc("r2c", [n], batch, "forward", "none"), x, want.size)
    assert route.startswith(f"xcd-r2c-solo[N={label}]") and launches == 1, route
    check(got, want, f"xcd-r2c-solo {label}", 1e-5)
    back, route, launches = emu.run_plan(_abi.make_desc("c2r", [n], batch, "inverse", "backward"), want, n * batch)
    assert route.startswith(f"xcd-c2r-solo[N={label}]") and launches == 1, route
    check(back, x, f"xcd-c2r-solo {label}", 1e-5)


def test_r2c_rejects_inverse_and_c2r_rejects_forward():
    for typ, direction, frag in (("r2c", "inverse", "forward"), ("c2r", "forward", "inverse")):
        desc = _abi.make_desc(typ, [16], 1, direction, "none")
        with pytest.raises(emu.EmuError) as e:
            emu.run_plan(desc, np.zeros(64, np.float32), 64)
        assert e.value.code == _abi.ERR_INVALID and frag in str(e.value)


def test_r2c_2d(oracle):
    shape, batch = [16, 4], 2
    n = 64
    x = oracle.random_real_batch(n, batch, 11).reshape(-1)
    p = shape[0] // 2 + 1
    want = []
    for b in range(batch):
        cplx = np.zeros(2 * n, np.float32)
        cplx[0::2] = x[b * n:(b + 1) * n]
        full = oracle.fftnd_ref(cplx, shape, "forward", "none").reshape(shape[1], shape[0], 2)
        want.append(full[:, :p, :].reshape(-1))
    want = np.concatenate(want)
    desc = _abi.make_desc("r2c", shape, batch, "forward", "none")
    got, _, _ = emu.run_plan(desc, x, want.size)
    check(got, want, "r2c 2-D")
    desc = _abi.make_desc("c2r", shape, batch, "inverse", "backward")
    back, _, _ = emu.run_plan(desc, want, n * batch)
    check(back, x, "c2r 2-D round trip")


def test_support_kernels_match_oracle_prng(oracle):
    import ctypes
    rows, row_floats = 3, 2 * 50
    out = np.zeros(rows * row_floats, np.float32)
    emu.lib().emu_fill_random(out.ctypes.data, row_floats, rows, 0x5EED0002, 7)
    want = oracle.random_complex_batch(50, rows, 0x5EED0002, b0=7).reshape(-1)
    assert np.array_equal(out, want)
    s = ctypes.c_double()
    emu.lib().emu_diff_sumsq(out.ctypes.data, None, 0.0, out.size, ctypes.byref(s))
    assert abs(s.value - float(np.sum(out.astype(np.float64) ** 2))) < 1e-9


@pytest.mark.parametrize("n", [32, 256])
def test_r2c_c2r_whdcn_lanes(oracle, n):
    """layout.whdcn on r2c / c2r: the real side and the packed side each resolve against their own physical shape
    (docs/API.md "resolves per-side against the physical side shape"); only the addressed lane is read / written.
    n = 256: the lane addressing rides the r2c / c2r line kernels (one launch); n = 32: gather / scatter passes"""
    from mi355fft.layout import resolve_plan_options
    batch, channels, cidx = 2, 3, 1
    p = n // 2 + 1
    x = oracle.random_real_batch(n, batch, 8800).reshape(-1)
    phys_in = np.full(batch * channels * n, 9.0, np.float32)
    for b in range(batch):
        phys_in[b * channels * n + cidx * n: b * channels * n + (cidx + 1) * n] = x[b * n:(b + 1) * n]
    r = resolve_plan_options({"type": "r2c", "shape": [n], "batch": batch, "direction": "forward",
                              "layout": {"interleavedComplex": True, "whdcn": {"channels": channels, "channelIndex": cidx}}})
    assert r["input_layout"] == {"strides": [1], "offset": cidx * n, "batch_stride": channels * n}
    assert r["output_layout"] == {"strides": [1], "offset": cidx * p, "batch_stride": channels * p}
    desc = _abi.make_desc("r2c", [n], batch, "forward", "none", input_layout=r["input_layout"], output_layout=r["output_layout"])
    sentinel = np.tile(np.array([77.0, -55.0], np.float32), batch * channels * p)
    got, route, _ = emu.run_plan(desc, phys_in, sentinel.size, out_init=sentinel)
    assert route.split() == ["lines-r2c-mapped[N=256]"] if n == 256 else ("gather" in route and "scatter" in route), route
    want = sentinel.copy()
    for b in range(batch):
        base = 2 * (b * channels * p + cidx * p)
        want[base:base + 2 * p] = oracle.r2c_ref_packed(x[b * n:(b + 1) * n], n, "none")
    check(got, want, "r2c whdcn lanes")
    # c2r back out of the lanes into real lanes
    r = resolve_plan_options({"type": "c2r", "shape": [n], "batch": batch, "direction": "inverse", "normalize": "backward",
                              "layout": {"interleavedComplex": True, "whdcn": {"channels": channels, "channelIndex": cidx}}})
    desc = _abi.make_desc("c2r", [n], batch, "inverse", "backward", input_layout=r["input_layout"], output_layout=r["output_layout"])
    rs = np.full(batch * channels * n, -3.0, np.float32)
    back, route, _ = emu.run_plan(desc, want, rs.size, out_init=rs)
    assert route.split() == ["lines-c2r-mapped[N=256]"] if n == 256 else ("gather" in route and "scatter" in route), route
    wantr = rs.copy()
    for b in range(batch):
        wantr[b * channels * n + cidx * n: b * channels * n + (cidx + 1) * n] = x[b * n:(b + 1) * n]
    assert float(np.max(np.abs(back - wantr))) < 2e-6


@pytest.mark.parametrize("cus,xcds,split,slots", [(2, 2, 1, 2), (4, 2, 1, 2), (6, 3, 1, 1), (3, 1, 1, 2), (8, 2, 2, 2), (8, 1, 4, 1), (6, 1, 4, 2), (9, 1, 8, 2), (5, 1, 0, 2)])
def test_c2c_xcd_fused_route(oracle, monkeypatch, cus, xcds, split, slots):
    """both passes in one persistent launch; workgroups grouped by (emulated) XCC id synchronise through global
    counters: blocks run concurrently under emulation.  N = 64 x 64 is the test instance of the kernel template."""
    monkeypatch.setenv("MI355_EMU_XCD_FUSED", "2")
    monkeypatch.setenv("MI355_EMU_CUS", str(cus))
    monkeypatch.setenv("MI355_EMU_XCDS", str(xcds))
    monkeypatch.setenv("MI355_EMU_XCD_SPLIT", str(split))
    monkeypatch.setenv("MI355_EMU_XCD_SLOTS", str(slots))
    n, batch = 4096, 5
    x = oracle.random_complex_batch(n, batch, 0xF00D + cus).reshape(-1)
    for direction, norm in (("forward", "none"), ("inverse", "backward")):
        desc = _abi.make_desc("c2c", [n], batch, direction, norm)
        got, route, launches = emu.run_plan(desc, x, x.size)
        assert route.startswith("xcd-fused[N=64x64]") and launches == 2
        check(got, oracle.c2c_ref_batch(x, [n], batch, direction, norm), f"xcd-fused {direction} cus={cus} xcds={xcds}")
    desc = _abi.make_desc("c2c", [n], batch, "forward", "unitary", in_place=True)
    got, route, _ = emu.run_plan(desc, x, x.size)
    check(got, oracle.c2c_ref_batch(x, [n], batch, "forward", "unitary"), "xcd-fused in place")


@pytest.mark.parametrize("lg,label,cus,xcds,split,slots", [(21, "1024x2048", 3, 1, 1, 1), (22, "2048x2048", 4, 2, 2, 2), (21, "2048x1024", 3, 1, 0, 0)])
def test_c2c_xcd_regtile(oracle, monkeypatch, lg, label, cus, xcds, split, slots):
    """2048-point sides on register-resident 16-line tiles (kern_regtile.hpp): 64-point DFT in registers, one exchange through LDS in
    two halves, radix-32 stage; 2^21 = LDS-resident pass A + register-tile pass B, 2^22 = register tiles on both passes.
    MI355_EMU_XCD_RT=0 is the route it replaces (8-line LDS tiles / two-pass)."""
    monkeypatch.setenv("MI355_EMU_XCD_FUSED", "1")
    monkeypatch.setenv("MI355_EMU_CUS", str(cus))
    monkeypatch.setenv("MI355_EMU_XCDS", str(xcds))
    monkeypatch.setenv("MI355_EMU_XCD_SPLIT", str(split))
    monkeypatch.setenv("MI355_EMU_XCD_SLOTS", str(slots))
    if lg == 21:                                          # c2c 2^21 ships on the LDS-resident instance; its two register-tile forms are opt-in
        monkeypatch.setenv("MI355_EMU_XCD_RT", "2" if label == "1024x2048" else "3")
    n, batch = 1 << lg, 3
    x = oracle.random_complex_batch(n, batch, 0x2700 + lg).reshape(-1)
    for direction, norm in (("forward", "none"), ("inverse", "backward")):
        desc = _abi.make_desc("c2c", [n], batch, direction, norm)
        got, route, launches = emu.run_plan(desc, x, x.size)
        assert route.startswith(f"xcd-fused-rt32[N={label}]" if label == "2048x1024" else f"xcd-fused-rt[N={label}]") and launches == 2, route
        check(got, oracle.c2c_ref_batch(x, [n], batch, direction, norm), f"xcd-fused-rt {label} {direction}")


@pytest.mark.parametrize("cus,xcds,split,slots,hx,label", [(2, 2, 1, 2, 1, "2wg"), (3, 1, 2, 1, 1, "2wg"), (2, 2, 1, 2, 2, "rt32"), (3, 1, 1, 1, 2, "rt32"), (2, 1, 2, 1, 3, "rt16x2")])
def test_c2c_xcd_two_workgroups_per_cu(oracle, monkeypatch, cus, xcds, split, slots, hx, label):
    """N = 2^20 on register tiles (kern_regtile.hpp): hx=1 the exchange in two 64 KB halves, 72 KB of LDS and 128 VGPRs per workgroup,
    two workgroups per CU (fft_xcd_hx_kernel); hx=2 tiles of 32 lines, 256-byte segments (fft_xcd_rt1k_kernel); hx=3 the same code on 16-line tiles
    with 256 threads, two workgroups per CU"""
    monkeypatch.setenv("MI355_EMU_XCD_FUSED", "1")
    monkeypatch.setenv("MI355_EMU_XCD_HX", str(hx))
    monkeypatch.setenv("MI355_EMU_CUS", str(cus))
    monkeypatch.setenv("MI355_EMU_XCDS", str(xcds))
    monkeypatch.setenv("MI355_EMU_XCD_SPLIT", str(split))
    monkeypatch.setenv("MI355_EMU_XCD_SLOTS", str(slots))
    n, batch = 1 << 20, 5
    x = oracle.random_complex_batch(n, batch, 0x2820 + cus).reshape(-1)
    for direction, norm in (("forward", "none"), ("inverse", "backward")):
        desc = _abi.make_desc("c2c", [n], batch, direction, norm)
        got, route, launches = emu.run_plan(desc, x, x.size)
        assert route.startswith(f"xcd-fused-{label}[N=1024x1024]") and launches == 2, route
        check(got, oracle.c2c_ref_batch(x, [n], batch, direction, norm), f"xcd-fused-{label} {direction}")


@pytest.mark.parametrize("lg,label,cus", [(15, "128x256", 3), (16, "256x256", 2), (17, "256x512", 5)])
def test_c2c_xcd_solo_sizes(oracle, monkeypatch, lg, label, cus):
    """transforms of at most 1 MiB: one workgroup walks a whole transform (both passes, its own workspace slot, no cross-
    workgroup synchronisation); more transforms than workgroups so that every slot is re-used"""
    monkeypatch.setenv("MI355_EMU_XCD_FUSED", "1")
    monkeypatch.setenv("MI355_EMU_LINE32K", "0")              # 2^15 is a single-workgroup line by default (kern_line32k.hpp)
    monkeypatch.setenv("MI355_EMU_SOLO_MAX_KB", "1024")       # 2^17 runs shared (XCD groups) by default since r02
    monkeypatch.setenv("MI355_EMU_CUS", str(cus))
    monkeypatch.setenv("MI355_EMU_MAX_GRID", str(cus))        # fewer workgroups than transforms
    n, batch = 1 << lg, 7
    x = oracle.random_complex_batch(n, batch, 0x5010 + lg).reshape(-1)
    for direction, norm in (("forward", "none"), ("inverse", "backward")):
        desc = _abi.make_desc("c2c", [n], batch, direction, norm)
        got, route, launches = emu.run_plan(desc, x, x.size)
        assert route.startswith(f"xcd-solo[N={label}]") and launches == 1, route
        check(got, oracle.c2c_ref_batch(x, [n], batch, direction, norm), f"xcd-solo {label} {direction}")


@pytest.mark.parametrize("shape,label,extra", [([256, 256], "xcd-2d-solo[256x256]", []), ([512, 512], "xcd-2d[512x512]", []), ([256, 256, 3], "xcd-2d-solo[256x256]", [3]),
                                               ([512, 256], "xcd-2d-solo[512x256]", []), ([1024, 512], "xcd-2d[1024x512]", []), ([512, 1024], "xcd-2d[512x1024]", [])])
def test_c2c_2d_fused(oracle, monkeypatch, shape, label, extra):
    """square power-of-two planes: axes 0 and 1 in one fused launch (columns, barrier, rows in natural order); a third axis follows"""
    monkeypatch.setenv("MI355_EMU_XCD_FUSED", "1")
    monkeypatch.setenv("MI355_EMU_CUS", "4")
    monkeypatch.setenv("MI355_EMU_XCDS", "2")
    batch = 3 if len(shape) == 2 else 2
    n = int(np.prod(shape))
    x = oracle.random_complex_batch(n, batch, 0x2D00 + n).reshape(-1)
    for direction, norm in (("forward", "none"), ("inverse", "backward")):
        desc = _abi.make_desc("c2c", shape, batch, direction, norm)
        got, route, launches = emu.run_plan(desc, x, x.size)
        assert route.startswith(label), route
        check(got, oracle.c2c_ref_batch(x, shape, batch, direction, norm), f"{label} {direction}", 3e-6)
    desc = _abi.make_desc("c2c", shape, batch, "forward", "unitary", in_place=True)
    got, route, _ = emu.run_plan(desc, x, x.size)
    check(got, oracle.c2c_ref_batch(x, shape, batch, "forward", "unitary"), f"{label} in place", 3e-6)


@pytest.mark.parametrize("lg,label", [(18, "512x512"), (19, "512x1024"), (21, "1024x2048")])
def test_c2c_xcd_fused_product_sizes(oracle, monkeypatch, lg, label):
    """the product instances of the fused kernel whose two passes use different tile widths (512x1024: 32-column
    tiles in pass A; 1024x2048: 8-row tiles and a three-stage row FFT in pass B), forward and inverse"""
    monkeypatch.setenv("MI355_EMU_XCD_FUSED", "1")
    monkeypatch.setenv("MI355_EMU_CUS", "4")
    monkeypatch.setenv("MI355_EMU_XCDS", "1")
    monkeypatch.setenv("MI355_EMU_XCD_SPLIT", "2")
    monkeypatch.setenv("MI355_EMU_XCD_RT", "0")          # 2^21: the LDS-resident instance (the register-tile one: test_c2c_xcd_regtile)
    n, batch = 1 << lg, 3
    x = oracle.random_complex_batch(n, batch, 0xBEEF + lg).reshape(-1)
    for direction, norm in (("forward", "none"), ("inverse", "backward")):
        desc = _abi.make_desc("c2c", [n], batch, direction, norm)
        got, route, launches = emu.run_plan(desc, x, x.size)
        assert route.startswith(f"xcd-fused[N={label}]") and launches == 2
        check(got, oracle.c2c_ref_batch(x, [n], batch, direction, norm), f"xcd-fused {label} {direction}")


@pytest.mark.parametrize("direction,depth,batch", [("forward", 4, 1), ("inverse", 2, 2), ("forward", 1, 2)])
def test_c2c_xcd_resident_2p20(oracle, monkeypatch, direction, depth, batch):
    """XCD-resident 1024 x 1024 kernel (kern_xcd_res.hpp): 32 emulated workgroups of 512 threads hold the transform between its
    passes and hand it over through the exchange channels; depth = channels in flight (1 and 2 exercise the READ-counter gating
    of a re-used channel buffer), batch 2 the re-use of buffers and LDS across transforms."""
    monkeypatch.setenv("MI355_EMU_XCD_RES", "1")
    monkeypatch.setenv("MI355_EMU_XCD_RES_DEPTH", str(depth))
    monkeypatch.setenv("MI355_EMU_CUS", "32")
    monkeypatch.setenv("MI355_EMU_XCDS", "1")
    n = 1 << 20
    x = oracle.random_complex_batch(n, batch, 0xE500 + depth).reshape(-1)
    desc = _abi.make_desc("c2c", [n], batch, direction, "backward")
    got, route, launches = emu.run_plan(desc, x, x.size)
    assert route.startswith("xcd-resident[N=1024x1024,depth=%d]" % depth) and launches == 2, route
    check(got, oracle.c2c_ref_batch(x, [n], batch, direction, "backward"), f"xcd-resident {direction} depth={depth}")

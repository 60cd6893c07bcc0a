"""GPU tier (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the same seeded
inputs; golden fixtures generated from the reference's math.js; size-independent properties at the
BASELINE.json sizes.

Tolerance (north_star: "within 1e-5 relative for f32"): norm-relative  ||a-e||2/||e||2 <= 1e-5 and
max|a-e|/max|e| <= 1e-5 per case, plus the reference's own per-element form
|a-e| <= atol + rtol*|e| at its own tolerances (3e-4 c2c, 8e-4 r2c, 2e-3 c2r, 4e-3..5e-3 fftconv:
test/complete.suite.js:674,1797,1813,4663,4831)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def fft():
    import mi355fft
    return mi355fft


@pytest.fixture(scope="module")
def dev(fft):
    d = fft.Device(0)
    yield d
    d.close()


def run_plan(fft, dev, opts, x, out_floats, kernel=None, out_init=None, in_place=False, use_graph=False):
    x = np.ascontiguousarray(x, dtype=np.float32)
    inp = fft.uploadComplex(dev, x)
    out = None
    if not in_place:
        out = dev.createBuffer({"size": max(4 * out_floats, 8)})
        if out_init is not None:
            dev.queue.writeBuffer(out, 0, np.asarray(out_init, dtype=np.float32))
    plan = fft.createPlan(dev, opts)
    enc = dev.createCommandEncoder()
    args = {"input": inp}
    if out is not None:
        args["output"] = out
    if kernel is not None:
        args["kernel"] = kernel
    plan.exec(enc, args)
    cb = enc.finish(use_graph=use_graph)
    dev.queue.submit([cb])
    dev.queue.onSubmittedWorkDone()
    got = fft.downloadF32(dev, inp if in_place else out, out_floats)
    route = plan.describe()
    cb.release()
    plan.destroy()
    inp.destroy()
    if out is not None:
        out.destroy()
    return got, route


def check(oracle, got, want, what, atol=3e-4, rtol=3e-4, tol=TOL):
    l2, mx = oracle.rel_l2(got, want), oracle.rel_max(got, want)
    assert l2 <= tol and mx <= tol, f"{what}: rel_l2={l2:.3e} rel_max={mx:.3e}"
    # elementwise |got - want| <= atol + rtol*|want| as the reference's suites do; their sizes keep the output rms near 1-20,
    # so for long unnormalised transforms (rms = sqrt(N/12): 418 at N = 2^21, where one f32 ulp is already 3e-5) atol is
    # scaled by the rms above 64 — otherwise the check measures the f32 rounding of the oracle, not parity
    rms = float(np.sqrt(np.mean(np.asarray(want, dtype=np.float64) ** 2)))
    oracle.assert_close_elementwise(got, want, atol * max(1.0, rms / 64.0), rtol, what)


# ---- c2c ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096])
def test_c2c_lines_all_sizes(fft, dev, oracle, n):
    batch = 37 if n <= 1024 else 5       # ragged vs the tile size
    x = oracle.random_complex_batch(n, batch, 0xA000 + n).reshape(-1)
    for direction, norm in (("forward", "none"), ("inverse", "backward"), ("forward", "unitary")):
        got, (route, launches) = run_plan(fft, dev, {"type": "c2c", "shape": [n], "batch": batch, "direction": direction, "normalize": norm}, x, x.size)
        assert route.startswith("lines[") and launches == 1
        check(oracle, got, oracle.c2c_ref_batch(x, [n], batch, direction, norm), f"c2c N={n} {direction} {norm}")


def test_c2c_golden_fixture_cfg1(fft, dev, oracle, manifest):
    """BASELINE config 1 (bench_1d_1024.js shape): N=1024 batch=1, against the reference's own output"""
    cases, _ = manifest
    c = cases["c2c_N1024_b1_forward_none"]
    x = np.fromfile(os.path.join(GOLDEN, c["in_file"]), dtype=np.float32)
    want = np.fromfile(os.path.join(GOLDEN, c["out_file"]), dtype=np.float32)
    got, _ = run_plan(fft, dev, {"type": "c2c", "shape": [1024], "batch": 1, "direction": "forward", "normalize": "none"}, x, x.size)
    check(oracle, got, want, "cfg1 golden")
    got, _ = run_plan(fft, dev, {"type": "c2c", "shape": [1024], "batch": 1, "direction": "forward", "normalize": "none"}, x, x.size, use_graph=True)
    check(oracle, got, want, "cfg1 golden via hipGraph")


def test_c2c_in_place_and_offsets(fft, dev, oracle):
    n, batch = 256, 6
    x = oracle.random_complex_batch(n, batch, 77).reshape(-1)
    got, _ = run_plan(fft, dev, {"type": "c2c", "shape": [n], "batch": batch, "direction": "forward", "normalize": "none", "inPlace": True}, x,
                      x.size, in_place=True)
    check(oracle, got, oracle.c2c_ref_batch(x, [n], batch, "forward"), "in-place")
    # inputOffsetBytes / outputOffsetBytes: transform the last 4 lines into the middle of a larger output
    inp = fft.uploadComplex(dev, x)
    out = dev.createBuffer({"size": x.nbytes * 2})
    dev.queue.writeBuffer(out, 0, np.full(2 * x.size, 3.0, np.float32))
    plan = fft.createPlan(dev, {"type": "c2c", "shape": [n], "batch": 4, "direction": "inverse", "normalize": "unitary"})
    enc = dev.createCommandEncoder()
    plan.exec(enc, {"input": inp, "output": out, "inputOffsetBytes": 2 * n * 8, "outputOffsetBytes": 1024})
    dev.queue.submit([enc.finish()])
    dev.queue.onSubmittedWorkDone()
    full = fft.downloadF32(dev, out, 2 * x.size)
    want = oracle.c2c_ref_batch(x[2 * n * 2:], [n], 4, "inverse", "unitary")
    check(oracle, full[256:256 + want.size], want, "offsets")
    assert np.all(full[:256] == 3.0) and np.all(full[256 + want.size:] == 3.0)
    with pytest.raises(fft.Mi355Error, match="multiples of 8"):
        plan.exec(dev.createCommandEncoder(), {"input": inp, "output": out, "inputOffsetBytes": 4})
    with pytest.raises(fft.Mi355Error, match="too small"):
        plan.exec(dev.createCommandEncoder(), {"input": inp, "output": out, "inputOffsetBytes": x.nbytes - 8})
    plan.destroy()
    with pytest.raises(fft.Mi355Error, match="plan destroyed"):
        plan.exec(dev.createCommandEncoder(), {"input": inp, "output": out})
    plan.destroy()  # idempotent
    inp.destroy()
    out.destroy()


FUSED_LG = (15, 16, 17, 18, 19, 20, 21, 22)   # 15-17: solo mode; 21, 22 (r03): register-tile instances   # MI355_XCD_KERNEL_LIST / MI355_XCD_RT_KERNEL_LIST (plan.hpp)


@pytest.mark.parametrize("fused", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("lg", [13, 14, 15, 16, 17, 18, 19, 20, 21, 22])
def test_c2c_two_pass(fft, dev, oracle, monkeypatch, lg, fused):
    """four-step sizes on both routes: the two-launch route and (where an instance exists) the XCD-fused launch"""
    if fused and lg not in FUSED_LG or (fused == 2 and lg != 20) or (fused >= 3 and lg != 21):
        pytest.skip("no fused instance")
    monkeypatch.setenv("MI355FFT_XCD_FUSED", str(min(fused, 1)))
    # fused=1: the shipped route (2^21: the LDS-resident 1024 x 2048 instance); fused=3 / 4: 2^21 on its register-tile forms, 1024 x 2048 / 2048 x 1024
    monkeypatch.setenv("MI355FFT_XCD_RT", "0" if fused == 2 else "2" if fused == 3 else "3" if fused == 4 else "1")   # fused=2: the LDS-resident instances the register tiles replaced (2^21: 8-line tiles;
    monkeypatch.setenv("MI355FFT_XCD_HX", "0" if fused == 2 else "2")   #          2^20: 16 x 1024 tiles in LDS)
    monkeypatch.setenv("MI355FFT_MAX_LINE", "4096")       # 2^13 and 2^14 would otherwise run as single-workgroup lines
    monkeypatch.setenv("MI355FFT_LINE32K", "0")           # ... and so would 2^15 (kern_line32k.hpp, test_c2c_line32k)
    monkeypatch.setenv("MI355FFT_SOLO_MAX_KB", "1024")    # solo mode up to 2^17 as in round 1 (default since r02: up to 2^16, 2^17 shared)
    n, batch = 1 << lg, 3 if lg <= 18 else 2
    x = oracle.random_complex_batch(n, batch, 0xB000 + lg).reshape(-1)
    for direction in ("forward", "inverse"):
        got, (route, _) = run_plan(fft, dev, {"type": "c2c", "shape": [n], "batch": batch, "direction": direction, "normalize": "backward"}, x, x.size)
        assert route.startswith(("xcd-solo[" if lg <= 17 else "xcd-fused-rt[" if (lg == 22 and fused == 1) or fused == 3 else "xcd-fused-rt32[" if (lg == 20 and fused == 1) or fused == 4 else "xcd-fused[") if fused else "two-pass["), route
        check(oracle, got, oracle.c2c_ref_batch(x, [n], batch, direction, "backward"), f"{route.strip()} 2^{lg} {direction}")


@pytest.mark.parametrize("hx,label", [(1, "2wg"), (2, "rt32"), (3, "rt16x2")])
def test_c2c_two_workgroups_per_cu(fft, dev, oracle, monkeypatch, hx, label):
    """N = 2^20 on the opt-in register-tile kernels (kern_regtile.hpp): hx=1 two workgroups per CU (fft_xcd_hx_kernel; measured slower
    than the shipped kernel, profiles/r03_headline_2wg_ab.log), hx=2 tiles of 32 lines (fft_xcd_rt1k_kernel): 19 transforms over the
    groups, both directions, against the oracle"""
    monkeypatch.setenv("MI355FFT_XCD_HX", str(hx))
    n, batch = 1 << 20, 19
    x = oracle.random_complex_batch(n, batch, 0xE520).reshape(-1)
    for direction in ("forward", "inverse"):
        got, (route, launches) = run_plan(fft, dev, {"type": "c2c", "shape": [n], "batch": batch, "direction": direction, "normalize": "backward"}, x, x.size)
        assert route.startswith(f"xcd-fused-{label}[N=1024x1024]") and launches == 2, route
        check(oracle, got, oracle.c2c_ref_batch(x, [n], batch, direction, "backward"), f"{route.strip()} {direction}")


@pytest.mark.parametrize("depth", [4, 2, 1])
def test_c2c_xcd_resident(fft, dev, oracle, monkeypatch, depth):
    """N = 2^20 on the XCD-resident kernel (kern_xcd_res.hpp): the transform is handed between the 32 workgroups of an XCD through
    L2-resident exchange channels (depth = channels in flight).  11 transforms over 8 XCD groups: ragged last round, buffers and
    counters re-used across transforms; every transform against the oracle, both directions."""
    if dev.info()["compute_units"] % 32:
        pytest.skip("needs whole XCDs of 32 CUs")
    monkeypatch.setenv("MI355FFT_XCD_RES", "1")
    monkeypatch.setenv("MI355FFT_XCD_RES_DEPTH", str(depth))
    n, batch = 1 << 20, 11
    x = oracle.random_complex_batch(n, batch, 0xE500 + depth).reshape(-1)
    for direction in ("forward", "inverse"):
        got, (route, launches) = run_plan(fft, dev, {"type": "c2c", "shape": [n], "batch": batch, "direction": direction, "normalize": "backward"}, x, x.size)
        assert route.startswith("xcd-resident[N=1024x1024,depth=%d]" % depth) and launches == 2, route
        check(oracle, got, oracle.c2c_ref_batch(x, [n], batch, direction, "backward"), f"{route.strip()} {direction}")


@pytest.mark.parametrize("fused", [0, 1])
@pytest.mark.parametrize("shape,batch", [([256, 256], 40), ([512, 512], 20), ([1024, 1024], 5), ([256, 256, 4], 3),
                                         ([512, 256], 20), ([256, 512], 20), ([512, 1024], 6), ([1024, 512], 6)])
def test_c2c_2d_planes(fft, dev, oracle, monkeypatch, shape, batch, fused):
    """power-of-two planes with sides of 256, 512 or 1024: both axes in one fused launch (columns, barrier, rows in natural order) vs one launch per axis"""
    monkeypatch.setenv("MI355FFT_XCD_2D", str(fused))
    n = int(np.prod(shape))
    x = oracle.random_complex_batch(n, batch, 0x2D00 + n).reshape(-1)
    for direction, norm in (("forward", "none"), ("inverse", "backward")):
        got, (route, launches) = run_plan(fft, dev, {"type": "c2c", "shape": shape, "batch": batch, "direction": direction, "normalize": norm}, x, x.size)
        assert route.startswith("xcd-2d") == bool(fused), route
        check(oracle, got, oracle.c2c_ref_batch(x, shape, batch, direction, norm), f"{shape} {direction} ({route.strip()})")


def test_r2c_c2r_2d_real_images(fft, dev, oracle):
    """2-D r2c / c2r of 512x512 and 256x1024 real planes: the second axis walks the packed bins (stride 257 / 129: ragged column
    groups); against the full complex oracle and back to the signal"""
    for shape, batch in (([512, 512], 6), ([256, 1024], 4), ([64, 64, 8], 3)):
        n, p = int(np.prod(shape)), shape[0] // 2 + 1
        x = oracle.random_real_batch(n, batch, 0x4A88 + n).reshape(-1)
        cplx = np.zeros(2 * n * batch, np.float32)
        cplx[0::2] = x
        full = oracle.c2c_ref_batch(cplx, shape, batch, "forward", "none").reshape(batch, n // shape[0], shape[0], 2)
        want = np.ascontiguousarray(full[:, :, :p, :]).reshape(-1)
        got, (route, _) = run_plan(fft, dev, {"type": "r2c", "shape": shape, "batch": batch, "direction": "forward", "normalize": "none"}, x, want.size)
        assert "columns-ragged[" in route, route
        check(oracle, got, want, f"r2c {shape} ({route.strip()})", 8e-4, 8e-4)
        back, (route, _) = run_plan(fft, dev, {"type": "c2r", "shape": shape, "batch": batch, "direction": "inverse", "normalize": "backward"}, want, n * batch)
        assert "columns-ragged[" in route, route
        check(oracle, back, x, f"c2r {shape} ({route.strip()})", 2e-3, 2e-3)


@pytest.mark.parametrize("n,reg", [(8192, 1), (8192, 0), (16384, 0), (16384, 2)])
def test_c2c_single_workgroup_long_lines(fft, dev, oracle, monkeypatch, n, reg):
    """N = 8192 / 16384: one workgroup per line — in LDS (last stage table from global memory) or, reg != 0, in the registers of N/64
    threads with the exchanges through LDS in halves (kern_line_reg.hpp; the default at 8192); one launch, one HBM round trip"""
    monkeypatch.setenv("MI355FFT_LINE32K", str(reg))
    batch = 37 if reg == 0 else 1500          # more lines than resident workgroups on the register route
    x = oracle.random_complex_batch(n, batch, 0xB16 + n).reshape(-1)
    for direction, norm in (("forward", "none"), ("inverse", "backward"), ("forward", "unitary")):
        got, (route, launches) = run_plan(fft, dev, {"type": "c2c", "shape": [n], "batch": batch, "direction": direction, "normalize": norm}, x, x.size)
        assert route.startswith(f"line-reg[N={n}]" if reg else f"lines[N={n}]") and launches == 1, route
        check(oracle, got, oracle.c2c_ref_batch(x, [n], batch, direction, norm), f"lines {n} {direction} {norm}")


@pytest.mark.parametrize("batch", [1, 5, 700])
def test_c2c_line32k(fft, dev, oracle, batch):
    """N = 2^15 in one workgroup (kern_line32k.hpp: the line in registers, both exchanges through LDS in halves): one launch, both
    directions, every normalisation, more lines than workgroups; and the real transforms of 2^16 points that ride it"""
    n = 1 << 15
    x = oracle.random_complex_batch(n, batch, 0xC800 + batch).reshape(-1)
    for direction in ("forward", "inverse"):
        for norm in (("none", "backward", "unitary") if batch == 5 else ("backward",)):
            got, (route, launches) = run_plan(fft, dev, {"type": "c2c", "shape": [n], "batch": batch, "direction": direction, "normalize": norm}, x, x.size)
            assert route.split() == ["line32k[N=32768]"] and launches == 1, route
            check(oracle, got, oracle.c2c_ref_batch(x, [n], batch, direction, norm), f"line32k x{batch} {direction} {norm}")
    if batch == 5:
        m = 1 << 16
        xr = oracle.random_real_batch(m, batch, 0xC900).reshape(-1)
        p = m // 2 + 1
        spec, (route, _) = run_plan(fft, dev, {"type": "r2c", "shape": [m], "batch": batch, "direction": "forward", "normalize": "none"}, xr, 2 * p * batch)
        assert route.split() == ["line32k[N=32768]", "r2c-split"], route
        want = np.concatenate([oracle.r2c_ref_packed(xr[b * m:(b + 1) * m], m, "none", use_pow2=True) for b in range(batch)])
        check(oracle, spec, want, "r2c 2^16 over line32k", 8e-4, 8e-4)
        back, (route, _) = run_plan(fft, dev, {"type": "c2r", "shape": [m], "batch": batch, "direction": "inverse", "normalize": "backward"}, want, m * batch)
        assert route.split() == ["c2r-split", "line32k[N=32768]"] or route.split() == ["line32k[N=32768]", "c2r-split"], route
        check(oracle, back, xr, "c2r 2^16 over line32k", 2e-3, 2e-3)


@pytest.mark.parametrize("lg,batch", [(15, 3000), (16, 1500), (17, 700), (18, 150), (19, 75), (21, 37), (22, 19)])
def test_c2c_fused_many_transforms(fft, dev, oracle, monkeypatch, lg, batch):
    """more transforms than groups: every group walks several transforms and alternates its two workspace slots"""
    monkeypatch.setenv("MI355FFT_LINE32K", "0")           # keep the solo four-step of 2^15 under test
    monkeypatch.setenv("MI355FFT_SOLO_MAX_KB", "1024")    # ... and of 2^17 (shared by default since r02)
    n = 1 << lg
    x = oracle.random_complex_batch(n, batch, 0xC000 + lg).reshape(-1)
    for direction in ("forward", "inverse"):
        got, (route, launches) = run_plan(fft, dev, {"type": "c2c", "shape": [n], "batch": batch, "direction": direction, "normalize": "none"}, x, x.size)
        assert (route.startswith("xcd-solo[") and launches == 1) if lg <= 17 else (route.startswith("xcd-fused-rt[" if lg == 22 else "xcd-fused[") and launches == 2), route
        check(oracle, got, oracle.c2c_ref_batch(x, [n], batch, direction, "none"), f"fused 2^{lg} x{batch} {direction}")


def test_c2c_large_golden_samples(fft, dev, oracle, manifest):
    """F3: N=2^20 single transform vs the reference's sampled bins + sum of squares"""
    cases, _ = manifest
    for name in ("c2c_N2p20_forward", "c2c_N2p20_inverse", "c2c_N2p16_forward", "c2c_N2p21_forward"):
        c = cases[name]
        n = c["shape"][0]
        x = oracle.random_complex_interleaved(n, c["seed"])
        got, _ = run_plan(fft, dev, {"type": "c2c", "shape": [n], "batch": 1, "direction": c["direction"], "normalize": "none"}, x, x.size)
        idx = np.asarray(c["sample_idx"])
        want = np.asarray(c["sample_vals"], dtype=np.float32).reshape(-1, 2)
        g = got.reshape(-1, 2)[idx]
        scale = float(np.sqrt(c["out_sumsq"] / n))           # rms magnitude of an output bin
        assert float(np.max(np.abs(g.astype(np.float64) - want))) <= 2e-5 * scale * 8, name
        assert abs(float(np.sum(got.astype(np.float64) ** 2)) / c["out_sumsq"] - 1.0) < 1e-5, name


@pytest.mark.parametrize("mixed", [0, 1])
@pytest.mark.parametrize("n", [3, 5, 6, 7, 11, 12, 13, 15, 21, 96, 105, 210, 1000, 1001, 1144, 2187, 3000, 4095, 3 * 4096])
def test_c2c_mixed_radix(fft, dev, oracle, monkeypatch, n, mixed):
    """mixed-radix lengths: the one-launch LDS line kernel (N <= 4096, >= 2 stages) and the global-memory stage route"""
    monkeypatch.setenv("MI355FFT_MIXED_LINES", str(2 * mixed))   # 2: also where the planner would keep the stage route
    monkeypatch.setenv("MI355FFT_MIXED_CT", "0")                 # (the compile-time-plan instances: test_c2c_mixed_radix_compile_time_plans)
    batch = 300 if n < 200 else (37 if n < 2000 else 5)   # several tiles per workgroup, ragged last tile; the O(N^2) oracle bounds the rest
    x = oracle.random_complex_batch(n, batch, 0xC000 + n).reshape(-1)
    for direction, norm in (("forward", "none"), ("inverse", "backward")):
        got, (route, launches) = run_plan(fft, dev, {"type": "c2c", "shape": [n], "batch": batch, "direction": direction, "normalize": norm}, x, x.size)
        if mixed and n <= 4096 and n not in (3, 5, 7, 11, 13):
            assert route.startswith("mixed-lines[") and launches == 1, route
        else:
            assert route.startswith("stages["), route
        check(oracle, got, oracle.c2c_ref_batch(x, [n], batch, direction, norm), f"N={n} {direction} {route}", 3e-3, 3e-3)


@pytest.mark.parametrize("n", [96, 192, 384, 768, 1536, 3072, 160, 320, 640, 1280, 2560, 1000, 2000, 3000, 105, 1001, 360, 1920, 2187, 500, 1500, 120, 240, 480, 720, 1440])
def test_c2c_mixed_radix_compile_time_plans(fft, dev, oracle, n):
    """every instance of kern_mixed_ct.hpp: radices, tile shape and threads as template constants (232 vs 85 GPoints/s at N=1000);
    batches with several tiles per workgroup and a ragged last tile, against the oracle's O(N^2) DFT"""
    batch = 301 if n < 400 else (37 if n < 2000 else 7)
    x = oracle.random_complex_batch(n, batch, 0xC700 + n).reshape(-1)
    for direction, norm in (("forward", "unitary"), ("inverse", "backward")):
        got, (route, launches) = run_plan(fft, dev, {"type": "c2c", "shape": [n], "batch": batch, "direction": direction, "normalize": norm}, x, x.size)
        assert route.startswith("mixed-ct[N=%d," % n) and launches == 1, route
        check(oracle, got, oracle.c2c_ref_batch(x, [n], batch, direction, norm), f"N={n} {direction} {route}", 3e-3, 3e-3)


@pytest.mark.parametrize("shape", [[96, 105], [24, 25, 27], [1000, 3, 5]])
def test_c2c_nd_mixed_radix(fft, dev, oracle, shape):
    """the reference's N-D mixed-radix shapes (complete.suite.js:876-913), every axis a single launch"""
    batch = 3
    n = int(np.prod(shape))
    x = oracle.random_complex_batch(n, batch, 0xC500 + n).reshape(-1)
    for direction, norm in (("forward", "none"), ("inverse", "backward")):
        got, (route, launches) = run_plan(fft, dev, {"type": "c2c", "shape": shape, "batch": batch, "direction": direction, "normalize": norm}, x, x.size)
        assert launches == len(shape), route
        check(oracle, got, oracle.c2c_ref_batch(x, shape, batch, direction, norm), f"{shape} {direction} {route}", 3e-3, 3e-3)


@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.parametrize("n", [17, 29, 34, 97, 2039, 6007, 100003])
def test_c2c_bluestein_lengths(fft, dev, oracle, monkeypatch, n, fused):
    """the reference's own prime test sizes (complete.suite.js:664-676) through the chirp-z route; fused=1: two line-kernel launches
    (chirp, embed and product on the forward launch, chirp and crop on the inverse one) up to a convolution length of 16384"""
    if not fused and n == 6007:
        pytest.skip("one long case on the five-launch form is enough")
    monkeypatch.setenv("MI355FFT_FUSE_VIEWS", str(fused))
    batch = 300 if n < 100 else (6 if n < 10000 else 1)      # (the O(N^2) oracle bounds the long ones)
    x = oracle.random_complex_batch(n, batch, 0xC100 + n).reshape(-1)
    for direction, norm in (("forward", "none"), ("inverse", "backward")):
        got, (route, launches) = run_plan(fft, dev, {"type": "c2c", "shape": [n], "batch": batch, "direction": direction, "normalize": norm}, x, x.size)
        assert (route.startswith("bluestein-lines[") and launches == 2) if (fused and n < 10000) else "bluestein[" in route, route
        if n < 10000:
            want = oracle.c2c_ref_batch(x, [n], batch, direction, norm)
        else:  # O(N^2) oracle is infeasible: independent f64 FFT
            c = x.astype(np.float64).view(np.complex128)
            w = np.fft.fft(c) if direction == "forward" else np.fft.ifft(c)
            want = np.stack([w.real, w.imag], axis=-1).reshape(-1)
        l2, mx = oracle.rel_l2(got, want), oracle.rel_max(got, want)
        # north_star's bar in both norms (r02 had loosened rel_max to 2e-5 here; measured against an f64 FFT the route is at
        # 1-5e-7 in both norms for every length of this list, fused or not: profiles/r03_bluestein_error.log)
        assert l2 <= 1e-5 and mx <= 1e-5, f"bluestein N={n} {direction}: {l2:.2e} {mx:.2e}"
        oracle.assert_close_elementwise(got, want, 3e-4 * max(1.0, float(np.max(np.abs(want))) / 30), 3e-4, f"bluestein N={n}")


@pytest.mark.parametrize("shape", [[8, 4], [16, 16], [4, 8, 2], [96, 105], [24, 25, 27], [1024, 8], [64, 64, 4], [17, 4], [8, 19]])
def test_c2c_nd(fft, dev, oracle, shape):
    n, batch = int(np.prod(shape)), 2
    x = oracle.random_complex_batch(n, batch, 0xD000 + n).reshape(-1)
    got, _ = run_plan(fft, dev, {"type": "c2c", "shape": shape, "batch": batch, "direction": "forward", "normalize": "unitary"}, x, x.size)
    check(oracle, got, oracle.c2c_ref_batch(x, shape, batch, "forward", "unitary"), f"nd {shape}", 3e-3, 3e-3)


def test_c2c_nd_golden(fft, dev, oracle, manifest):
    cases, _ = manifest
    for name in ("c2c_nd_8x4_forward", "c2c_nd_16x16_inverse", "c2c_nd_4x8x2_forward"):
        c = cases[name]
        n = int(np.prod(c["shape"]))
        x = oracle.random_complex_interleaved(n, c["seed"])
        got, _ = run_plan(fft, dev, {"type": "c2c", "shape": c["shape"], "batch": 1, "direction": c["direction"], "normalize": c["normalize"]}, x, x.size)
        want = oracle.fftnd_ref(x, c["shape"], c["direction"], c["normalize"])
        assert format(oracle.fnv1a64(want), "016x") == c["out_fnv1a64"]
        check(oracle, got, want, name)


@pytest.mark.parametrize("n", [64, 60, 1 << 18, 1 << 20])
def test_strided_layout_whdcn(fft, dev, oracle, n):
    """layout.whdcn channel lanes on c2c: only the addressed lane is read / written.  Power-of-two lines run straight on the
    line kernel with the lane pitch; other lengths go through a gather / scatter pair"""
    batch, channels, cidx = 3, 4, 2
    logical = oracle.random_complex_batch(n, batch, 0xF00).reshape(-1)
    phys = np.full(2 * batch * channels * n, 9.0, np.float32)
    for b in range(batch):
        base = b * channels * n + cidx * n
        phys[2 * base:2 * (base + n)] = logical[2 * b * n:2 * (b + 1) * n]
    sentinel = np.full(phys.size, -4.0, np.float32)
    opts = {"type": "c2c", "shape": [n], "batch": batch, "direction": "forward", "normalize": "none",
            "layout": {"interleavedComplex": True, "whdcn": {"channels": channels, "channelIndex": cidx}}}
    got, (route, _) = run_plan(fft, dev, opts, phys, phys.size, out_init=sentinel)
    # (r03: four-step sizes ride the fused kernels with the lane pitch too)
    assert (route.startswith("lines[N=64,pitch=256/256]") if n == 64 else "lanes[pitch=" in route and "gather" not in route if n > 64 else ("gather" in route and "scatter" in route)), route
    want = sentinel.copy()
    ref = oracle.c2c_ref_batch(logical, [n], batch, "forward")
    for b in range(batch):
        base = b * channels * n + cidx * n
        want[2 * base:2 * (base + n)] = ref[2 * b * n:2 * (b + 1) * n]
    check(oracle, got, want, "whdcn lanes")


@pytest.mark.parametrize("fuse", [1, 0])
def test_r2c_c2r_ioview_and_zeropad(fft, dev, oracle, monkeypatch, fuse):
    """views on the real transforms: the real side's window lives on the real domain, the spectrum side's on the packed one;
    checked against the emu-tier numpy restatement (embed -> zero -> oracle -> zero -> extract).  fuse=1: the sides ride the r2c / c2r
    line kernels and the neighbouring c2c axis (no staging launch); fuse=0: the staging route"""
    monkeypatch.setenv("MI355FFT_FUSE_VIEWS", str(fuse))
    from test_emu_ioview import _embed, _extract, _zero_outside
    from mi355fft.layout import resolve_plan_options
    shape, batch = [256, 8], 3
    packed = [129, 8]
    opts = {"type": "r2c", "shape": shape, "batch": batch, "direction": "forward", "normalize": "none",
            "ioView": {"input": {"shape": [200, 5], "offset": [20, 1]}, "output": {"shape": [140, 9], "offset": [-4, 0], "clearOutside": True}},
            "zeroPad": {"read": {"start": [30, 0], "end": [210, 8]}, "write": {"start": [0, 1], "end": [100, 8]}}}
    r = resolve_plan_options(opts)
    x = oracle.random_real_batch(200 * 5, batch, 9911).reshape(-1)
    out_floats = 2 * 140 * 9 * batch
    sentinel = np.tile(np.array([77.0, -55.0], np.float32), out_floats // 2)
    got, (route, _) = run_plan(fft, dev, opts, x, out_floats, out_init=sentinel)
    if fuse:
        assert route.split() == ["lines-r2c-mapped[N=256]", "columns-mapped[N=8,S=129]"], route
    else:
        assert "embed" in route and "extract" in route and "zero-read" in route and "zero-write" in route
    logical = _embed(x, shape, r["io_view"]["input"], batch, 1)
    _zero_outside(logical, shape, r["zero_pad"]["read"])
    cplx = np.zeros((batch, *reversed(shape), 2), np.float32)
    cplx[..., 0] = logical[..., 0]
    y = oracle.c2c_ref_batch(cplx.reshape(-1), shape, batch, "forward", "none").reshape(batch, *reversed(shape), 2)[..., :129, :].copy()
    _zero_outside(y, packed, r["zero_pad"]["write"])
    want = _extract(y, packed, r["io_view"]["output"], batch, 2, sentinel)
    assert float(np.max(np.abs(got.astype(np.float64) - want))) <= 2e-5 * max(1.0, float(np.max(np.abs(want))))
    # c2r of a low-passed spectrum into a window of a larger real array
    n = 4096
    sig = oracle.random_real_batch(n, batch, 9912).reshape(-1)
    spec = np.concatenate([oracle.r2c_ref_packed(sig[b * n:(b + 1) * n], n, "none") for b in range(batch)]).reshape(batch, n // 2 + 1, 2)
    low = np.ascontiguousarray(spec[:, :300, :]).reshape(-1)
    opts = {"type": "c2r", "shape": [n], "batch": batch, "direction": "inverse", "normalize": "backward",
            "ioView": {"input": {"shape": [300]}, "output": {"shape": [5000], "offset": [-100]}}}
    sentinel = np.full(5000 * batch, 77.0, np.float32)
    got, (route, _) = run_plan(fft, dev, opts, low, 5000 * batch, out_init=sentinel)
    assert route.split() == (["lines-c2r-mapped[N=4096]"] if fuse else ["embed", "lines-c2r[N=4096]", "extract"]), route
    lp = spec.copy()
    lp[:, 300:, :] = 0
    want = sentinel.reshape(batch, 5000).copy()
    for b in range(batch):
        want[b, 100:100 + n] = oracle.c2r_ref_from_packed(lp[b].reshape(-1), n, "backward")
    assert float(np.max(np.abs(got.astype(np.float64) - want.reshape(-1)))) <= 3e-5 * max(1.0, float(np.max(np.abs(want)))), route
    assert np.count_nonzero(got == 77.0) == (5000 - n) * batch


@pytest.mark.parametrize("lg,fuse", [(20, 1), (20, 0), (17, 1), (18, 1), (19, 1), (21, 1)])
def test_c2c_view_of_a_four_step_line(fft, dev, oracle, monkeypatch, lg, fuse):
    """r03: rank-1 ioView + zeroPad of a 2^20-point line as predicates of the fused kernel's loads and stores (fuse=1: control-block reset +
    one launch, route free of embed / zero / extract) against the staging route (fuse=0) and the numpy restatement of the semantics"""
    monkeypatch.setenv("MI355FFT_FUSE_VIEWS", str(fuse))
    n, batch = 1 << lg, 11
    vin = {"shape": [n - 3000], "offset": [1000]}
    vout = {"shape": [n // 2 + 77], "offset": [-50], "clearOutside": False}
    zr, zw = {"start": [5000], "end": [n - 100]}, {"start": [64], "end": [n // 2 - 5]}
    rng = np.random.default_rng(12)
    x = rng.standard_normal(2 * vin["shape"][0] * batch).astype(np.float32)
    out_init = rng.standard_normal(2 * vout["shape"][0] * batch).astype(np.float32)
    opts = {"type": "c2c", "shape": [n], "batch": batch, "direction": "forward", "normalize": "none",
            "ioView": {"input": vin, "output": vout}, "zeroPad": {"read": zr, "write": zw}}
    got, (route, launches) = run_plan(fft, dev, opts, x, out_init.size, out_init=out_init)
    if fuse:
        assert "xcd-fused-view[N=" in route and launches == 2 and not any(w in route for w in ("embed", "extract", "zero-")), route
    else:
        assert "embed" in route and "extract" in route, route
    logical = np.zeros((batch, n, 2), np.float32)
    logical[:, 1000:n - 2000] = x.reshape(batch, -1, 2)
    logical[:, :zr["start"][0]] = 0
    logical[:, zr["end"][0]:] = 0
    y = oracle.c2c_ref_batch(logical.reshape(-1), [n], batch, "forward", "none").reshape(batch, n, 2)
    y[:, :zw["start"][0]] = 0
    y[:, zw["end"][0]:] = 0
    want = out_init.reshape(batch, -1, 2).copy()
    want[:, 50:] = y[:, :vout["shape"][0] - 50]
    check(oracle, got, want.reshape(-1), route, 2e-3, 2e-3)
    assert np.array_equal(got.reshape(batch, -1, 2)[:, :50], out_init.reshape(batch, -1, 2)[:, :50])


@pytest.mark.parametrize("fuse", [1, 0])
def test_c2c_ioview_and_zeropad(fft, dev, oracle, monkeypatch, fuse):
    """pad-in-read + embed-in-write (clearOutside) + range zeroing, checked against the emu-tier numpy restatement.  fuse=1: both
    sides ride the first load / last store of the line kernels (no separate passes); fuse=0: the staging route"""
    from test_emu_ioview import reference
    from mi355fft.layout import resolve_plan_options
    monkeypatch.setenv("MI355FFT_FUSE_VIEWS", str(fuse))
    shape, batch = [64, 8], 3
    opts = {"type": "c2c", "shape": shape, "batch": batch, "direction": "forward", "normalize": "unitary",
            "ioView": {"input": {"shape": [40, 8], "placement": "center"}, "output": {"shape": [80, 10], "placement": "center", "clearOutside": True}},
            "zeroPad": {"read": {"start": [4, 0], "end": [60, 8]}, "write": {"start": [0, 1], "end": [64, 7]}}}
    r = resolve_plan_options(opts)
    x = oracle.random_complex_interleaved(40 * 8 * batch, 4711)
    out_floats = 2 * 80 * 10 * batch
    sentinel = np.tile(np.array([77.0, -55.0], np.float32), out_floats // 2)
    got, (route, _) = run_plan(fft, dev, opts, x, out_floats, out_init=sentinel)
    if fuse:
        assert route.split() == ["lines-mapped[N=64]", "columns-mapped[N=8,S=64]"], route
    else:
        assert "embed" in route and "extract" in route and "zero-read" in route and "zero-write" in route
    want = reference(oracle, x, shape, batch, "forward", "unitary", r["io_view"]["input"], r["io_view"]["output"], r["zero_pad"]["read"],
                     r["zero_pad"]["write"], sentinel)
    assert float(np.max(np.abs(got.astype(np.float64) - want))) <= 2e-5 * max(1.0, float(np.max(np.abs(want))))
    # without clearOutside the surrounding elements keep their sentinel
    opts["ioView"]["output"]["clearOutside"] = False
    got, _ = run_plan(fft, dev, opts, x, out_floats, out_init=sentinel)
    r = resolve_plan_options(opts)
    want = reference(oracle, x, shape, batch, "forward", "unitary", r["io_view"]["input"], r["io_view"]["output"], r["zero_pad"]["read"],
                     r["zero_pad"]["write"], sentinel)
    assert float(np.max(np.abs(got.astype(np.float64) - want))) <= 2e-5 * max(1.0, float(np.max(np.abs(want))))
    assert np.count_nonzero(got == 77.0) > 0


# ---- r2c / c2r ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("lines_r2c", [1, 0])
@pytest.mark.parametrize("n", [2, 4, 8, 16, 64, 128, 1024, 4096, 8192, 1 << 14, 1 << 15, 1 << 16, 1 << 20, 6, 10, 30, 9, 15, 21, 17])
def test_r2c_c2r(fft, dev, oracle, monkeypatch, n, lines_r2c):
    """every r2c / c2r route by length; lines_r2c: the split fused into the line kernel (one launch, half lengths 64..16384 for
    r2c, 2..16384 for c2r) or the two-launch route"""
    monkeypatch.setenv("MI355FFT_LINES_R2C", str(lines_r2c))
    monkeypatch.setenv("MI355FFT_LINES_C2R", str(2 * lines_r2c))  # 2: the c2r twin for every length (default: N <= 2^14)
    batch = 37 if n <= 4096 else 2
    x = oracle.random_real_batch(n, batch, 0xE000 + n).reshape(-1)
    p = n // 2 + 1
    want = np.concatenate([oracle.r2c_ref_packed(x[b * n:(b + 1) * n], n, "none") for b in range(batch)])
    got, _ = run_plan(fft, dev, {"type": "r2c", "shape": [n], "batch": batch, "direction": "forward", "normalize": "none"}, x, 2 * p * batch)
    check(oracle, got, want, f"r2c N={n}", 8e-4, 8e-4)
    back, _ = run_plan(fft, dev, {"type": "c2r", "shape": [n], "batch": batch, "direction": "inverse", "normalize": "backward"}, want, n * batch)
    check(oracle, back, x, f"c2r(r2c) N={n}", 2e-3, 2e-3)


@pytest.mark.parametrize("fused", [0, 1])
@pytest.mark.parametrize("lg,batch", [(15, 1200), (16, 600), (17, 300), (18, 70), (19, 37), (20, 35), (21, 19), (22, 18)])
def test_r2c_four_step_sizes(fft, dev, oracle, monkeypatch, lg, batch, fused):
    """long power-of-two r2c on both routes: XCD-fused real four-step (one launch, more transforms than groups so that
    the workspace slots alternate) and the half-length c2c + split route"""
    monkeypatch.setenv("MI355FFT_XCD_FUSED", str(fused))
    monkeypatch.setenv("MI355FFT_LINES_R2C", "0")        # 2^15 would otherwise be one line-kernel launch (tested in test_r2c_c2r)
    monkeypatch.setenv("MI355FFT_LINE32K", "0")          # ... and 2^16 the half-length route over the 2^15 line kernel (test_c2c_line32k)
    n = 1 << lg
    p = n // 2 + 1
    x = oracle.random_real_batch(n, batch, 0xE100 + lg).reshape(-1)
    for norm in ("none", "unitary"):
        got, (route, launches) = run_plan(fft, dev, {"type": "r2c", "shape": [n], "batch": batch, "direction": "forward", "normalize": norm}, x, 2 * p * batch)
        assert route.startswith("xcd-r2c-rt[" if lg >= 21 else "xcd-r2c") == bool(fused), route   # 2^21, 2^22 (r03): real four-step on register tiles (kern_regtile.hpp)
        want = np.concatenate([oracle.r2c_ref_packed(x[b * n:(b + 1) * n], n, norm, use_pow2=True) for b in range(batch)])
        check(oracle, got, want, f"r2c 2^{lg} {norm} ({route.strip()})", 8e-4, 8e-4)


@pytest.mark.parametrize("fused", [0, 1])
@pytest.mark.parametrize("lg,batch", [(15, 1200), (16, 600), (17, 300), (18, 70), (19, 37), (20, 35), (21, 19), (22, 18)])
def test_c2r_four_step_sizes(fft, dev, oracle, monkeypatch, lg, batch, fused):
    """long power-of-two c2r on both routes (XCD-fused Hermitian four-step / half-length pre-split + inverse c2c):
    against the oracle's c2r of the oracle's own spectrum, and the round trip back to the signal"""
    monkeypatch.setenv("MI355FFT_XCD_FUSED", str(fused))
    monkeypatch.setenv("MI355FFT_LINES_R2C", "0")
    monkeypatch.setenv("MI355FFT_LINES_C2R", "3")    # 2^15 is a line-kernel launch by default since r02: keep the four-step under test
    monkeypatch.setenv("MI355FFT_LINE32K", "0")      # 2^16: likewise (half-length route over the 2^15 line kernel)
    n = 1 << lg
    p = n // 2 + 1
    x = oracle.random_real_batch(n, batch, 0xE200 + lg).reshape(-1)
    spec = np.concatenate([oracle.r2c_ref_packed(x[b * n:(b + 1) * n], n, "none", use_pow2=True) for b in range(batch)])
    got, (route, launches) = run_plan(fft, dev, {"type": "c2r", "shape": [n], "batch": batch, "direction": "inverse", "normalize": "backward"}, spec, n * batch)
    assert route.startswith("xcd-c2r-rt[" if lg >= 21 else "xcd-c2r") == bool(fused), route   # 2^21, 2^22 (r03): register tiles
    check(oracle, got, x, f"c2r(r2c) 2^{lg} ({route.strip()})", 2e-3, 2e-3)
    want = np.concatenate([oracle.c2r_ref_from_packed(spec[2 * b * p:2 * (b + 1) * p], n, "backward", use_pow2=True) for b in range(3)])
    check(oracle, got[:3 * n], want, f"c2r vs oracle 2^{lg}", 2e-3, 2e-3)


def test_r2c_golden_fixtures(fft, dev, oracle, manifest):
    cases, _ = manifest
    seen = 0
    for c in cases.values():
        if c["kind"] == "r2c_dft":
            x = oracle.random_real(c["N"], c["seed"])
            want = np.fromfile(os.path.join(GOLDEN, c["out_file"]), dtype=np.float32)
            got, _ = run_plan(fft, dev, {"type": "r2c", "shape": [c["N"]], "batch": 1, "direction": "forward", "normalize": c["normalize"]}, x, want.size)
            check(oracle, got, want, c["name"], 8e-4, 8e-4)
            seen += 1
        elif c["kind"] == "c2r_dft":
            xin = np.fromfile(os.path.join(GOLDEN, c["in_file"]), dtype=np.float32)
            want = np.fromfile(os.path.join(GOLDEN, c["out_file"]), dtype=np.float32)
            got, _ = run_plan(fft, dev, {"type": "c2r", "shape": [c["N"]], "batch": 1, "direction": "inverse", "normalize": c["normalize"]}, xin, want.size)
            check(oracle, got, want, c["name"], 2e-3, 2e-3)
            seen += 1
        elif c["kind"] == "r2c_pow2" and c["N"] <= (1 << 22):
            x = oracle.random_real(c["N"], c["seed"])
            got, _ = run_plan(fft, dev, {"type": "r2c", "shape": [c["N"]], "batch": 1, "direction": "forward", "normalize": "none"}, x, c["N"] + 2)
            idx = np.asarray(c["sample_idx"])
            want = np.asarray(c["sample_vals"], dtype=np.float32).reshape(-1, 2)
            scale = float(np.sqrt(np.mean(want.astype(np.float64) ** 2))) + 1e-30
            assert float(np.max(np.abs(got.reshape(-1, 2)[idx].astype(np.float64) - want))) <= 2e-4 * scale, c["name"]
            seen += 1
    assert seen >= 20


# ---- fftconv --------------------------------------------------------------------------------------------
def test_fftconv_golden_fixtures(fft, dev, oracle, manifest):
    cases, _ = manifest
    ran = 0
    for c in cases.values():
        if c["kind"] != "fftconv":
            continue
        shape, batch, K = c["shape"], c["batch"], c["kernelCount"]
        ks = c["kernelShape"] or shape
        fshape = shape if c["boundary"] == "circular" else [s + k - 1 for s, k in zip(shape, ks)]
        n, kn = int(np.prod(shape)), int(np.prod(ks))
        x = oracle.random_complex_interleaved(n * batch, c["seed"])
        kern = oracle.random_complex_interleaved(kn * K, c["kernel_seed"])
        want = np.fromfile(os.path.join(GOLDEN, c["out_file"]), dtype=np.float32)
        opts = {"type": "fftconv", "shape": shape, "batch": batch,
                "fftConv": {"mode": c["mode"], "boundary": c["boundary"], "kernelCount": K, "kernelShape": c["kernelShape"]}}
        kernels = [kern[2 * k * kn:2 * (k + 1) * kn] for k in range(K)]      # array-of-kernels form
        got, _ = run_plan(fft, dev, opts, x, want.size, kernel=kernels)
        check(oracle, got, want, c["name"], 4e-3, 4e-3)
        ran += 1
    assert ran >= 10


@pytest.mark.parametrize("shape,ks,boundary,mode,K", [([8192], None, "circular", "convolution", 2), ([16384], [100], "circular", "correlation", 1),
                                                       ([100], [29], "linear-full", "convolution", 3), ([1000], [25], "linear-same", "correlation", 2),
                                                       ([4000], [97], "linear-valid", "convolution", 1)])
def test_fftconv_product_fused_into_forward_lines(fft, dev, oracle, monkeypatch, shape, ks, boundary, mode, K):
    """1-D, power-of-two FFT length: the kernel-spectrum product rides the forward line FFT (fft_lines_mul_kernel); against the
    oracle's fftConvRef restatement and against the forward + pointwise route"""
    batch = 3
    n, kn = shape[0], (ks or shape)[0]
    x = oracle.random_complex_interleaved(n * batch, 0xC0DE + n)
    kern = oracle.random_complex_interleaved(kn * K, 0xC1DE + kn)
    opts = {"type": "fftconv", "shape": shape, "batch": batch, "fftConv": {"mode": mode, "boundary": boundary, "kernelCount": K, "kernelShape": ks}}
    want = np.concatenate([oracle.fftconv_ref(x, kern[2 * k * kn:2 * (k + 1) * kn], shape, batch, mode, boundary, ks)[0] for k in range(K)])
    kernels = [kern[2 * k * kn:2 * (k + 1) * kn] for k in range(K)]
    got, (route, _) = run_plan(fft, dev, opts, x, want.size, kernel=kernels)
    assert ("lines-mul-mapped[" if boundary != "circular" else "lines-mul[") in route, route   # linear modes: embed and crop ride the launches
    check(oracle, got, want, route, 4e-3, 4e-3)
    assert oracle.rel_l2(got, want) < 1e-5, route
    monkeypatch.setenv("MI355FFT_CONV_LINES", "0")
    old, (route0, _) = run_plan(fft, dev, opts, x, want.size, kernel=kernels)
    assert "lines-mul" not in route0, route0
    assert oracle.rel_l2(got, old) < 1e-6


@pytest.mark.parametrize("ks,mode,K,layout,batch", [(None, "convolution", 1, "kernel-major", 37), ([1000], "correlation", 3, "batch-major", 5)])
def test_fftconv_pipeline_2p20(fft, dev, oracle, monkeypatch, ks, mode, K, layout, batch):
    """2^20-point circular lines: forward transform, products and inverse transforms in ONE persistent launch (kern_regtile.hpp
    fft_xcd_conv1m_kernel; more data lines than groups, K = 3 re-runs the middle and last phase per kernel); against the oracle's
    fftConvRef restatement and against the forward + pointwise + inverse route (MI355FFT_CONV_PIPELINE=0)"""
    n = 1 << 20
    kn = (ks or [n])[0]
    x = oracle.random_complex_interleaved(n * batch, 0xC4DE)
    kern = oracle.random_complex_interleaved(kn * K, 0xC5DE)
    opts = {"type": "fftconv", "shape": [n], "batch": batch, "fftConv": {"mode": mode, "boundary": "circular", "kernelCount": K, "kernelShape": ks, "outputLayout": layout}}
    kernels = [kern[2 * k * kn:2 * (k + 1) * kn] for k in range(K)]
    got, (route, _) = run_plan(fft, dev, opts, x, 2 * n * batch * K, kernel=kernels)
    assert "fftconv-pipeline[N=1024x1024,K=%d]" % K in route, route
    nb = min(batch, 3)                     # the oracle on the first data lines ...
    per = [oracle.fftconv_ref(x[:2 * n * nb], kernels[k], [n], nb, mode, "circular", ks, use_pow2=True)[0].reshape(nb, 2 * n) for k in range(K)]
    g = got.reshape(K, batch, 2 * n) if layout == "kernel-major" else got.reshape(batch, K, 2 * n).transpose(1, 0, 2)
    for k in range(K):
        check(oracle, g[k, :nb].reshape(-1), per[k].reshape(-1), f"{route.strip()} kernel {k}", 4e-3, 4e-3)
        assert oracle.rel_l2(g[k, :nb].reshape(-1), per[k].reshape(-1)) < 1e-5, route
    monkeypatch.setenv("MI355FFT_CONV_PIPELINE", "0")   # ... and every line against the composed route
    old, (route0, _) = run_plan(fft, dev, opts, x, 2 * n * batch * K, kernel=kernels)
    assert "fftconv-pipeline" not in route0, route0
    assert oracle.rel_l2(got, old) < 1e-6


def test_fftconv_cfg4_channel_lane_preset(fft, dev, oracle, manifest):
    """BASELINE config 4 with sentinel preservation (mirror of complete.suite.js:4812-4830)"""
    cases, _ = manifest
    c = cases["fftconv_cfg4_N256_b4_k3"]
    preset = fft.createFftConvKernelMajorChannelLanePreset({"shape": [256], "batch": 4, "kernelCount": 3, "input": {"channels": 64},
                                                            "output": {"channels": 128, "kernelStepChannels": 16}})
    n, batch, K = 256, 4, 3
    logical = oracle.random_complex_interleaved(n * batch, c["seed"])
    kern = oracle.random_complex_interleaved(n * K, c["kernel_seed"])
    phys_in = np.full(2 * 4 * 64 * 256, 5.0, np.float32)
    for b in range(batch):
        phys_in[2 * b * 16384:2 * (b * 16384 + n)] = logical[2 * b * n:2 * (b + 1) * n]
    out_elems = 4 * 128 * 256
    sentinel = np.empty(2 * out_elems, np.float32)
    sentinel[0::2], sentinel[1::2] = 77.0, -55.0
    got, (route, launches) = run_plan(fft, dev, dict(preset, type="fftconv"), phys_in, sentinel.size, kernel=kern, out_init=sentinel)
    gold = np.fromfile(os.path.join(GOLDEN, c["out_file"]), dtype=np.float32).reshape(K, batch, 2 * n)
    want = sentinel.copy()
    lanes = np.zeros(out_elems, bool)
    for k in range(K):
        for b in range(batch):
            lane = k * 16 * 256 + b * 32768
            want[2 * lane:2 * (lane + n)] = gold[k, b]
            lanes[lane:lane + n] = True
    oracle.assert_close_elementwise(got, want, 5e-3, 5e-3, "cfg4 lanes")
    assert oracle.rel_l2(got.reshape(-1, 2)[lanes], want.reshape(-1, 2)[lanes]) < TOL
    assert np.array_equal(got.reshape(-1, 2)[~lanes], sentinel.reshape(-1, 2)[~lanes])


# ---- properties at the BASELINE sizes ---------------------------------------------------------------------
def _full_size_properties(fft, dev, oracle, n, batch, seed):
    """forward then inverse(backward) returns the input; Parseval; a sampled transform matches the oracle"""
    info = dev.info()
    bytes_needed = 3 * n * batch * 8
    if info["hbm_free"] < bytes_needed + (2 << 30):
        pytest.skip(f"needs {bytes_needed >> 30} GiB of HBM")
    a = dev.createBuffer({"size": n * batch * 8})
    b = dev.createBuffer({"size": n * batch * 8})
    c = dev.createBuffer({"size": n * batch * 8})
    dev.fillRandom(a, 0, 2 * n, batch, seed, 0)
    fwd = fft.createPlan(dev, {"type": "c2c", "shape": [n], "batch": batch, "direction": "forward", "normalize": "none"})
    inv = fft.createPlan(dev, {"type": "c2c", "shape": [n], "batch": batch, "direction": "inverse", "normalize": "backward"})
    enc = dev.createCommandEncoder()
    fwd.exec(enc, {"input": a, "output": b})
    inv.exec(enc, {"input": b, "output": c})          # two execs in one encoder run in order
    dev.queue.submit([enc.finish()])
    dev.queue.onSubmittedWorkDone()
    e_in = dev.sumsq(a, 0, 2 * n * batch)
    e_out = dev.sumsq(b, 0, 2 * n * batch)
    assert abs(e_out / (n * e_in) - 1.0) < 1e-5, "Parseval"
    rt = dev.diffSumsq(c, 0, a, 0, 1.0, 2 * n * batch)
    assert np.sqrt(rt / e_in) < 1e-5, f"round trip rel_l2={np.sqrt(rt / e_in):.3e}"
    # sampled transforms against the oracle (first, middle, last)
    for t in (0, batch // 2, batch - 1):
        x = oracle.random_complex_interleaved(n, oracle.stream_seed(seed, t))
        dev_in = fft.downloadF32(dev, a, 2 * n, t * n * 8)
        assert np.array_equal(dev_in, x), "device PRNG twin"
        got = fft.downloadF32(dev, b, 2 * n, t * n * 8)
        want = oracle.fft1d_ref(x, n, "forward")
        l2, mx = oracle.rel_l2(got, want), oracle.rel_max(got, want)
        assert l2 <= TOL and mx <= TOL, f"transform {t}: rel_l2={l2:.3e} rel_max={mx:.3e}"
    for p in (fwd, inv):
        p.destroy()
    for buf in (a, b, c):
        buf.destroy()


def test_cfg2_full_size_properties(fft, dev, oracle):
    """BASELINE config 2: N=1024 batch=65536"""
    _full_size_properties(fft, dev, oracle, 1024, 65536, 0x5EED0002)


def test_cfg3_full_size_properties(fft, dev, oracle):
    """BASELINE config 3 (north-star metric): N=2^20 batch=4096 — 32 GiB per buffer"""
    _full_size_properties(fft, dev, oracle, 1 << 20, 4096, 0x5EED0003)


@pytest.mark.parametrize("lg,batch", [(21, 1024), (22, 512)])
def test_c2c_register_tile_sizes_full_size_properties(fft, dev, oracle, lg, batch):
    """the register-tile instances (kern_regtile.hpp) at bench-sized batches — 16 GiB per buffer: every group walks 64-128 transforms"""
    _full_size_properties(fft, dev, oracle, 1 << lg, batch, 0x5EED0021 + lg)


def test_fftconv_pipeline_full_size_matches_composed_route(fft, dev, oracle, monkeypatch):
    """fftconv of 2^20-point lines x 384 data lines, one full-length kernel: the one-launch pipeline against the forward + pointwise +
    inverse route over the whole batch (device-side difference), plus conv(x, delta) = x as an absolute anchor"""
    n, batch = 1 << 20, 384
    x = dev.createBuffer({"size": n * batch * 8})
    dev.fillRandom(x, 0, 2 * n, batch, 0x5EED0C01, 0)
    kern = oracle.random_complex_interleaved(n, 0x5EED0C02)
    opts = {"type": "fftconv", "shape": [n], "batch": batch, "fftConv": {"mode": "convolution", "boundary": "circular", "kernelCount": 1}}
    outs = []
    for pipeline in ("1", "0"):
        monkeypatch.setenv("MI355FFT_CONV_PIPELINE", pipeline)
        plan = fft.createPlan(dev, opts)
        assert ("fftconv-pipeline" in plan.describe()[0]) == (pipeline == "1"), plan.describe()
        y = dev.createBuffer({"size": n * batch * 8})
        kb = fft.uploadComplex(dev, kern)
        enc = dev.createCommandEncoder()
        plan.exec(enc, {"input": x, "output": y, "kernel": kb})
        dev.queue.submit([enc.finish()])
        dev.queue.onSubmittedWorkDone()
        outs.append(y)
        plan.destroy()
        kb.destroy()
    e = dev.sumsq(outs[1], 0, 2 * n * batch)
    d = dev.diffSumsq(outs[0], 0, outs[1], 0, 1.0, 2 * n * batch)
    assert np.sqrt(d / e) < 1e-6, f"pipeline vs composed rel_l2={np.sqrt(d / e):.3e}"
    # identity kernel: y = x
    delta = np.zeros(2 * n, np.float32)
    delta[0] = 1.0
    monkeypatch.setenv("MI355FFT_CONV_PIPELINE", "1")
    plan = fft.createPlan(dev, opts)
    kb = fft.uploadComplex(dev, delta)
    enc = dev.createCommandEncoder()
    plan.exec(enc, {"input": x, "output": outs[0], "kernel": kb})
    dev.queue.submit([enc.finish()])
    dev.queue.onSubmittedWorkDone()
    e_in = dev.sumsq(x, 0, 2 * n * batch)
    d = dev.diffSumsq(outs[0], 0, x, 0, 1.0, 2 * n * batch)
    assert np.sqrt(d / e_in) < 1e-5, f"conv with delta rel_l2={np.sqrt(d / e_in):.3e}"
    plan.destroy()
    kb.destroy()
    for b in outs + [x]:
        b.destroy()


def test_cfg5_shard_full_size_properties(fft, dev, oracle):
    """BASELINE config 5, one GPU's shard: r2c N=2^22 x 1024 transforms (16 GiB in, 16 GiB out): c2r(r2c(x)) = x, Parseval
    for the packed spectrum, and sampled transforms against the oracle"""
    n, batch, seed = 1 << 22, 1024, 0x5EED0005
    p = n // 2 + 1
    need = (2 * n * batch * 4) + p * batch * 8
    if dev.info()["hbm_free"] < need + (4 << 30):
        pytest.skip(f"needs {need >> 30} GiB of HBM")
    x = dev.createBuffer({"size": n * batch * 4})
    spec = dev.createBuffer({"size": p * batch * 8})
    back = dev.createBuffer({"size": n * batch * 4})
    dev.fillRandom(x, 0, n, batch, seed, 0)
    fwd = fft.createPlan(dev, {"type": "r2c", "shape": [n], "batch": batch, "direction": "forward", "normalize": "none"})
    inv = fft.createPlan(dev, {"type": "c2r", "shape": [n], "batch": batch, "direction": "inverse", "normalize": "backward"})
    enc = dev.createCommandEncoder()
    fwd.exec(enc, {"input": x, "output": spec})
    inv.exec(enc, {"input": spec, "output": back})
    dev.queue.submit([enc.finish()])
    dev.queue.onSubmittedWorkDone()
    e_in = dev.sumsq(x, 0, n * batch)
    rt = dev.diffSumsq(back, 0, x, 0, 1.0, n * batch)
    assert np.sqrt(rt / e_in) < 1e-5, f"round trip rel_l2={np.sqrt(rt / e_in):.3e}"
    for t in (0, batch // 2, batch - 1):
        xin = fft.downloadF32(dev, x, n, t * n * 4)
        assert np.array_equal(xin, oracle.random_real(n, oracle.stream_seed(seed, t)))
        got = fft.downloadF32(dev, spec, 2 * p, t * p * 8)
        want = oracle.r2c_ref_packed(xin, n, "none", use_pow2=True)
        l2, mx = oracle.rel_l2(got, want), oracle.rel_max(got, want)
        assert l2 <= TOL and mx <= TOL, f"transform {t}: rel_l2={l2:.3e} rel_max={mx:.3e}"
        # Parseval on the packed half spectrum: sum|x|^2 = (|X0|^2 + |X_{N/2}|^2 + 2 sum_{0<k<N/2} |X_k|^2) / N
        g = got.astype(np.float64).reshape(-1, 2)
        pw = (g ** 2).sum(axis=1)
        assert abs((pw[0] + pw[-1] + 2 * pw[1:-1].sum()) / n / float(np.sum(xin.astype(np.float64) ** 2)) - 1.0) < 1e-5
    for pl in (fwd, inv):
        pl.destroy()
    for b in (x, spec, back):
        b.destroy()


def test_copy_buffer_to_buffer_streaming_kernel(fft, dev):
    """copyBufferToBuffer: 16-byte-aligned ranges of 1 MiB and more take the one-shot streaming kernel (ragged last slab, offsets on both
    sides), anything else the runtime copy; bit-exact, bytes around the destination range untouched"""
    rng = np.random.default_rng(7)
    for nbytes, so, do in (((1 << 20) + 16 * 37, 48, 16), ((5 << 20) + 16, 0, 1024), ((1 << 20) + 4, 16, 16), (4096, 4, 8)):
        n = nbytes // 4
        src = rng.standard_normal(n + 64).astype(np.float32)
        sb = dev.createBuffer({"size": src.nbytes})
        db = dev.createBuffer({"size": src.nbytes + 4096})
        dev.queue.writeBuffer(sb, 0, src)
        guard = np.full((src.nbytes + 4096) // 4, -7.0, dtype=np.float32)
        dev.queue.writeBuffer(db, 0, guard)
        enc = dev.createCommandEncoder()
        enc.copyBufferToBuffer(sb, so, db, do, nbytes)
        dev.queue.submit([enc.finish()])
        dev.queue.onSubmittedWorkDone()
        got = fft.downloadF32(dev, db, guard.size, 0)
        want = guard.copy()
        want[do // 4:do // 4 + n] = src[so // 4:so // 4 + n]
        assert np.array_equal(got, want), (nbytes, so, do)
        sb.destroy()
        db.destroy()


def test_batches_of_2p24_lines(fft, dev, oracle):
    """16 Mi lines in one plan: one-shot grids are capped below HIP's 2^32-threads-per-launch limit and the kernels walk the rest
    (r2c N=64 x 2^24 used to fail with "invalid configuration argument" in its split launch).  Round trip over the whole batch,
    first / last transforms against the oracle."""
    n, batch, seed = 64, 1 << 24, 0x5EED0009
    p = n // 2 + 1
    x = dev.createBuffer({"size": n * batch * 4})
    spec = dev.createBuffer({"size": p * batch * 8})
    back = dev.createBuffer({"size": n * batch * 4})
    dev.fillRandom(x, 0, n, batch, seed, 0)
    fwd = fft.createPlan(dev, {"type": "r2c", "shape": [n], "batch": batch, "direction": "forward", "normalize": "none"})
    inv = fft.createPlan(dev, {"type": "c2r", "shape": [n], "batch": batch, "direction": "inverse", "normalize": "backward"})
    enc = dev.createCommandEncoder()
    fwd.exec(enc, {"input": x, "output": spec})
    inv.exec(enc, {"input": spec, "output": back})
    dev.queue.submit([enc.finish()])
    dev.queue.onSubmittedWorkDone()
    e_in = dev.sumsq(x, 0, n * batch)
    rt = dev.diffSumsq(back, 0, x, 0, 1.0, n * batch)
    assert np.sqrt(rt / e_in) < 1e-5, f"round trip rel_l2={np.sqrt(rt / e_in):.3e}"
    for t in (0, batch - 1):
        xin = fft.downloadF32(dev, x, n, t * n * 4)
        got = fft.downloadF32(dev, spec, 2 * p, t * p * 8)
        want = oracle.r2c_ref_packed(xin, n, "none", use_pow2=True)
        assert oracle.rel_l2(got, want) <= TOL, f"transform {t}"
    for pl in (fwd, inv):
        pl.destroy()
    for b in (x, spec, back):
        b.destroy()
    # c2c lines of 8 points, 2^25 of them: the one-shot line grid
    n, batch = 8, 1 << 25
    a = dev.createBuffer({"size": n * batch * 8})
    b2 = dev.createBuffer({"size": n * batch * 8})
    dev.fillRandom(a, 0, 2 * n, batch, seed + 1, 0)
    pl = fft.createPlan(dev, {"type": "c2c", "shape": [n], "batch": batch, "direction": "forward", "normalize": "none"})
    enc = dev.createCommandEncoder()
    pl.exec(enc, {"input": a, "output": b2})
    dev.queue.submit([enc.finish()])
    dev.queue.onSubmittedWorkDone()
    for t in (0, batch - 1):
        xin = fft.downloadComplex(dev, a, n, t * n * 8)
        got = fft.downloadComplex(dev, b2, n, t * n * 8)
        assert oracle.rel_l2(got, oracle.c2c_ref_batch(xin, [n], 1, "forward")) <= TOL, f"line {t}"
    pl.destroy()
    a.destroy()
    b2.destroy()


def test_linearity_at_2p20(fft, dev, oracle):
    n, batch = 1 << 20, 4
    xs = [oracle.random_complex_batch(n, batch, 0x7100 + i).reshape(-1) for i in range(2)]
    opts = {"type": "c2c", "shape": [n], "batch": batch, "direction": "forward", "normalize": "none"}
    fa, _ = run_plan(fft, dev, opts, xs[0], xs[0].size)
    fb, _ = run_plan(fft, dev, opts, xs[1], xs[1].size)
    fs, _ = run_plan(fft, dev, opts, (2.0 * xs[0] - 0.5 * xs[1]).astype(np.float32), xs[0].size)
    assert oracle.rel_l2(fs, 2.0 * fa.astype(np.float64) - 0.5 * fb.astype(np.float64)) < 1e-5


def test_impulse_and_constant(fft, dev, oracle):
    n = 1 << 20
    x = np.zeros(2 * n, np.float32)
    x[2 * 5] = 1.0                                    # delta at n=5 -> e^{-2 pi i 5k/N}
    got, _ = run_plan(fft, dev, {"type": "c2c", "shape": [n], "batch": 1, "direction": "forward", "normalize": "none"}, x, x.size)
    k = np.arange(n, dtype=np.float64)
    want = np.exp(-2j * np.pi * 5 * k / n)
    g = got.astype(np.float64).view(np.complex128)
    assert np.max(np.abs(g - want)) < 2e-6
    x = np.zeros(2 * n, np.float32)
    x[0::2] = 1.0
    got, _ = run_plan(fft, dev, {"type": "c2c", "shape": [n], "batch": 1, "direction": "forward", "normalize": "none"}, x, x.size)
    assert abs(got[0] - n) < 1e-3 * n and np.max(np.abs(got[2:])) < 1e-2


@pytest.mark.parametrize("lg", [16, 20, 21])
def test_accuracy_against_f64(fft, dev, oracle, monkeypatch, lg):
    """error against an f64 FFT: both four-step routes stay within a few f32 roundings (measured 1.6e-7 ... 2.4e-7; the
    oracle, which keeps f64 twiddles and rounds its data to f32 per stage, sits at 1.0e-7 ... 1.2e-7) — 40x inside the
    1e-5 parity tolerance"""
    n = 1 << lg
    x = oracle.random_complex_batch(n, 1, 0xACC0 + lg).reshape(-1)
    exact = np.fft.fft(x.astype(np.float64).view(np.complex128)).view(np.float64)
    ref_err = oracle.rel_l2(oracle.c2c_ref_batch(x, [n], 1, "forward", "none"), exact)
    errs = {}
    for fused in (0, 1):
        monkeypatch.setenv("MI355FFT_XCD_FUSED", str(fused))
        got, (route, _) = run_plan(fft, dev, {"type": "c2c", "shape": [n], "batch": 1, "direction": "forward", "normalize": "none"}, x, x.size)
        errs[route.strip()] = oracle.rel_l2(got, exact)
    print(f"N=2^{lg}: rel_l2 vs f64: {errs}, oracle {ref_err:.3e}")
    for route, e in errs.items():
        assert e < 4e-7 and e < 4.0 * ref_err, (route, e, ref_err)


@pytest.mark.parametrize("typ", ["dct1", "dct2", "dct3", "dct4", "dst1", "dst2", "dst3", "dst4"])
def test_dct_dst(fft, dev, oracle, typ):
    """real-to-real transforms (dct_fft.js) through the pre / FFT / post route: 1-D sizes that hit the line, mixed-radix and
    Bluestein FFTs underneath, and an N-D array; against the oracle's restatement of math.js:291-409"""
    for shape, batch in (([16], 5), ([17], 3), ([1000], 4), ([4096], 2), ([12, 10, 3], 2)):
        n = int(np.prod(shape))
        x = oracle.random_real_batch(n, batch, 0x7A16 + n).reshape(-1)
        for direction, norm in (("forward", "none"), ("inverse", "backward")):
            got, (route, _) = run_plan(fft, dev, {"type": typ, "shape": shape, "batch": batch, "direction": direction, "normalize": norm,
                                                   "layout": {"interleavedComplex": False}}, x, x.size)
            want = oracle.trig_ref_batch(x, shape, batch, typ, direction, norm)
            scale = max(1.0, float(np.max(np.abs(want))))
            # the reference's own tolerance for these plans is 2e-3 (complete.suite.js:3932); f32 FFTs of length 2N do far better
            assert float(np.max(np.abs(got.astype(np.float64) - want))) <= 1e-4 * scale, (typ, shape, direction, route)


@pytest.mark.parametrize("typ", ["dct1", "dct2", "dct3", "dct4", "dst1", "dst2", "dst3", "dst4"])
@pytest.mark.parametrize("lg", [13, 16, 20])
def test_dct_dst_real_fft_route_long(fft, dev, oracle, monkeypatch, typ, lg):
    """long dense lines take the real-FFT route (kern_trig.hpp kinds 8..15) over lines-r2c / the real four-step; too long for the
    O(N^2) oracle, so: agreement with the general 2N-point route (itself oracle-checked above) and with scipy's f64 transforms"""
    sfft = pytest.importorskip("scipy.fft")
    n, batch = (1 << lg) + {"dct1": 1, "dst1": -1}.get(typ, 0), 3       # dct1 / dst1: the extension 2(N -/+ 1) is the power of two
    x = oracle.random_real_batch(n, batch, 0x7C00 + lg).reshape(-1)
    opts = {"type": typ, "shape": [n], "batch": batch, "direction": "forward", "normalize": "none", "layout": {"interleavedComplex": False}}
    got, (route, launches) = run_plan(fft, dev, opts, x, x.size)
    if lg == 13 and typ[3] in "23":       # half length 4096: the whole DCT-II / DCT-III (DST) is ONE line-kernel launch (r02)
        assert route.strip() == "lines-%s[N=%d]" % (typ, n) and launches == 1, route
        monkeypatch.setenv("MI355FFT_TRIG_FUSED", "0")
        three, (route3, _) = run_plan(fft, dev, opts, x, x.size)
        assert route3.startswith("trig-real["), route3
        assert float(np.max(np.abs(got.astype(np.float64) - three.astype(np.float64)))) <= 2e-5 * float(np.sqrt(np.mean(three.astype(np.float64) ** 2))) * 8
    else:
        assert route.startswith("trig-real["), route
    monkeypatch.setenv("MI355FFT_TRIG_REAL", "0")
    old, (route0, _) = run_plan(fft, dev, opts, x, x.size)
    assert "trig-real" not in route0, route0
    xs = x.reshape(batch, n).astype(np.float64)
    ref = {"dct2": lambda v: sfft.dct(v, type=2) / 2, "dct3": lambda v: sfft.dct(v, type=3) / 2, "dct4": lambda v: sfft.dct(v, type=4) / 2,
           "dst2": lambda v: sfft.dst(v, type=2) / 2, "dst3": lambda v: sfft.dst(v, type=3) / 2, "dst4": lambda v: sfft.dst(v, type=4) / 2,
           "dct1": lambda v: sfft.dct(v, type=1), "dst1": lambda v: sfft.dst(v, type=1) / 2}[typ](xs).reshape(-1)
    rms = float(np.sqrt(np.mean(ref ** 2)))
    assert float(np.max(np.abs(got.astype(np.float64) - ref))) <= 2e-5 * rms * 8, (typ, lg, route)
    assert float(np.sqrt(np.mean((got.astype(np.float64) - ref) ** 2))) <= 1e-6 * rms, (typ, lg, route)
    assert float(np.max(np.abs(got.astype(np.float64) - old.astype(np.float64)))) <= 2e-5 * rms * 8, (typ, lg, route, route0)


@pytest.mark.parametrize("typ", ["dct1", "dct2", "dct3", "dct4", "dst1", "dst2", "dst3", "dst4"])
def test_dct_dst_nd_strided_axes(fft, dev, oracle, monkeypatch, typ):
    """N-D real-to-real transforms: axes >= 1 go through the tiled permutation / phase passes (kern_trig.hpp) around dense FFTs.
    Small arrays against the oracle; 1024 x 512 against scipy's f64 dctn / dstn and against the general route"""
    sfft = pytest.importorskip("scipy.fft")
    for shape, batch in (([40, 36], 2), ([6, 70, 4], 2)):
        n = int(np.prod(shape))
        x = oracle.random_real_batch(n, batch, 0x7D00 + n).reshape(-1)
        opts = {"type": typ, "shape": shape, "batch": batch, "direction": "forward", "normalize": "backward", "layout": {"interleavedComplex": False}}
        got, (route, _) = run_plan(fft, dev, opts, x, x.size)
        assert route.count("trig-real[") + route.count("lines-d") >= 2, route
        want = oracle.trig_ref_batch(x, shape, batch, typ, "forward", "backward")
        assert float(np.max(np.abs(got.astype(np.float64) - want))) <= 3e-5 * max(1.0, float(np.max(np.abs(want)))), (typ, shape, route)
    shape, batch = [1024, 512], 3
    n = shape[0] * shape[1]
    x = oracle.random_real_batch(n, batch, 0x7D77).reshape(-1)
    opts = {"type": typ, "shape": shape, "batch": batch, "direction": "forward", "normalize": "none", "layout": {"interleavedComplex": False}}
    got, (route, _) = run_plan(fft, dev, opts, x, x.size)
    assert route.count("trig-real[") + route.count("lines-d") == 2, route   # axis 0 may be the one-launch DCT (lines-dct2 ...)
    monkeypatch.setenv("MI355FFT_TRIG_REAL", "0")
    old, (route0, _) = run_plan(fft, dev, opts, x, x.size)
    assert "trig-real" not in route0, route0
    xs = x.reshape(batch, shape[1], shape[0]).astype(np.float64)
    k = int(typ[3])
    f = sfft.dctn if typ.startswith("dct") else sfft.dstn
    ref = (f(xs, type=k, axes=(1, 2)) / (1.0 if typ == "dct1" else 4.0)).reshape(-1)
    rms = float(np.sqrt(np.mean(ref ** 2)))
    assert float(np.sqrt(np.mean((got.astype(np.float64) - ref) ** 2))) <= 1e-6 * rms, (typ, route)
    assert float(np.max(np.abs(got.astype(np.float64) - ref))) <= 2e-4 * rms, (typ, route)
    assert float(np.max(np.abs(got.astype(np.float64) - old.astype(np.float64)))) <= 2e-4 * rms, (typ, route, route0)


def test_torch_fft_cross_check(fft, dev, oracle):
    """independent f32 FFT (torch.fft on the same GPU) agrees to f32 rounding — not the parity oracle"""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("torch sees no GPU")
    n, batch = 1 << 16, 4
    x = oracle.random_complex_batch(n, batch, 0x7777).reshape(-1)
    got, _ = run_plan(fft, dev, {"type": "c2c", "shape": [n], "batch": batch, "direction": "forward", "normalize": "none"}, x, x.size)
    t = torch.from_numpy(x.reshape(batch, n, 2).copy()).cuda()
    ref = torch.view_as_real(torch.fft.fft(torch.view_as_complex(t), dim=1)).cpu().numpy().reshape(-1)
    assert oracle.rel_l2(got, ref) < 2e-6


# ---- safety paths of the kernels whose workgroups wait for each other ---------------------------------------
def test_xcd_timeout_fails_the_readback_then_falls_back(fft, dev, oracle, monkeypatch):
    """A bounded wait that gives up (forced: one poll) must never be read back as a result: submit -> downloadComplex is the
    reference's normal flow (no onSubmittedWorkDone), so the readback itself reports it.  After that the device plans, and the
    same plan at its next exec runs, the routes without cross-workgroup synchronisation — and the result is right."""
    import mi355fft
    d2 = mi355fft.Device(0)          # own device object: the fallback state is per device and must not leak into other tests
    try:
        monkeypatch.setenv("MI355FFT_XCD_SPIN_LIMIT", "1")
        n, batch = 1 << 20, 16
        x = oracle.random_complex_batch(n, batch, 0x7117).reshape(-1)
        inp = mi355fft.uploadComplex(d2, x)
        out = d2.createBuffer({"size": x.nbytes})
        plan = mi355fft.createPlan(d2, {"type": "c2c", "shape": [n], "batch": batch, "direction": "forward", "normalize": "none"})
        assert plan.describe()[0].startswith("xcd-fused"), plan.describe()
        enc = d2.createCommandEncoder()
        plan.exec(enc, {"input": inp, "output": out})
        cb = enc.finish()
        d2.queue.submit([cb])
        with pytest.raises(mi355fft.Mi355Error, match="gave up waiting"):
            mi355fft.downloadF32(d2, out, 2 * n)                       # NOT preceded by a queue wait
        monkeypatch.delenv("MI355FFT_XCD_SPIN_LIMIT")
        enc = d2.createCommandEncoder()
        plan.exec(enc, {"input": inp, "output": out})                  # re-planned here
        assert plan.describe()[0].startswith("two-pass["), plan.describe()
        with pytest.raises(mi355fft.Mi355Error, match="destroyed plan or buffer"):
            d2.queue.submit([cb])                                      # the old list points into the old tables
        d2.queue.submit([enc.finish()])
        got = mi355fft.downloadF32(d2, out, 2 * n * batch)
        check(oracle, got, oracle.c2c_ref_batch(x, [n], batch, "forward", "none"), "after the fallback")
        p2 = mi355fft.createPlan(d2, {"type": "c2c", "shape": [1 << 19], "batch": 2, "direction": "forward", "normalize": "none"})
        assert p2.describe()[0].startswith("two-pass["), p2.describe()   # new plans on this device too
        p3 = mi355fft.createPlan(d2, {"type": "c2c", "shape": [1 << 16], "batch": 2, "direction": "forward", "normalize": "none"})
        assert p3.describe()[0].startswith("xcd-solo["), p3.describe()   # no cross-workgroup waits: stays
        for p in (plan, p2, p3):
            p.destroy()
        inp.destroy()
        out.destroy()
    finally:
        d2.close()


def test_submit_after_destroy_is_refused(fft, dev, oracle):
    """a recorded command list holds raw pointers into the plan's tables and the caller's buffers: once either is destroyed the
    list can no longer be submitted (the reference raises a validation error; here MI355FFT_ERR_DESTROYED)"""
    n = 1024
    x = oracle.random_complex_batch(n, 4, 0xDE57).reshape(-1)
    for victim in ("plan", "output"):
        inp = fft.uploadComplex(dev, x)
        out = dev.createBuffer({"size": x.nbytes})
        plan = fft.createPlan(dev, {"type": "c2c", "shape": [n], "batch": 4, "direction": "forward", "normalize": "none"})
        enc = dev.createCommandEncoder()
        plan.exec(enc, {"input": inp, "output": out})
        cb = enc.finish()
        dev.queue.submit([cb])
        dev.queue.onSubmittedWorkDone()
        (plan if victim == "plan" else out).destroy()
        with pytest.raises(fft.Mi355Error, match="destroyed plan or buffer"):
            dev.queue.submit([cb])
        cb.release()
        plan.destroy()
        inp.destroy()
        if victim == "plan":
            out.destroy()

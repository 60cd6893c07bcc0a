"""CPU tier: DCT-I..IV / DST-I..IV plans (kern_trig.hpp pre / post passes around the complex FFT routes) under host emulation,
against the oracle's restatement of the reference's dct*Ref / dst*Ref (math.js:291-409, pinned by tests/golden trig_* fixtures)."""
import numpy as np
import pytest

import emu_harness as emu
from mi355fft import _abi
from mi355fft.layout import resolve_plan_options

TYPES = ["dct1", "dct2", "dct3", "dct4", "dst1", "dst2", "dst3", "dst4"]


def _desc(opts):
    r = resolve_plan_options(opts)
    return _abi.make_desc(r["type"], r["shape"], r["batch"], r["direction"], r["normalize"], r["inPlace"], r["input_layout"], r["output_layout"],
                          None, r["io_view"], r["zero_pad"]), r


@pytest.mark.parametrize("typ", TYPES)
@pytest.mark.parametrize("n", [2, 3, 16, 17, 30, 64])
def test_trig_1d(oracle, typ, n):
    batch = 3
    x = oracle.random_real_batch(n, batch, 0x7A16 + n).reshape(-1)
    for direction, norm in (("forward", "none"), ("inverse", "backward"), ("forward", "unitary")):
        desc, _ = _desc({"type": typ, "shape": [n], "batch": batch, "direction": direction, "normalize": norm, "layout": {"interleavedComplex": False}})
        got, route, _ = emu.run_plan(desc, x, x.size)
        want = oracle.trig_ref_batch(x, [n], batch, typ, direction, norm)
        scale = max(1.0, float(np.max(np.abs(want))))
        assert float(np.max(np.abs(got.astype(np.float64) - want))) <= 2e-5 * scale, (typ, n, direction, route)


@pytest.mark.parametrize("typ", TYPES)
@pytest.mark.parametrize("n,fused,backend", [(4, "1", "r2c-split"), (100, "1", "r2c-split"), (256, "1", "lines-r2c"), (4096, "1", "lines-r2c"),
                                             (4096, "2", "xcd-r2c"), (256, "1", "lines-dct2"), (2048, "1", "lines-dct2"), (2048, "1", "lines-dct2-plain-shape"),
                                             (4096, "1", "lines-dct2"), (8192, "1", "lines-dct2")])
def test_trig_real_fft_route(oracle, monkeypatch, typ, n, fused, backend):
    """dense axis 0 (kern_trig.hpp kinds 8..15): dct2/dst2/dct3/dst3 as Makhoul permutation + a real FFT of length N over each
    r2c / c2r back-end, dct4/dst4 as a complex FFT of N/2, dct1/dst1 as the r2c of the real extension; the general 2N route
    (MI355_EMU_TRIG_REAL=0) must agree with all of them"""
    monkeypatch.setenv("MI355_EMU_XCD_FUSED", fused)
    # backend "lines-dct2": the DCT-II / DST-II computation (dct2 / dst2 forward, dct3 / dst3 inverse) as ONE line-kernel launch —
    # permutation in the LDS staging, real FFT, phase in the split (kern_lines.hpp fft_lines_r2c_kernel<C, TRIG>) — and the
    # DCT-III / DST-III computation the same way on the c2r line kernel; the other back-ends run with that fusion off
    # N = 2048 runs on the alternate ROW shape kept for these launches (line_kernels.def LINE_ROW_TRIG); "-plain-shape" turns that off
    one_launch = backend.startswith("lines-dct2")
    monkeypatch.setenv("MI355_EMU_TRIG_ALT", "0" if backend.endswith("plain-shape") else "1")
    monkeypatch.setenv("MI355_EMU_TRIG_FUSED", "1" if one_launch else "0")
    if one_launch and typ[3] in "14":
        pytest.skip("DCT-II / DST-II computations only")
    if fused == "2":
        monkeypatch.setenv("MI355_EMU_CUS", "4")
        monkeypatch.setenv("MI355_EMU_LINES_R2C", "0")
        monkeypatch.setenv("MI355_EMU_LINES_C2R", "0")
    batch = 2
    x = oracle.random_real_batch(n, batch, 0x7B16 + n).reshape(-1)
    for direction in ("forward", "inverse"):
        desc, _ = _desc({"type": typ, "shape": [n], "batch": batch, "direction": direction, "normalize": "unitary", "layout": {"interleavedComplex": False}})
        got, route, launches = emu.run_plan(desc, x, x.size)
        is_type2 = (typ[3] == "2") == (direction == "forward")         # dct2 forward / dct3 inverse compute a DCT-II
        if one_launch:
            assert route.strip() == "lines-d%st%d[N=%d]" % (typ[1], 2 if is_type2 else 3, n) and launches == 1, route
        else:
            assert route.startswith("trig-real[") and "trig[" not in route, route
            assert ("r2c" in route or "c2r" in route) == (typ[3] != "4"), route
            if direction == "forward" and typ in ("dct2", "dst2"):
                assert backend in route, route
        want = oracle.trig_ref_batch(x, [n], batch, typ, direction, "unitary")
        scale = max(1.0, float(np.max(np.abs(want))))
        assert float(np.max(np.abs(got.astype(np.float64) - want))) <= 2e-5 * scale, (typ, n, direction, route)
        if n <= 256:
            monkeypatch.setenv("MI355_EMU_TRIG_REAL", "0")
            old, route0, _ = emu.run_plan(desc, x, x.size)
            monkeypatch.delenv("MI355_EMU_TRIG_REAL")
            assert "trig[" in route0 and "trig-real" not in route0, route0
            assert float(np.max(np.abs(got.astype(np.float64) - old))) <= 2e-5 * scale


@pytest.mark.parametrize("typ", ["dct2", "dst3", "dct1", "dst4"])
def test_trig_nd(oracle, typ):
    shape, batch = [8, 5, 4], 2
    n = int(np.prod(shape))
    x = oracle.random_real_batch(n, batch, 0x7A99).reshape(-1)
    for direction in ("forward", "inverse"):
        desc, _ = _desc({"type": typ, "shape": shape, "batch": batch, "direction": direction, "normalize": "backward", "layout": {"interleavedComplex": False}})
        got, route, _ = emu.run_plan(desc, x, x.size)
        want = oracle.trig_ref_batch(x, shape, batch, typ, direction, "backward")
        assert float(np.max(np.abs(got.astype(np.float64) - want))) <= 3e-5 * max(1.0, float(np.max(np.abs(want)))), (typ, direction, route)


@pytest.mark.parametrize("typ", TYPES)
@pytest.mark.parametrize("shape", [[40, 36], [6, 70, 4], [33, 6, 8]])
def test_trig_nd_strided_axes(oracle, typ, shape):
    """axes >= 1 (stride > 1) of an N-D real array on the shortened routes: the tiled permutation / phase passes
    (trig_real_*_tiled_kernel, ragged 32 x 32 tiles) around the dense FFTs; odd lengths fall back to the general route per axis"""
    batch = 2
    n = int(np.prod(shape))
    x = oracle.random_real_batch(n, batch, 0x7AA0 + n).reshape(-1)
    for direction in ("forward", "inverse"):
        desc, _ = _desc({"type": typ, "shape": shape, "batch": batch, "direction": direction, "normalize": "unitary", "layout": {"interleavedComplex": False}})
        got, route, _ = emu.run_plan(desc, x, x.size)
        assert route.count("trig-real[") >= 2, route
        want = oracle.trig_ref_batch(x, shape, batch, typ, direction, "unitary")
        assert float(np.max(np.abs(got.astype(np.float64) - want))) <= 3e-5 * max(1.0, float(np.max(np.abs(want)))), (typ, direction, route)


def test_trig_round_trips(oracle):
    """dct2 -> dct3 and dst2 -> dst3 invert each other up to 2/N; dct4 / dst4 are their own inverses up to 2/N"""
    n, batch = 32, 2
    x = oracle.random_real_batch(n, batch, 0x7AAA).reshape(-1)
    for a, b, factor in (("dct2", "dct3", 2.0 / n), ("dst2", "dst3", 2.0 / n), ("dct4", "dct4", 2.0 / n), ("dst4", "dst4", 2.0 / n),
                         ("dct1", "dct1", 1.0 / (2 * (n - 1))), ("dst1", "dst1", 2.0 / (n + 1))):
        d1, _ = _desc({"type": a, "shape": [n], "batch": batch, "layout": {"interleavedComplex": False}})
        y, _, _ = emu.run_plan(d1, x, x.size)
        d2, _ = _desc({"type": b, "shape": [n], "batch": batch, "layout": {"interleavedComplex": False}})
        back, _, _ = emu.run_plan(d2, y, x.size)
        assert float(np.max(np.abs(back * factor - x))) < 2e-6, (a, b)


def test_trig_validation():
    with pytest.raises(ValueError, match="real buffers"):
        resolve_plan_options({"type": "dct2", "shape": [8], "layout": {"interleavedComplex": True}})
    with pytest.raises(ValueError, match="dimensions must be >= 2"):
        resolve_plan_options({"type": "dst1", "shape": [8, 1], "layout": {"interleavedComplex": False}})
    with pytest.raises(ValueError, match="inPlace is not supported"):
        resolve_plan_options({"type": "dct4", "shape": [8], "inPlace": True, "layout": {"interleavedComplex": False}})
    assert resolve_plan_options({"type": "dct3", "shape": [8], "layout": {"interleavedComplex": False}})["direction"] == "forward"

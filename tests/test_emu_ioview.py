"""CPU tier: c2c ioView (pad-in-read / crop / embed-in-write, clearOutside) and zeroPad (read / write ranges)
under host emulation, against a numpy restatement of the reference's embed / extract / zero semantics
(src/kernels/ioview.js generateEmbedComplexWGSL / generateExtractComplexWGSL, src/kernels/zero_pad.js) around the oracle."""
import numpy as np
import pytest

import emu_harness as emu
from mi355fft import _abi
from mi355fft.layout import resolve_plan_options


def _desc(opts):
    r = resolve_plan_options(opts)
    return _abi.make_desc(r["type"], r["shape"], r["batch"], r["direction"], r["normalize"], r["inPlace"], r["input_layout"], r["output_layout"],
                          r["conv"], r["io_view"], r["zero_pad"]), r


def _as_nd(flat, shape, batch):
    return flat.reshape(batch, *reversed(shape), 2).copy()          # axis 0 fastest -> last numpy axis before re/im


def _box(shape, start, end):
    return tuple(slice(start[d], end[d]) for d in reversed(range(len(shape))))


def reference(oracle, x, shape, batch, direction, normalize, io_in=None, io_out=None, zr=None, zw=None, out_init=None):
    rank = len(shape)
    if io_in:
        v = _as_nd(x, io_in["shape"], batch)
        logical = np.zeros((batch, *reversed(shape), 2), np.float32)
        for idx in np.ndindex(*reversed(shape)):
            c = idx[::-1]
            vc = [c[d] - io_in["offset"][d] for d in range(rank)]
            if all(0 <= vc[d] < io_in["shape"][d] for d in range(rank)):
                logical[(slice(None), *idx)] = v[(slice(None), *vc[::-1])]
    else:
        logical = _as_nd(x, shape, batch)
    if zr:
        keep = np.zeros(tuple(reversed(shape)), bool)
        keep[_box(shape, zr["start"], zr["end"])] = True
        logical[:, ~keep] = 0
    y = oracle.c2c_ref_batch(logical.reshape(-1), shape, batch, direction, normalize).reshape(batch, *reversed(shape), 2)
    if zw:
        keep = np.zeros(tuple(reversed(shape)), bool)
        keep[_box(shape, zw["start"], zw["end"])] = True
        y[:, ~keep] = 0
    if not io_out:
        return y.reshape(-1)
    out = _as_nd(np.asarray(out_init, np.float32), io_out["shape"], batch)
    if io_out.get("clearOutside"):
        out[...] = 0
    for idx in np.ndindex(*reversed(io_out["shape"])):
        vc = idx[::-1]
        lc = [vc[d] + io_out["offset"][d] for d in range(rank)]
        if all(0 <= lc[d] < shape[d] for d in range(rank)):
            out[(slice(None), *idx)] = y[(slice(None), *lc[::-1])]
    return out.reshape(-1)


CASES = [
    # (shape, ioView, zeroPad)
    ([16], {"input": {"shape": [10]}}, None),                                                   # pad-in-read at the start
    ([16], {"input": {"shape": [10], "placement": "center"}}, None),                            # centred
    ([16], {"input": {"shape": [24], "offset": [-3]}}, None),                                   # view larger than the domain: crop
    ([16], {"output": {"shape": [8], "offset": [4]}}, None),                                    # sub-region write, rest untouched
    ([16], {"output": {"shape": [24], "placement": "center", "clearOutside": True}}, None),     # embed into a larger output, cleared
    ([16], {"output": {"shape": [24], "placement": "center"}}, None),                           # ... or left untouched
    ([8, 4], {"input": {"shape": [5, 3], "offset": [2, 1]}, "output": {"shape": [6, 6], "offset": [1, -1]}}, None),
    ([16], None, {"read": {"start": [4], "end": [12]}, "write": {"start": [2], "end": [14]}}),  # the docs/API.md example
    ([8, 4], {"input": {"shape": [6, 4]}}, {"read": {"start": [1, 0], "end": [7, 3]}}),
    ([12, 5], None, {"write": {"start": [0, 1], "end": [12, 4]}}),
]


@pytest.mark.parametrize("shape,io_view,zero_pad", CASES)
def test_c2c_ioview_zeropad(oracle, shape, io_view, zero_pad):
    batch = 2
    opts = {"type": "c2c", "shape": shape, "batch": batch, "direction": "forward", "normalize": "unitary"}
    if io_view:
        opts["ioView"] = io_view
    if zero_pad:
        opts["zeroPad"] = zero_pad
    desc, r = _desc(opts)
    in_shape = r["io_view"]["input"]["shape"] if r["io_view"]["input"] else shape
    out_shape = r["io_view"]["output"]["shape"] if r["io_view"]["output"] else shape
    x = oracle.random_complex_interleaved(int(np.prod(in_shape)) * batch, 31337 + sum(shape))
    out_floats = 2 * int(np.prod(out_shape)) * batch
    sentinel = np.tile(np.array([77.0, -55.0], np.float32), out_floats // 2)
    got, route, _ = emu.run_plan(desc, x, out_floats, out_init=sentinel)
    want = reference(oracle, x, shape, batch, "forward", "unitary", r["io_view"]["input"], r["io_view"]["output"],
                     r["zero_pad"]["read"], r["zero_pad"]["write"], sentinel)
    assert got.shape == want.shape
    scale = max(1.0, float(np.max(np.abs(want))))
    assert float(np.max(np.abs(got.astype(np.float64) - want))) <= 2e-5 * scale, route


FUSED_CASES = [
    # power-of-two axes: every side feature rides the first load / last store of the line kernels (no staging launch at all)
    ([64], {"input": {"shape": [40], "placement": "center"}, "output": {"shape": [48], "offset": [5]}}, {"read": {"start": [4], "end": [60]}, "write": {"start": [2], "end": [62]}}, None),
    ([16], {"output": {"shape": [24], "placement": "center", "clearOutside": True}}, None, None),
    ([64, 16], {"input": {"shape": [50, 12], "offset": [3, 2]}, "output": {"shape": [70, 10], "offset": [-2, 4]}}, {"write": {"start": [1, 0], "end": [63, 15]}}, None),
    ([32, 8, 4], {"input": {"shape": [20, 8, 3]}}, {"read": {"start": [0, 1, 0], "end": [32, 7, 4]}}, None),
    ([128, 4], None, {"read": {"start": [8, 0], "end": [120, 3]}, "write": {"start": [0, 1], "end": [128, 4]}}, None),
    # strided physical layouts on both sides (padded rows, permuted batch pitch) together with views
    ([32, 16], {"input": {"shape": [24, 16]}}, None, {"input": {"strides": [1, 40], "offsetElements": 7, "batchStrideElements": 700},
                                                      "output": {"strides": [2, 70], "offsetElements": 3, "batchStrideElements": 1200}}),
]


@pytest.mark.parametrize("fuse", [1, 0])
@pytest.mark.parametrize("shape,io_view,zero_pad,layouts", FUSED_CASES)
def test_c2c_sides_fused_into_the_line_kernels(oracle, monkeypatch, shape, io_view, zero_pad, layouts, fuse):
    """with power-of-two axes the strided layout / ioView / zeroPad of each side is the address map of the
    first / last line-kernel launch (route: only *-mapped[...] launches); MI355_EMU_FUSE_VIEWS=0 is the staging route
    (gather / embed / zero / extract / scatter passes) — both against the numpy restatement of the reference semantics"""
    monkeypatch.setenv("MI355_EMU_FUSE_VIEWS", str(fuse))
    batch = 3
    rank = len(shape)
    for direction in ("forward", "inverse"):
        opts = {"type": "c2c", "shape": shape, "batch": batch, "direction": direction, "normalize": "backward"}
        if io_view:
            opts["ioView"] = io_view
        if zero_pad:
            opts["zeroPad"] = zero_pad
        if layouts:
            opts["layout"] = {"interleavedComplex": True}
            for side, lay in layouts.items():
                opts["layout"][side + "Strides"] = lay["strides"]
                opts["layout"][side + "OffsetElements"] = lay["offsetElements"]
                opts["layout"][side + "BatchStrideElements"] = lay["batchStrideElements"]
        desc, r = _desc(opts)
        in_shape = r["io_view"]["input"]["shape"] if r["io_view"]["input"] else shape
        out_shape = r["io_view"]["output"]["shape"] if r["io_view"]["output"] else shape
        li, lo = (layouts or {}).get("input"), (layouts or {}).get("output")

        def extent(lay, shp):
            if not lay:
                return int(np.prod(shp)) * batch
            return lay["offsetElements"] + (batch - 1) * lay["batchStrideElements"] + sum((shp[d] - 1) * lay["strides"][d] for d in range(rank)) + 1

        def to_dense(phys, lay, shp):          # physical buffer -> dense [batch][shape] view contents
            if not lay:
                return phys.copy()
            dense = np.zeros((batch, *reversed(shp), 2), np.float32)
            p2 = phys.reshape(-1, 2)
            for bi in range(batch):
                for idx in np.ndindex(*reversed(shp)):
                    c = idx[::-1]
                    dense[(bi, *idx)] = p2[lay["offsetElements"] + bi * lay["batchStrideElements"] + sum(c[d] * lay["strides"][d] for d in range(rank))]
            return dense.reshape(-1)

        xin = oracle.random_complex_interleaved(extent(li, in_shape), 4242 + sum(shape))
        out_floats = 2 * extent(lo, out_shape)
        sentinel = np.tile(np.array([77.0, -55.0], np.float32), out_floats // 2)
        got, route, launches = emu.run_plan(desc, xin, out_floats, out_init=sentinel)
        want_dense = reference(oracle, to_dense(xin, li, in_shape), shape, batch, direction, "backward", r["io_view"]["input"], r["io_view"]["output"],
                               r["zero_pad"]["read"], r["zero_pad"]["write"], to_dense(sentinel, lo, out_shape))
        got_dense = to_dense(got, lo, out_shape)
        scale = max(1.0, float(np.max(np.abs(want_dense))))
        assert float(np.max(np.abs(got_dense.astype(np.float64) - want_dense))) <= 2e-5 * scale, route
        if lo:   # elements of the physical output that no logical / view element maps to stay untouched
            touched = np.zeros(out_floats // 2, bool)
            for bi in range(batch):
                for idx in np.ndindex(*reversed(out_shape)):
                    c = idx[::-1]
                    touched[lo["offsetElements"] + bi * lo["batchStrideElements"] + sum(c[d] * lo["strides"][d] for d in range(rank))] = True
            assert np.array_equal(got.reshape(-1, 2)[~touched], sentinel.reshape(-1, 2)[~touched]), route
        staging = [w for w in ("gather", "embed", "zero-read", "zero-write", "extract", "scatter") if w in route]
        if fuse:
            assert not staging and "mapped[" in route, route
            assert launches == len([a for a in shape if a > 1]) + (1 if (io_view or {}).get("output", {}).get("clearOutside") else 0), route
        else:
            assert staging and "mapped[" not in route, route


@pytest.mark.parametrize("lg,label,clear", [(20, "1024x1024", False), (20, "1024x1024", True), (17, "256x512", False), (18, "512x512", True), (19, "512x1024", False), (21, "1024x2048", False)])
def test_c2c_view_of_a_four_step_line(oracle, monkeypatch, lg, label, clear):
    """r03: a rank-1 view of a 2^20-point line — pad-in-read (the input view is shorter and shifted), zeroPad.read / .write ranges, crop +
    embed-in-write (the output view is a shifted window, optionally cleared outside) — as predicates of the fused kernel's loads and stores
    (kern_regtile.hpp fft_xcd_rt1k_kernel<.., VIEW> at 2^20, kern_xcd.hpp fft_xcd_fused_kernel<.., VIEW> at 2^17 .. 2^19): control-block reset + ONE launch (+ the clearOutside memset), no embed / zero / extract"""
    monkeypatch.setenv("MI355_EMU_XCD_FUSED", "1")
    monkeypatch.setenv("MI355_EMU_CUS", "3")
    monkeypatch.setenv("MI355_EMU_XCDS", "1")
    n, batch = 1 << lg, 2
    vin = {"shape": [n - 3000], "offset": [1000]}            # logical i <- view element i - 1000
    vout = {"shape": [n // 2 + 77], "offset": [-50], "clearOutside": clear}   # view element j <- logical j - 50
    zr, zw = {"start": [5000], "end": [n - 100]}, {"start": [64], "end": [n // 2 - 5]}
    rng = np.random.default_rng(11)
    x = rng.standard_normal(2 * vin["shape"][0] * batch).astype(np.float32)
    out_init = rng.standard_normal(2 * vout["shape"][0] * batch).astype(np.float32)
    for direction, norm in (("forward", "none"), ("inverse", "backward")):
        desc, _ = _desc({"type": "c2c", "shape": [n], "batch": batch, "direction": direction, "normalize": norm,
                         "ioView": {"input": vin, "output": vout}, "zeroPad": {"read": zr, "write": zw}})
        got, route, launches = emu.run_plan(desc, x, out_init.size, out_init=out_init)
        assert f"xcd-fused-view[N={label}]" in route and launches == (3 if clear else 2), (route, launches)
        assert not any(w in route for w in ("embed", "extract", "zero-", "gather", "scatter")), route
        # numpy restatement (rank 1, vectorised)
        logical = np.zeros((batch, n, 2), np.float32)
        v = x.reshape(batch, -1, 2)
        lo, hi = max(0, vin["offset"][0]), min(n, vin["offset"][0] + vin["shape"][0])
        logical[:, lo:hi] = v[:, lo - vin["offset"][0]:hi - vin["offset"][0]]
        logical[:, :zr["start"][0]] = 0
        logical[:, zr["end"][0]:] = 0
        y = oracle.c2c_ref_batch(logical.reshape(-1), [n], batch, direction, norm).reshape(batch, n, 2)
        y[:, :zw["start"][0]] = 0
        y[:, zw["end"][0]:] = 0
        want = out_init.reshape(batch, -1, 2).copy()
        if clear:
            want[...] = 0
        j0 = max(0, -vout["offset"][0])
        j1 = min(vout["shape"][0], n - vout["offset"][0])
        want[:, j0:j1] = y[:, j0 + vout["offset"][0]:j1 + vout["offset"][0]]
        from oracle import oracle as orc
        assert orc.rel_l2(got, want.reshape(-1)) < 1e-5 and orc.rel_max(got, want.reshape(-1)) < 1e-5, route
        untouched = np.ones(vout["shape"][0], bool)
        untouched[j0:j1] = False
        if not clear:
            assert np.array_equal(got.reshape(batch, -1, 2)[:, untouched], out_init.reshape(batch, -1, 2)[:, untouched])


def test_noop_views_resolve_to_nothing():
    _, r = _desc({"type": "c2c", "shape": [8, 4], "direction": "forward", "ioView": {"input": {"shape": [8, 4]}, "output": {"shape": [8, 4], "offset": [0, 0]}},
                  "zeroPad": {"read": {"start": [0, 0], "end": [8, 4]}}})
    assert r["io_view"] == {"input": None, "output": None} and r["zero_pad"] == {"read": None, "write": None}


def test_view_validation_messages():
    bad = [({"ioView": {"input": {"shape": [8]}}}, "ioView.input.shape must be an array of 2 positive ints"),
           ({"ioView": {"output": {"shape": [8, 4], "placement": "middle"}}}, 'placement must be "start"|"center"'),
           ({"zeroPad": {"read": {"start": [0, 0], "end": [9, 4]}}}, "zeroPad.read.end[0] must be <= shape[0] (8); got 9"),
           ({"zeroPad": {"write": {"start": [3, 0], "end": [2, 4]}}}, "zeroPad.write: start[0] must be <= end[0]")]
    for extra, frag in bad:
        with pytest.raises(ValueError) as e:
            resolve_plan_options(dict({"type": "c2c", "shape": [8, 4], "direction": "forward"}, **extra))
        assert frag in str(e.value)
    with pytest.raises(ValueError) as e:    # fftconv takes zeroPad (on its FFT domain) but no ioView
        resolve_plan_options({"type": "fftconv", "shape": [8], "fftConv": {"kernelCount": 1}, "ioView": {"input": {"shape": [4]}}})
    assert "ioView is not an fftconv option" in str(e.value)
    with pytest.raises(ValueError) as e:    # r2c writes the packed domain: shape[0] // 2 + 1 = 5 bins
        resolve_plan_options({"type": "r2c", "shape": [8], "direction": "forward", "zeroPad": {"write": {"start": [0], "end": [6]}}})
    assert "zeroPad.write.end[0] must be <= shape[0] (5); got 6" in str(e.value)


# ---- r2c / c2r: the views of the real side live on the real domain, those of the spectrum side on the packed one ----
def _embed(x, lshape, view, batch, width):
    """physical view array -> zero-filled logical domain (width = 1 real / 2 complex floats per element)"""
    rank = len(lshape)
    logical = np.zeros((batch, *reversed(lshape), width), np.float32)
    if not view:
        return x.reshape(logical.shape).copy()
    v = x.reshape(batch, *reversed(view["shape"]), width)
    for idx in np.ndindex(*reversed(lshape)):
        c = idx[::-1]
        vc = [c[d] - view["offset"][d] for d in range(rank)]
        if all(0 <= vc[d] < view["shape"][d] for d in range(rank)):
            logical[(slice(None), *idx)] = v[(slice(None), *vc[::-1])]
    return logical


def _extract(y, lshape, view, batch, width, out_init):
    rank = len(lshape)
    if not view:
        return y.reshape(-1)
    out = np.asarray(out_init, np.float32).reshape(batch, *reversed(view["shape"]), width).copy()
    if view.get("clearOutside"):
        out[...] = 0
    for idx in np.ndindex(*reversed(view["shape"])):
        vc = idx[::-1]
        lc = [vc[d] + view["offset"][d] for d in range(rank)]
        if all(0 <= lc[d] < lshape[d] for d in range(rank)):
            out[(slice(None), *idx)] = y[(slice(None), *lc[::-1])]
    return out.reshape(-1)


def _zero_outside(a, lshape, z):
    if z:
        keep = np.zeros(tuple(reversed(lshape)), bool)
        keep[_box(lshape, z["start"], z["end"])] = True
        a[:, ~keep] = 0


REAL_CASES = [
    ([16], {"input": {"shape": [10]}}, None),
    ([16], {"input": {"shape": [20], "offset": [-2]}, "output": {"shape": [5], "offset": [2]}}, None),
    ([16], {"output": {"shape": [12], "placement": "center", "clearOutside": True}}, {"read": {"start": [3], "end": [13]}}),
    ([16, 4], {"input": {"shape": [12, 3], "offset": [1, 1]}}, {"write": {"start": [1, 0], "end": [8, 3]}}),
    ([64], None, {"read": {"start": [8], "end": [56]}, "write": {"start": [0], "end": [20]}}),
    # line-kernel sizes: both sides ride the r2c line kernel (rank 1) / the r2c kernel and the last c2c axis (rank > 1)
    ([128], {"input": {"shape": [100], "offset": [-5]}, "output": {"shape": [70], "offset": [-2]}}, {"read": {"start": [3], "end": [120]}, "write": {"start": [1], "end": [60]}}),
    ([256, 4], {"input": {"shape": [200, 3], "offset": [10, 1]}, "output": {"shape": [140, 6], "offset": [-4, -1], "clearOutside": True}}, {"write": {"start": [0, 1], "end": [129, 4]}}),
    ([128, 8, 2], None, {"read": {"start": [0, 1, 0], "end": [128, 7, 2]}}),
    ([512], None, {"write": {"start": [0], "end": [100]}}),
]
STAGING_WORDS = ("gather", "embed", "zero-read", "zero-write", "extract", "scatter")


def _check_fused_route(route, launches, shape, fuse, clear):
    """shape[0] >= 128 (a fused r2c / c2r line kernel exists): no staging launch with fusing on, one launch per axis"""
    if shape[0] < 128:
        return
    staging = [w for w in STAGING_WORDS if w in route]
    if fuse:
        assert not staging and "mapped[" in route, route
        assert launches == len([a for a in shape if a > 1]) + (1 if clear else 0), route
    else:
        assert staging and "mapped[" not in route, route


@pytest.mark.parametrize("fuse", [1, 0])
@pytest.mark.parametrize("shape,io_view,zero_pad", REAL_CASES)
def test_r2c_ioview_zeropad(oracle, monkeypatch, shape, io_view, zero_pad, fuse):
    monkeypatch.setenv("MI355_EMU_FUSE_VIEWS", str(fuse))
    batch = 2
    opts = {"type": "r2c", "shape": shape, "batch": batch, "direction": "forward", "normalize": "none"}
    if io_view:
        opts["ioView"] = io_view
    if zero_pad:
        opts["zeroPad"] = zero_pad
    desc, r = _desc(opts)
    packed = [shape[0] // 2 + 1] + shape[1:]
    vin, vout = r["io_view"]["input"], r["io_view"]["output"]
    in_shape = vin["shape"] if vin else shape
    out_shape = vout["shape"] if vout else packed
    x = oracle.random_real_batch(int(np.prod(in_shape)), batch, 4242 + sum(shape)).reshape(-1)
    out_floats = 2 * int(np.prod(out_shape)) * batch
    sentinel = np.tile(np.array([77.0, -55.0], np.float32), out_floats // 2)
    got, route, launches = emu.run_plan(desc, x, out_floats, out_init=sentinel)
    _check_fused_route(route, launches, shape, fuse, bool((vout or {}).get("clearOutside")))
    logical = _embed(x, shape, vin, batch, 1)
    _zero_outside(logical, shape, r["zero_pad"]["read"])
    cplx = np.zeros((batch, *reversed(shape), 2), np.float32)
    cplx[..., 0] = logical[..., 0]
    full = oracle.c2c_ref_batch(cplx.reshape(-1), shape, batch, "forward", "none").reshape(batch, *reversed(shape), 2)
    y = full[..., :packed[0], :].copy()
    _zero_outside(y, packed, r["zero_pad"]["write"])
    want = _extract(y, packed, vout, batch, 2, sentinel)
    scale = max(1.0, float(np.max(np.abs(want))))
    assert got.shape == want.shape and float(np.max(np.abs(got.astype(np.float64) - want))) <= 2e-5 * scale, route


@pytest.mark.parametrize("shape,io_view,zero_pad", [
    ([16], {"output": {"shape": [10], "offset": [3]}}, None),
    ([16], {"input": {"shape": [6]}}, {"write": {"start": [2], "end": [14]}}),                  # low-pass: only the first 6 bins given
    ([16, 4], {"output": {"shape": [20, 4], "placement": "center", "clearOutside": True}}, {"read": {"start": [0, 0], "end": [5, 4]}}),
    # line-kernel sizes: the packed side rides the first inverse c2c axis / the c2r kernel's loads, the real side its store pass
    ([128], {"input": {"shape": [40]}, "output": {"shape": [100], "offset": [10]}}, {"write": {"start": [5], "end": [120]}}),
    ([256, 4], {"output": {"shape": [300, 4], "placement": "center", "clearOutside": True}}, {"read": {"start": [0, 0], "end": [100, 4]}}),
    ([128, 4, 2], {"input": {"shape": [60, 4, 2], "offset": [0, 0, 0]}}, None),
])
@pytest.mark.parametrize("fuse", [1, 0])
def test_c2r_ioview_zeropad(oracle, monkeypatch, shape, io_view, zero_pad, fuse):
    monkeypatch.setenv("MI355_EMU_FUSE_VIEWS", str(fuse))
    batch = 2
    opts = {"type": "c2r", "shape": shape, "batch": batch, "direction": "inverse", "normalize": "backward"}
    if io_view:
        opts["ioView"] = io_view
    if zero_pad:
        opts["zeroPad"] = zero_pad
    desc, r = _desc(opts)
    packed = [shape[0] // 2 + 1] + shape[1:]
    vin, vout = r["io_view"]["input"], r["io_view"]["output"]
    in_shape = vin["shape"] if vin else packed
    out_shape = vout["shape"] if vout else shape
    # a Hermitian-consistent spectrum: r2c of a random real signal, cropped to the physical input view
    n = int(np.prod(shape))
    sig = oracle.random_real_batch(n, batch, 777 + sum(shape)).reshape(batch, *reversed(shape))
    cplx = np.zeros((batch, *reversed(shape), 2), np.float32)
    cplx[..., 0] = sig
    spec = oracle.c2c_ref_batch(cplx.reshape(-1), shape, batch, "forward", "none").reshape(batch, *reversed(shape), 2)[..., :packed[0], :].copy()
    if vin:      # physical input = the view's window of the packed spectrum (zeros where the window leaves it)
        phys = np.zeros((batch, *reversed(in_shape), 2), np.float32)
        for idx in np.ndindex(*reversed(in_shape)):
            lc = [idx[::-1][d] + vin["offset"][d] for d in range(len(shape))]
            if all(0 <= lc[d] < packed[d] for d in range(len(shape))):
                phys[(slice(None), *idx)] = spec[(slice(None), *lc[::-1])]
        x = phys.reshape(-1)
    else:
        x = spec.reshape(-1)
    out_floats = int(np.prod(out_shape)) * batch
    sentinel = np.full(out_floats, 77.0, np.float32)
    got, route, launches = emu.run_plan(desc, x, out_floats, out_init=sentinel)
    _check_fused_route(route, launches, shape, fuse, bool((vout or {}).get("clearOutside")))
    logical = _embed(x, packed, vin, batch, 2)
    _zero_outside(logical, packed, r["zero_pad"]["read"])
    p = packed[0]
    lines = logical.reshape(-1, p, 2)
    if len(shape) == 1:
        y = np.stack([oracle.c2r_ref_from_packed(l.reshape(-1), shape[0], "backward") for l in lines]).reshape(batch, shape[0], 1)
    else:
        # N-D: inverse c2c over the other axes of the Hermitian-extended array, via the full complex oracle
        full = np.zeros((batch, *reversed(shape), 2), np.float32)
        full[..., :p, :] = logical
        for k in range(1, shape[0] - p + 1):
            src = logical[..., p - 1 - k if shape[0] % 2 == 0 else p - k, :]
            # mirror along every other axis as well: X[N0-k0, (N1-k1)%N1, ...] = conj X[k0, k1, ...]
            mirrored = src
            for ax in range(1, len(shape)):
                a = len(shape) - ax          # numpy axis of logical axis `ax` (batch is numpy axis 0)
                mirrored = np.roll(np.flip(mirrored, axis=a), 1, axis=a)
            full[..., p - 1 + k, 0] = mirrored[..., 0]
            full[..., p - 1 + k, 1] = -mirrored[..., 1]
        y = oracle.c2c_ref_batch(full.reshape(-1), shape, batch, "inverse", "backward").reshape(batch, *reversed(shape), 2)[..., 0:1].copy()
    y = y.reshape(batch, *reversed(shape), 1)
    _zero_outside(y, shape, r["zero_pad"]["write"])
    want = _extract(y, shape, vout, batch, 1, sentinel)
    scale = max(1.0, float(np.max(np.abs(want))))
    assert got.shape == want.shape and float(np.max(np.abs(got.astype(np.float64) - want))) <= 3e-5 * scale, route


@pytest.mark.parametrize("fuse", [1, 0])
def test_r2c_c2r_strided_sides(oracle, monkeypatch, fuse):
    """strided physical layouts on the real transforms: non-unit element strides on the real side (4-byte accesses in the mapped
    kernels, no pair fast path) and on the packed side, odd offsets, padded batch pitches; elements no logical index maps to stay
    untouched.  fuse=1: one mapped launch each way; fuse=0: gather / scatter passes"""
    monkeypatch.setenv("MI355_EMU_FUSE_VIEWS", str(fuse))
    n, batch = 256, 3
    p = n // 2 + 1
    li = {"strides": [3], "offset": 5, "batch_stride": 3 * n + 11}
    lo = {"strides": [2], "offset": 1, "batch_stride": 2 * p + 7}
    x = oracle.random_real_batch(n, batch, 0xABCD).reshape(batch, n)
    phys_in = np.full(li["offset"] + (batch - 1) * li["batch_stride"] + (n - 1) * 3 + 1, 9.0, np.float32)
    for b in range(batch):
        phys_in[li["offset"] + b * li["batch_stride"] + 3 * np.arange(n)] = x[b]
    out_elems = lo["offset"] + (batch - 1) * lo["batch_stride"] + (p - 1) * 2 + 1
    sentinel = np.tile(np.array([77.0, -55.0], np.float32), out_elems)
    desc = _abi.make_desc("r2c", [n], batch, "forward", "none", input_layout=li, output_layout=lo)
    got, route, launches = emu.run_plan(desc, phys_in, sentinel.size, out_init=sentinel)
    assert (route.split() == ["lines-r2c-mapped[N=256]"] and launches == 1) if fuse else ("gather" in route and "scatter" in route), route
    want = sentinel.copy().reshape(-1, 2)
    for b in range(batch):
        want[lo["offset"] + b * lo["batch_stride"] + 2 * np.arange(p)] = oracle.r2c_ref_packed(x[b], n, "none").reshape(-1, 2)
    assert float(np.max(np.abs(got.astype(np.float64) - want.reshape(-1)))) <= 2e-5 * float(np.max(np.abs(want))), route
    # c2r back: packed side strided by 2, real side strided by 3
    ri = lo
    ro = {"strides": [3], "offset": 2, "batch_stride": 3 * n + 5}
    real_elems = ro["offset"] + (batch - 1) * ro["batch_stride"] + (n - 1) * 3 + 1
    rs = np.full(real_elems, -3.0, np.float32)
    desc = _abi.make_desc("c2r", [n], batch, "inverse", "backward", input_layout=ri, output_layout=ro)
    back, route, launches = emu.run_plan(desc, got, rs.size, out_init=rs)
    assert (route.split() == ["lines-c2r-mapped[N=256]"] and launches == 1) if fuse else ("gather" in route and "scatter" in route), route
    wantr = rs.copy()
    for b in range(batch):
        wantr[ro["offset"] + b * ro["batch_stride"] + 3 * np.arange(n)] = x[b]
    assert float(np.max(np.abs(back - wantr))) < 3e-6, route

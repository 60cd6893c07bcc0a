"""CPU tier: seeded differential fuzz of the planner + kernels (host emulation) against the oracle over random ranks,
shapes (ones, primes, powers of two, mixed radix), batches, directions and normalisations — the edge cases the
reference's suite covers with hand-picked sizes (ragged, length-1 axes, odd r2c lengths)."""
import numpy as np
import pytest

import emu_harness as emu
from mi355fft import _abi

DIMS = [1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 12, 13, 15, 16, 17, 19, 20, 32, 34, 64]


def _rand_shape(rng, max_points):
    rank = int(rng.integers(1, 4))
    while True:
        shape = [int(rng.choice(DIMS)) for _ in range(rank)]
        if int(np.prod(shape)) <= max_points:
            return shape


@pytest.mark.parametrize("seed", range(24))
def test_fuzz_c2c(oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    shape = _rand_shape(rng, 2048)
    batch = int(rng.integers(1, 4))
    direction = str(rng.choice(["forward", "inverse"]))
    normalize = str(rng.choice(["none", "backward", "unitary"]))
    in_place = bool(rng.integers(0, 2))
    n = int(np.prod(shape))
    x = oracle.random_complex_batch(n, batch, 5000 + seed).reshape(-1)
    desc = _abi.make_desc("c2c", shape, batch, direction, normalize, in_place=in_place)
    got, route, _ = emu.run_plan(desc, x, x.size)
    want = oracle.c2c_ref_batch(x, shape, batch, direction, normalize, anysize=True)
    err = float(np.max(np.abs(got.astype(np.float64) - want)))
    assert err <= 3e-5 * max(1.0, float(np.max(np.abs(want)))), f"shape={shape} batch={batch} {direction} {normalize} inplace={in_place} route={route} err={err}"


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_r2c_c2r(oracle, seed):
    rng = np.random.default_rng(2000 + seed)
    while True:
        shape = _rand_shape(rng, 1024)
        if shape[0] >= 2:
            break
    batch = int(rng.integers(1, 4))
    normalize = str(rng.choice(["none", "unitary"]))
    n = int(np.prod(shape))
    p = shape[0] // 2 + 1
    packed_n = p * (n // shape[0])
    x = oracle.random_real_batch(n, batch, 6000 + seed).reshape(-1)
    # reference definition: full complex transform of the real data, first N0/2+1 bins along axis 0
    want = []
    for b in range(batch):
        c = np.zeros(2 * n, np.float32)
        c[0::2] = x[b * n:(b + 1) * n]
        full = oracle.fftnd_ref(c, shape, "forward", normalize, anysize=True).reshape(n // shape[0], shape[0], 2)
        want.append(full[:, :p, :].reshape(-1))
    want = np.concatenate(want)
    desc = _abi.make_desc("r2c", shape, batch, "forward", normalize)
    got, route, _ = emu.run_plan(desc, x, 2 * packed_n * batch)
    err = float(np.max(np.abs(got.astype(np.float64) - want)))
    assert err <= 3e-5 * max(1.0, float(np.max(np.abs(want)))), f"r2c shape={shape} batch={batch} {normalize} route={route} err={err}"
    # c2r of the exact spectrum returns the signal (backward normalisation)
    spec = []
    for b in range(batch):
        c = np.zeros(2 * n, np.float32)
        c[0::2] = x[b * n:(b + 1) * n]
        full = oracle.fftnd_ref(c, shape, "forward", "none", anysize=True).reshape(n // shape[0], shape[0], 2)
        spec.append(full[:, :p, :].reshape(-1))
    spec = np.concatenate(spec)
    desc = _abi.make_desc("c2r", shape, batch, "inverse", "backward")
    back, route, _ = emu.run_plan(desc, spec, n * batch)
    err = float(np.max(np.abs(back.astype(np.float64) - x)))
    assert err <= 3e-5, f"c2r shape={shape} batch={batch} route={route} err={err}"


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_fftconv(oracle, seed):
    rng = np.random.default_rng(3000 + seed)
    rank = int(rng.integers(1, 3))
    shape = [int(rng.choice([4, 5, 6, 8, 9, 12, 16])) for _ in range(rank)]
    boundary = str(rng.choice(["circular", "linear-full", "linear-same", "linear-valid"]))
    kshape = [int(rng.integers(1, s + 1)) for s in shape]
    mode = str(rng.choice(["convolution", "correlation"]))
    K = int(rng.integers(1, 4))
    batch = int(rng.integers(1, 4))
    layout = str(rng.choice(["kernel-major", "batch-major"]))
    n, kn = int(np.prod(shape)), int(np.prod(kshape))
    x = oracle.random_complex_interleaved(n * batch, 7000 + seed)
    kern = oracle.random_complex_interleaved(kn * K, 7100 + seed)
    from mi355fft.layout import resolve_plan_options
    r = resolve_plan_options({"type": "fftconv", "shape": shape, "batch": batch,
                              "fftConv": {"mode": mode, "boundary": boundary, "kernelShape": kshape, "kernelCount": K, "outputLayout": layout}})
    desc = _abi.make_desc(r["type"], r["shape"], r["batch"], r["direction"], r["normalize"], r["inPlace"], r["input_layout"], r["output_layout"], r["conv"])
    on = int(np.prod(r["outputShape"]))
    got, route, _ = emu.run_plan(desc, x, 2 * on * batch * K, kernel=kern)
    want = np.empty((K, batch, 2 * on), np.float32)
    for k in range(K):
        ref, osh = oracle.fftconv_ref(x, kern[2 * k * kn:2 * (k + 1) * kn], shape, batch, mode, boundary, kshape)
        assert osh == r["outputShape"]
        want[k] = ref.reshape(batch, 2 * on)
    if layout == "batch-major":
        want = want.transpose(1, 0, 2)
    err = float(np.max(np.abs(got.astype(np.float64) - want.reshape(-1))))
    assert err <= 5e-5 * max(1.0, float(np.max(np.abs(want)))), f"fftconv shape={shape} k={kshape} {boundary} {mode} K={K} b={batch} {layout} route={route} err={err}"

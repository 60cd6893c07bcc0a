"""Pins oracle/oracle.c (the CPU restatement) against fixtures produced by the reference's own
src/utils/math.js (oracle/gen_fixtures.mjs imported it under Node in the build container).

Radix-2 path: bit-exact (FNV-1a-64 of the output bytes).  O(N^2) DFT path: 2e-6 relative."""
import os
import struct

import numpy as np
import pytest

from conftest import GOLDEN


def _load(name):
    return np.fromfile(os.path.join(GOLDEN, name), dtype=np.float32)


def _hex_f64(h):
    return struct.unpack("<d", bytes.fromhex(h))[0]


def test_stage_twiddles_match_v8_libm(manifest, oracle):
    cases, _ = manifest
    for e in cases["stage_twiddles"]["entries"]:
        re, im = oracle.stage_twiddle(e["len"], e["inverse"])
        assert re == _hex_f64(e["cos"]) and im == _hex_f64(e["sin"]), e


def test_rng_twins(manifest, oracle):
    cases, _ = manifest
    c = cases["rng"]
    want = [_hex_f64(h) for h in c["first_f64"]]
    got_np = oracle.rng_f64(c["seed"], 0, 8)
    got_c = [oracle.lib().oracle_rng_at(c["seed"], i) for i in range(8)]
    assert list(got_np) == want and got_c == want
    rc = oracle.random_complex_interleaved(16, c["seed_complex"])
    assert rc.tolist() == c["complex16"]
    import ctypes
    buf = np.empty(32, dtype=np.float32)
    oracle.lib().oracle_random_complex_interleaved(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), 16, c["seed_complex"])
    assert buf.tolist() == c["complex16"]


def test_normalize_scale_factor(manifest, oracle):
    cases, _ = manifest
    for r in cases["normalize_scale"]["rows"]:
        got = oracle.lib().oracle_normalize_scale_factor(oracle.NORMALIZE[r["normalize"]], 1 if r["direction"] == "inverse" else 0,
                                                         float(r["nTotal"]))
        assert got == _hex_f64(r["value"]), r


def test_c2c_pow2_bit_exact(manifest, oracle):
    cases, _ = manifest
    n_checked = 0
    for c in cases.values():
        if c["kind"] != "c2c_pow2":
            continue
        n = int(np.prod(c["shape"]))
        x = oracle.random_complex_interleaved(n * c["batch"], c["seed"])
        out = oracle.c2c_ref_batch(x, c["shape"], c["batch"], c["direction"], c["normalize"], anysize=False)
        assert format(oracle.fnv1a64(out), "016x") == c["out_fnv1a64"], c["name"]
        assert out[:8].tolist()[: len(c["out_head"])] == c["out_head"], c["name"]
        if "out_file" in c:
            assert np.array_equal(_load(c["in_file"]), x)
            assert np.array_equal(_load(c["out_file"]), out)
        n_checked += 1
    assert n_checked >= 140


@pytest.mark.parametrize("name", ["c2c_N2p16_forward", "c2c_N2p20_forward", "c2c_N2p20_inverse", "c2c_N2p21_forward"])
def test_c2c_large_bit_exact(manifest, oracle, name):
    cases, _ = manifest
    c = cases[name]
    n = c["shape"][0]
    x = oracle.random_complex_interleaved(n, c["seed"])
    out = oracle.fft1d_ref(x, n, c["direction"])
    assert format(oracle.fnv1a64(out), "016x") == c["out_fnv1a64"]
    idx = np.asarray(c["sample_idx"])
    vals = np.asarray(c["sample_vals"], dtype=np.float32).reshape(-1, 2)
    assert np.array_equal(out.reshape(-1, 2)[idx], vals)


def test_r2c_pow2_bit_exact(manifest, oracle):
    cases, _ = manifest
    for c in cases.values():
        if c["kind"] != "r2c_pow2":
            continue
        x = oracle.random_real(c["N"], c["seed"])
        out = oracle.r2c_ref_packed(x, c["N"], "none", use_pow2=True)
        assert format(oracle.fnv1a64(out), "016x") == c["out_fnv1a64"], c["name"]


def _close(a, e, tol=2e-6):
    scale = max(1.0, float(np.max(np.abs(e))))
    assert a.shape == e.shape
    assert float(np.max(np.abs(a.astype(np.float64) - e.astype(np.float64)))) <= tol * scale


def test_dft_path(manifest, oracle):
    cases, _ = manifest
    for c in cases.values():
        if c["kind"] == "dft":
            x = oracle.random_complex_interleaved(c["N"], c["seed"])
            _close(oracle.dft1d_ref(x, c["N"], c["direction"]), _load(c["out_file"]))
        elif c["kind"] == "r2c_dft":
            x = oracle.random_real(c["N"], c["seed"])
            _close(oracle.r2c_ref_packed(x, c["N"], c["normalize"], use_pow2=False), _load(c["out_file"]))
        elif c["kind"] == "c2r_dft":
            _close(oracle.c2r_ref_from_packed(_load(c["in_file"]), c["N"], c["normalize"], use_pow2=False), _load(c["out_file"]))


def test_trig_refs(manifest, oracle):
    """DCT-I..IV / DST-I..IV references against the reference's dct*Ref / dst*Ref (math.js:291-409)"""
    cases, _ = manifest
    seen = 0
    for c in cases.values():
        if c["kind"] != "trig":
            continue
        x = oracle.random_real(c["N"], c["seed"])
        _close(oracle.trig1d_ref(x, c["N"], c["type"], c["direction"]), _load(c["out_file"]))
        seen += 1
    assert seen == 8 * 6 * 2


def test_fftconv_ref(manifest, oracle):
    cases, _ = manifest
    seen = 0
    for c in cases.values():
        if c["kind"] != "fftconv":
            continue
        n = int(np.prod(c["shape"]))
        ks = c["kernelShape"] or c["shape"]
        kn = int(np.prod(ks))
        x = oracle.random_complex_interleaved(n * c["batch"], c["seed"])
        kern = oracle.random_complex_interleaved(kn * c["kernelCount"], c["kernel_seed"])
        want = _load(c["out_file"]).reshape(c["kernelCount"], -1)
        for k in range(c["kernelCount"]):
            got, _ = oracle.fftconv_ref(x, kern[2 * k * kn: 2 * (k + 1) * kn], c["shape"], c["batch"], c["mode"], c["boundary"], ks)
            _close(got, want[k], 4e-6)
        seen += 1
    assert seen >= 9


def test_pow2_and_dft_routes_agree(oracle):
    """the two oracle routes agree to f32 rounding at small pow-2 N"""
    x = oracle.random_complex_interleaved(256, 99)
    a = oracle.fft1d_ref(x, 256, "forward")
    b = oracle.dft1d_ref(x, 256, "forward")
    assert oracle.rel_l2(a, b) < 5e-7

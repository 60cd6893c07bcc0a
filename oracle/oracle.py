"""ctypes loader for oracle/liboracle.so plus numpy twins of the shared PRNG.

TEST INFRASTRUCTURE ONLY (see oracle.c header): imported by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg — never by the product path.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

NORMALIZE = {"none": 0, "backward": 1, "unitary": 2}
MODE = {"convolution": 0, "correlation": 1}
BOUNDARY = {"circular": 0, "linear-full": 1, "linear-same": 2, "linear-valid": 3}


def build():
    """(Re)build liboracle.so with the committed Makefile."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(path):
        build()
    L = ctypes.CDLL(path)
    f32p = ctypes.POINTER(ctypes.c_float)
    i64p = ctypes.POINTER(ctypes.c_int64)
    L.oracle_rng_at.restype = ctypes.c_double
    L.oracle_rng_at.argtypes = [ctypes.c_uint32, ctypes.c_uint64]
    L.oracle_stream_seed.restype = ctypes.c_uint32
    L.oracle_stream_seed.argtypes = [ctypes.c_uint32, ctypes.c_uint64]
    L.oracle_random_complex_interleaved.argtypes = [f32p, ctypes.c_uint64, ctypes.c_uint32]
    L.oracle_random_real.argtypes = [f32p, ctypes.c_uint64, ctypes.c_uint32]
    L.oracle_normalize_scale_factor.restype = ctypes.c_double
    L.oracle_normalize_scale_factor.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double]
    L.oracle_fft1d_ref.argtypes = [f32p, f32p, ctypes.c_int64, ctypes.c_int]
    L.oracle_dft1d_ref.argtypes = [f32p, f32p, ctypes.c_int64, ctypes.c_int]
    L.oracle_stage_twiddle.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    L.oracle_fftnd_ref.argtypes = [f32p, f32p, i64p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.oracle_fftnd_anysize_ref.argtypes = [f32p, f32p, i64p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.oracle_r2c_ref_packed.argtypes = [f32p, f32p, ctypes.c_int64, ctypes.c_int, ctypes.c_int]
    L.oracle_c2r_ref_from_packed.argtypes = [f32p, f32p, ctypes.c_int64, ctypes.c_int, ctypes.c_int]
    L.oracle_fftconv_ref.argtypes = [f32p, f32p, f32p, i64p, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                     i64p, ctypes.c_int, i64p]
    L.oracle_fft1d_ref_batch.argtypes = [f32p, f32p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int]
    L.oracle_trig1d_ref.argtypes = [f32p, f32p, ctypes.c_int64, ctypes.c_int]
    L.oracle_trig_nd_ref.argtypes = [f32p, f32p, i64p, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.oracle_fnv1a64.restype = ctypes.c_uint64
    L.oracle_fnv1a64.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
    _LIB = L
    return L


def _f32p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _shape(shape):
    arr = (ctypes.c_int64 * len(shape))(*[int(s) for s in shape])
    return arr


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed with code {rc}")


# ---- numpy twins of the PRNG (bit-identical to oracle.c / gen_fixtures.mjs) ---------------------
def mulberry32_u32(seed, n0, count):
    """uint32 outputs of draws n0 .. n0+count-1 of stream `seed`."""
    n = np.arange(n0 + 1, n0 + count + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        t = (np.uint64(seed) + n * np.uint64(0x6D2B79F5)).astype(np.uint32)
        t = (t ^ (t >> np.uint32(15))) * (t | np.uint32(1))
        t ^= t + (t ^ (t >> np.uint32(7))) * (t | np.uint32(61))
        return t ^ (t >> np.uint32(14))


def rng_f64(seed, n0, count):
    return mulberry32_u32(seed, n0, count).astype(np.float64) / 4294967296.0


def random_complex_interleaved(length_complex, seed):
    """math.js:150-158 with the seeded PRNG: float32[2*length_complex]."""
    r = rng_f64(seed, 0, 2 * length_complex)
    return ((r * 2.0 - 1.0) * 0.5).astype(np.float32)


def random_real(length, seed):
    r = rng_f64(seed, 0, length)
    return ((r * 2.0 - 1.0) * 0.5).astype(np.float32)


def stream_seed(seed0, b):
    return int(lib().oracle_stream_seed(ctypes.c_uint32(seed0 & 0xFFFFFFFF), ctypes.c_uint64(b)))


def random_complex_batch(n, batch, seed0, b0=0):
    """[batch, 2n] float32; transform b uses the independent stream stream_seed(seed0, b0+b)."""
    out = np.empty((batch, 2 * n), dtype=np.float32)
    for b in range(batch):
        out[b] = random_complex_interleaved(n, stream_seed(seed0, b0 + b))
    return out


def random_real_batch(n, batch, seed0, b0=0):
    out = np.empty((batch, n), dtype=np.float32)
    for b in range(batch):
        out[b] = random_real(n, stream_seed(seed0, b0 + b))
    return out


# ---- transforms -----------------------------------------------------------------------------------
def fft1d_ref(x, n, direction):
    x = np.ascontiguousarray(x, dtype=np.float32)
    assert x.size == 2 * n
    out = np.empty_like(x)
    _chk(lib().oracle_fft1d_ref(_f32p(x), _f32p(out), n, 1 if direction == "inverse" else 0), "fft1d_ref")
    return out


def dft1d_ref(x, n, direction):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    _chk(lib().oracle_dft1d_ref(_f32p(x), _f32p(out), n, 1 if direction == "inverse" else 0), "dft1d_ref")
    return out


def fftnd_ref(x, shape, direction, normalize="none", anysize=False):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    fn = lib().oracle_fftnd_anysize_ref if anysize else lib().oracle_fftnd_ref
    _chk(fn(_f32p(x), _f32p(out), _shape(shape), len(shape), 1 if direction == "inverse" else 0, NORMALIZE[normalize]), "fftnd_ref")
    return out


def c2c_ref_batch(x, shape, batch, direction, normalize="none", anysize=None):
    """batch outermost; pow-2 dims use the radix-2 oracle unless anysize is forced."""
    n = int(np.prod(shape))
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(batch, 2 * n)
    if anysize is None:
        anysize = not all(s >= 2 and (s & (s - 1)) == 0 for s in shape)
    out = np.empty_like(x)
    for b in range(batch):
        out[b] = fftnd_ref(x[b], shape, direction, normalize, anysize)
    return out.reshape(-1)


def fft1d_ref_batch(x, n, batch, direction, nthreads=1):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    _chk(lib().oracle_fft1d_ref_batch(_f32p(x), _f32p(out), n, batch, 1 if direction == "inverse" else 0, nthreads), "fft1d_ref_batch")
    return out


def r2c_ref_packed(x, n, normalize="none", use_pow2=None):
    x = np.ascontiguousarray(x, dtype=np.float32)
    if use_pow2 is None:
        use_pow2 = n >= 2 and (n & (n - 1)) == 0
    out = np.empty(2 * (n // 2 + 1), dtype=np.float32)
    _chk(lib().oracle_r2c_ref_packed(_f32p(x), _f32p(out), n, NORMALIZE[normalize], 1 if use_pow2 else 0), "r2c_ref_packed")
    return out


def c2r_ref_from_packed(xp, n, normalize="none", use_pow2=None):
    xp = np.ascontiguousarray(xp, dtype=np.float32)
    if use_pow2 is None:
        use_pow2 = n >= 2 and (n & (n - 1)) == 0
    out = np.empty(n, dtype=np.float32)
    _chk(lib().oracle_c2r_ref_from_packed(_f32p(xp), _f32p(out), n, NORMALIZE[normalize], 1 if use_pow2 else 0), "c2r_ref_from_packed")
    return out


def fftconv_ref(x, kernel, shape, batch=1, mode="convolution", boundary="circular", kernel_shape=None, use_pow2=False):
    """One kernel; returns (out float32[2*batch*prod(outShape)], outShape)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    kernel = np.ascontiguousarray(kernel, dtype=np.float32)
    rank = len(shape)
    ks = list(kernel_shape) if kernel_shape is not None else list(shape)
    fft_shape = list(shape) if boundary == "circular" else [s + k - 1 for s, k in zip(shape, ks)]
    out_shape = {"circular": list(shape), "linear-full": fft_shape, "linear-same": list(shape),
                 "linear-valid": [s - k + 1 for s, k in zip(shape, ks)]}[boundary]
    out = np.empty(2 * batch * int(np.prod(out_shape)), dtype=np.float32)
    osh = (ctypes.c_int64 * rank)()
    _chk(lib().oracle_fftconv_ref(_f32p(x), _f32p(kernel), _f32p(out), _shape(shape), rank, batch, MODE[mode], BOUNDARY[boundary],
                                  _shape(ks), 1 if use_pow2 else 0, osh), "fftconv_ref")
    return out, list(osh)


def trig_kind(typ, direction):
    """runtime/plans/dct_fft.js:48-57: dct3 = dct2 with the directions exchanged, dst3 likewise"""
    fwd = direction == "forward"
    return {"dct1": 0, "dct2": 1 if fwd else 2, "dct3": 2 if fwd else 1, "dct4": 3,
            "dst1": 4, "dst2": 5 if fwd else 6, "dst3": 6 if fwd else 5, "dst4": 7}[typ]


def trig1d_ref(x, n, typ, direction="forward"):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty(n, dtype=np.float32)
    _chk(lib().oracle_trig1d_ref(_f32p(x), _f32p(out), n, trig_kind(typ, direction)), "trig1d_ref")
    return out


def trig_ref_batch(x, shape, batch, typ, direction="forward", normalize="none"):
    """DCT / DST of `batch` real arrays of `shape` (axis 0 fastest): the 1-D reference along every axis, one final scale"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    _chk(lib().oracle_trig_nd_ref(_f32p(x), _f32p(out), _shape(shape), len(shape), batch, trig_kind(typ, direction),
                                  1 if direction == "inverse" else 0, NORMALIZE[normalize]), "trig_nd_ref")
    return out


def fnv1a64(arr):
    arr = np.ascontiguousarray(arr)
    return int(lib().oracle_fnv1a64(arr.ctypes.data_as(ctypes.c_void_p), arr.nbytes))


def stage_twiddle(length, inverse):
    re, im = ctypes.c_double(), ctypes.c_double()
    lib().oracle_stage_twiddle(length, 1 if inverse else 0, ctypes.byref(re), ctypes.byref(im))
    return re.value, im.value


# ---- parity metrics ---------------------------------------------------------
def rel_l2(a, e):
    a = np.asarray(a, dtype=np.float64)
    e = np.asarray(e, dtype=np.float64)
    d = np.linalg.norm(a - e)
    n = np.linalg.norm(e)
    return d / n if n > 0 else d


def rel_max(a, e):
    a = np.asarray(a, dtype=np.float64)
    e = np.asarray(e, dtype=np.float64)
    m = np.max(np.abs(e)) if e.size else 0.0
    d = np.max(np.abs(a - e)) if e.size else 0.0
    return d / m if m > 0 else d


def assert_close_elementwise(a, e, atol, rtol, what=""):
    """the reference's own per-element form (test/complete.node.test.js:14-25)"""
    a = np.asarray(a, dtype=np.float64)
    e = np.asarray(e, dtype=np.float64)
    bad = np.abs(a - e) > atol + rtol * np.abs(e)
    if bad.any():
        i = int(np.argmax(bad))
        raise AssertionError(f"{what}: element {i}: got {a[i]} expected {e[i]} (atol={atol}, rtol={rtol}); {int(bad.sum())} bad")

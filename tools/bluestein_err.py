"""Prints rel_l2 / rel_max of the chirp-z route per length, direction and launch form against an f64 FFT (numpy) — to see which term puts
the maximum error between 1e-5 and 2e-5.  GPU box: python3 tools/bluestein_err.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "webgpu-fft_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402
import mi355fft as fft  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from test_gpu_parity import run_plan  # noqa: E402

dev = fft.Device(0)
for fused in (1, 0):
    os.environ["MI355FFT_FUSE_VIEWS"] = str(fused)
    for n in (17, 29, 34, 97, 257, 1009, 2039, 6007, 100003):
        batch = 300 if n < 100 else (6 if n < 10000 else 1)
        x = orc.random_complex_batch(n, batch, 0xC100 + n).reshape(-1)
        for direction, norm in (("forward", "none"), ("inverse", "backward")):
            got, (route, launches) = run_plan(fft, dev, {"type": "c2c", "shape": [n], "batch": batch, "direction": direction, "normalize": norm}, x, x.size)
            c = x.astype(np.float64).view(np.complex128).reshape(batch, n)
            w = np.fft.fft(c, axis=1) if direction == "forward" else np.fft.ifft(c, axis=1)
            want = np.stack([w.real, w.imag], axis=-1).reshape(-1)
            g = got.astype(np.float64)
            l2 = np.sqrt(np.sum((g - want) ** 2) / np.sum(want ** 2))
            mx = np.max(np.abs(g - want)) / np.max(np.abs(want))
            print(f"fused={fused} N={n:6d} {direction:7s} rel_l2={l2:.2e} rel_max={mx:.2e} {route.strip()}", flush=True)

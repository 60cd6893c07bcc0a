import os, sys, numpy as np
sys.path.insert(0, "webgpu-fft_amd/python"); sys.path.insert(0, "tests")
import torch, mi355fft
dev = mi355fft.Device(0)
rng = np.random.default_rng(7)
for n in (4096, 8192, 16384, 32768):
    batch = 37
    x = (rng.random(2 * n * batch, dtype=np.float32) - 0.5).astype(np.float32)
    z = x.astype(np.float64).reshape(batch, n, 2); z = z[..., 0] + 1j * z[..., 1]
    for direction in ("forward", "inverse"):
        ref = np.fft.fft(z, axis=1) if direction == "forward" else np.fft.ifft(z, axis=1)
        inp = mi355fft.uploadComplex(dev, x); out = dev.createBuffer({"size": x.nbytes})
        plan = mi355fft.createPlan(dev, {"type": "c2c", "shape": [n], "batch": batch, "direction": direction, "normalize": "backward"})
        enc = dev.createCommandEncoder(); plan.exec(enc, {"input": inp, "output": out}); dev.queue.submit([enc.finish()])
        g = mi355fft.downloadF32(dev, out, 2 * n * batch).astype(np.float64).reshape(batch, n, 2); g = g[..., 0] + 1j * g[..., 1]
        print(n, direction, plan.describe()[0], "rel_l2 %.3e" % (np.linalg.norm(g - ref) / np.linalg.norm(ref)))

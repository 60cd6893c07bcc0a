#!/usr/bin/env python3
"""Phase breakdown of the XCD-resident kernel from its in-kernel stamps (diagnostic variants MI355FFT_XCD_RES=3 / 4):
    MI355FFT_XCD_RES=3 [MI355FFT_XCD_RES_DEPTH=d] python3 tools/res_stamps.py [batch]
The plan runs on a caller-provided temp buffer so that the control block (with the stamps) can be read back."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "webgpu-fft_amd", "python"))
import numpy as np
import torch  # noqa: F401
import mi355fft

PHASES = ["x loads + radix-32", "LDS exchange + radix-32 + four-step roots", "wait: channel free", "push stores complete", "wait: all pushed",
          "payload read from L2", "pass B stage 0", "transposes + stage 1 + output stores issued"]


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    n = 1 << 20
    dev = mi355fft.Device(0, use_graph=False)
    inp = dev.createBuffer({"size": n * batch * 8})
    out = dev.createBuffer({"size": n * batch * 8})
    dev.fillRandom(inp, 0, 2 * n, batch, 0x5EED0003, 0)
    plan = mi355fft.createPlan(dev, {"type": "c2c", "shape": [n], "batch": batch, "direction": "forward", "normalize": "none"})
    route, _ = plan.describe()
    wbytes = plan.getWorkspaceSizeBytes()
    temp = dev.createBuffer({"size": wbytes})
    for rep in range(2):   # second run: warm
        enc = dev.createCommandEncoder()
        plan.exec(enc, {"input": inp, "output": out, "temp": temp})
        dev.queue.submit([enc.finish()])
        dev.queue.onSubmittedWorkDone()
    ctl_off = 16 * 4 * (1 << 20)
    words = mi355fft.downloadF32(dev, temp, (68 + 512 * 64) // 4 + 16, ctl_off).view(np.uint32)
    bar = words[17:17 + 512 * 16].reshape(512, 16)
    st = bar[256:512]
    k = st[:, 8].astype(np.float64)
    ok = k > 0
    print(f"route {route.strip()}  batch {batch}  workgroups with stamps {int(ok.sum())}")
    per = st[ok, :8].astype(np.float64) / k[ok, None] / 100.0    # us per transform (100 MHz ticks)
    tot = per.sum(axis=1)
    print(f"us per transform per workgroup: mean {tot.mean():.2f}  min {tot.min():.2f}  max {tot.max():.2f}   -> {tot.mean() / 8:.2f} us per transform chip-wide")
    for i, name in enumerate(PHASES):
        print(f"  {i} {name:52s} mean {per[:, i].mean():7.2f}  min {per[:, i].min():7.2f}  max {per[:, i].max():7.2f}")


if __name__ == "__main__":
    main()

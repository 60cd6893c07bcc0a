#!/usr/bin/env python3
"""Latency-bound BASELINE configs (1: c2c N=1024 batch=1; 4: fftconv [256] batch 4, 64->128 channel lanes, 3 kernels):
µs per exec (submit -> done) and kernel launches per exec, for the op-list and the hipGraph executors."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "webgpu-fft_amd", "python")]
import numpy as np  # noqa: E402

try:
    import torch  # noqa: F401,E402
except Exception:
    pass
import mi355fft  # noqa: E402


def measure(dev, plan, args, iters=200, warm=10):
    out = {}
    for name, graph in (("op_list", False), ("hipgraph", True)):
        enc = dev.createCommandEncoder()
        plan.exec(enc, args)
        cmds = enc.finish(use_graph=graph)
        for _ in range(warm):
            dev.queue.submit([cmds])
        dev.queue.onSubmittedWorkDone()
        # (a) the reference's bench shape (bench_1d_1024.js:26-65): N submits, one wait
        t0 = time.perf_counter()
        for _ in range(iters):
            dev.queue.submit([cmds])
        dev.queue.onSubmittedWorkDone()
        pipelined = (time.perf_counter() - t0) / iters * 1e6
        # (b) submit -> done round trip
        t0 = time.perf_counter()
        for _ in range(iters):
            dev.queue.submit([cmds])
            dev.queue.onSubmittedWorkDone()
        roundtrip = (time.perf_counter() - t0) / iters * 1e6
        out[name] = {"us_per_exec_pipelined": pipelined, "us_per_exec_roundtrip": roundtrip}
        cmds.release()
    return out


def main():
    dev = mi355fft.Device(0)
    res = {}
    x = np.zeros(2 * 1024, np.float32)
    x[2] = 1.0
    inp = mi355fft.uploadComplex(dev, x)
    outb = dev.createBuffer({"size": x.nbytes})
    plan = mi355fft.createPlan(dev, {"type": "c2c", "shape": [1024], "batch": 1, "direction": "forward", "normalize": "none"})
    route, launches = plan.describe()
    res["cfg1_c2c_N1024_b1"] = dict(measure(dev, plan, {"input": inp, "output": outb}), route=route.strip(), launches_per_exec=launches)
    plan.destroy()

    preset = mi355fft.createFftConvKernelMajorChannelLanePreset({"shape": [256], "batch": 4, "kernelCount": 3, "input": {"channels": 64},
                                                                 "output": {"channels": 128, "kernelStepChannels": 16}})
    phys = mi355fft.uploadComplex(dev, np.random.default_rng(0).standard_normal(2 * 4 * 64 * 256).astype(np.float32))
    outp = dev.createBuffer({"size": 4 * 128 * 256 * 8})
    kern = mi355fft.uploadComplex(dev, np.random.default_rng(1).standard_normal(2 * 3 * 256).astype(np.float32))
    for label, env in (("fused", None), ("composed", "1")):
        if env:
            os.environ["MI355FFT_FORCE_GENERIC"] = env
        plan = mi355fft.createPlan(dev, dict(preset, type="fftconv"))
        os.environ.pop("MI355FFT_FORCE_GENERIC", None)
        route, launches = plan.describe()
        res[f"cfg4_fftconv_256_b4_k3_{label}"] = dict(measure(dev, plan, {"input": phys, "output": outp, "kernel": kern}), route=route.strip(),
                                                      launches_per_exec=launches)
        plan.destroy()
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()

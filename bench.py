#!/usr/bin/env python3
"""bench.py — headline benchmark: batched 1-D c2c f32, N=2^20, batch=4096 per GPU (BASELINE.json config 3).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--no-graph] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the hot path (plan.exec recorded once into a command list, replayed per step) over
one batch of synthetic complex input already resident in HBM (device-side twin of the oracle's seeded PRNG,
components uniform in (-0.5, 0.5) as math.js:150-158).  One process per GPU; ranks shard the batch dimension
(rank r owns transforms [r*B, (r+1)*B)), no data-path collective — torch.distributed (RCCL)
only brackets the timed region with barriers and reduces the elapsed time with MAX.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "webgpu-fft_amd", "python"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md "Chip-level parameters")

ND_SHAPE = []   # set by an N-D probe workload
LIN_KERNEL = []   # kernel length of a fftconvlin probe
WORKLOADS = {
    # name: (type, N, batch per GPU, algorithmic bytes per point, description)
    "c2c_2p20_b4096": ("c2c", 1 << 20, 4096, 16, "1D c2c N=2^20 batch=4096 f32 forward, out-of-place (BASELINE config 3, north-star metric)"),
    "c2c_1024_b65536": ("c2c", 1024, 65536, 16, "1D c2c N=1024 batch=65536 f32 forward, out-of-place (BASELINE config 2)"),
    "c2c_2p20_b512": ("c2c", 1 << 20, 512, 16, "1D c2c N=2^20 batch=512 (reduced batch; NOT the headline config)"),
    "r2c_2p22_b1024": ("r2c", 1 << 22, 1024, 8, "1D r2c N=2^22, 1024 transforms per GPU (BASELINE config 5 shard)"),
}


class HipEvents:
    """hipEvent timing on the library's own stream (torch.cuda.Event would only see torch's stream)."""

    def __init__(self):
        self.hip = None
        for name in ("libamdhip64.so.7", "libamdhip64.so"):
            try:
                self.hip = ctypes.CDLL(name)
                break
            except OSError:
                continue
        if self.hip is None:
            raise RuntimeError("libamdhip64 not loadable")
        self.hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self.hip.hipEventSynchronize.argtypes = [ctypes.c_void_p]
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        self.hip.hipEventDestroy.argtypes = [ctypes.c_void_p]

    def create(self):
        e = ctypes.c_void_p()
        rc = self.hip.hipEventCreate(ctypes.byref(e))
        if rc != 0:
            raise RuntimeError(f"hipEventCreate failed: {rc}")
        return e

    def record(self, ev, stream):
        rc = self.hip.hipEventRecord(ev, stream)
        if rc != 0:
            raise RuntimeError(f"hipEventRecord failed: {rc}")

    def elapsed_ms(self, a, b):
        self.hip.hipEventSynchronize(b)
        ms = ctypes.c_float()
        rc = self.hip.hipEventElapsedTime(ctypes.byref(ms), a, b)
        if rc != 0:
            raise RuntimeError(f"hipEventElapsedTime failed: {rc}")
        return float(ms.value)


def cpu_baseline(n, seconds_target=12.0):
    """the oracle (C restatement of math.js:25-88, kind "port") on the host cores, bounded sample"""
    import numpy as np
    from oracle import oracle as orc
    cores = os.cpu_count() or 1
    threads = max(1, min(cores, 64))
    if n >= (1 << 16):
        sample = max(threads, 32)
    else:
        sample = max(threads * 64, 4096)
    x = orc.random_complex_batch(n, min(sample, 64), 0x5EED00C0).reshape(-1)
    reps = sample // min(sample, 64)
    x = np.tile(x, reps)
    sample = x.size // (2 * n)
    t0 = time.perf_counter()
    orc.fft1d_ref_batch(x, n, sample, "forward", nthreads=threads)
    dt = time.perf_counter() - t0
    rounds = 1
    # repeat the same sample until ~seconds_target of CPU work has been timed
    extra = int(min(max(seconds_target / max(dt, 1e-6) - 1, 0), 200))
    t1 = time.perf_counter()
    for _ in range(extra):
        orc.fft1d_ref_batch(x, n, sample, "forward", nthreads=threads)
    dt_total = dt + (time.perf_counter() - t1)
    rounds += extra
    pts = float(n) * sample * rounds
    return {"value": pts / dt_total / 1e9, "unit": "GPoints/s", "cores": threads, "kind": "port",
            "sample": f"{sample} transforms of N={n} x {rounds} rounds in {dt_total:.1f}s, oracle/oracle.c fft1d_ref (radix-2, f32 storage, f64 twiddles), {threads} pthreads"}


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _usable_cores():
    try:
        return len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return os.cpu_count() or 1


def cpu_baseline_node(n, seconds_target=10.0):
    """The reference's Node.js CPU correctness path (src/utils/math.js:25-88 fft1dRefInterleaved, single-threaded JS) as restated
    in oracle/oracle.mjs (pinned bit-exact to reference-generated fixtures), timed on the host cores: one Node child process per
    usable core (Node 12 here: child processes, no worker_threads needed), each transforming its own seeded slice of the
    workload for ~seconds_target; value = total points / the longest child's time."""
    import shutil
    node = shutil.which("node")
    if node is None:
        raise RuntimeError("node not on PATH")
    total, usable = os.cpu_count() or 1, _usable_cores()
    used = max(1, min(usable, 64))
    per = 1 if n >= (1 << 18) else max(1, (1 << 18) // n)       # transforms per child per round (~2^18 points)
    worker = os.path.join(ROOT, "oracle", "cpu_baseline_worker.mjs")
    t0 = time.perf_counter()
    procs = [subprocess.Popen([node, worker, str(n), str(per), str(0x5EED00C0 + i), str(seconds_target)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for i in range(used)]
    pts, longest, version, rounds = 0.0, 0.0, None, 0
    for p in procs:
        so, se = p.communicate(timeout=seconds_target * 6 + 60)
        if p.returncode != 0:
            raise RuntimeError(f"node worker failed: {se.strip()[-200:]}")
        d = json.loads(so.strip().splitlines()[-1])
        pts += d["points"]
        longest = max(longest, d["seconds"])
        rounds += d["rounds"]
        version = d["node"]
    wall = time.perf_counter() - t0
    value = pts / longest / 1e9
    return {"value": value, "unit": "GPoints/s", "cores": used, "kind": "node", "node": version, "cpu_model": _cpu_model(),
            "cores_total": total, "cores_used": used, "per_core": value / used,
            "sample": f"{used} Node processes x {per} transform(s) of N={n} x {rounds} rounds in all, {longest:.1f}s each ({wall:.1f}s wall), "
                      f"oracle/oracle.mjs fft1dRef = restatement of the reference's src/utils/math.js:25-88 (radix-2, Float32Array storage, f64 twiddle recurrence)"}


def pmc_traffic(workload):
    """HBM bytes per step from the committed PMC passes (profiles/r*_pmc_traffic_<workload>.json), newest round"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_traffic_{workload}.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        d = json.load(f)
    return d.get("hbm_bytes_per_step"), os.path.relpath(files[-1], ROOT)


def fabric_ceiling(dev, ev, src, dst, nbytes, reps=5):
    """This chip's copy rate, measured live through the product's own copyBufferToBuffer (one-shot 16-byte nontemporal streaming
    kernel, kern_generic.hpp stream_copy_kernel): GB/s of read + write traffic.  It is the ceiling for ANY kernel's fabric traffic on
    this box — what `roofline.attainable` prices the transform against, next to (never instead of) the 8 TB/s roofline."""
    nbytes = (min(nbytes, 8 << 30) // 16) * 16
    enc = dev.createCommandEncoder()
    enc.copyBufferToBuffer(src, 0, dst, 0, nbytes)
    cmds = enc.finish()
    dev.queue.submit([cmds])
    dev.queue.onSubmittedWorkDone()
    a, b = ev.create(), ev.create()
    ev.record(a, dev.stream)
    for _ in range(reps):
        dev.queue.submit([cmds])
    ev.record(b, dev.stream)
    dev.queue.onSubmittedWorkDone()
    ms = ev.elapsed_ms(a, b)
    cmds.release()
    return 2.0 * nbytes * reps / (ms / 1e3) / 1e9, nbytes


def time_single_pass(mi355fft, dev, ev, opts, inp, out, which, reps):
    """live hipEvent timing of ONE of the two pass kernels (MI355FFT_ONLY_PASS planner aid): avg us per launch"""
    os.environ["MI355FFT_ONLY_PASS"] = str(which)
    try:
        plan = mi355fft.createPlan(dev, opts)
    finally:
        del os.environ["MI355FFT_ONLY_PASS"]
    _, launches = plan.describe()
    enc = dev.createCommandEncoder()
    plan.exec(enc, {"input": inp, "output": out})
    cmds = enc.finish(use_graph=False)
    dev.queue.submit([cmds])
    dev.queue.onSubmittedWorkDone()
    a, b = ev.create(), ev.create()
    ev.record(a, dev.stream)
    for _ in range(reps):
        dev.queue.submit([cmds])
    ev.record(b, dev.stream)
    ms = ev.elapsed_ms(a, b)
    cmds.release()
    plan.destroy()
    return ms * 1e3 / (reps * max(launches, 1)), launches


def launch_ranks(n):
    """parent side of `bench.py --gpus N`: torch.distributed.run with one rank per GPU on this node, as a child process"""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def stub_worker(args):
    """MI355FFT_BENCH_STUB=1: the rank plumbing of a bench run without a GPU (tests/test_bench_launcher.py)"""
    from mi355fft.sharding import Group, rank_info, shard_range
    rank, local_rank, world = rank_info()
    group = Group("gloo")
    seen = int(round(group.reduce_sum([1.0])[0]))
    first, last = shard_range(rank, world, 4096 * world)
    spans = group.reduce_sum([float(last - first)])
    group.barrier()
    if rank == 0:
        print(json.dumps({"metric": "stub", "value": None, "n_gpus": world, "ranks_seen": seen, "collective_backend": group.backend or "none",
                          "global_batch": int(spans[0]), "steps": args.steps, "warmup": args.warmup}), flush=True)
    group.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=os.environ.get("MI355FFT_BENCH_WORKLOAD", "c2c_2p20_b4096"))
    ap.add_argument("--no-graph", action="store_true", help="replay the op list instead of a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    gpus_given = any(a == "--gpus" or a.startswith("--gpus=") for a in sys.argv[1:])

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: start the N ranks as CHILD processes (one per GPU) before anything here has
        # touched the GPU — this parent never imports torch or opens a device — and exit with the launcher's code.
        sys.exit(launch_ranks(args.gpus))
    if "WORLD_SIZE" in os.environ and not gpus_given:
        args.gpus = int(os.environ["WORLD_SIZE"])       # started under torch.distributed.run without --gpus: adopt its world size
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks")
    if os.environ.get("MI355FFT_BENCH_STUB"):   # CPU-tier test of the launcher plumbing: no device work at all
        return stub_worker(args)

    import torch  # first: its bundled HIP runtime is the one the library then binds to
    import mi355fft
    from mi355fft.sharding import Group, rank_info, shard_range
    rank, local_rank, world = rank_info()
    share_gpu = bool(os.environ.get("MI355FFT_BENCH_SHARE_GPU"))
    if share_gpu:   # rehearsal on a 1-GPU box: all ranks use device 0 (RCCL refuses two ranks on one device, gloo takes over)
        local_rank = 0
    if world > 1:
        torch.cuda.set_device(local_rank)
    # RCCL over xGMI; barrier + scalar reductions only.  On a real multi-GPU run a failing nccl init is an error, not a fallback.
    group = (Group("gloo") if share_gpu else Group("nccl", torch.device("cuda", local_rank)))
    ranks_seen = int(round(group.reduce_sum([1.0])[0]))
    if ranks_seen != world:
        raise SystemExit(f"collective saw {ranks_seen} ranks, expected {world}")

    if args.workload not in WORKLOADS:   # probes (never the headline line): c2c_2pL_bB, r2c_n1000_bB, c2c_s1024x1024_bB ...
        import re
        m = re.fullmatch(r"(c2c|r2c|c2r|fftconv|dct[1-4]|dst[1-4])_(2p|n)(\d+)_b(\d+)", args.workload)
        nd = re.fullmatch(r"(c2c|r2c|dct[1-4]|dst[1-4])_s((?:\d+x)*\d+)_b(\d+)(_view)?", args.workload)      # N-D: axis 0 first; _view: padded read + cropped write
        lin = re.fullmatch(r"fftconvlin_n(\d+)k(\d+)_b(\d+)", args.workload)   # linear-full convolution, data n, kernel k (probe)
        if lin:
            WORKLOADS[args.workload] = ("fftconvlin", int(lin.group(1)), int(lin.group(3)), 16,
                                        f"1D fftconv linear-full N={lin.group(1)} kernel={lin.group(2)} batch={lin.group(3)} (probe; NOT a BASELINE config)")
            LIN_KERNEL[:] = [int(lin.group(2))]
        elif nd:
            ND_SHAPE[:] = [int(v) for v in nd.group(2).split("x")]
            tot = 1
            for v in ND_SHAPE:
                tot *= v
            WORKLOADS[args.workload] = (nd.group(1), tot, int(nd.group(3)), 16 if nd.group(1) == "c2c" else 8,
                                        f"{len(ND_SHAPE)}-D {nd.group(1)} {nd.group(2)} batch={nd.group(3)} (probe; NOT a BASELINE config)")
        elif m:
            nn = 1 << int(m.group(3)) if m.group(2) == "2p" else int(m.group(3))
            WORKLOADS[args.workload] = (m.group(1), nn, int(m.group(4)), 16 if m.group(1) in ("c2c", "fftconv") else 8,
                                        f"1D {m.group(1)} N={nn} batch={m.group(4)} (probe; NOT a BASELINE config)")
        else:
            raise SystemExit(f"unknown workload {args.workload}")
    typ, n, batch, bytes_per_point, desc = WORKLOADS[args.workload]
    dev = mi355fft.Device(local_rank, use_graph=False if args.no_graph else "auto")
    info = dev.info()
    if typ == "c2c":
        in_bytes = out_bytes = n * batch * 8
        in_row_floats = 2 * n
        opts = {"type": "c2c", "shape": list(ND_SHAPE) if ND_SHAPE else [n], "batch": batch, "direction": "forward", "normalize": "none"}
        if args.workload.endswith("_view"):
            # probe of the sides fused into the line kernels: the input is a smaller array padded into the
            # logical domain on read (centred), the output a cropped window, with a zero range on each side
            vshape = [max(1, (v * 15) // 16) for v in opts["shape"]]
            opts["ioView"] = {"input": {"shape": vshape, "placement": "center"}, "output": {"shape": vshape, "placement": "center"}}
            opts["zeroPad"] = {"read": {"start": [1] * len(vshape), "end": [v - 1 for v in opts["shape"]]},
                               "write": {"start": [0] * len(vshape), "end": [v - 2 for v in opts["shape"]]}}
            vn = 1
            for v in vshape:
                vn *= v
            in_bytes = out_bytes = vn * batch * 8
            in_row_floats = 2 * vn
    elif typ == "fftconvlin":   # linear-full convolution: zero-padded embed of data and kernel, product, inverse, full output
        kl = LIN_KERNEL[0]
        in_bytes, out_bytes = n * batch * 8, (n + kl - 1) * batch * 8
        in_row_floats = 2 * n
        opts = {"type": "fftconv", "shape": [n], "batch": batch, "fftConv": {"mode": "convolution", "boundary": "linear-full", "kernelCount": 1, "kernelShape": [kl]}}
    elif typ == "fftconv":   # circular convolution with one full-length kernel (probe): forward FFTs, product, inverse FFT
        in_bytes = out_bytes = n * batch * 8
        in_row_floats = 2 * n
        opts = {"type": "fftconv", "shape": [n], "batch": batch, "fftConv": {"mode": "convolution", "boundary": "circular", "kernelCount": 1}}
    elif typ[:3] in ("dct", "dst"):   # real-to-real probe
        in_bytes = out_bytes = n * batch * 4
        in_row_floats = n
        opts = {"type": typ, "shape": list(ND_SHAPE) if ND_SHAPE else [n], "batch": batch, "direction": "forward", "normalize": "none",
                "layout": {"interleavedComplex": False}}
    elif typ == "r2c":
        shp = list(ND_SHAPE) if ND_SHAPE else [n]
        packed = (shp[0] // 2 + 1) * (n // shp[0])
        in_bytes, out_bytes = n * batch * 4, packed * batch * 8
        in_row_floats = n
        opts = {"type": "r2c", "shape": shp, "batch": batch, "direction": "forward", "normalize": "none"}
        if args.workload.endswith("_view"):   # as the c2c probe: padded read of a smaller real array, cropped write of the packed bins, zero ranges
            pk = [shp[0] // 2 + 1] + shp[1:]
            vin = [max(1, (v * 15) // 16) for v in shp]
            vout = [max(1, (v * 15) // 16) for v in pk]
            opts["ioView"] = {"input": {"shape": vin, "placement": "center"}, "output": {"shape": vout, "placement": "start"}}
            opts["zeroPad"] = {"read": {"start": [1] * len(shp), "end": [v - 1 for v in shp]}, "write": {"start": [0] * len(shp), "end": [max(1, v - 2) for v in pk]}}
            vi = vo = 1
            for a_, b_ in zip(vin, vout):
                vi *= a_
                vo *= b_
            in_bytes, out_bytes = vi * batch * 4, vo * batch * 8
            in_row_floats = vi
    else:   # c2r probe: random packed spectra (not Hermitian-consistent in bins 0 and N/2; irrelevant for timing)
        in_bytes, out_bytes = (n // 2 + 1) * batch * 8, n * batch * 4
        in_row_floats = 2 * (n // 2 + 1)
        opts = {"type": "c2r", "shape": [n], "batch": batch, "direction": "inverse", "normalize": "none"}
    need = in_bytes + out_bytes + (1 << 30)
    if info["hbm_free"] < need:
        raise SystemExit(f"workload {args.workload} needs {need >> 30} GiB of HBM, {info['hbm_free'] >> 30} GiB free")

    inp = dev.createBuffer({"size": in_bytes})
    out = dev.createBuffer({"size": out_bytes})
    # synthetic input, generated on the device: this rank's shard [first, first+batch) of the global batch
    first, last = shard_range(rank, world, batch * world)
    assert last - first == batch
    dev.fillRandom(inp, 0, in_row_floats, batch, 0x5EED0003, first)
    plan = mi355fft.createPlan(dev, opts)
    route, launches = plan.describe()
    enc = dev.createCommandEncoder()
    exec_args = {"input": inp, "output": out}
    if typ in ("fftconv", "fftconvlin"):
        kn = n if typ == "fftconv" else LIN_KERNEL[0]
        kbuf = dev.createBuffer({"size": kn * 8})
        dev.fillRandom(kbuf, 0, 2 * kn, 1, 0x5EED0004, 0)
        exec_args["kernel"] = kbuf
    plan.exec(enc, exec_args)
    cmds = enc.finish()
    ev = HipEvents()
    e0, e1 = ev.create(), ev.create()

    def barrier():
        dev.queue.onSubmittedWorkDone()
        torch.cuda.synchronize()
        group.barrier()

    def timed_steps():
        for _ in range(args.warmup):
            dev.queue.submit([cmds])
        if not share_gpu:
            barrier()
        t0 = time.perf_counter()
        ev.record(e0, dev.stream)
        for _ in range(args.steps):
            dev.queue.submit([cmds])
        ev.record(e1, dev.stream)
        dev.queue.onSubmittedWorkDone()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, ev.elapsed_ms(e0, e1)

    if share_gpu and world > 1:
        # rehearsal of the N-rank path on ONE GPU: the fused kernels need every CU to themselves (two of them from two processes on one
        # device starve each other of CUs until their bounded waits give up), so the ranks take turns; `value` then says nothing about
        # scaling — the rehearsal checks the launcher, the sharding, ranks_seen and the reductions
        wall = dev_ms = 0.0
        for turn in range(world):
            if turn == rank:
                wall, dev_ms = timed_steps()
            barrier()
    else:
        wall, dev_ms = timed_steps()
        barrier()

    # MAX over ranks of the wall time around the K steps (and of the device-event time)
    wall_max, dev_max = group.reduce_max([wall, dev_ms / 1e3])

    per_kernel = None
    if rank == 0 and typ == "c2c" and "two-pass" in route:
        # per-kernel averages, measured live with hipEvents on the library's stream (must agree with the rocprofv3
        # kernel-trace summary committed under profiles/).  Pass B alone reads whatever pass A left in the workspace.
        ua, la = time_single_pass(mi355fft, dev, ev, opts, inp, out, 1, 3)
        ub, lb = time_single_pass(mi355fft, dev, ev, opts, inp, out, 2, 3)
        pts_launch = float(n) * batch / max(la, 1)
        per_kernel = [
            {"kernel": "fft_lines_kernel PASS_A (column FFT, N1=%d)" % (1 << (n.bit_length() - 1) // 2), "avg_launch_us": ua, "launches_per_step": la,
             "moved_bytes_per_launch": 16 * pts_launch, "moved_GBps": 16 * pts_launch / ua / 1e3},
            {"kernel": "fft_lines_kernel PASS_B (twiddle + row FFT + transposed store)", "avg_launch_us": ub, "launches_per_step": lb,
             "moved_bytes_per_launch": 16 * pts_launch, "moved_GBps": 16 * pts_launch / ub / 1e3},
        ]

    if rank == 0:
        points_per_step = float(n) * batch * world
        ms_per_step = wall_max / args.steps * 1e3
        value = points_per_step / (wall_max / args.steps) / 1e9
        # roofline of the transform on ONE GPU: algorithmic bytes of a step / device time of a step (hip events)
        step_dev_s = dev_max / args.steps
        achieved = bytes_per_point * float(n) * batch / step_dev_s / 1e9
        traffic, traffic_src = pmc_traffic(args.workload)
        fused = ("xcd-fused" in route or "xcd-r2c" in route or "xcd-c2r" in route) and launches == 2
        resident = "xcd-resident" in route and launches == 2
        if resident:
            dominant, dominant_launches = "fft_xcd_res_kernel (transform resident in one XCD's registers + LDS between its passes, hand-offs through the L2)", 1
        elif fused:
            kname = ("fft_xcd_rt_r2c_kernel" if "xcd-r2c-rt" in route else "fft_xcd_rt_c2r_kernel" if "xcd-c2r-rt" in route else "fft_xcd_rt_kernel" if "xcd-fused-rt" in route
                     else "fft_xcd_rt1k_kernel" if "xcd-fused-rt32" in route else "fft_xcd_hx_kernel" if "xcd-fused-2wg" in route else "fft_xcd_r2c_kernel" if "xcd-r2c" in route else "fft_xcd_c2r_kernel" if "xcd-c2r" in route else "fft_xcd_fused_kernel")
            dominant, dominant_launches = kname + " (pass A + XCD barrier + pass B in one persistent launch)", 1
        elif "xcd-fused" in route:
            dominant, dominant_launches = "fft_xcd_fused_kernel, then " + route.split("]")[-1].strip() + " kernel", launches - 1
        elif "two-pass" in route:
            dominant, dominant_launches = "fft_lines_kernel (pass A + pass B per chunk)", launches
        else:
            dominant, dominant_launches = "fft_lines_kernel", launches
        line = {
            "metric": "1D c2c f32 GPoints/s at N=2^20 batch=4096" if args.workload == "c2c_2p20_b4096" else f"GPoints/s ({args.workload})",
            "value": value, "unit": "GPoints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (device-side seeded PRNG twin of the oracle, uniform (-0.5,0.5), resident in HBM)",
            "collective_backend": (group.backend or "none (single rank)"), "ranks_seen": ranks_seen,
            "config": {"workload": desc, "type": typ, "N": n, "batch_per_gpu": batch, "global_batch": batch * world,
                       "sharding": f"batch-sharded x{world}, no data-path collective", "route": route.strip(),
                       "launches_per_step": launches, "executor": "op-list replay" if (args.no_graph or launches < 8) else "hipGraph replay",
                       "arch": info["arch"], "compute_units": info["compute_units"]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": dominant, "algorithmic_bytes_per_point": bytes_per_point, "device_ms_per_step": step_dev_s * 1e3,
                         "avg_launch_us": step_dev_s * 1e6 / max(dominant_launches, 1)},
        }
        if resident:
            line["roofline"]["note"] = ("one fft_xcd_res_kernel launch per step carries the whole batch; the only other launch of a step is the 36 KiB zero_kernel "
                                        "that resets its control block (<3 us), so the step's device time is that kernel's launch duration")
        if fused:
            line["roofline"]["note"] = ("one " + kname + " launch per step carries the whole batch (column FFTs -> per-XCD workspace slot -> "
                                        "XCD barrier -> four-step roots + row FFTs); the only other launch of a step is the 8 KiB zero_kernel "
                                        "that resets its control block (<3 us), so the step's device time is that kernel's launch duration")
        try:
            ceil_gbs, ceil_bytes = fabric_ceiling(dev, ev, inp, out, min(in_bytes, out_bytes))
            moved = (traffic / (float(n) * batch)) if traffic else None          # bytes per point through the fabric (PMC passes)
            line["roofline"]["attainable"] = {
                "fabric_ceiling_GBps": ceil_gbs, "fabric_ceiling_source": f"copyBufferToBuffer of {ceil_bytes >> 20} MiB timed in this run (read + write bytes / device time)",
                "bytes_per_point_moved": moved,
                "moved_GBps": (moved * float(n) * batch / step_dev_s / 1e9) if moved else None,
                "frac_of_attainable": (moved * float(n) * batch / step_dev_s / 1e9 / ceil_gbs) if moved else None,
                "ceiling_for_this_traffic_frac_of_peak": (bytes_per_point / moved * ceil_gbs / HBM_PEAK_GBS) if moved else None,
                "note": "frac (of the 8 TB/s roofline, algorithmic bytes) is the figure of merit; this block says how much of the gap is the chip's copy ceiling "
                        "and how much is traffic beyond the algorithmic bytes (the four-step intermediate crosses the fabric once in each direction)"}
        except Exception as e:   # never lose the line over the side measurement
            line["roofline"]["attainable"] = {"error": str(e)}
        if per_kernel:
            line["roofline"]["per_kernel"] = per_kernel
            line["roofline"]["note"] = ("a launch in the roofline sense is the pass A + pass B pair that moves each point in and out once: "
                                        "algorithmic 16 B/point over the pair; each pass alone moves 16 B/point through the fabric")
        if world > 1:
            line["cpu_baseline"] = None
            line["cpu_baseline_note"] = "reported at N=1 only (rank 0 of a single-GPU run): see the n_gpus=1 line"
        elif not args.no_cpu_baseline:
            # the baseline the north_star names: the reference's Node.js CPU correctness path on the host cores; the C/pthreads
            # port of the same algorithm is kept beside it.  Reported numbers, never a reason to lose the GPU line.
            try:
                line["cpu_baseline"] = cpu_baseline_node(n)
            except Exception as e:
                line["cpu_baseline"] = {"value": None, "unit": "GPoints/s", "cores": 0, "kind": "node", "sample": f"failed: {e}"}
            try:
                line["cpu_baseline_port"] = cpu_baseline(n, seconds_target=6.0)
            except Exception as e:
                line["cpu_baseline_port"] = {"value": None, "unit": "GPoints/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(line), flush=True)

    cmds.release()
    plan.destroy()
    inp.destroy()
    out.destroy()
    dev.close()
    group.barrier()   # rank 0 may still be timing the single-pass plans / CPU baseline
    group.close()


if __name__ == "__main__":
    main()

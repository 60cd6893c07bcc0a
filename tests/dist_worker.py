"""worker for tests/test_distributed_cpu.py: one rank of a batch-sharded transform on the CPU (gloo).
The FFT itself runs through the host emulation of the HIP kernels; what is under test is the sharding
(contiguous per-rank ranges, per-transform PRNG streams) and the reduce of error norms / timings."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "webgpu-fft_amd", "python"), os.path.join(ROOT, "tests")]

import emu_harness as emu  # noqa: E402
from mi355fft import _abi  # noqa: E402
from mi355fft.sharding import Group, shard_range  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main():
    n, global_batch, seed0 = int(sys.argv[1]), int(sys.argv[2]), 0x5EED0003
    group = Group("gloo")
    first, last = shard_range(group.rank, group.world, global_batch)
    batch = last - first
    x = orc.random_complex_batch(n, batch, seed0, b0=first).reshape(-1)
    desc = _abi.make_desc("c2c", [n], batch, "forward", "none")
    got, route, _ = emu.run_plan(desc, x, x.size)
    want = orc.c2c_ref_batch(x, [n], batch, "forward", "none")
    d = got.astype(np.float64) - want
    group.barrier()
    err2, ref2, count = group.reduce_sum([float(np.sum(d * d)), float(np.sum(want.astype(np.float64) ** 2)), float(batch)])
    (tmax,) = group.reduce_max([0.001 * (group.rank + 1)])
    # a global checksum that only matches if every transform was processed exactly once, by the right rank
    local = sum(float(np.sum(got[2 * n * i:2 * n * (i + 1)].astype(np.float64))) * (first + i + 1) for i in range(batch))
    (checksum,) = group.reduce_sum([local])
    if group.rank == 0:
        print(json.dumps({"rel_l2": (err2 / ref2) ** 0.5, "count": count, "tmax": tmax, "checksum": checksum, "route": route, "world": group.world}))
    group.close()


if __name__ == "__main__":
    main()

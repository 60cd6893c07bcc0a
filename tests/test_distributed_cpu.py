"""N>1 path on the CPU: world_size 2 and 3 over gloo (127.0.0.1), batch-sharded exactly as bench.py shards
on GPUs; the result must equal the single-process run."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT
from mi355fft.sharding import shard_range


def test_shard_range_is_a_partition():
    for world in (1, 2, 3, 8):
        for gb in (1, 7, 8, 4096, 8192 + 3):
            spans = [shard_range(r, world, gb) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _launch(world, n, gb, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if world == 1:
        cmd = [sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(n), str(gb)]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py"), str(n), str(gb)]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


@pytest.mark.parametrize("world", [2, 3])
def test_batch_sharded_transform_matches_single_process(world):
    n, gb = 256, 11                      # ragged: ranks get different shard sizes
    single = _launch(1, n, gb, 29611)
    multi = _launch(world, n, gb, 29612 + world)
    assert multi["world"] == world and multi["count"] == gb == single["count"]
    assert multi["rel_l2"] < 1e-5 and single["rel_l2"] < 1e-5
    assert abs(multi["checksum"] - single["checksum"]) <= 1e-9 * max(1.0, abs(single["checksum"]))
    assert abs(multi["tmax"] - 0.001 * world) < 1e-12   # MAX over ranks

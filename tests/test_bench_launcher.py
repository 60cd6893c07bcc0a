"""bench.py --gpus N starts the N ranks itself (the driver's other command shape wraps the script in
torch.distributed.run).  CPU tier: the launcher path with the stub worker (MI355FFT_BENCH_STUB=1: rank plumbing over gloo,
no device work) must come back with one JSON line that saw both ranks."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, extra_env=None):
    env = dict(os.environ)
    env["MI355FFT_BENCH_STUB"] = "1"
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=300)
    return r


def test_gpus_flag_launches_that_many_ranks():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout          # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["collective_backend"] == "gloo"
    assert d["global_batch"] == 2 * 4096       # weak scaling: every rank keeps the configured batch


def test_world_size_and_gpus_flag_must_agree():
    r = _run(["--gpus", "4", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_single_rank_stub_runs_in_process():
    r = _run(["--steps", "1", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1

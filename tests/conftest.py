import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "webgpu-fft_amd", "python")):
    if p not in sys.path:
        sys.path.insert(0, p)

# torch ships its own libamdhip64.so.7; importing it BEFORE libmi355fft.so is loaded makes the
# dynamic loader hand that same runtime to our library (matching SONAME) instead of loading a
# second HIP runtime into the process when a later test imports torch.distributed.
try:  # pragma: no cover
    import torch  # noqa: F401
except Exception:  # torch is plumbing for the multi-process tests only
    torch = None

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_present():
    if os.environ.get("MI355FFT_FORCE_NO_GPU"):
        return False
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (/dev/kfd missing)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def manifest():
    import json
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        m = json.load(f)
    return {c["name"]: c for c in m["cases"]}, m


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.lib()
    return orc

"""The JavaScript host (webgpu-fft_amd/js: Node ESM over the N-API addon) — run under the system Node.
CPU tier: host logic + JS oracle twin vs the golden fixtures.  GPU tier: the reference-style parity suite."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

NODE = shutil.which("node")
ADDON = os.path.join(ROOT, "webgpu-fft_amd", "lib", "mi355fft.node")
needs_node = pytest.mark.skipif(NODE is None, reason="node is not installed on this machine")


def _run(script, timeout):
    if not os.path.exists(ADDON):
        pytest.skip("N-API addon not built (run __graft_entry__.build())")
    p = subprocess.run([NODE, os.path.join(ROOT, "webgpu-fft_amd", "js", "test", script)], cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    print(p.stdout)
    print(p.stderr)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]


@needs_node
def test_js_host_logic():
    _run("host_logic.test.mjs", 300)


@needs_node
@pytest.mark.gpu
def test_js_gpu_parity():
    _run("gpu_parity.test.mjs", 600)

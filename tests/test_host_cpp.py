"""The C++ host mirror (host/eppk_host.hpp): EndpointPicker / RoundRobinPicker / GpuPicker micro-batcher.
CPU mode uses a fake backend (no GPU); GPU mode drives the real C ABI through concurrent Pick() calls."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "gateway-api-inference-extension_amd", "host", "test_host")


def _exe():
    if not os.path.exists(EXE):
        import __graft_entry__ as g
        g.build()
    return EXE


def test_host_mirror_cpu():
    out = subprocess.run([_exe(), "cpu"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "host cpu ok" in out.stdout


@pytest.mark.gpu
def test_host_mirror_gpu():
    out = subprocess.run([_exe(), "gpu"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    assert "host gpu ok" in out.stdout

"""GpuPicker (host/eppk_host.hpp) over the real backend against the ORACLE: concurrent picks on a frozen index, the index learning on
the device through the pipelined staging sets, slot reuse under churn (tests/cpp/test_gpupicker.cpp; links libeppk AND liboracle:
test infrastructure)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_gpupicker.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "test_gpupicker")
PKG = os.path.join(ROOT, "gateway-api-inference-extension_amd")


def _build():
    import __graft_entry__ as g
    g.build()
    deps = [SRC, os.path.join(PKG, "host", "eppk_host.hpp"), os.path.join(ROOT, "include", "eppk.h"), os.path.join(ROOT, "oracle", "oracle.h")]
    if not g._newer(EXE, deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-pthread", SRC, "-o", EXE, f"-L{PKG}", "-leppk", f"-L{os.path.join(ROOT, 'oracle')}", "-loracle",
                        f"-Wl,-rpath,{PKG}", f"-Wl,-rpath,{os.path.join(ROOT, 'oracle')}"], check=True)
        g._stamp(EXE, deps)
    return EXE


def test_gpupicker_test_compiles():
    _build()


@pytest.mark.gpu
def test_gpupicker_against_the_oracle():
    out = subprocess.run([_build()], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    assert "gpupicker 1 ok" in out.stdout and "gpupicker 2 ok" in out.stdout and "gpupicker 3 ok" in out.stdout
    assert "gpupicker 4 ok (2 members)" in out.stdout and "gpupicker 4 ok (4 members)" in out.stdout

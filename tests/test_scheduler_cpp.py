"""The C++ scheduling cycle (host/eppk_host.hpp: Scheduler / ProfileHandler / SchedulingResult) run in the shape of the reference's
examples/example.yaml (prefill + decode profiles, best-score and random-top-3 pickers) end to end on the GPU against the oracle:
tests/cpp/test_scheduler.cpp (links libeppk AND liboracle: test infrastructure)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_scheduler.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "test_scheduler")
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


def test_scheduler_test_compiles():
    _build()


@pytest.mark.gpu
def test_two_profile_scheduling_cycle_against_the_oracle():
    out = subprocess.run([_build()], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    assert "scheduler ok" in out.stdout

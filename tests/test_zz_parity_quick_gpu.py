"""tests/cpp/parity_quick.hip: open-loop and closed-loop parity of the C5 workload against the oracle through the device entry points, in
a Python-free binary that runs in seconds -- the check to run after every kernel edit (NEXT.md).  Needs a GPU.
(The file name sorts last on purpose: the harness is younger than the suite it abbreviates.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gateway-api-inference-extension_amd")
SRC = os.path.join(ROOT, "tests", "cpp", "parity_quick.hip")
EXE = os.path.join(ROOT, "tests", "cpp", "parity_quick")


def build():
    import __graft_entry__ as g
    g.build()
    deps = [SRC, os.path.join(ROOT, "include", "eppk.h"), os.path.join(ROOT, "oracle", "oracle.h")]
    if not g._newer(EXE, deps):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", SRC, "-o", EXE, f"-L{PKG}", "-leppk", f"-L{os.path.join(ROOT, 'oracle')}", "-loracle",
                        f"-Wl,-rpath,{PKG}", f"-Wl,-rpath,{os.path.join(ROOT, 'oracle')}", "-pthread"], check=True)
        g._stamp(EXE, deps)
    return EXE


def test_parity_quick_compiles():
    build()


def test_parity_quick_selftest(tmp_path):
    """The harness itself, on the CPU: the library and HIP replaced by stand-ins over a second oracle instance (-DPARITY_QUICK_SELFTEST) --
    file parsing, fresh tails per generation, epoch ticks, a real eviction and the index-size bookkeeping run end to end on C3."""
    import __graft_entry__ as g
    g.load_oracle().build()
    exe = str(tmp_path / "parity_quick_selftest")
    subprocess.run(["g++", "-O2", "-std=c++17", "-x", "c++", "-DPARITY_QUICK_SELFTEST", SRC, "-o", exe, f"-L{os.path.join(ROOT, 'oracle')}", "-loracle",
                    f"-Wl,-rpath,{os.path.join(ROOT, 'oracle')}", "-pthread"], check=True)
    subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "dump_workload.py"), "--config", "3", "--out", str(tmp_path / "c3")], check=True, timeout=300)
    out = subprocess.run([exe, str(tmp_path / "c3"), "4"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "parity_quick ok" in out.stdout and "closed loop, generation 3:" in out.stdout
    evicted = [ln for ln in out.stdout.splitlines() if "evicted" in ln]
    assert evicted and int(evicted[-1].split()[1]) > 0, out.stdout          # the eviction really dropped something


@pytest.mark.gpu
def test_parity_quick(tmp_path):
    exe = build()
    subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "dump_workload.py"), "--config", "5", "--out", str(tmp_path)], check=True, timeout=300)
    out = subprocess.run([exe, str(tmp_path), "4"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "parity_quick ok" in out.stdout

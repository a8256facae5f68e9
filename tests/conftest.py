import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "timeout: per-test time limit (pytest-timeout; tests that drive the resident kernel)")


def _have_gpu() -> bool:
    """True iff a HIP device is usable (the kernel driver node exists and the runtime sees a device)."""
    if not os.path.exists("/dev/kfd"):
        return False
    try:
        import torch
        return bool(torch.cuda.is_available())
    except Exception:
        return True     # no torch: let libeppk's own eppk_create decide (fails loudly with EPPK_ERR_DEVICE)


def pytest_collection_modifyitems(config, items):
    # `pytest -m gpu` on the GPU box runs them; a plain `pytest` on a host without a GPU skips them instead of failing with
    # EPPK_ERR_DEVICE (libeppk has no CPU path, so there is nothing for them to run on)
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device on this host (libeppk has no CPU path)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# ---- library modes --------------------------------------------------------------------------------------------------------
# libeppk reads three switches when a context is created: EPPK_QUAD_MIN (smallest batch that takes pick_quad_kernel; 4 = every
# eligible batch), EPPK_QUAD=0 (never: every request through pick_fast_kernel), EPPK_LISTS=0 (the pick kernels' list routes off:
# every request takes the dense route, which expands listed sets through LDS).  The parity-critical GPU modules run once per mode,
# so that ONE `pytest -m gpu` run -- the driver's -- exercises every route against the oracle, not only the default one.
# (tests/test_gpu_quad.py sets its own switches; the full-size closed-loop / config tests run in the default mode only.)
LIBRARY_MODES = {"default": {}, "quadmin4": {"EPPK_QUAD_MIN": "4"}, "quad0": {"EPPK_QUAD": "0"}, "lists0": {"EPPK_LISTS": "0"}}
MODE_MODULES = {"test_gpu_parity", "test_gpu_fuzz", "test_gpu_holes", "test_gpu_pickers", "test_gpu_subset", "test_gpu_staging",
                "test_gpu_device_rows", "test_gpu_resident"}


def pytest_generate_tests(metafunc):
    if metafunc.module.__name__.split(".")[-1] in MODE_MODULES and "eppk_mode" in metafunc.fixturenames:
        metafunc.parametrize("eppk_mode", list(LIBRARY_MODES), indirect=True)


@pytest.fixture(autouse=True)
def eppk_mode(request, monkeypatch):
    mode = getattr(request, "param", "default")
    for k, v in LIBRARY_MODES[mode].items():
        monkeypatch.setenv(k, v)
    return mode


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as g
    # build() is mtime-aware (seconds when nothing changed): always call it, so that a stale prebuilt libeppk.so is never what
    # gets tested after a source edit.  On a box without hipcc (none known) the prebuilt library is used as it is.
    if os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        g.build()
    elif not os.path.exists(os.path.join(g.PKG_DIR, "libeppk.so")):
        raise RuntimeError("libeppk.so is missing and there is no hipcc to build it")
    return g.load_package()


@pytest.fixture(scope="session")
def orc():
    import __graft_entry__ as g
    return g.load_oracle()

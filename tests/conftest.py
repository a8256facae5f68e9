import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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

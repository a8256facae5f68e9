"""The experiment tools under scripts/ keep working against the current sources (CPU only, nothing is compiled or run on a GPU):
 * scripts/dump_workload.py: the raw files scripts/micro/pickbench reads agree with the workload generator."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dump_workload_round_trips(tmp_path, pkg):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "dump_workload.py"), "--config", "3", "--out", str(tmp_path)],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    wl = pkg.workload.make_workload(3)
    meta = [int(x) for x in open(tmp_path / "meta.txt").read().split()]
    assert meta[:6] == [wl.R, wl.P, wl.B, len(wl.index_hashes), wl.index_slots, len(wl.chain)]
    assert meta[6:] == [int(v) for kw in wl.chain for v in kw]
    assert np.array_equal(np.fromfile(tmp_path / "reqs.bin", dtype="<u8").reshape(wl.R, 1 + wl.B), wl.reqs)
    assert np.fromfile(tmp_path / "pods.bin", dtype=np.uint8).size == wl.P * 64
    assert np.array_equal(np.fromfile(tmp_path / "index_hashes.bin", dtype="<u8"), wl.index_hashes)
    assert np.array_equal(np.fromfile(tmp_path / "index_pods.bin", dtype="<u4"), wl.index_pods)

"""The experiment tools under scripts/ keep working against the current sources (CPU only, nothing is compiled or run on a GPU):
 * scripts/micro/gen_insert_variants.py: every text edit it applies to csrc/eppk_kernels.hip.h still matches exactly once (the variants
   are how maintenance-kernel changes are measured standalone before they touch the product: NEXT.md);
 * scripts/dump_workload.py: the raw files scripts/micro/pickbench reads agree with the workload generator."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_insert_variant_generator_applies_to_the_current_header():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "micro", "gen_insert_variants.py")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    gen = os.path.join(ROOT, "scripts", "micro", "_gen")
    names = sorted(ln.split()[1] for ln in out.stdout.splitlines() if ln.startswith("wrote"))
    assert {"v1", "v2", "f1", "f2", "e1"} <= set(names)
    for n in names:
        text = open(os.path.join(gen, f"eppk_kernels_{n}.hip.h")).read()
        assert f"namespace eppk_{n} {{" in text and "#pragma once" not in text
    src = open(os.path.join(ROOT, "gateway-api-inference-extension_amd", "csrc", "eppk_kernels.hip.h")).read()
    f2 = open(os.path.join(gen, "eppk_kernels_f2.hip.h")).read()
    assert "s_cnt" in f2 and "s_cnt" not in src                    # the product header is read, never written


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

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/r3topk
rm -rf $OUT; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_staging.py tests/test_gpu_pickers.py tests/test_gpupicker_cpp.py "tests/test_gpu_parity.py::test_topk_fallbacks" -m gpu -q -x -k "default or not (quadmin4 or quad0 or lists0)" 2>&1 | tail -3 | tee $OUT/pytest_sel.txt
python - <<'P' | tee $OUT/topk_latency.txt
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import __graft_entry__ as graft
pkg = graft.load_package()
wl = pkg.workload.make_workload(5)
for zc in ("0", None):
    if zc is None: os.environ.pop("EPPK_ZERO_COPY_MAX", None)
    else: os.environ["EPPK_ZERO_COPY_MAX"] = zc
    pk = pkg.BatchedPicker(wl.chain, max_pods=wl.P, max_blocks=wl.B, max_batch=wl.R, index_slots=wl.index_slots)
    pk.publish(wl.pods); pk.index_insert(wl.index_hashes, wl.index_pods)
    for n in (16, 128, 2048):
        rows = wl.reqs[:n].copy()
        for _ in range(20): pk.pick_topk(rows, 4)
        lat = []
        for i in range(300):
            t0 = time.perf_counter(); pk.pick_topk(rows, 4); lat.append(time.perf_counter() - t0)
        lat = np.asarray(lat) * 1e6
        print(f"EPPK_ZERO_COPY_MAX={zc}: eppk_pick_topk k=4 n={n:5d}: p50 {np.percentile(lat, 50):6.1f} us  p99 {np.percentile(lat, 99):6.1f} us", flush=True)
    pk.close()
P

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/r3zc
mkdir -p $OUT
for zc in 1024 2048 4096 1024 2048 4096; do echo "== EPPK_ZERO_COPY_MAX=$zc"; EPPK_ZERO_COPY_MAX=$zc timeout 120 python scripts/gpu_small_batch_latency.py 2>&1 | grep "n=" | sed -n 4,7p; done | tee $OUT/latency_zc_threshold2.txt

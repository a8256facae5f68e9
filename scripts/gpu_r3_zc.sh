#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/r3zc
mkdir -p $OUT
for zc in 0 1000000 0 1000000; do echo "== EPPK_ZERO_COPY_MAX=$zc"; EPPK_ZERO_COPY_MAX=$zc timeout 120 python scripts/gpu_small_batch_latency.py 2>&1 | grep "n="; done | tee $OUT/latency_fresh_rows.txt

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/r3zc
mkdir -p $OUT
for intr in 1 0 1 0; do echo "== HSA_ENABLE_INTERRUPT=$intr"; HSA_ENABLE_INTERRUPT=$intr timeout 120 python scripts/gpu_small_batch_latency.py 2>&1 | grep "n=" | head -6; done | tee $OUT/latency_interrupt.txt

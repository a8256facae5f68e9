#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r3f
for rep in 1 2; do for b in ab/insertbreak_*; do echo "=== $b"; timeout 60 $b 2>&1 | grep -E "^new|^step|^evict"; done; done | tee gpurun_out/r3f/variants.txt

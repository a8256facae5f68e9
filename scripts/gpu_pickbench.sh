#!/bin/bash
# A/B of library variants in seconds of GPU budget (no Python, no torch): every ab/<name>/libeppk.so that scripts/abq.sh (or ab.sh)
# built, next to the tree's own library, through scripts/micro/pickbench.  Before the gpurun call, on the build box:
#   python scripts/dump_workload.py                      # scripts/micro/_gen/c5/ (17 MB, git-ignored, travels with gpurun)
#   hipcc --offload-arch=gfx950 -O2 -std=c++17 -o scripts/micro/pickbench scripts/micro/pickbench.hip -ldl
# then:  gpurun --timeout 120 -- 'bash scripts/gpu_pickbench.sh [pickbench flags, e.g. --inflight 1 | --closed-loop | --profile]'
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
LIBS="gateway-api-inference-extension_amd/libeppk.so $(ls ab/*/libeppk.so 2>/dev/null)"
timeout 100 ./scripts/micro/pickbench scripts/micro/_gen/c5 $LIBS "$@" 2>&1 | tee gpurun_out/pickbench.txt

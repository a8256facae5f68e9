#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/r3zc
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_staging.py tests/test_gpupicker_cpp.py tests/test_host_cpp.py tests/test_scheduler_cpp.py -m gpu -q -x 2>&1 | tail -5 | tee $OUT/pytest_sel2.txt

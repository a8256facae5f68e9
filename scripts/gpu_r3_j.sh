#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/r3j
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpupicker_cpp.py tests/test_host_cpp.py tests/test_scheduler_cpp.py tests/test_gpu_staging.py -m gpu -q -x > $OUT/pytest_sel.txt 2>&1; tail -12 $OUT/pytest_sel.txt

#!/bin/bash
# after a late change of the hashed sources: a test subset, the stamps, the round's bench lines
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r3end
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_staging.py tests/test_zz_parity_quick_gpu.py tests/test_host_cpp.py -m gpu -q -k "default or not (quadmin4 or quad0 or lists0)" > gpurun_out/r3end/pytest_subset.txt 2>&1; tail -2 gpurun_out/r3end/pytest_subset.txt
bash scripts/gpu_stamps_r3.sh 2>&1 | grep -E "^\[" | tail -4
FINAL_DIR=r3end_bench bash scripts/gpu_r3_final_bench.sh 2>&1 | tail -22

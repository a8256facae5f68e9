#!/bin/bash
# end of round 3 (second half): the whole GPU suite as the driver runs it, the stamps, the round's bench lines
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r3end
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r3end/pytest_gpu.txt 2>&1; tail -4 gpurun_out/r3end/pytest_gpu.txt
bash scripts/gpu_stamps_r3.sh 2>&1 | grep -E "^\[|M\s" | tail -8
FINAL_DIR=r3end_bench bash scripts/gpu_r3_final_bench.sh 2>&1 | tail -24

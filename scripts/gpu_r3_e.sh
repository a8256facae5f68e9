#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r3e
export EPPK_SELFCHECK_VERBOSE=1
timeout 600 python -m pytest "tests/test_gpu_group.py::test_learn_applies_the_gathered_update_on_every_member" -m gpu -q -x -s > gpurun_out/r3e/out.txt 2>&1
grep "selfcheck\]" gpurun_out/r3e/out.txt | head -20

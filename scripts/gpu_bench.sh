#!/bin/bash
# Run on the GPU box (via gpurun): bench + rocprofv3 kernel trace summary.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
python bench.py "$@" > gpurun_out/bench.json 2> gpurun_out/bench.err
tail -3 gpurun_out/bench.err
cat gpurun_out/bench.json

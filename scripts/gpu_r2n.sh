#!/bin/bash
# subset filter on the device: GPU suite + route timings (incl. the subset leg)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r2n
rm -rf $OUT; mkdir -p $OUT
T0=$(date +%s)
lap() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 600 python -m pytest tests/test_gpu_subset.py -q 2>&1 | tail -30 | tee $OUT/pytest_subset.txt; lap subset
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 | tee $OUT/pytest.txt; lap pytest
timeout 300 python scripts/gpu_route_times.py > $OUT/route_times.json 2> $OUT/route_times.err; tail -5 $OUT/route_times.err; cat $OUT/route_times.json; lap routes

#!/bin/bash
# Round 2, quick GPU session: GPU suite + route timings.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r2c
rm -rf $OUT; mkdir -p $OUT
T0=$(date +%s)
lap() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee $OUT/pytest.txt; lap pytest
timeout 300 python scripts/gpu_route_times.py > $OUT/route_times.json 2> $OUT/route_times.err; tail -2 $OUT/route_times.err; cat $OUT/route_times.json; lap routes
ROUTE_R=8192 timeout 300 python scripts/gpu_route_times.py > $OUT/route_times_8k.json 2>/dev/null; cat $OUT/route_times_8k.json; lap routes8k
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --host-path 0 --p99-samples 0 --no-cold-ref > $OUT/bench_c5.json 2>/dev/null; cut -c1-200 $OUT/bench_c5.json; lap bench

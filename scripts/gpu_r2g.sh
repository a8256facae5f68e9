#!/bin/bash
# Round 2: validation of pickers / assumed load / scheduler + a regular bench run.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r2g
rm -rf $OUT; mkdir -p $OUT
T0=$(date +%s)
lap() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee $OUT/pytest.txt; lap pytest
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --host-path 0 --no-cold-ref > $OUT/bench_c5.json 2>/dev/null; cut -c1-200 $OUT/bench_c5.json; lap bench
timeout 300 python bench.py --steps 200 --warmup 20 --inflight 1 --no-cpu-baseline --host-path 0 --no-cold-ref > $OUT/bench_c5_inflight1.json 2>/dev/null; cut -c1-200 $OUT/bench_c5_inflight1.json; lap bench1

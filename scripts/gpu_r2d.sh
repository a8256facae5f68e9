#!/bin/bash
# Round 2: lean kernel validation -- GPU suite with the lean kernel (default) and without (EPPK_LEAN=0: every request through
# pick_fast_kernel, as before), bench in both modes, cold reference, closed loop.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r2d
rm -rf $OUT; mkdir -p $OUT
T0=$(date +%s)
lap() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee $OUT/pytest_lean.txt; lap pytest-lean
EPPK_LEAN=0 timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $OUT/pytest_nolean.txt; lap pytest-nolean
for inf in 2 1; do
timeout 300 python bench.py --steps 200 --warmup 20 --inflight $inf --no-cpu-baseline --host-path 0 --no-cold-ref > $OUT/bench_lean_inflight$inf.json 2>/dev/null; cut -c1-200 $OUT/bench_lean_inflight$inf.json
EPPK_LEAN=0 timeout 300 python bench.py --steps 200 --warmup 20 --inflight $inf --no-cpu-baseline --host-path 0 --no-cold-ref > $OUT/bench_nolean_inflight$inf.json 2>/dev/null; cut -c1-200 $OUT/bench_nolean_inflight$inf.json
done; lap bench
timeout 300 python bench.py --steps 100 --warmup 10 --inflight 1 --groups 65536 --zipf 0 --batches 4 --no-cpu-baseline --host-path 0 --no-cold-ref --p99-samples 0 > $OUT/bench_cold_lean.json 2>/dev/null; cut -c1-250 $OUT/bench_cold_lean.json
EPPK_LEAN=0 timeout 300 python bench.py --steps 100 --warmup 10 --inflight 1 --groups 65536 --zipf 0 --batches 4 --no-cpu-baseline --host-path 0 --no-cold-ref --p99-samples 0 > $OUT/bench_cold_nolean.json 2>/dev/null; cut -c1-250 $OUT/bench_cold_nolean.json; lap cold
timeout 400 python bench.py --closed-loop --steps 60 --warmup 10 > $OUT/bench_closed_loop.json 2>/dev/null; cut -c1-250 $OUT/bench_closed_loop.json; lap closed

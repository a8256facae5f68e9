#!/bin/bash
# Round 3: "lists first" index (a set with at most 24 members lives only in its list).  Parity first, then the closed loop and the
# standalone harness.  Output -> gpurun_out/r3c/.
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/r3c
rm -rf $OUT; mkdir -p $OUT
T0=$(date +%s)
lap() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 300 python -m pytest tests/test_zz_parity_quick_gpu.py -m gpu -x -q > $OUT/parity_quick.txt 2>&1; tail -3 $OUT/parity_quick.txt; lap parity-quick
timeout 900 python -m pytest tests/test_gpu_closed_loop.py tests/test_gpu_fuzz.py tests/test_gpu_holes.py -m gpu -x -q > $OUT/pytest_index.txt 2>&1; tail -3 $OUT/pytest_index.txt; lap index-tests
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_all.txt 2>&1; tail -15 $OUT/pytest_all.txt; lap all-tests
timeout 300 python bench.py --closed-loop --no-cpu-baseline > $OUT/bench_closed_loop.json 2> $OUT/bench_closed_loop.err; cut -c1-400 $OUT/bench_closed_loop.json; tail -3 $OUT/bench_closed_loop.err; lap closed-loop
[ -x scripts/micro/insertbreak ] && timeout 60 ./scripts/micro/insertbreak > $OUT/insertbreak.txt 2>&1; cat $OUT/insertbreak.txt; lap harness
timeout 300 python bench.py --no-cpu-baseline --no-cold-ref --host-path 0 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; cut -c1-300 $OUT/bench_c5.json; lap bench

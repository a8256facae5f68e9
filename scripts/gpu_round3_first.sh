#!/bin/bash
# First GPU call of round 3, after `git merge r3/insert-counters` (NEXT.md): does the index with per-workgroup counters still behave,
# and what did the closed loop gain?  ~3 GPU-minutes.  Output -> gpurun_out/r3first/.
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/r3first
rm -rf $OUT; mkdir -p $OUT
T0=$(date +%s)
lap() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
# 0. seconds: open-loop and closed-loop parity of C5 against the oracle in a Python-free binary
timeout 300 python -m pytest tests/test_zz_parity_quick_gpu.py -m gpu -x -q > $OUT/parity_quick.txt 2>&1; tail -3 $OUT/parity_quick.txt; lap parity-quick
# 1. the index maintenance and closed-loop tests first: they are what the change can break
timeout 900 python -m pytest tests/test_gpu_closed_loop.py tests/test_gpu_fuzz.py tests/test_gpu_holes.py -m gpu -x -q > $OUT/pytest_index.txt 2>&1; tail -3 $OUT/pytest_index.txt; lap index-tests
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "index or insert or evict or trim or capacity or full" > $OUT/pytest_parity_index.txt 2>&1; tail -2 $OUT/pytest_parity_index.txt; lap parity-index
# 2. the closed loop and the standalone harness (library row = the merged kernel now)
timeout 300 python bench.py --closed-loop --no-cpu-baseline > $OUT/bench_closed_loop.json 2> $OUT/bench_closed_loop.err; cut -c1-400 $OUT/bench_closed_loop.json; lap closed-loop
lap harness-skipped
# 3. everything else
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_all.txt 2>&1; tail -3 $OUT/pytest_all.txt; lap all-tests
timeout 300 python bench.py > $OUT/bench_c5.json 2> $OUT/bench_c5.err; cut -c1-300 $OUT/bench_c5.json; lap bench

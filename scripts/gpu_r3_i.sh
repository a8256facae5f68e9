#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/r3i
rm -rf $OUT; mkdir -p $OUT
T0=$(date +%s)
lap() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 900 python -m pytest tests/test_gpu_staging.py tests/test_gpu_quad.py tests/test_gpu_subset.py -m gpu -q -x > $OUT/pytest_sel.txt 2>&1; tail -8 $OUT/pytest_sel.txt; lap tests
timeout 300 python bench.py --no-cold-ref > $OUT/bench_c5.json 2> $OUT/bench_c5.err; python -c "
import json;d=json.load(open('$OUT/bench_c5.json'));print(d['value'], d['ms_per_step']);print(json.dumps(d['host_path'],indent=0)); print(d['cpu_baseline']['value'])"; tail -3 $OUT/bench_c5.err; lap bench

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/r3m
rm -rf $OUT; mkdir -p $OUT
T0=$(date +%s)
lap() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 900 python -m pytest tests/test_gpu_closed_loop.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x > $OUT/pytest_sel.txt 2>&1; tail -8 $OUT/pytest_sel.txt; lap tests
timeout 300 python bench.py --closed-loop --no-cpu-baseline > $OUT/bench_closed_loop.json 2> $OUT/bench_closed_loop.err; python -c "
import json;d=json.loads(open('$OUT/bench_closed_loop.json').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step']);print(json.dumps(d['roofline_closed_loop']['step_parts_ms']))"; lap closed-loop
timeout 60 ./scripts/micro/insertbreak 2>&1 | tail -7; lap harness

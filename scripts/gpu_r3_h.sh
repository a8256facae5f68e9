#!/bin/bash
# whole GPU suite with the library modes parametrised + closed-loop JSON with its roofline object
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/r3h
rm -rf $OUT; mkdir -p $OUT
T0=$(date +%s)
lap() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 2400 python -m pytest tests -m gpu -q -x > $OUT/pytest_all.txt 2>&1; tail -15 $OUT/pytest_all.txt; lap all-tests
timeout 300 python bench.py --closed-loop --no-cpu-baseline > $OUT/bench_closed_loop.json 2> $OUT/bench_closed_loop.err; python -c "
import json;d=json.load(open('$OUT/bench_closed_loop.json'));print(d['value'], d['ms_per_step']);print(json.dumps(d['roofline_closed_loop'])[:1500])"; tail -3 $OUT/bench_closed_loop.err; lap closed-loop

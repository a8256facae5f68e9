#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/r3k
rm -rf $OUT; mkdir -p $OUT
T0=$(date +%s)
lap() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 900 python -m pytest tests/test_gpu_group.py -m gpu -q -x > $OUT/pytest_group.txt 2>&1; tail -6 $OUT/pytest_group.txt; lap tests
for m in 1 2 4 8; do timeout 300 python bench.py --group $m > $OUT/bench_group_$m.json 2> $OUT/bench_group_$m.err; python -c "
import json;d=json.load(open('$OUT/bench_group_$m.json'));print($m, d['value'], d['ms_per_step'], d['completion_latency']['p50_ms'], d['parity'], d['config']['ranks_seen'])" || tail -5 $OUT/bench_group_$m.err; done; lap group-bench
timeout 300 python bench.py --force-dist --steps 64 --warmup 16 --no-cpu-baseline --no-cold-ref --host-path 0 > $OUT/bench_forcedist.json 2> $OUT/bench_forcedist.err; python -c "
import json;d=json.load(open('$OUT/bench_forcedist.json'));print(d['value'], d['ms_per_step'], d.get('completion_latency'))" || tail -5 $OUT/bench_forcedist.err; lap forcedist

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3o
rm -rf $OUT; mkdir -p $OUT/prof
T0=$(date +%s)
lap() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
last() { python -c "
import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'], d['config'].get('quad_route'))"; }
timeout 900 python -m pytest tests/test_gpu_quad.py tests/test_zz_parity_quick_gpu.py tests/test_gpu_closed_loop.py -m gpu -q -x > $OUT/pytest_quad.txt 2>&1; tail -5 $OUT/pytest_quad.txt; lap quad-tests
timeout 300 python bench.py --no-cpu-baseline --no-cold-ref --host-path 0 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; last $OUT/bench_c5.json; lap bench
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cold-ref --host-path 0 > $OUT/bench_c5_20.json 2> $OUT/bench_c5_20.err; last $OUT/bench_c5_20.json; lap bench20
EPPK_QUAD_TAIL=0 timeout 300 python bench.py --no-cpu-baseline --no-cold-ref --host-path 0 > $OUT/bench_c5_notail.json 2>/dev/null; last $OUT/bench_c5_notail.json; lap bench-notail
timeout 300 python bench.py --inflight 1 --no-cpu-baseline --no-cold-ref --host-path 0 > $OUT/bench_c5_inflight1.json 2>/dev/null; last $OUT/bench_c5_inflight1.json; lap inflight1
timeout 300 python bench.py --closed-loop --no-cpu-baseline > $OUT/bench_closed_loop.json 2>/dev/null; last $OUT/bench_closed_loop.json; lap closed-loop
( cd /tmp; timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline --host-path 0 --no-cold-ref --p99-samples 0 > $OUT/prof/bench_under_rocprof.json 2> $OUT/prof/prof.err )
f=$(find $OUT/prof -name "trace_kernel_stats.csv" | head -1); python - "$f" <<'P'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:4]:
    print(r['Name'][:90].ljust(90), r['Calls'].rjust(5), f"{float(r['AverageNs'])/1e3:9.1f} us", r['Percentage'])
P
rm -f $(find $OUT -name "*agent_info.csv") $(find $OUT -name "*kernel_trace.csv"); lap stats

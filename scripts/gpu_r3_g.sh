#!/bin/bash
# closed loop under rocprofv3: per-kernel times of a step
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3g
rm -rf $OUT; mkdir -p $OUT/prof
( cd /tmp; timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --closed-loop --steps 60 --warmup 10 --no-cpu-baseline --cl-verify 0 > $OUT/prof/bench_under_rocprof.json 2> $OUT/prof/prof.err )
f=$(find $OUT/prof -name "trace_kernel_stats.csv" | head -1); python - "$f" <<'P'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(5), f"{float(r['AverageNs'])/1e3:9.1f} us", r['Percentage'])
P
rm -f $(find $OUT -name "*agent_info.csv") $(find $OUT -name "*kernel_trace.csv")

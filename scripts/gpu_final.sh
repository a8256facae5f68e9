#!/bin/bash
# GPU box: final numbers of the round -- default bench (two batches in flight), the single-stream variant, rocprofv3 kernel stats
# of the default command.  Output in gpurun_out/final/.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/final
rm -rf $OUT; mkdir -p $OUT/prof
python bench.py --alone-ref > $OUT/bench_default_with_alone_ref.json 2>/dev/null
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 1800 $OUT/bench_default.json; echo
python bench.py --inflight 1 --no-cpu-baseline --host-path 0 > $OUT/bench_inflight1.json 2>/dev/null
( cd /tmp; timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 6 --no-cpu-baseline --host-path 0 > $OUT/prof/bench_under_rocprof.json 2> $OUT/prof/prof.err )
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); head -3 "$f" | cut -c1-80,330-420
rm -f $OUT/prof/*agent_info.csv

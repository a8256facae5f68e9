#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats of the bench command.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof
cd /tmp
timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_under_prof.json 2> $OUT/prof.err
tail -2 $OUT/prof.err
ls -R $OUT | head -20
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
echo "== $f"; head -12 "$f"

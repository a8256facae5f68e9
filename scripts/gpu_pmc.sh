#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 PMC passes (counters only, with --kernel-trace) of a short bench run.
# usage: gpu_pmc.sh "<bench args>" "<counters pass 1>" "<counters pass 2>" ...
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc
rm -rf $OUT/*
BARGS="$1"; shift
cd /tmp
i=0
for ctrs in "$@"; do
  i=$((i+1))
  timeout -k 5 90 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT -o pass$i -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline $BARGS > $OUT/bench_pass$i.json 2> $OUT/pass$i.err
  tail -1 $OUT/pass$i.err
done
python $GRAFT_REPO_ROOT/scripts/pmc_summary.py $OUT pick_fast_kernel | tee $OUT/summary.csv

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/r3z2
rm -rf $OUT; mkdir -p $OUT
for extra in "" "--gather-every 8" "--steps 200 --warmup 20"; do
  for rep in 1 2; do
  echo -n "== $extra: "
  EPPK_BENCH_HOSTTIME=1 timeout 300 python bench.py --force-dist --scaling weak --steps 20 --warmup 5 --no-cpu-baseline --no-cold-ref --host-path 0 --p99-samples 0 $extra 2>&1 >/dev/null | grep "host time"
  done
done | tee $OUT/hosttime.txt

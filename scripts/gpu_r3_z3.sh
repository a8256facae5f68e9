#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/r3z3
rm -rf $OUT; mkdir -p $OUT
line() { python -c "
import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);s=d.get('strong') or {};cl=d.get('completion_latency') or {}
print('%s value %.3f G  %.2f us/step | strong %.3f G %.2f us/step | bucket latency p50 %s' % (d['scaling'], d['value']/1e9, d['ms_per_step']*1e3, (s.get('value') or 0)/1e9, (s.get('ms_per_step') or 0)*1e3, cl.get('p50_ms')))"; }
for rep in 1 2; do
  for intr in 1 0; do
    HSA_ENABLE_INTERRUPT=$intr timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cold-ref --host-path 0 --p99-samples 0 > $OUT/single_$intr.json 2>/dev/null; echo -n "N=1 K=20 HSA_ENABLE_INTERRUPT=$intr: "; line $OUT/single_$intr.json
    EPPK_BENCH_HOSTTIME=1 HSA_ENABLE_INTERRUPT=$intr timeout 300 python bench.py --force-dist --scaling weak --steps 20 --warmup 5 --no-cpu-baseline --no-cold-ref --host-path 0 --p99-samples 0 2>&1 >$OUT/weak_$intr.json | grep "host time"; echo -n "weak K=20 HSA_ENABLE_INTERRUPT=$intr: "; line $OUT/weak_$intr.json
  done
done 2>&1 | tee $OUT/lines.txt

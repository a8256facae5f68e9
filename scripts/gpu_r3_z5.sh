#!/bin/bash
# N > 1 path on one GPU after the restructure (weak headline, per-bucket events, waits only at bucket ends, barrier outside the clock, no cast)
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/r3z5
rm -rf $OUT; mkdir -p $OUT
line() { python -c "
import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);s=d.get('strong') or {};cl=d.get('completion_latency') or {}
print('%s value %.3f G  %.2f us/step | strong %.3f G %.2f us/step | bucket latency p50 %s' % (d['scaling'], d['value']/1e9, d['ms_per_step']*1e3, (s.get('value') or 0)/1e9, (s.get('ms_per_step') or 0)*1e3, cl.get('p50_ms')))"; }
for rep in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cold-ref --host-path 0 --p99-samples 0 > $OUT/single_short.json 2>/dev/null; echo -n "N=1 --steps 20 --warmup 5:            "; line $OUT/single_short.json
  EPPK_BENCH_HOSTTIME=1 timeout 300 python bench.py --force-dist --steps 20 --warmup 5 --no-cpu-baseline --no-cold-ref --host-path 0 --p99-samples 0 2>$OUT/fd.err > $OUT/fd_short.json; echo -n "--force-dist --steps 20 --warmup 5:   "; line $OUT/fd_short.json; grep "host time" $OUT/fd.err
done
timeout 300 python bench.py --force-dist --steps 400 --warmup 40 --no-cpu-baseline --no-cold-ref --host-path 0 --p99-samples 0 > $OUT/fd_long.json 2>/dev/null; echo -n "--force-dist --steps 400 --warmup 40: "; line $OUT/fd_long.json
timeout 300 python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-cold-ref --host-path 0 --p99-samples 0 > $OUT/single_long.json 2>/dev/null; echo -n "N=1 --steps 400 --warmup 40:          "; line $OUT/single_long.json

#!/bin/bash
# after the eviction scan change (4 chunks in flight) and the N > 1 bench restructure (weak headline, per-bucket events, int16 payload, barrier
# outside the clock): maintenance tests, the N > 1 path on one GPU at the driver's --steps 20, the N = 1 line at --steps 20, closed loop
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/r3z
rm -rf $OUT; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_closed_loop.py tests/test_gpu_fuzz.py tests/test_zz_parity_quick_gpu.py -m gpu -q -x 2>&1 | tail -4 | tee $OUT/pytest_sel.txt
line() { python -c "
import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);s=d.get('strong') or {};cl=d.get('completion_latency') or {}
print('%s value %.3f G  %.2f us/step | strong %.3f G %.2f us/step | bucket latency p50 %s' % (d['scaling'], d['value']/1e9, d['ms_per_step']*1e3, (s.get('value') or 0)/1e9, (s.get('ms_per_step') or 0)*1e3, cl.get('p50_ms')))"; }
for rep in 1 2; do
  timeout 300 python bench.py --force-dist --steps 20 --warmup 5 --no-cpu-baseline --no-cold-ref --host-path 0 --p99-samples 0 > $OUT/fd_short_$rep.json 2>$OUT/fd_short.err; echo -n "force-dist K=20 (#$rep): "; line $OUT/fd_short_$rep.json
done
timeout 300 python bench.py --force-dist --steps 400 --warmup 40 --no-cpu-baseline --no-cold-ref --host-path 0 --p99-samples 0 > $OUT/fd_long.json 2>>$OUT/fd_short.err; echo -n "force-dist K=400: "; line $OUT/fd_long.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cold-ref --host-path 0 > $OUT/single_short.json 2>>$OUT/fd_short.err; echo -n "N=1 K=20: "; line $OUT/single_short.json
timeout 200 python bench.py --closed-loop --no-cpu-baseline 2>/dev/null > $OUT/closed_loop.json; python -c "
import sys,json; d=json.loads(open('$OUT/closed_loop.json').read().strip().splitlines()[-1]); p=d['roofline_closed_loop']['step_parts_ms']
print('closed loop M/s=%.1f'%(d['value']/1e6), 'pick=%.1f update=%.1f ageing=%.1f us'%(p['pick']*1e3, p['index_update']*1e3, p['ageing_per_step']*1e3), d['closed_loop'].get('picks_equal_oracle'))"

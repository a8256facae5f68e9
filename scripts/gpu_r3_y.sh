#!/bin/bash
# (1) claimcost with the one-line layouts, (2) insertbreak with 1/4/8 eviction chunks in flight, (3) closed loop: ab/ variants of the main unit,
# (4) what a SHORT run (--steps 20 --warmup 5, the driver's) makes of the N > 1 code path: strong at the per-rank load of 8 GPUs, weak
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/r3y
rm -rf $OUT; mkdir -p $OUT
timeout 60 ./scripts/micro/_bin/claimcost > $OUT/claimcost.txt 2>&1; grep -E "^A0 |^C0|^C1|^A0l|^E0 |^E3" $OUT/claimcost.txt
for c in 1 4 8; do for rep in 1 2; do echo -n "insertbreak chunks=$c: "; timeout 60 ./scripts/micro/_bin/insertbreak_c$c 2>&1 | grep -E "^evict" ; done; done | tee $OUT/insertbreak_evict.txt
for round in 1 2; do
  for so in ab/libeppk_c1.so ab/libeppk_c8.so; do
    EPPK_LIB=$PWD/$so timeout 200 python bench.py --closed-loop --no-cpu-baseline --cl-verify 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['roofline_closed_loop']['step_parts_ms']
print('$so', 'M/s=%.1f'%(d['value']/1e6), 'pick=%.1f update=%.1f ageing=%.1f us'%(p['pick']*1e3, p['index_update']*1e3, p['ageing_per_step']*1e3))"
  done
done | tee $OUT/ab.txt
last() { python -c "
import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print('value %.3f G  %.2f us/step  launch_reqs %s  weak %s' % (d['value']/1e9, d['ms_per_step']*1e3, d['config'].get('requests_per_launch'), (d.get('weak') or {}).get('value')))"; }
for R in 8192 65536; do
  timeout 300 python bench.py --force-dist --requests $R --steps 20 --warmup 5 --no-cpu-baseline --no-cold-ref --host-path 0 --p99-samples 0 > $OUT/fd_short_$R.json 2>$OUT/fd_short_$R.err; echo -n "force-dist K=20 R=$R: "; last $OUT/fd_short_$R.json
done

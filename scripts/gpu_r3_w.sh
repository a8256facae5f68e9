#!/bin/bash
# A/B of main-unit variants (scripts/abm.sh) through the closed loop of bench.py, two rounds on one box
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/r3w
rm -rf $OUT; mkdir -p $OUT
for round in 1 2; do
  for so in ab/*.so; do
    EPPK_LIB=$PWD/$so timeout 200 python bench.py --closed-loop --no-cpu-baseline --cl-verify 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['roofline_closed_loop']['step_parts_ms']
print('$so', 'M/s=%.1f'%(d['value']/1e6), 'pick=%.1f update=%.1f ageing=%.1f us'%(p['pick']*1e3, p['index_update']*1e3, p['ageing_per_step']*1e3))"
  done
done | tee $OUT/ab.txt

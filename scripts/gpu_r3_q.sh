#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/r3q
rm -rf $OUT; mkdir -p $OUT
T0=$(date +%s)
lap() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
last() { python -c "
import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['config'].get('quad_route',{}).get('launches'))"; }
for cfg in 2 3 4; do for qm in default 4; do
  if [ $qm = default ]; then unset EPPK_QUAD_MIN; else export EPPK_QUAD_MIN=$qm; fi
  timeout 300 python bench.py --config $cfg --no-cpu-baseline --no-cold-ref --host-path 0 --p99-samples 0 > $OUT/b_c${cfg}_$qm.json 2>/dev/null; echo -n "config $cfg quad_min $qm: "; last $OUT/b_c${cfg}_$qm.json
done; done; unset EPPK_QUAD_MIN; lap configs
for R in 4096 8192 16384; do for qm in default 4; do
  if [ $qm = default ]; then unset EPPK_QUAD_MIN; else export EPPK_QUAD_MIN=$qm; fi
  timeout 300 python bench.py --requests $R --no-cpu-baseline --no-cold-ref --host-path 0 --p99-samples 0 > $OUT/b_r${R}_$qm.json 2>/dev/null; echo -n "C5 R=$R quad_min $qm: "; last $OUT/b_r${R}_$qm.json
done; done; unset EPPK_QUAD_MIN; lap sizes
timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest_all.txt 2>&1; tail -4 $OUT/pytest_all.txt; lap all-tests

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r2i
rm -rf $OUT; mkdir -p $OUT
T0=$(date +%s)
lap() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 | tee $OUT/pytest.txt; lap pytest
timeout 300 python scripts/gpu_route_times.py > $OUT/route_times.json 2> $OUT/route_times.err; tail -2 $OUT/route_times.err; cat $OUT/route_times.json; lap routes
timeout 400 python bench.py --steps 200 --warmup 20 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; tail -2 $OUT/bench_c5.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2i/bench_c5.json').read().strip().splitlines()[-1])
print('value %.4g ms/step %.4f'%(d['value'], d['ms_per_step']))
print({k:v for k,v in d.get('roofline_cold',{}).items() if k not in ('bytes_definition','workload')})
PY
lap bench
for inf in 3 4; do true; done

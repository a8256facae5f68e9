#!/bin/bash
# Round 2, session 3e: suite with the quad route on and off (incl. tests/test_gpu_quad.py), the full bench line.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3e
rm -rf $OUT; mkdir -p $OUT
T0=$(date +%s)
lap() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest.txt; lap pytest
EPPK_QUAD=0 timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_quad.py 2>&1 | tail -4 | tee $OUT/pytest_quad_off.txt; lap pytest-quad-off
timeout 500 python bench.py > $OUT/bench_c5.json 2> $OUT/bench_c5.err; tail -3 $OUT/bench_c5.err; python -c "
import json; d=json.load(open('$OUT/bench_c5.json'))
print('value %.1f M/s step %.4f ms'%(d['value']/1e6, d['ms_per_step'])); print({k:d['roofline'][k] for k in ('kernel','kernel_avg_ms','kernel_p99_ms','frac','l2_frac_of_gather_ceiling')})
print('cold', {k:d['roofline_cold'].get(k) for k in ('kernel','kernel_avg_ms','value','frac','frac_of_gather_ceiling')}); print(d.get('parity'), d['config'].get('quad_route')); print(d.get('cpu_baseline'))"; lap bench
timeout 300 python bench.py --inflight 1 --no-cold-ref --no-cpu-baseline --host-path 0 > $OUT/bench_c5_inflight1.json 2>/dev/null; lap inflight1

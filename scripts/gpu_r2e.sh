#!/bin/bash
# Round 2: does the measurement window matter?  (shader clocks under short bursts) -- bench with 200 / 5000 / 30000 steps, clock micro-benchmark.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r2e
rm -rf $OUT; mkdir -p $OUT
./scripts/micro/clockcal | tee $OUT/clockcal.txt
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk" | head -5 | tee -a $OUT/clockcal.txt
for st in 200 5000 30000; do
EPPK_LEAN=0 timeout 300 python bench.py --steps $st --warmup 20 --inflight 1 --no-cpu-baseline --host-path 0 --no-cold-ref --p99-samples 0 > $OUT/bench_nolean_steps$st.json 2>/dev/null; python -c "
import json,sys; d=json.loads(open('$OUT/bench_nolean_steps$st.json').read().strip().splitlines()[-1]); print('nolean steps',$st,'value %.4g ms/step %.4f kernel %.4f'%(d['value'],d['ms_per_step'],d['roofline']['kernel_avg_ms']))"
done
timeout 300 python bench.py --steps 30000 --warmup 20 --inflight 1 --no-cpu-baseline --host-path 0 --no-cold-ref --p99-samples 0 > $OUT/bench_lean_steps30000.json 2>/dev/null; python -c "
import json,sys; d=json.loads(open('$OUT/bench_lean_steps30000.json').read().strip().splitlines()[-1]); print('lean steps 30000 value %.4g ms/step %.4f kernel %.4f'%(d['value'],d['ms_per_step'],d['roofline']['kernel_avg_ms']))"
rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -3

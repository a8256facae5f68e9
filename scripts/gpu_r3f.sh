#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; OUT=$PWD/gpurun_out/r3f; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
B="--no-cold-ref --no-cpu-baseline --host-path 0 --p99-samples 0"
for i in 1 2; do for inf in 1 2; do timeout 300 python bench.py --inflight $inf $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('inflight=$inf', 'kernel_ms=%.4f'%d['roofline']['kernel_avg_ms'], 'step_ms=%.4f'%d['ms_per_step'], 'Mdec/s=%.1f'%(d['value']/1e6))"; done; done

#!/bin/bash
# Round 2, session 3j: the quad kernel scoring its own deferred requests (no second launch) vs the work-list launch.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; OUT=$PWD/gpurun_out/r3j; rm -rf $OUT; mkdir -p $OUT
EPPK_QUAD_MIN=4 timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
EPPK_QUAD_MIN=4 EPPK_QUAD_INLINE=0 timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
B="--no-cold-ref --no-cpu-baseline --host-path 0 --p99-samples 0"
for rep in 1 2; do for il in 1 0; do for inf in 1 2; do
EPPK_QUAD_INLINE=$il timeout 300 python bench.py --inflight $inf $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('inline=$il inflight=$inf', 'kernel_ms=%.4f p99=%.4f'%(d['roofline']['kernel_avg_ms'], d['roofline']['kernel_p99_ms']), 'step_ms=%.4f'%d['ms_per_step'], 'Mdec/s=%.1f'%(d['value']/1e6))"
done; done; done | tee $OUT/inline.txt
for r in 8192 16384; do for m in 4 24576; do
EPPK_QUAD_MIN=$m timeout 300 python bench.py --config 5 --requests $r $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('R=$r quad_min=$m', d['roofline']['kernel'], 'kernel_ms=%.4f'%d['roofline']['kernel_avg_ms'], 'step_ms=%.4f'%d['ms_per_step'], 'Mdec/s=%.1f'%(d['value']/1e6))"
done; done | tee $OUT/small.txt

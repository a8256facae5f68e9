#!/bin/bash
# GPU box: run each ab/libeppk_tNNN.so with EPPK_FAST_THREADS=NNN
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for round in 1 2; do
for so in ab/libeppk_t*.so; do
  t=$(basename $so .so); t=${t#libeppk_t}; t=${t%%_*}
  EPPK_LIB=$PWD/$so EPPK_FAST_THREADS=$t python bench.py --steps 100 --warmup 10 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$so', 'kernel_ms=%.4f'%d['roofline']['kernel_avg_ms'], 'Mdec/s=%.1f'%(d['value']/1e6), d.get('parity'))"
done; done | tee gpurun_out/ab_threads.txt

#!/bin/bash
# run every ab/libeppk_*.so through the bench, twice; a name ending in _tNNN sets EPPK_QUAD_THREADS=NNN
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
B="--no-cold-ref --no-cpu-baseline --host-path 0 --p99-samples 0 --steps 100 --warmup 10"
for rep in 1 2; do
  for so in ab/*.so; do
    t=$(echo $so | grep -o "_t[0-9]*\.so" | tr -dc 0-9); t=${t:-512}
    for inf in 1 2; do
    EPPK_QUAD_THREADS=$t EPPK_LIB=$PWD/$so timeout 200 python bench.py $B --inflight $inf "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$so inflight=$inf', 'kernel_ms=%.4f'%d['roofline']['kernel_avg_ms'], 'step_ms=%.4f'%d['ms_per_step'], 'Mdec/s=%.1f'%(d['value']/1e6))"
    done
  done
done | tee gpurun_out/abq.txt

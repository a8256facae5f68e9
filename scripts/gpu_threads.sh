#!/bin/bash
# GPU box: parity tests, then the bench at several fast-kernel workgroup sizes (EPPK_FAST_THREADS tuning knob).
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
for t in ${THREADS:-256 512 1024}; do
  for extra in "" "--groups 65536 --zipf 0"; do
    EPPK_FAST_THREADS=$t python bench.py --steps 100 --warmup 10 --no-cpu-baseline $extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('threads=$t $extra', 'kernel_ms=%.4f'%d['roofline']['kernel_avg_ms'], 'Mdec/s=%.1f'%(d['value']/1e6), 'frac=%.3f'%d['roofline']['frac'])"
  done
done | tee gpurun_out/threads.txt

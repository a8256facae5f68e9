#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/r3dma
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_staging.py -m gpu -q -x -k "default or not (quadmin4 or quad0 or lists0)" 2>&1 | tail -3 | tee $OUT/pytest_sel.txt
for th in 0 1 3 7 3; do
EPPK_COPY_THREADS=$th timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cold-ref --host-path 300 --p99-samples 0 2>/dev/null > $OUT/b_$th.json
python - <<P
import json
d=json.loads(open('$OUT/b_$th.json').read().strip().splitlines()[-1]); h=d['host_path']
print('EPPK_COPY_THREADS=$th pageable p50 %.3f p99 %.3f ms | staged p50 %.3f p99 %.3f | pipelined %.1f M/s' % (h['p50_ms'], h['p99_ms'], h['staged']['p50_ms'], h['staged']['p99_ms'], h['pipelined']['decisions_per_s']/1e6))
P
done | tee $OUT/copy_threads.txt

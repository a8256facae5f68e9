#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/r3dma
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_staging.py tests/test_gpupicker_cpp.py tests/test_host_cpp.py -m gpu -q -x -k "default or not (quadmin4 or quad0 or lists0)" 2>&1 | tail -3 | tee $OUT/pytest_sel.txt
timeout 120 python scripts/gpu_small_batch_latency.py 2>&1 | grep "n=" | tee $OUT/latency.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cold-ref --host-path 400 --p99-samples 0 2>/dev/null > $OUT/b.json
python - <<P
import json
d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1]); h=d['host_path']
print('pageable p50 %.3f p99 %.3f ms | staged p50 %.3f p99 %.3f | pipelined %.1f M/s, %.3f ms per batch, p50 %.3f p99 %.3f' % (h['p50_ms'], h['p99_ms'], h['staged']['p50_ms'], h['staged']['p99_ms'], h['pipelined']['decisions_per_s']/1e6, h['pipelined']['ms_per_batch'], h['pipelined']['p50_ms'], h['pipelined']['p99_ms']))
P

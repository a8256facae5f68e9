#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/r3zc2
rm -rf $OUT; mkdir -p $OUT
for zc in 0 1000000 0 1000000; do
  EPPK_ZERO_COPY_MAX=$zc timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cold-ref --host-path 300 --p99-samples 0 2>/dev/null > $OUT/b_$zc.json
  python - <<P
import json
d=json.loads(open('$OUT/b_$zc.json').read().strip().splitlines()[-1]); h=d['host_path']
print('zc_max=$zc pageable p50 %.3f p99 %.3f ms | staged p50 %.3f p99 %.3f | pipelined %.1f M/s, %.3f ms per batch, p50 %.3f p99 %.3f' % (h['p50_ms'], h['p99_ms'], h['staged']['p50_ms'], h['staged']['p99_ms'], h['pipelined']['decisions_per_s']/1e6, h['pipelined']['ms_per_batch'], h['pipelined']['p50_ms'], h['pipelined']['p99_ms']))
P
done | tee $OUT/host_path.txt

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r3s
last() { python -c "
import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'], d['config'].get('requests_per_launch'), d.get('completion_latency',{}).get('p50_ms'))"; }
for R in 1024 8192; do
timeout 300 python bench.py --force-dist --requests $R --steps 1600 --warmup 160 --no-cpu-baseline --no-cold-ref --host-path 0 --p99-samples 0 > gpurun_out/r3s/fd_$R.json 2>/dev/null; echo -n "force-dist R=$R: "; last gpurun_out/r3s/fd_$R.json
done

#!/bin/bash
# Run on the GPU box: bench lines for the other BASELINE configs + cold-cache index variant + host path.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for c in 2 3 4; do
  python bench.py --config $c --steps 200 --warmup 20 > gpurun_out/bench_c$c.json 2>/dev/null; cut -c1-400 gpurun_out/bench_c$c.json
done
python bench.py --steps 100 --warmup 10 --host-path 60 > gpurun_out/bench_c5_host.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/bench_c5_host.json')); print(d['value'], d['roofline']['kernel_avg_ms'], d['host_path'])"
python bench.py --steps 50 --warmup 5 --groups 65536 --zipf 0 --no-cpu-baseline > gpurun_out/bench_c5_cold.json 2>gpurun_out/cold.err; tail -2 gpurun_out/cold.err; cut -c1-1200 gpurun_out/bench_c5_cold.json

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/r3t
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_quad.py tests/test_gpu_parity.py tests/test_gpu_pickers.py -m gpu -q -x -k "fallback or topk or mask or quad or picker" > $OUT/pytest_sel.txt 2>&1; tail -5 $OUT/pytest_sel.txt
timeout 300 python scripts/gpu_route_times.py > $OUT/route_times.json 2> $OUT/route_times.err; python -c "
import json;d=json.load(open('$OUT/route_times.json'))
for k,v in d.items():
    if isinstance(v,dict) and 'kernel_avg_us' in v: print(k, round(v['kernel_avg_us'],1), v.get('quad_launches_so_far'), v.get('quad_deferred_so_far'))"

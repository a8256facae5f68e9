#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
last() { python -c "
import json,sys;d=json.loads(open('$1').read().strip().splitlines()[-1]);print(d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'])"; }
mkdir -p gpurun_out/r3p
cp gateway-api-inference-extension_amd/libeppk.so /tmp/lib_main.so
for v in main noreport; do
  [ $v = noreport ] && cp ab/noreport/libeppk.so gateway-api-inference-extension_amd/libeppk.so
  for inf in 1 2; do
    timeout 300 python bench.py --inflight $inf --no-cpu-baseline --no-cold-ref --host-path 0 > gpurun_out/r3p/b_${v}_$inf.json 2>/dev/null; echo -n "$v inflight $inf: "; last gpurun_out/r3p/b_${v}_$inf.json
  done
done
cp /tmp/lib_main.so gateway-api-inference-extension_amd/libeppk.so

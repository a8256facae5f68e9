#!/bin/bash
# A/B kernel variants. Build here (CPU container):   bash scripts/ab.sh build "name:-DFLAG=1 -DX=2" ...
# Run on the GPU box (via gpurun):                   bash scripts/ab.sh run [bench args]
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname $0)/..}"
CS=gateway-api-inference-extension_amd/csrc
mkdir -p ab
if [ "$1" = build ]; then
  shift; rm -f ab/*.so
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-function $flags -o ab/libeppk_$name.so $CS/eppk.hip $CS/eppk_host.cpp &
  done
  wait; ls -la ab/
else
  shift
  mkdir -p gpurun_out
  for round in 1 2; do
    for so in ab/*.so; do
      EPPK_LIB=$PWD/$so python bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$so', 'kernel_ms=%.4f'%d['roofline']['kernel_avg_ms'], 'Mdec/s=%.1f'%(d['value']/1e6))"
    done
  done | tee gpurun_out/ab.txt
fi

#!/bin/bash
# A/B kernel variants. Build here (CPU container):   bash scripts/ab.sh build "name:-DFLAG=1 -DX=2" ...
# Run on the GPU box (via gpurun):                   bash scripts/ab.sh run [bench args]
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname $0)/..}"
CS=gateway-api-inference-extension_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function"
mkdir -p ab
if [ "$1" = build ]; then
  shift; rm -rf ab/*.so ab/obj
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    mkdir -p ab/obj/$name
    for u in $CS/eppk.hip $CS/eppk_host.cpp $CS/eppk_pick_*.hip; do
      /opt/rocm/bin/hipcc $FLAGS $flags -c -o ab/obj/$name/$(basename ${u%.*}).o $u &
    done
  done
  wait
  for spec in "$@"; do
    name=${spec%%:*}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libeppk_$name.so ab/obj/$name/*.o
  done
  rm -rf ab/obj; ls -la ab/
else
  shift
  mkdir -p gpurun_out
  for round in 1 2; do
    for so in ab/*.so; do
      EPPK_LIB=$PWD/$so python bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$so', 'kernel_ms=%.4f'%d['roofline']['kernel_avg_ms'], 'Mdec/s=%.1f'%(d['value']/1e6))"
    done
  done | tee gpurun_out/ab.txt
fi

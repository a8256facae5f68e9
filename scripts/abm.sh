#!/bin/bash
# A/B variants of the MAIN unit only (eppk.hip: the index maintenance kernels, the ABI): every other unit comes from the tree's build.
#   bash scripts/abm.sh "name:-DFLAG=1" ...   -> ab/libeppk_<name>.so
set -e
cd "$(dirname $0)/.."
CS=gateway-api-inference-extension_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function"
mkdir -p ab/base; rm -f ab/*.so
python -c "import __graft_entry__ as g; g.build()" 2>/dev/null >/dev/null
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc $FLAGS $flags -c -o ab/base/main_$name.o $CS/eppk.hip 2>/dev/null
    objs=$(ls $CS/build/*.o | grep -v "/eppk.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libeppk_$name.so $objs ab/base/main_$name.o ) &
done
wait; rm -rf ab/base; ls ab/

#!/bin/bash
# A/B variants of pick_quad_kernel only: every other unit is compiled once, each variant recompiles ONE unit (UNIT=..., default
# eppk_pick_quad; the headline's one-launch form is eppk_pick_quad_tail) with its flags and links its own library.
#   [UNIT=eppk_pick_quad_tail] bash scripts/abq.sh "name:-DFLAG=1" ...   -> ab/libeppk_<name>.so
set -e
cd "$(dirname $0)/.."
CS=gateway-api-inference-extension_amd/csrc
UNIT=${UNIT:-eppk_pick_quad}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function"
mkdir -p ab/base; rm -f ab/*.so
[ -n "$NOBUILD" ] || python -c "import __graft_entry__ as g; g.build()" 2>/dev/null >/dev/null   # (NOBUILD=1: the other units as they were last built)
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc $FLAGS $flags -c -o ab/base/quad_$name.o $CS/$UNIT.hip 2>/dev/null
    objs=$(ls $CS/build/*.o | grep -v "/$UNIT.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libeppk_$name.so $objs ab/base/quad_$name.o ) &
done
wait; rm -rf ab/base; ls ab/

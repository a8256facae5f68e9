#!/bin/bash
# quick: the tests that exercise concurrent first inserts of a key + the harness
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/r3d
rm -rf $OUT; mkdir -p $OUT
for lib in "$@"; do
  echo "=== $lib"
  cp $lib gateway-api-inference-extension_amd/libeppk.so
  timeout 600 python -m pytest tests/test_gpu_group.py "tests/test_gpu_closed_loop.py::test_eight_closed_loop_generations_at_full_size[8192-2097152-False]" tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -4
done
[ -x scripts/micro/insertbreak ] && timeout 60 ./scripts/micro/insertbreak 2>&1 | tail -8

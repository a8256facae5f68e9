#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r2j
rm -rf $OUT; mkdir -p $OUT
./scripts/micro/linegather | tee $OUT/linegather.txt
for inf in 1 2 3 4; do
python - <<PY
import re
PY
done

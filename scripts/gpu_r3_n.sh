#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/r3n
rm -rf $OUT; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest_all.txt 2>&1; tail -6 $OUT/pytest_all.txt
timeout 60 ./scripts/micro/insertbreak > $OUT/insertbreak.txt 2>&1; cat $OUT/insertbreak.txt

#!/bin/bash
# rocprofv3 --kernel-trace --stats of masked batches (scripts/mask_probe.py at 50 % / 12.5 % / 6 % density) and of the subset-filter shapes
# (scripts/subset_route_probe.py): gpurun -- 'bash scripts/gpu_maskprof.sh' -> gpurun_out/r6_maskprof/
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6_maskprof; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for d in 50 12 6; do
  MASK_ONLY="${d}%" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/m$d -o t -- python $R/scripts/mask_probe.py > $OUT/m$d.txt 2>&1
  grep "^mask" $OUT/m$d.txt
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/subset -o t -- python $R/scripts/subset_route_probe.py > $OUT/subset.txt 2>&1
grep "candidates\|top-3" $OUT/subset.txt | tail -6
ls $OUT/m12

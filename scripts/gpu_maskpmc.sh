#!/bin/bash
# HBM traffic of the MASKED pick kernel by counters (separate --pmc passes, --kernel-trace only): scripts/mask_probe.py at 50 % and 12.5 % density.
#   gpurun -- 'bash scripts/gpu_maskpmc.sh' -> gpurun_out/r6_maskpmc/{m50,m12}_summary.csv
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6_maskpmc; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
CTRS=("WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS")
for d in 50 12; do
  mkdir -p $OUT/m$d; i=0
  for ctrs in "${CTRS[@]}"; do i=$((i+1))
    MASK_ONLY="${d}%" timeout -k 5 150 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/m$d -o pass$i -- python $R/scripts/mask_probe.py > $OUT/m$d/pass$i.txt 2>&1
  done
  ( cd $R; python scripts/pmc_summary.py $OUT/m$d pick_quad --by-kernel | tee $OUT/m${d}_summary.csv | cut -c1-200 )
  rm -f $(find $OUT/m$d -name "*agent_info.csv") $(find $OUT/m$d -name "*kernel_trace.csv")
done

#!/bin/bash
# PMC passes over the closed loop (what bounds index_insert_picks_kernel / index_evict_kernel): SQ instruction mix, wave cycles, TA busy
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3x
rm -rf $OUT; mkdir -p $OUT/pmc
CARGS="--closed-loop --steps 12 --warmup 6 --no-cpu-baseline --cl-verify 0"
i=0
for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "TA_BUSY_avr TA_TA_BUSY_sum TCC_BUSY_avr" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum"; do
  i=$((i+1))
  ( cd /tmp; timeout -k 5 150 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/pmc -o pass$i -- python $GRAFT_REPO_ROOT/bench.py $CARGS > $OUT/pmc/bench_pass$i.json 2> $OUT/pmc/pass$i.err )
done
python scripts/pmc_summary.py $OUT/pmc index_ --by-kernel | tee $OUT/pmc_summary.csv | cut -c1-260
rm -f $(find $OUT -name "*agent_info.csv") $(find $OUT -name "*kernel_trace.csv")
du -sh $OUT

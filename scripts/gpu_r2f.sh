#!/bin/bash
# Round 2 experiment: why is the lean kernel (8 waves per SIMD) not faster?  Kernel trace + SQ counters of the lean pair, in the
# worktree of commit a51441b (.lean_exp/).
cd "${GRAFT_REPO_ROOT:-.}/.lean_exp"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2f
rm -rf $OUT; mkdir -p $OUT/pmc $OUT/prof
BARGS="--steps 30 --warmup 4 --inflight 1 --no-cpu-baseline --host-path 0 --no-cold-ref --p99-samples 0"
( cd /tmp; timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o lean -- python $GRAFT_REPO_ROOT/.lean_exp/bench.py $BARGS > $OUT/prof/bench.json 2> $OUT/prof/err.txt )
f=$(find $OUT/prof -name "lean_kernel_stats.csv" | head -1); head -5 "$f" | cut -c1-300
i=0
for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  ( cd /tmp; timeout -k 5 100 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/pmc -o pass$i -- python $GRAFT_REPO_ROOT/.lean_exp/bench.py $BARGS > $OUT/pmc/bench_pass$i.json 2> $OUT/pmc/pass$i.err )
done
python $GRAFT_REPO_ROOT/scripts/pmc_summary.py $OUT/pmc pick_ --by-kernel | cut -c1-60,300- | tee $OUT/pmc_summary.csv
rm -f $(find $OUT -name "*agent_info.csv") $(find $OUT -name "*kernel_trace.csv")

#!/bin/bash
# Run on the GPU box (via gpurun): the round's measurement session -- tests, headline bench, other configs, rocprofv3 kernel
# stats and PMC passes.  Everything lands in gpurun_out/round/.  Every rocprofv3 run sits under its own `timeout`: a counter set
# the hardware cannot collect in one pass makes rocprofv3 abort and then hang in its signal handler (FETCH_SIZE + WRITE_SIZE
# together cost this round 25 GPU-minutes).  Set ONLY_PMC=1 to run just the counter passes.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/round
[ -z "$ONLY_PMC" ] && rm -rf $OUT; rm -rf $OUT/pmc; mkdir -p $OUT/pmc $OUT/prof
if [ -z "$ONLY_PMC" ]; then
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $OUT/pytest.txt
python bench.py --steps 200 --warmup 20 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; cut -c1-300 $OUT/bench_c5.json
for c in 2 3 4; do python bench.py --config $c --steps 200 --warmup 20 > $OUT/bench_c$c.json 2>/dev/null; done
python bench.py --steps 50 --warmup 5 --groups 65536 --zipf 0 --no-cpu-baseline > $OUT/bench_c5_cold.json 2>/dev/null
EPPK_LISTS=0 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --host-path 0 > $OUT/bench_c5_dense_rows_only.json 2>/dev/null
python bench.py --steps 200 --warmup 20 --force-dist --no-cpu-baseline --host-path 0 > $OUT/bench_c5_force_dist.json 2>/dev/null
( cd /tmp; timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --host-path 0 > $OUT/prof/bench_under_rocprof.json 2> $OUT/prof/prof.err )
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); head -6 "$f"
fi
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  ( cd /tmp; timeout -k 5 90 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/pmc -o pass$i -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --host-path 0 > $OUT/pmc/bench_pass$i.json 2> $OUT/pmc/pass$i.err )
done
python scripts/pmc_summary.py $OUT/pmc pick_fast_kernel | tee $OUT/pmc_summary.csv
rm -f $OUT/pmc/*agent_info.csv

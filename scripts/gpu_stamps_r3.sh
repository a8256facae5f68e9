#!/bin/bash
# Round 3: re-measure what bench.py's stamped JSONs hold (after the last change of the hashed sources): the bench line and its variants,
# rocprofv3 kernel stats, PMC passes of the headline (one-launch quad form) and of the cold reference.  Everything -> gpurun_out/r3stamps/;
# scripts/make_pmc_json.py turns the passes into the stamped profiles/*.json (run it here, in the build container, afterwards).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3stamps
rm -rf $OUT; mkdir -p $OUT/pmc $OUT/pmc_cold $OUT/prof $OUT/prof_cl
T0=$(date +%s)
lap() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
( cd /tmp; timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline --host-path 0 --no-cold-ref --p99-samples 0 > $OUT/prof/bench_under_rocprof.json 2> $OUT/prof/prof.err )
( cd /tmp; timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_cl -o trace -- python $GRAFT_REPO_ROOT/bench.py --closed-loop --steps 60 --warmup 10 --no-cpu-baseline --cl-verify 0 > $OUT/prof_cl/bench_under_rocprof.json 2> $OUT/prof_cl/prof.err )
lap stats
BARGS="--steps 6 --warmup 10 --no-cpu-baseline --host-path 0 --no-cold-ref --p99-samples 0"
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TA_BUSY_avr TA_TA_BUSY_sum TCC_BUSY_avr"; do
  i=$((i+1))
  ( cd /tmp; timeout -k 5 100 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/pmc -o pass$i -- python $GRAFT_REPO_ROOT/bench.py $BARGS --inflight 1 > $OUT/pmc/bench_pass$i.json 2> $OUT/pmc/pass$i.err )
done
python scripts/pmc_summary.py $OUT/pmc pick_ --by-kernel | tee $OUT/pmc_summary.csv | cut -c1-200; lap pmc
CARGS="--steps 6 --warmup 10 --no-cpu-baseline --host-path 0 --no-cold-ref --p99-samples 0 --groups 262144 --zipf 0 --pods-per-group 4 --batches 4 --inflight 1"
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum"; do
  i=$((i+1))
  ( cd /tmp; timeout -k 5 150 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/pmc_cold -o pass$i -- python $GRAFT_REPO_ROOT/bench.py $CARGS > $OUT/pmc_cold/bench_pass$i.json 2> $OUT/pmc_cold/pass$i.err )
done
python scripts/pmc_summary.py $OUT/pmc_cold pick_ --by-kernel | tee $OUT/pmc_cold_summary.csv | cut -c1-200; lap pmc-cold
rm -f $(find $OUT -name "*agent_info.csv") $(find $OUT -name "*kernel_trace.csv")
du -sh $OUT; lap done

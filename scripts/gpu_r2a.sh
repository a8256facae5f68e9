#!/bin/bash
# Round 2, first GPU session: the whole GPU suite, the reworked bench line (+ cold reference), the closed-loop baseline, the N>1
# code path on one GPU, FETCH_SIZE calibration, rocprofv3 kernel stats and a first set of PMC passes.  Everything -> gpurun_out/r2a/.
# Every rocprofv3 run sits under its own `timeout` (NEXT.md: a counter set the hardware cannot collect hangs rocprofv3).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r2a
rm -rf $OUT; mkdir -p $OUT/pmc $OUT/prof $OUT/cal
T0=$(date +%s)
lap() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest.txt; lap pytest
timeout 400 python bench.py --steps 200 --warmup 20 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; tail -3 $OUT/bench_c5.err; cut -c1-600 $OUT/bench_c5.json; lap bench
timeout 300 python bench.py --steps 200 --warmup 20 --inflight 1 --no-cold-ref --no-cpu-baseline --host-path 0 > $OUT/bench_c5_inflight1.json 2>/dev/null; lap inflight1
timeout 400 python bench.py --closed-loop --steps 60 --warmup 10 > $OUT/bench_closed_loop.json 2> $OUT/bench_closed_loop.err; tail -3 $OUT/bench_closed_loop.err; cut -c1-400 $OUT/bench_closed_loop.json; lap closed
timeout 300 python bench.py --steps 200 --warmup 20 --force-dist --no-cpu-baseline --host-path 0 > $OUT/bench_c5_force_dist.json 2> $OUT/force_dist.err; tail -2 $OUT/force_dist.err; cut -c1-300 $OUT/bench_c5_force_dist.json; lap forcedist
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (scripts/micro/fetchcal.hip)
( cd scripts/micro && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o fetchcal fetchcal.hip 2>&1 | tail -2; ./fetchcal | tee $OUT/cal/fetchcal_plain.txt )
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  ( cd /tmp; timeout -k 5 120 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/cal -o cal$i -- $GRAFT_REPO_ROOT/scripts/micro/fetchcal > $OUT/cal/run$i.txt 2> $OUT/cal/err$i.txt )
done
python scripts/pmc_summary.py $OUT/cal cal_ --by-kernel | tee $OUT/cal/summary.csv; lap fetchcal
# kernel stats of the headline and of the closed loop
( cd /tmp; timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --host-path 0 --no-cold-ref --p99-samples 0 > $OUT/prof/bench_under_rocprof.json 2> $OUT/prof/prof.err )
f=$(find $OUT/prof -name "trace_kernel_stats.csv" | head -1); head -8 "$f"
( cd /tmp; timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o cl -- python $GRAFT_REPO_ROOT/bench.py --closed-loop --steps 30 --warmup 6 --cl-verify 0 > $OUT/prof/bench_cl_under_rocprof.json 2> $OUT/prof/cl.err )
f=$(find $OUT/prof -name "cl_kernel_stats.csv" | head -1); head -8 "$f"; lap stats
rm -f $(find $OUT/prof -name "*kernel_trace.csv") $(find $OUT/prof -name "*agent_info.csv")
# PMC passes on the headline (16 rotating batches for the traffic counters)
BARGS="--steps 6 --warmup 2 --no-cpu-baseline --host-path 0 --no-cold-ref --p99-samples 0"
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  ( cd /tmp; timeout -k 5 100 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/pmc -o pass$i -- python $GRAFT_REPO_ROOT/bench.py $BARGS > $OUT/pmc/bench_pass$i.json 2> $OUT/pmc/pass$i.err )
done
python scripts/pmc_summary.py $OUT/pmc pick_fast_kernel | tee $OUT/pmc_summary.csv
rm -f $(find $OUT -name "*agent_info.csv") $(find $OUT -name "*kernel_trace.csv")
du -sh $OUT; lap done

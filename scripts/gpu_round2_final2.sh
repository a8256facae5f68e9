#!/bin/bash
# Round 2, final measurement session with pick_quad_kernel in front (run on the GPU box via gpurun): the GPU suite in five modes,
# the bench line and its variants, rocprofv3 kernel stats, PMC passes of the headline and of the cold reference.
# Everything -> gpurun_out/r2final2/; scripts/make_pmc_json.py turns the passes into the stamped profiles/*.json.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r2final2
rm -rf $OUT; mkdir -p $OUT/pmc $OUT/pmc_cold $OUT/prof
T0=$(date +%s)
lap() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $OUT/pytest.txt; lap pytest
EPPK_QUAD_MIN=4 timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $OUT/pytest_quad_everywhere.txt; lap pytest-quad-everywhere
EPPK_QUAD=0 timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_quad.py 2>&1 | tail -3 | tee $OUT/pytest_quad_off.txt; lap pytest-quad-off
EPPK_LISTS=0 timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_quad.py 2>&1 | tail -3 | tee $OUT/pytest_lists_off.txt; lap pytest-lists-off
if [ -f ab/libeppk_nouniform.so ]; then EPPK_LIB=$PWD/ab/libeppk_nouniform.so EPPK_QUAD=0 timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_quad.py 2>&1 | tail -3 | tee $OUT/pytest_no_uniform.txt; lap pytest-no-uniform; fi
timeout 500 python bench.py > $OUT/bench_c5.json 2> $OUT/bench_c5.err; tail -2 $OUT/bench_c5.err; cut -c1-300 $OUT/bench_c5.json; lap bench
timeout 300 python bench.py --inflight 1 --no-cold-ref --no-cpu-baseline --host-path 0 > $OUT/bench_c5_inflight1.json 2>/dev/null; lap inflight1
EPPK_QUAD=0 timeout 300 python bench.py --no-cold-ref --no-cpu-baseline --host-path 0 --p99-samples 0 > $OUT/bench_c5_quad_off.json 2>/dev/null; lap quad-off
for c in 2 3 4; do timeout 200 python bench.py --config $c --no-cold-ref --host-path 0 > $OUT/bench_c$c.json 2>/dev/null; done; lap configs
timeout 400 python bench.py --closed-loop --steps 100 --warmup 10 > $OUT/bench_closed_loop.json 2>/dev/null; cut -c1-200 $OUT/bench_closed_loop.json; lap closed
timeout 300 python bench.py --force-dist --no-cpu-baseline --host-path 0 > $OUT/bench_c5_force_dist.json 2>/dev/null; lap forcedist
timeout 300 python scripts/gpu_route_times.py > $OUT/route_times.json 2>/dev/null; lap routes
( cd /tmp; timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --host-path 0 --no-cold-ref --p99-samples 0 > $OUT/prof/bench_under_rocprof.json 2> $OUT/prof/prof.err )
f=$(find $OUT/prof -name "trace_kernel_stats.csv" | head -1); head -4 "$f" | cut -c1-60,200-330; lap stats
BARGS="--steps 6 --warmup 2 --no-cpu-baseline --host-path 0 --no-cold-ref --p99-samples 0"
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TA_BUSY_avr TA_TA_BUSY_sum TCC_BUSY_avr"; do
  i=$((i+1))
  ( cd /tmp; timeout -k 5 100 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/pmc -o pass$i -- python $GRAFT_REPO_ROOT/bench.py $BARGS --inflight 1 > $OUT/pmc/bench_pass$i.json 2> $OUT/pmc/pass$i.err )
done
python scripts/pmc_summary.py $OUT/pmc pick_ --by-kernel | tee $OUT/pmc_summary.csv; lap pmc
CARGS="--steps 6 --warmup 2 --no-cpu-baseline --host-path 0 --no-cold-ref --p99-samples 0 --groups 262144 --zipf 0 --pods-per-group 4 --batches 4 --inflight 1"
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum"; do
  i=$((i+1))
  ( cd /tmp; timeout -k 5 150 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/pmc_cold -o pass$i -- python $GRAFT_REPO_ROOT/bench.py $CARGS > $OUT/pmc_cold/bench_pass$i.json 2> $OUT/pmc_cold/pass$i.err )
done
python scripts/pmc_summary.py $OUT/pmc_cold pick_ --by-kernel | tee $OUT/pmc_cold_summary.csv; lap pmc-cold
rm -f $(find $OUT -name "*agent_info.csv") $(find $OUT -name "*kernel_trace.csv")
du -sh $OUT; lap done

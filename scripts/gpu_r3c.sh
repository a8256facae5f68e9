#!/bin/bash
# Round 2, session 3c: pick_quad_kernel v2 (quad-transposed gathers, 20 keys ahead, interleaved tier planes): suite, bench, counters.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3c
rm -rf $OUT; mkdir -p $OUT/pmc
T0=$(date +%s)
lap() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
show() { python -c "import sys,json; d=json.load(open('$1')); print('$1'.split('/')[-1], 'kernel_ms=%.4f'%d['roofline']['kernel_avg_ms'], 'step_ms=%.4f'%d['ms_per_step'], 'Mdec/s=%.1f'%(d['value']/1e6))"; }
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $OUT/pytest.txt; lap pytest
B="--no-cold-ref --no-cpu-baseline --host-path 0 --p99-samples 0"
timeout 300 python bench.py --inflight 1 $B > $OUT/bench_quad_inflight1.json 2> $OUT/err1.txt; tail -2 $OUT/err1.txt; show $OUT/bench_quad_inflight1.json
timeout 300 python bench.py $B > $OUT/bench_quad_inflight2.json 2>/dev/null; show $OUT/bench_quad_inflight2.json; lap bench
for t in 256 1024; do EPPK_QUAD_THREADS=$t timeout 300 python bench.py $B > $OUT/bench_quad_t$t.json 2>/dev/null; show $OUT/bench_quad_t$t.json; done; lap threads
( cd /tmp; timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --inflight 1 $B > $OUT/bench_under_rocprof.json 2> $OUT/prof.err )
f=$(find $OUT/prof -name "trace_kernel_stats.csv" | head -1); head -4 "$f" | cut -c1-60,200-400; cp "$f" $OUT/kernel_stats.csv; rm -rf $OUT/prof; lap stats
P="--no-cold-ref --no-cpu-baseline --host-path 0 --p99-samples 0 --inflight 1 --steps 6 --warmup 2"
i=0
for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TA_BUSY_avr TA_TA_BUSY_sum TCC_BUSY_avr"; do
  i=$((i+1))
  ( cd /tmp; timeout -k 5 100 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/pmc -o pass$i -- python $GRAFT_REPO_ROOT/bench.py $P > $OUT/pmc/bench_pass$i.json 2> $OUT/pmc/pass$i.err ) || echo "pass $i failed: $ctrs"
done
python scripts/pmc_summary.py $OUT/pmc pick_quad_kernel | tee $OUT/pmc_summary_quad.csv; lap pmc
rm -f $(find $OUT -name "*agent_info.csv") $(find $OUT -name "*kernel_trace.csv")

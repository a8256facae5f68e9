#!/bin/bash
# Round 2, session 3b: is the pick bound by L2 channel hot spots (Zipf over 256 prefix groups)?  zipf 1 vs 0, quad vs fast kernel;
# the counters rocprofv3 offers for the vector memory pipe; a PMC pass of the quad kernel.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3b
rm -rf $OUT; mkdir -p $OUT/pmc
T0=$(date +%s)
lap() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
B="--no-cold-ref --no-cpu-baseline --host-path 0 --p99-samples 0 --inflight 1 --steps 100 --warmup 10"
show() { python -c "import sys,json; d=json.load(open('$1')); print('$1'.split('/')[-1], 'kernel_ms=%.4f'%d['roofline']['kernel_avg_ms'], 'step_ms=%.4f'%d['ms_per_step'], 'Mdec/s=%.1f'%(d['value']/1e6))"; }
for q in 1 0; do for z in 1 0; do
  EPPK_QUAD=$q timeout 200 python bench.py $B --zipf $z > $OUT/bench_q${q}_z${z}.json 2>/dev/null; show $OUT/bench_q${q}_z${z}.json
done; done; lap zipf
# groups 4096 (16x more hot lines, still L2-resident per XCD? 4096*16*128 B = 8 MB: no -> Infinity Cache)
for q in 1 0; do EPPK_QUAD=$q timeout 200 python bench.py $B --groups 1024 > $OUT/bench_q${q}_g1024.json 2>/dev/null; show $OUT/bench_q${q}_g1024.json; done; lap groups
( cd /tmp; timeout 60 rocprofv3 -L 2>/dev/null | grep -o "\b\(TCP\|TA\|TD\|TCC\)_[A-Za-z_0-9]*" | sort -u | tr '\n' ' ' > $OUT/counters_available.txt ); wc -c $OUT/counters_available.txt; lap list
P="--no-cold-ref --no-cpu-baseline --host-path 0 --p99-samples 0 --inflight 1 --steps 6 --warmup 2"
i=0
for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TA_BUSY_avr TA_TA_BUSY_sum TCP_TA_TCP_STATE_READ_sum" "TCC_TAG_STALL_sum TCC_BUSY_avr"; do
  i=$((i+1))
  ( cd /tmp; timeout -k 5 100 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/pmc -o pass$i -- python $GRAFT_REPO_ROOT/bench.py $P > $OUT/pmc/bench_pass$i.json 2> $OUT/pmc/pass$i.err ) || echo "pass $i failed: $ctrs"
done
python scripts/pmc_summary.py $OUT/pmc pick_quad_kernel | tee $OUT/pmc_summary_quad.csv; lap pmc
rm -f $(find $OUT -name "*agent_info.csv") $(find $OUT -name "*kernel_trace.csv")
du -sh $OUT

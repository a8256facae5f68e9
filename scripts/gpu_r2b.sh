#!/bin/bash
# Round 2, second GPU session: GPU suite (holes, closed loop at full size, groups, device-row validation), closed-loop bench after the
# index-maintenance rework, its kernel stats.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r2b
rm -rf $OUT; mkdir -p $OUT/prof
T0=$(date +%s)
lap() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee $OUT/pytest.txt; lap pytest
timeout 400 python bench.py --closed-loop --steps 100 --warmup 10 > $OUT/bench_closed_loop.json 2> $OUT/bench_closed_loop.err; tail -3 $OUT/bench_closed_loop.err; cut -c1-300 $OUT/bench_closed_loop.json; lap closed
( cd /tmp; timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o cl -- python $GRAFT_REPO_ROOT/bench.py --closed-loop --steps 30 --warmup 6 --cl-verify 0 > $OUT/prof/bench_cl_under_rocprof.json 2> $OUT/prof/cl.err )
f=$(find $OUT/prof -name "cl_kernel_stats.csv" | head -1); head -8 "$f"; lap stats
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --host-path 0 --p99-samples 0 > $OUT/bench_c5.json 2>/dev/null; cut -c1-200 $OUT/bench_c5.json; lap bench
rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_EA0_RDREQ[A-Za-z0-9_]*\|TCC_EA0_WRREQ[A-Za-z0-9_]*\|TCC_BUBBLE[A-Za-z0-9_]*\|TCC_EA0_RD_UNCACHED[A-Za-z0-9_]*" | sort -u | tr '\n' ' ' > $OUT/tcc_counters.txt; cat $OUT/tcc_counters.txt; echo
rm -f $(find $OUT -name "*agent_info.csv") $(find $OUT -name "*kernel_trace.csv")
du -sh $OUT; lap done

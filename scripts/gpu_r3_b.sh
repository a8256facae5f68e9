#!/bin/bash
# Round 3, second GPU call: the work-list pass sized by the reports + sampled profiling.  Quick parity, the bench line, kernel stats
# under rocprofv3, the C harness beside bench.py.  Output -> gpurun_out/r3b/.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3b
rm -rf $OUT; mkdir -p $OUT/prof
T0=$(date +%s)
lap() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 600 python -m pytest tests/test_gpu_quad.py tests/test_zz_parity_quick_gpu.py -m gpu -x -q > $OUT/pytest_quad.txt 2>&1; tail -2 $OUT/pytest_quad.txt; lap quad-tests
timeout 300 python bench.py --no-cpu-baseline --no-cold-ref --host-path 0 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; cut -c1-330 $OUT/bench_c5.json; lap bench
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cold-ref --host-path 0 > $OUT/bench_c5_20.json 2> $OUT/bench_c5_20.err; cut -c1-330 $OUT/bench_c5_20.json; lap bench20
timeout 300 python bench.py --inflight 1 --no-cpu-baseline --no-cold-ref --host-path 0 > $OUT/bench_c5_inflight1.json 2>/dev/null; cut -c1-330 $OUT/bench_c5_inflight1.json; lap inflight1
( cd /tmp; timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --host-path 0 --no-cold-ref --p99-samples 0 > $OUT/prof/bench_under_rocprof.json 2> $OUT/prof/prof.err )
f=$(find $OUT/prof -name "trace_kernel_stats.csv" | head -1); head -5 "$f" | cut -c1-60,200-330; lap stats
[ -x scripts/micro/pickbench ] && [ -d scripts/micro/_gen/c5 ] && timeout 100 ./scripts/micro/pickbench scripts/micro/_gen/c5 gateway-api-inference-extension_amd/libeppk.so > $OUT/pickbench.txt 2>&1; tail -5 $OUT/pickbench.txt; lap pickbench
rm -f $(find $OUT -name "*agent_info.csv") $(find $OUT -name "*kernel_trace.csv")

#!/bin/bash
# Round 2, session 3a: first run of pick_quad_kernel (four requests per wavefront) -- GPU suite, then the bench with it and without.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3a
rm -rf $OUT; mkdir -p $OUT
T0=$(date +%s)
lap() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee $OUT/pytest.txt; lap pytest
B="--no-cold-ref --no-cpu-baseline --host-path 0 --p99-samples 0"
timeout 300 python bench.py --inflight 1 $B > $OUT/bench_quad_inflight1.json 2> $OUT/err1.txt; tail -3 $OUT/err1.txt; cut -c1-400 $OUT/bench_quad_inflight1.json; lap quad1
EPPK_QUAD=0 timeout 300 python bench.py --inflight 1 $B > $OUT/bench_noquad_inflight1.json 2>/dev/null; cut -c1-400 $OUT/bench_noquad_inflight1.json; lap noquad1
timeout 300 python bench.py $B > $OUT/bench_quad_inflight2.json 2>/dev/null; cut -c1-400 $OUT/bench_quad_inflight2.json; lap quad2
for t in 256 1024; do EPPK_QUAD_THREADS=$t timeout 300 python bench.py --inflight 1 $B > $OUT/bench_quad_t$t.json 2>/dev/null; cut -c1-200 $OUT/bench_quad_t$t.json; done; lap threads
( cd /tmp; timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --inflight 1 $B > $OUT/bench_under_rocprof.json 2> $OUT/prof.err )
f=$(find $OUT/prof -name "trace_kernel_stats.csv" | head -1); head -5 "$f" | cut -c1-60,200-400; cp "$f" $OUT/kernel_stats.csv; rm -rf $OUT/prof; lap stats

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/r3l
rm -rf $OUT; mkdir -p $OUT
PYTHONFAULTHANDLER=1 timeout 400 python -u -c "
import sys, traceback
sys.argv=['bench.py','--force-dist','--steps','64','--warmup','16','--no-cpu-baseline','--no-cold-ref','--host-path','0']
import bench
try:
    bench.main()
    print('MAIN RETURNED', file=sys.stderr)
except BaseException as e:
    traceback.print_exc()
    print('EXC', repr(e), file=sys.stderr)
" > $OUT/bench_forcedist.json 2> $OUT/bench_forcedist.err; echo "rc=$?"; tail -c 1500 $OUT/bench_forcedist.err; head -c 300 $OUT/bench_forcedist.json

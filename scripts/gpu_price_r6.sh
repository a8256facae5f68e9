#!/bin/bash
# Round-6 pricing run (GPU box): what would a pick cost if a request fetched ONE pod-set line instead of one per hit?
# ab/libeppk_onelist.so = the library with -DEPPK_DBGQ_ONE_LIST in eppk_pick_quad_tail.hip (timing experiment; exact for identical lists).
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r6_price; mkdir -p $OUT
echo "== l2gather (64-byte vs 128-byte lines)"; timeout 60 ./scripts/micro/_bin/l2gather 2>&1 | tee $OUT/micro_l2gather.txt
for lib in base onelist; do
  [ $lib = base ] && unset EPPK_LIB || export EPPK_LIB=$PWD/ab/libeppk_$lib.so
  echo "== $lib: new / returning 64k batch (scripts/revisit_probe.py)"
  timeout 200 python scripts/revisit_probe.py 65536 2>&1 | tail -2 | tee $OUT/revisit_$lib.txt
  echo "== $lib: headline + cold"
  timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --host-path 0 --no-closed-loop-leg --no-revisit-leg --p99-samples 0 --inflight 1 > $OUT/bench_$lib.json 2> $OUT/bench_$lib.err
  python - $OUT/bench_$lib.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r, c = d["roofline"], d.get("roofline_cold", {})
print(f"   headline kernel {r['kernel_avg_ms'] * 1e3:.2f} us  step {d['ms_per_step'] * 1e3:.2f} us  parity {d.get('parity')}")
print(f"   cold kernel {c.get('kernel_avg_ms', 0) * 1e3:.2f} us  value {c.get('value', 0) / 1e6:.0f} M/s")
PY
done

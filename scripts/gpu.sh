#!/bin/bash
# ONE runner for everything that needs the GPU box:   gpurun --timeout N -- 'bash scripts/gpu.sh <tag> <stage> [<stage> ...]'
# Every stage writes into gpurun_out/<tag>/ (merged back by gpurun) and prints a short digest.  Stages:
#   smoke                 __graft_entry__.smoke()
#   tests[:<pytest -k expression>]        the GPU suite (or a subset), -x -q
#   quick                 tests/cpp/parity_quick (C5 open loop + 4 closed-loop generations vs the oracle, seconds)
#   bench20 / bench200    the headline line at the driver's protocol (--steps 20 --warmup 5) / at the default; full line kept
#   benchq20 / benchq200  the same without cold reference, host path and CPU baseline (A/B runs)
#   closed                bench.py --closed-loop (oracle-verified generations + step parts)
#   closedq               ... without verification (A/B runs)
#   configs               bench lines of BASELINE configs 2-4
#   routes                scripts/gpu_route_times.py (masked / top-k route times at 64k)
#   small                 scripts/gpu_small_batch_latency.py
#   doorbell claim evictloop insertbreak     scripts/micro/_bin/<name> (built in the build container)
#   stats / stats_cl      rocprofv3 --kernel-trace --stats of the headline / the closed loop
#   pmc / pmc_cold / pmc_cl   the PMC passes behind profiles/pmc_*.json (then scripts/make_pmc_json.py / make_pmc_cl_json.py in the build container)
#   set:VAR=value / unset:VAR   environment for the stages that follow
#   benchq:"<args>"       bench.py without the side legs, any arguments;  trace20 = rocprofv3 timeline of the driver-protocol region;  routes_nopause
# Environment variables pass through (EPPK_QUAD_TAIL=0 bash scripts/gpu.sh ...).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
T0=$(date +%s)
lap() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
digest() { python - "$1" <<'EOF'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("   (no JSON line:", e, ")"); sys.exit(0)
r = d.get("roofline") or {}
s = f"   value {d['value'] / 1e6:9.1f} M/s  {d['ms_per_step'] * 1e3:7.2f} us/step  kernel avg {1e3 * (r.get('kernel_avg_ms') or 0):6.2f} us  hbm_frac {r.get('hbm_frac') or r.get('frac') or 0:.3f}"
p = d.get("parity") or {}
if p: s += f"  parity {p}"
cl = d.get("closed_loop") or {}
if cl: s += f"\n   closed_loop {({k: cl[k] for k in cl if k in ('value', 'ms_per_step', 'generations_verified', 'picks_equal_oracle', 'scores_bitwise_equal_oracle', 'index_size', 'index_size_oracle')})}"
rc = d.get("roofline_closed_loop") or (cl.get("roofline") if isinstance(cl, dict) else None) or {}
if rc: s += f"\n   step parts {rc.get('step_parts_ms')}"
h = d.get("host_path") or {}
if h:
    s += "\n   host_path " + str({k: (round(v.get('decisions_per_s', v.get('decisions_per_s_p50', 0)) / 1e6, 1) if isinstance(v, dict) else v) for k, v in h.items() if k in ('staged', 'pipelined', 'pipelined_learn', 'decisions_per_s_p50')})
    lb = (h.get("latency_by_batch") or {}).get("requests")
    if lb: s += "\n   latency_by_batch " + str({n: {k: (round(v, 1) if isinstance(v, float) else v) for k, v in e.items() if not k.startswith("resident_")} for n, e in lb.items()})
rv = (d.get("revisit") or {}).get("by_fraction") or {}
if rv: s += "\n   revisit " + str({f: (round(v.get("kernel_us_per_batch") or 0, 1), round(v.get("deferred_per_launch") or 0), v.get("picks_and_scores_equal_oracle")) for f, v in rv.items()})
rcold = d.get("roofline_cold") or {}
if rcold: s += f"\n   cold {rcold.get('value', 0) / 1e6:.0f} M/s kernel {1e3 * (rcold.get('kernel_avg_ms') or 0):.1f} us frac {rcold.get('frac') or 0:.3f} strict {rcold.get('frac_strict') or 0:.3f}"
s += f"\n   config keys {len(d.get('config') or {})}"
print(s)
EOF
}
for stage in "$@"; do
  name=${stage%%:*}; arg=""; [[ "$stage" == *:* ]] && arg=${stage#*:}
  case $name in
    set) export "$arg"; echo "export $arg" ;;
    unset) unset "$arg" ;;
    smoke) timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 ;;
    tests) if [ -n "$arg" ]; then timeout 1500 python -m pytest tests -m gpu -x -q --tb=short -k "$arg" 2>&1 | tail -30 | cut -c1-400 | tee $OUT/pytest_${arg//[^a-zA-Z0-9]/_}.txt
           else timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $OUT/pytest_gpu.txt; fi ;;
    quick) timeout 200 python scripts/dump_workload.py --config 5 --out /tmp/c5 > /dev/null 2>&1; timeout 120 ./tests/cpp/parity_quick /tmp/c5 4 2>&1 | tail -4 ;;
    bench20)  timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err; digest $OUT/bench_steps20.json ;;
    bench200) timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; digest $OUT/bench.json ;;
    benchq20)  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --host-path 0 --no-cold-ref --no-revisit-leg --p99-samples 0 > $OUT/benchq20.json 2> $OUT/benchq20.err; digest $OUT/benchq20.json ;;
    benchq200) timeout 300 python bench.py --no-cpu-baseline --host-path 0 --no-cold-ref --no-revisit-leg > $OUT/benchq200.json 2> $OUT/benchq200.err; digest $OUT/benchq200.json ;;
    closed)  timeout 600 python bench.py --closed-loop > $OUT/bench_closed_loop.json 2> $OUT/bench_closed_loop.err; digest $OUT/bench_closed_loop.json ;;
    closedq) f=$OUT/closedq${arg//[^a-zA-Z0-9]/_}; timeout 300 python bench.py --closed-loop --cl-verify 0 $arg > $f.json 2> $f.err; digest $f.json ;;
    configs) for c in 2 3 4; do timeout 300 python bench.py --config $c --no-cold-ref --no-revisit-leg > $OUT/bench_c$c.json 2> $OUT/bench_c$c.err; echo "config $c:"; digest $OUT/bench_c$c.json; done ;;
    routes) timeout 300 python scripts/gpu_route_times.py > $OUT/route_times.json 2> $OUT/route_times.err; tail -c 1500 $OUT/route_times.json ;;
    routes_nopause) EPPK_QUAD_PAUSE=0 timeout 300 python scripts/gpu_route_times.py > $OUT/route_times_nopause.json 2> $OUT/route_times_nopause.err; tail -c 1500 $OUT/route_times_nopause.json ;;
    benchq) f=$OUT/benchq${arg//[^a-zA-Z0-9]/_}; timeout 300 python bench.py --no-cpu-baseline --host-path 0 --no-cold-ref --no-revisit-leg --p99-samples 0 --no-closed-loop-leg $arg > $f.json 2> $f.err; digest $f.json; grep "host time" $f.err ;;
    trace20) ( cd /tmp; EPPK_BENCH_HOSTTIME=1 timeout -k 5 240 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace20 -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --host-path 0 --no-cold-ref --no-revisit-leg --p99-samples 0 --no-closed-loop-leg $arg > $OUT/trace20_bench.json 2> $OUT/trace20.err )
           digest $OUT/trace20_bench.json; grep "host time" $OUT/trace20.err
           python - $OUT/trace20 <<'EOF3' | tee $OUT/trace20_timeline.txt
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-70:]) for r in csv.DictReader(open(f))]
    rows.sort()
    picks = [x for x in rows if "pick_quad_kernel" in x[2]]
    if not picks: continue
    # the timed region = the 20 consecutive launches behind the last idle gap > 200 us that has >= 20 launches behind it ... print everything, mark gaps
    t_first = picks[0][0]; prev_end = None
    for s, e, n in picks:
        gap = "" if prev_end is None else f"gap to previous END {(s - prev_end) / 1e3:8.1f} us"
        print(f"start {(s - t_first) / 1e3:10.1f} us  dur {(e - s) / 1e3:6.1f} us  {gap}")
        prev_end = e
EOF3
           rm -f $(find $OUT/trace20 -name "*agent_info.csv") ;;
    linegather2) timeout 120 ./scripts/micro/_bin/linegather2 2>&1 | tee $OUT/micro_linegather2.txt
      mkdir -p $OUT/lg2; i=0
      for ctrs in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum" "FETCH_SIZE"; do i=$((i+1))
        ( cd /tmp; timeout -k 5 150 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/lg2 -o pass$i -- $GRAFT_REPO_ROOT/scripts/micro/_bin/linegather2 > /dev/null 2> $OUT/lg2/pass$i.err )
      done
      python scripts/pmc_summary.py $OUT/lg2 gather2 --by-kernel | tee $OUT/micro_linegather2_pmc.csv | cut -c1-160
      rm -f $(find $OUT/lg2 -name "*agent_info.csv") $(find $OUT/lg2 -name "*kernel_trace.csv") ;;
    dist1) # the N > 1 code path of bench.py on ONE GPU: RCCL at world size 1, through torch.distributed.run as the driver launches it
      timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --force-dist --steps 20 --warmup 5 --no-cpu-baseline --host-path 0 --no-cold-ref --no-revisit-leg $arg > $OUT/bench_force_dist.json 2> $OUT/bench_force_dist.err
      digest $OUT/bench_force_dist.json; tail -3 $OUT/bench_force_dist.err | cut -c1-300
      python - $OUT/bench_force_dist.json <<'EOF4'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("   metric:", d["metric"]); print("   scaling_note:", d.get("scaling_note")); c = d["config_detail"]
    print("   ", {k: c.get(k) for k in ("ranks_seen", "per_rank_kernel_us", "collective_us", "sharding")}); print("   strong:", {k: (d.get("strong") or {}).get(k) for k in ("value", "ms_per_step")}, " completion_latency:", d.get("completion_latency"))
except Exception as e:
    print("   (", e, ")")
EOF4
      ;;
    small) timeout 300 python scripts/gpu_small_batch_latency.py 2>&1 | tee $OUT/small_batch_latency.txt | tail -12 ;;
    doorbell|claim|claim2|evictloop|insertbreak)
      bin=$name; [ $name = claim ] && bin=claimcost; [ $name = claim2 ] && bin=claimcost2
      timeout 60 ./scripts/micro/_bin/$bin $arg 2>&1 | tee $OUT/micro_$name.txt | tail -40 ;;
    stats) ( cd /tmp; timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline --host-path 0 --no-cold-ref --no-revisit-leg --p99-samples 0 > $OUT/prof_bench_under_rocprof.json 2> $OUT/prof.err )
           head -6 $OUT/prof/*kernel_stats.csv 2>/dev/null | cut -c1-200; rm -f $(find $OUT/prof -name "*agent_info.csv") $(find $OUT/prof -name "*kernel_trace.csv") ;;
    stats_cl) ( cd /tmp; timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_cl -o trace -- python $GRAFT_REPO_ROOT/bench.py --closed-loop --steps 60 --warmup 10 --no-cpu-baseline --cl-verify 0 > $OUT/prof_cl_bench_under_rocprof.json 2> $OUT/prof_cl.err )
           head -8 $OUT/prof_cl/*kernel_stats.csv 2>/dev/null | cut -c1-200
           python - $OUT/prof_cl <<'EOF2' | tee $OUT/prof_cl_durations.txt
import csv, glob, sys
import numpy as np
for f in glob.glob(sys.argv[1] + "/*kernel_trace.csv"):
    d = {}
    for r in csv.DictReader(open(f)):
        d.setdefault(r["Kernel_Name"].split("(")[0][:60], []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    for k, v in d.items():
        if len(v) < 20: continue
        v.sort(); t = np.asarray([x[1] for x in v[10:]]) / 1e3        # in launch order, warm-up launches dropped
        print(f"{k:60s} n={t.size:4d} deciles us: " + " ".join(f"{np.percentile(t, q):6.1f}" for q in (0, 10, 25, 50, 75, 90, 100)) + f"   even/odd launches: {t[0::2].mean():6.1f} / {t[1::2].mean():6.1f}")
EOF2
           rm -f $(find $OUT/prof_cl -name "*agent_info.csv") $(find $OUT/prof_cl -name "*kernel_trace.csv") ;;
    pmc|pmc_cold)
      if [ $name = pmc ]; then ARGS="--steps 6 --warmup 10 --no-cpu-baseline --host-path 0 --no-cold-ref --no-revisit-leg --p99-samples 0 --inflight 1"
        CTRS=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TA_BUSY_avr TA_TA_BUSY_sum TCC_BUSY_avr")
      else ARGS="--steps 6 --warmup 10 --no-cpu-baseline --host-path 0 --no-cold-ref --no-revisit-leg --p99-samples 0 --groups 262144 --zipf 0 --pods-per-group 4 --batches 4 --inflight 1"
        CTRS=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum"); fi
      mkdir -p $OUT/$name; i=0
      for ctrs in "${CTRS[@]}"; do i=$((i+1))     # (counters in passes of their own, with --kernel-trace only: /opt/skills/guides/MI355X_MICROARCH.md)
        ( cd /tmp; timeout -k 5 150 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/$name -o pass$i -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/$name/bench_pass$i.json 2> $OUT/$name/pass$i.err )
      done
      python scripts/pmc_summary.py $OUT/$name pick_ --by-kernel | tee $OUT/${name}_summary.csv | cut -c1-200
      rm -f $(find $OUT/$name -name "*agent_info.csv") $(find $OUT/$name -name "*kernel_trace.csv") ;;
    pmc_cl) # counter passes of the closed loop's kernels (LEARN pick, update, eviction) -> scripts/make_pmc_cl_json.py in the build container
      mkdir -p $OUT/pmc_cl; i=0
      for ctrs in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum" "WRITE_SIZE" "FETCH_SIZE"; do i=$((i+1))
        ( cd /tmp; timeout -k 5 200 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/pmc_cl -o pass$i -- python $GRAFT_REPO_ROOT/bench.py --closed-loop --steps 12 --warmup 6 --cl-verify 0 --no-cpu-baseline > $OUT/pmc_cl/bench_pass$i.json 2> $OUT/pmc_cl/pass$i.err )
      done
      python scripts/pmc_summary.py $OUT/pmc_cl _kernel --by-kernel | grep -v "snap_\|lists_fill\|hash_" | tee $OUT/pmc_cl_summary.csv | cut -c1-200
      rm -f $(find $OUT/pmc_cl -name "*agent_info.csv") $(find $OUT/pmc_cl -name "*kernel_trace.csv") ;;
    *) echo "unknown stage $stage" ;;
  esac
  lap $stage
done
du -sh $OUT

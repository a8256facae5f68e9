#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$PWD/gpurun_out/${FINAL_DIR:-r3final}
rm -rf $OUT; mkdir -p $OUT
timeout 500 python bench.py > $OUT/bench_c5.json 2> $OUT/bench_c5.err; tail -2 $OUT/bench_c5.err
timeout 500 python bench.py --steps 20 --warmup 5 > $OUT/bench_c5_steps20.json 2> /dev/null
timeout 300 python bench.py --closed-loop --no-cpu-baseline > $OUT/bench_closed_loop.json 2>/dev/null
for c in 2 3 4; do timeout 200 python bench.py --config $c --no-cold-ref --host-path 0 > $OUT/bench_c$c.json 2>/dev/null; done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/'+__import__('os').environ.get('FINAL_DIR','r3final')+'/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    r=d.get('roofline',{})
    print(f.split('/')[-1], round(d['value']/1e6,1),'M/s', round(d['ms_per_step']*1e3,2),'us', r.get('bound'), round(r.get('frac') or 0,3), 'hbm', round(r.get('hbm_frac') or 0,3), 'traffic', r.get('traffic'), (d.get('cpu_baseline') or {}).get('value'))
    if 'roofline_cold' in d: rc=d['roofline_cold']; print('  cold', round(rc['value']/1e6,1), round(rc['frac'],3), rc.get('traffic_frac'), rc.get('kernel_avg_ms'))
    if 'host_path' in d:
        print('  host', {k:(round(v['decisions_per_s']/1e6,1) if 'decisions_per_s' in v else round(v['decisions_per_s_p50']/1e6,1)) for k,v in d['host_path'].items() if isinstance(v,dict) and ('decisions_per_s' in v or 'decisions_per_s_p50' in v)}, round(d['host_path']['decisions_per_s_p50']/1e6,1))
        print('  latency_by_batch', (d['host_path'].get('latency_by_batch') or {}).get('requests'))
    if 'roofline_closed_loop' in d: print('  cl', d['roofline_closed_loop']['step_parts_ms'], round(d['roofline_closed_loop']['frac'],3), d['roofline_closed_loop']['frac_of_random_line_floor'])
P

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; OUT=$PWD/gpurun_out/r3i; rm -rf $OUT; mkdir -p $OUT
show() { python -c "
import json
s=open('$1').read(); d=json.loads([l for l in s.split('\n') if l.startswith('{')][-1])
print('$1'.split('/')[-1], 'strong %.1f M/s (%.4f ms/step)'%(d['value']/1e6, d['ms_per_step']), 'weak %.1f'%(d.get('weak',{}).get('value',0)/1e6), d['config'].get('requests_per_launch'), d.get('parity'))"; }
B="--force-dist --no-cpu-baseline --host-path 0 --no-cold-ref --p99-samples 0 --scaling strong"
for r in 8192 16384 32768 65536; do for g in 8 16; do
timeout 300 python bench.py $B --requests $r --gather-every $g > $OUT/fd_${r}_g$g.json 2>/dev/null; show $OUT/fd_${r}_g$g.json
done; done
timeout 300 python bench.py --force-dist --host-path 0 --no-cold-ref --p99-samples 0 --steps 64 --warmup 16 > $OUT/fd_full_parity.json 2>/dev/null; show $OUT/fd_full_parity.json

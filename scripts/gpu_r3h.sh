#!/bin/bash
# Round 2, session 3h: the N > 1 code path on one GPU with and without launch groups (strong scaling: one launch per gather bucket),
# at the per-rank load of 8 GPUs (8192 requests per step) and at the full batch; suite with the batch-size threshold.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; OUT=$PWD/gpurun_out/r3h; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
show() { python -c "
import json
s=open('$1').read(); d=json.loads([l for l in s.split('\n') if l.startswith('{')][-1])
print('$1'.split('/')[-1], 'strong %.1f M/s (%.4f ms/step)'%(d['value']/1e6, d['ms_per_step']), 'weak %.1f'%(d.get('weak',{}).get('value',0)/1e6), d['config'].get('requests_per_launch'), d.get('parity'))"; }
B="--force-dist --no-cpu-baseline --host-path 0 --no-cold-ref --p99-samples 0"
timeout 300 python bench.py $B > $OUT/fd_full_grouped.json 2>/dev/null; show $OUT/fd_full_grouped.json
timeout 300 python bench.py $B --no-launch-groups > $OUT/fd_full_nogroups.json 2>/dev/null; show $OUT/fd_full_nogroups.json
for r in 8192 16384 32768; do
timeout 300 python bench.py $B --requests $r > $OUT/fd_${r}_grouped.json 2>/dev/null; show $OUT/fd_${r}_grouped.json
timeout 300 python bench.py $B --requests $r --no-launch-groups > $OUT/fd_${r}_nogroups.json 2>/dev/null; show $OUT/fd_${r}_nogroups.json
done

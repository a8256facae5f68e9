#!/bin/bash
# kernel timeline of a 20-step weak-scaling region on one GPU (where do the 100 us beyond the single-GPU region go?)
# (as run in round 3 the int16 cast was the default and the variant without it was called "nocast": gpurun_out/r3z4_gaps.txt)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3z4
rm -rf $OUT; mkdir -p $OUT
for dbg in none nolat nopack nogather; do   # (all but "nopack" with --pack16: the cast kernel is what the experiment is about)
rm -rf $OUT/trace
( cd /tmp; EPPK_BENCH_DBG=$dbg timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o weak -- python $GRAFT_REPO_ROOT/bench.py --force-dist --scaling weak --steps 20 --warmup 5 --no-cpu-baseline --no-cold-ref --host-path 0 --p99-samples 0 $( [ $dbg = nopack ] || echo --pack16 ) > $OUT/weak.json 2> $OUT/weak.err )
DBGN=$dbg python - <<'P'
import csv, glob, os
f = glob.glob(os.environ['OUT'] + '/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
picks = [r for r in rows if 'pick_quad' in r['Kernel_Name']][-20:]
t0 = int(picks[0]['Start_Timestamp'])
gaps = []
end = int(picks[0]['End_Timestamp'])
for r in picks[1:]:
    st = int(r['Start_Timestamp'])
    if st > end: gaps.append(((st - end) / 1e3, (end - t0) / 1e3))
    end = max(end, int(r['End_Timestamp']))
print(os.environ['DBGN'], 'pick span %.1f us; idle gaps between picks (us, at): %s' % ((end - t0) / 1e3, [(round(g, 1), round(a)) for g, a in gaps]))
P
done
rm -rf $OUT/trace

#!/bin/bash
# Round 2, session 3d: A/B of pick_quad_kernel register targets (4 vs 5 wavefronts per SIMD) and workgroup sizes.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3d
rm -rf $OUT; mkdir -p $OUT
B="--no-cold-ref --no-cpu-baseline --host-path 0 --p99-samples 0 --steps 100 --warmup 10"
run() { # name lib threads inflight
  EPPK_LIB=$PWD/ab/libeppk_$2.so EPPK_QUAD_THREADS=$3 timeout 200 python bench.py $B --inflight $4 2>/dev/null > $OUT/$1.json
  python -c "import sys,json; d=json.load(open('$OUT/$1.json')); print('$1', 'kernel_ms=%.4f'%d['roofline']['kernel_avg_ms'], 'step_ms=%.4f'%d['ms_per_step'], 'Mdec/s=%.1f'%(d['value']/1e6), 'parity', d['config'].get('parity_full_batch'))"
}
for rep in 1 2; do
run w4_t512_i1 a_w4 512 1
run w4_t512_i2 a_w4 512 2
run w4_t1024_i2 a_w4 1024 2
run w5_t640_i1 b_w5 640 1
run w5_t640_i2 b_w5 640 2
run w5_t320_i2 b_w5 320 2
done 2>&1 | tee $OUT/ab.txt

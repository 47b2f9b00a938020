#!/bin/bash
# Round 6, call F: fused SwiGLU backward — [gate | up] batches in flight (G4_SB_DEPTH 1 / 2 / 3) and the grouped launch on the persistent
# 4-wave walk (LMOD_GEMM_SB4G); kernel level (alternating processes), then the new tests, then the step.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=$PWD/gpurun_out/${1:-r6f}
mkdir -p $OUT
for rep in 1 2; do
  for v in sb1 default sb3; do
    lib=""; [ $v != default ] && lib=$PWD/alt_libs/liblmod_$v.so
    for g in 1 0; do
      echo "== depth $v SB4G=$g" | tee -a $OUT/swiglu_bwd_ab.txt
      LMOD_HIP_LIB=$lib LMOD_GEMM_SB4G=$g timeout 200 python tools/bench_swiglu_bwd.py 2>/dev/null | tee -a $OUT/swiglu_bwd_ab.txt
    done
  done
done
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "swiglu" > $OUT/pytest_swiglu.txt 2>&1; echo "pytest swiglu rc=$?"; tail -3 $OUT/pytest_swiglu.txt
for v in default sb1 sb3 default sb1 sb3; do
  lib=""; [ $v != default ] && lib=$PWD/alt_libs/liblmod_$v.so
  LMOD_HIP_LIB=$lib timeout 400 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2> $OUT/bench_$v.err | grep '^{' | tail -1 > $OUT/bench_${v}_$RANDOM.json
done
for f in $OUT/bench_*.json; do python -c "
import json,sys
r=json.load(open('$f')); print('$f'.split('/')[-1], r['value'], r['ms_per_step'], r['config']['final_loss'], r['roofline']['in_step']['families'].get('gemm4_kernel<4> / gemm_256_kernel<4> fused SwiGLU backward'))" | cut -c1-400; done

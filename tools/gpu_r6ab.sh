#!/bin/bash
# Round 6: the round's kernel library against round 5's (alt_libs/liblmod_r05.so, built from `git archive c6a068d llava-mod_amd/csrc`) under the
# SAME host code, on ONE box, alternating processes — the headline delta without box-to-box spread
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=$PWD/gpurun_out/${1:-r6ab}
mkdir -p $OUT
for rep in 1 2 3 4; do
  for lib in r05 r06; do
    L="X=1"; [ $lib = r05 ] && L="LMOD_HIP_LIB=$PWD/alt_libs/liblmod_r05.so"
    env $L timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); f = r['roofline']['in_step']['families']
print(json.dumps({'lib': '$lib', 'samples_per_s': r['value'], 'ms_per_step': r['ms_per_step'], 'final_loss': r['config']['final_loss'],
                  'attn_bwd_ms': f['attn_bwd2_kernel (dQ + dK/dV + delta)']['ms'], 'swiglu_bwd_ms': f['gemm4_kernel<4> / gemm_256_kernel<4> fused SwiGLU backward']['ms'],
                  'residual_ms': f['gemm4_kernel<8> + residual add']['ms']}))" | tee -a $OUT/r05_vs_r06.jsonl
  done
done

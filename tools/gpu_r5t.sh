#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5t; mkdir -p $O
timeout 200 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attn_fwd_bwd or fused_rope" 2>&1 | tail -2
cd tools
for v in shipped b2nospread shipped b2nospread; do
  lib=../alt_libs/liblmod_$v.so; [ -f $lib ] || lib=../llava-mod_amd/llavamod/_lib/liblmod_hip.so
  echo "== $v" >> ../$O/hd64_spread_ab.txt
  LMOD_HIP_LIB=$PWD/$lib timeout 120 python bench_attn.py --hd64 2>/dev/null | grep attn_bwd >> ../$O/hd64_spread_ab.txt
done
cat ../$O/hd64_spread_ab.txt | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('=='): print(l); continue
    d=json.loads(l); print('  ', d['B'], d['S'], d['nh'], d['causal'], d['tflops_algo(2.5x fwd)'])
"

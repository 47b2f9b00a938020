#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5w; mkdir -p $O
cd tools
for v in shipped f2pk shipped f2pk; do
  lib=../alt_libs/liblmod_$v.so; [ -f $lib ] || lib=../llava-mod_amd/llavamod/_lib/liblmod_hip.so
  echo "== $v" >> ../$O/fwd_pk_ab.txt
  LMOD_HIP_LIB=$PWD/$lib timeout 120 python bench_attn.py --hd64 2>/dev/null | grep attn_fwd >> ../$O/fwd_pk_ab.txt
  LMOD_HIP_LIB=$PWD/$lib timeout 120 python bench_attn.py 2>/dev/null | grep attn_fwd >> ../$O/fwd_pk_ab.txt
done
cat ../$O/fwd_pk_ab.txt | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('=='): print(l); continue
    d=json.loads(l); print('  ', d['hd'], d['B'], d['S'], d['nh'], d['causal'], d.get('ragged'), d['tflops'])
"

#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/abl
echo base; timeout 200 python tools/bench_attn.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print(' ', r['B'], r['S'], r['nh'], 'C' if r['causal'] else 'F', r['ms'], r['tflops'])
"
for v in ${1:-1 2 3 4 5 6}; do
  echo abl$v; LMOD_HIP_LIB=$PWD/alt_libs/liblmod_abl$v.so timeout 200 python tools/bench_attn.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print(' ', r['B'], r['S'], r['nh'], 'C' if r['causal'] else 'F', r['ms'], r['tflops'])
"
done

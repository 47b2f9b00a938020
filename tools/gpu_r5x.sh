#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5x; mkdir -p $O
timeout 200 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attn" 2>&1 | tail -2
cd tools
for v in 1 0 1 0; do
  echo "== LMOD_ATTN_XCD=$v" >> ../$O/xcd_ab.txt
  LMOD_ATTN_XCD=$v timeout 120 python bench_attn.py --hd64 2>/dev/null >> ../$O/xcd_ab.txt
done
timeout 120 python bench_attn.py --bwd-only 2>/dev/null > ../$O/hd128.jsonl
cat ../$O/xcd_ab.txt ../$O/hd128.jsonl | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('=='): print(l); continue
    d=json.loads(l); print('  ', d['kernel'], d['hd'], d['B'], d['S'], d['nh'], d['causal'], d.get('tflops', d.get('tflops_algo(2.5x fwd)')))
"

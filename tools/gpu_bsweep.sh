#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/bsweep
for b in ${1:-1 4 8 16 24}; do
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --micro-batch $b 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print(json.dumps({'micro_batch': $b, 'grad_accum': r['config']['grad_accum'], 'samples_per_s': r['value'], 'ms_per_step': r['ms_per_step'], 'whole_step_frac': r['roofline']['whole_step']['frac'], 'peak_hbm_gb': r['config']['peak_hbm_gb']}))
" | tee -a gpurun_out/bsweep/sweep.jsonl
done

#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/ragged
for v in "" "--ragged" "--ragged --unpad"; do
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $v 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print('$v', '|', r['value'], 'samples/s', r['ms_per_step'], 'ms', r['config']['batch_shape'][:60], 'loss', r['config']['final_loss'], 'hbm', r['config']['peak_hbm_gb'])
" | tee -a gpurun_out/ragged/summary.txt
done

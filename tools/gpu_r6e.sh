#!/bin/bash
# Round 6, call E: full GPU suite on the dS-spill build + step-level A/B (LMOD_ATTN_DS=1/0 alternating) + the 2-rank bench tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-r6e}
mkdir -p $OUT
for ds in 1 0 1 0; do
  LMOD_ATTN_DS=$ds timeout 400 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2> $OUT/bench_ds$ds.err | grep '^{' | tail -1 > $OUT/bench_ds${ds}_$RANDOM.json
done
for f in $OUT/bench_ds*.json; do python -c "
import json,sys
r=json.load(open('$f')); print('$f'.split('/')[-1], r['value'], r['ms_per_step'], r['config']['final_loss'], r['roofline']['kernel'][:40], r['roofline']['achieved'])"; done
timeout 2400 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.txt 2>&1; echo "pytest gpu rc=$?"; tail -8 $OUT/pytest_gpu.txt

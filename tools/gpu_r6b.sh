#!/bin/bash
# Round 6, call B: the dS-spill form of the attention backward — parity first, then A/B against the two-kernel form (kernel level,
# per-kernel durations from a rocprofv3 trace, and in the step).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-r6b}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "attn" > $OUT/pytest_attn.txt 2>&1; echo "pytest attn rc=$?"; tail -5 $OUT/pytest_attn.txt
for ds in 1 0 1 0; do
  LMOD_ATTN_DS=$ds timeout 300 python tools/bench_attn.py --bwd-only 2>/dev/null | grep attn_bwd | tee -a $OUT/attn_bwd_ds_ab.jsonl
done
(cd /tmp && LMOD_ATTN_DS=1 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_ds1 -o a --output-format csv -- python $OLDPWD/tools/bench_attn.py --bwd-one > $OUT/trace_ds1.log 2>&1)
(cd /tmp && LMOD_ATTN_DS=0 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_ds0 -o a --output-format csv -- python $OLDPWD/tools/bench_attn.py --bwd-one > $OUT/trace_ds0.log 2>&1)
head -8 $OUT/trace_ds1/*/a_kernel_stats.csv 2>/dev/null || find $OUT/trace_ds1 -name "*kernel_stats.csv" | head -1 | xargs head -8
for ds in 1 0 1 0; do
  LMOD_ATTN_DS=$ds timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2> $OUT/bench_ds$ds.err | grep '^{' | tail -1 > $OUT/bench_ds${ds}_$RANDOM.json
done
for f in $OUT/bench_ds*.json; do python -c "
import json,sys
r=json.load(open('$f')); print('$f'.split('/')[-1], r['value'], r['ms_per_step'], r['config']['final_loss'])"; done

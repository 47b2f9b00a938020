#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-r6c}
mkdir -p $OUT
for a in "2 768 2 2 0" "1 512 2 2 1" "2 768 4 2 1" "1 512 1 1 0"; do echo "== $a"; timeout 200 python tools/dbg_ds.py $a 2>&1 | grep -v amdgpu.ids | tail -12; done > $OUT/dbg_ds.txt 2>&1
cat $OUT/dbg_ds.txt | cut -c1-400
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "attn" > $OUT/pytest_attn.txt 2>&1; echo "pytest attn rc=$?"; tail -8 $OUT/pytest_attn.txt
for ds in 1 0; do
  LMOD_ATTN_DS=$ds timeout 300 python tools/bench_attn.py --bwd-only 2>/dev/null | grep attn_bwd | tee -a $OUT/attn_bwd_ds_ab.jsonl
done
(cd /tmp && LMOD_ATTN_DS=1 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_ds1 -o a --output-format csv -- python $OLDPWD/tools/bench_attn.py --bwd-one > $OUT/trace_ds1.log 2>&1)
head -5 $OUT/trace_ds1/a_kernel_stats.csv | cut -c1-200

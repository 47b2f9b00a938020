#!/bin/bash
# rocprofv3 kernel trace of the default bench workload (teacher pipelining off so kernel durations do not overlap)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-prof}
mkdir -p $OUT
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o a --output-format csv -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-teacher-prefetch > $OUT/bench.log 2>&1)
ls $OUT | head; tail -2 $OUT/bench.log | cut -c1-300
python tools/profile_md.py $OUT/a_kernel_stats.csv > $OUT/kernel_stats.md 2>/dev/null; head -40 $OUT/kernel_stats.md

#!/bin/bash
# per-kernel durations of the attention backward (kernel trace) at: causal B16, non-causal B8, causal S8192 B4
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-b4}
mkdir -p $OUT
for cfg in "16 2048 1" "8 2048 0" "4 8192 1"; do
  set -- $cfg
  tag=B$1_S$2_c$3
  (cd /tmp && A1_B=$1 A1_S=$2 A1_CAUSAL=$3 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/$tag -o a --output-format csv -- python $OLDPWD/tools/attn_one.py bwd > $OUT/$tag.log 2>&1)
  python - <<PY
import csv, glob
for f in glob.glob("$OUT/$tag/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "attn" in r["Name"]: print("$tag", r["Name"][:48], r["Calls"], "avg_us", round(float(r["AverageNs"])/1e3, 1))
PY
done
